"""Importable alias of the package directory ``point-cloud-process_amd`` (its name has a hyphen)."""
import importlib
import sys

_pkg = importlib.import_module("point-cloud-process_amd")
sys.modules[__name__] = _pkg
