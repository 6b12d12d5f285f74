"""CPU-only checks: the C-ABI library loads and exports every symbol include/pcr.h declares;
host-only entry points (Procrustes, pose utils) run without a GPU; compute entry points fail loudly."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT, load_golden


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pcr.h")).read()
    return sorted(set(re.findall(r"PCR_API\s+[\w\s\*]+?\b(pcr_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pcp):
    L = pcp._lib
    lib = L.lib()
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pcr.h but not exported by libpcr.so"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature"
    assert set(L.SIGNATURES) == set(names)


def test_no_gpu_means_loud_failure_not_fallback(pcp):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError) as e:
        pcp.Context(0)
    assert "no HIP device" in str(e.value) or "status" in str(e.value)


def test_host_only_entry_points(pcp):
    g = load_golden("procrustes.npz")
    R, t, cost = pcp.procrustes_transformation(g["K500_A"], g["K500_B"])
    assert np.abs(R - g["K500_R"]).max() < 1e-9
    assert np.abs(t - g["K500_t"]).max() < 1e-9
    p = load_golden("pose_utils.npz")
    for T, tq in zip(p["T"], p["tq"]):
        assert np.allclose(pcp.homo2tq(T), tq, rtol=0, atol=1e-15)


def test_procrustes_rank_deficient_inputs(pcp):
    """Planar, collinear and single-point correspondences (H of rank 2, 1, 0: main.py:131-141 does not special-case them, numpy's
    SVD completes U and V somehow): R must be orthogonal and the transform must reproduce B on the data itself.  These are the inputs
    of pcr_linalg.h's completion of U, the code that was rewritten (selects on scalars instead of variable-index stores) to keep
    every ICP kernel free of scratch memory."""
    rng = np.random.default_rng(3)

    def rot(axis, ang):
        a = np.asarray(axis, float)
        a /= np.linalg.norm(a)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K

    for case in range(120):
        kind = case % 4
        if kind == 0:    # a plane in general position, or an axis-aligned one
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            e1 = np.cross(n, [1.0, 0.0, 0.0])
            e1 /= np.linalg.norm(e1)
            A = np.outer(e1, rng.normal(size=40)) + np.outer(np.cross(n, e1), rng.normal(size=40))
            if case % 8 == 0:
                A = np.vstack([rng.normal(size=40), rng.normal(size=40), np.zeros(40)])
        elif kind == 1:  # a line
            d = rng.normal(size=3) if case % 8 != 1 else np.array([0.0, 1.0, 0.0])
            A = np.outer(d, rng.normal(size=40))
        elif kind == 2:  # one point, repeated
            A = np.outer(rng.normal(size=3), np.ones(40))
        else:            # full rank, for comparison
            A = rng.normal(size=(3, 40))
        R0 = rot(rng.normal(size=3), rng.uniform(0.0, 3.0))
        B = R0 @ A + rng.normal(size=(3, 1))
        R, t, cost = pcp.procrustes_transformation(A, B)
        assert np.isfinite(R).all() and np.isfinite(t).all() and np.isfinite(cost), (case, kind)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12, (case, kind)
        assert np.abs(R @ A + t - B).max() < 1e-9 * max(1.0, np.abs(B).max()), (case, kind, np.abs(R @ A + t - B).max())
        assert cost < 1e-6, (case, kind, cost)


def test_csv_format_matches_reference(pcp, tmp_path):
    p = load_golden("pose_utils.npz")
    rows = np.zeros((len(p["T"]), 9))
    rows[:, 0] = np.arange(len(rows))
    rows[:, 1] = np.arange(len(rows)) + 100
    rows[:, 2:] = [pcp.homo2tq(T) for T in p["T"]]
    f = tmp_path / "reg_result.txt"
    pcp.write_reg_result(str(f), rows)
    assert f.read_text() == str(p["csv"])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "point-cloud-process_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                assert "oracle" not in open(os.path.join(dirpath, fn)).read().lower().replace("host oracle", ""), fn


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/pcr.h must be consumable by a C compiler (the drop-in boundary is a C ABI): compile a C99 translation unit
    against it with -pedantic and link it against libpcr.so; calling a host-only entry point proves the linkage."""
    import subprocess

    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include "pcr.h"\n'
                   "int main(void) {\n"
                   "    pcr_icp_params p; double T[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1}, out[7];\n"
                   "    pcr_icp_default_params(&p);\n"
                   "    if (pcr_homo2tq(T, out) != PCR_OK || out[3] != 1.0) return 2;\n"
                   '    printf("%s %d\\n", pcr_version(), p.max_iter);\n'
                   "    return 0;\n}\n")
    exe = tmp_path / "abi"
    lib_dir = os.path.join(ROOT, "point-cloud-process_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", lib_dir, "-l:libpcr.so", "-Wl,-rpath," + lib_dir], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=lib_dir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[-1] == "100"     # main.py:97 max_iteration default
