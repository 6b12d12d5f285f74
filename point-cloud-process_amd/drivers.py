"""The registration driver loop of Registration/main.py:183-222 (and icp_template.py:226-272) as a function.

Reads the pair list (rows ``trg,src,...`` after a header, main.py:186-194), loads ``<root>/<id>.bin`` clouds of
6 x float32 records (main.py:10-17), registers every pair with ``icp_point2point`` from the given initial guesses
(``init="global"`` computes them like main.py:196-203: voxel 2.0 m down-sample, FPFH, RANSAC) and writes the
result CSV in the reference's format (main.py:220-222).
"""
from __future__ import annotations

import os

import numpy as np

from .batch import register_batch
from .registration import homo2tq, read_bin_velodyne, write_reg_result

__all__ = ["read_pair_list", "run_registration"]


def read_pair_list(path):
    """[(trg_id, src_id), ...] from a reg_result-style file (header skipped, main.py:186-194)."""
    pairs = []
    with open(path) as f:
        for line in f.readlines()[1:]:
            parts = line.split(",")
            if len(parts) >= 2 and parts[0].strip():
                pairs.append((int(float(parts[0])), int(float(parts[1]))))
    return pairs


def run_registration(pair_list_path, cloud_root, out_path, init=None, mode="compat", streams=2, voxel_size=2.0, **kw):
    """Register every listed pair; ``init`` maps (trg, src) -> 4x4 initial guess (default identity), or is the string
    "global" to run the reference's own initialisation (prepare_dataset + execute_global_registration, main.py:196-203,
    voxel_size = 2.0) for every pair.  Returns the (n, 9) result table that was written."""
    pairs = read_pair_list(pair_list_path)
    cache = {}

    def cloud(i):
        if i not in cache:
            cache[i] = read_bin_velodyne(os.path.join(cloud_root, f"{i}.bin"))
        return cache[i]

    if isinstance(init, str):
        if init != "global":
            raise ValueError("init must be a dict, None or 'global'")
        from .global_registration import execute_global_registration, preprocess_point_cloud
        from .registration import PointCloud

        prep = {}

        def pre(i):
            if i not in prep:
                prep[i] = preprocess_point_cloud(PointCloud(cloud(i)), voxel_size)
            return prep[i]

        init = {}
        for trg, src in pairs:
            (s_down, s_f), (t_down, t_f) = pre(src), pre(trg)
            init[(trg, src)] = execute_global_registration(s_down, t_down, s_f, t_f, voxel_size).transformation
    work = [(cloud(src), cloud(trg), None if init is None else init.get((trg, src))) for trg, src in pairs]
    results = register_batch(work, streams=streams, mode=mode, **kw)
    table = np.zeros((len(pairs), 9))
    for row, (trg, src), res in zip(table, pairs, results):
        row[:2] = (trg, src)
        row[2:] = homo2tq(res["T"])
    write_reg_result(out_path, table)
    return table
