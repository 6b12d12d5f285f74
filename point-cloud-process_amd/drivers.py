"""The registration driver loop of Registration/main.py:183-222 (and icp_template.py:226-272) as a function.

Reads the pair list (rows ``trg,src,...`` after a header, main.py:186-194), loads ``<root>/<id>.bin`` clouds of
6 x float32 records (main.py:10-17), registers every pair with ``icp_point2point`` from the given initial guesses
(``init="global"`` computes them like main.py:196-203: voxel 2.0 m down-sample, FPFH, RANSAC) and writes the
result CSV in the reference's format (main.py:220-222).

Inside an initialised torch.distributed process group the pair list is dealt to the ranks in contiguous blocks
(batch.shard_range): a rank reads only the scans its own pairs use -- each once, however many of its pairs share it
(Registration/reg_result.txt: 342 pairs over 504 scans) --, computes the global initialisation of its OWN pairs only,
registers them, and one all_gather exchanges the result records.
"""
from __future__ import annotations

import os

import numpy as np

from .batch import register_batch, shard_range
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


def _rank_world(group=None):
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(group), dist.get_world_size(group)
    except Exception:
        pass
    return 0, 1


def run_registration(pair_list_path, cloud_root, out_path, init=None, mode="compat", streams=8, voxel_size=2.0, seed=0, group=None, **kw):
    """Register every listed pair; ``init`` maps (trg, src) -> 4x4 initial guess (default identity), or is the string
    "global" to run the reference's own initialisation (prepare_dataset + execute_global_registration, main.py:196-203,
    voxel_size = 2.0) for every pair.  Returns the (n, 9) result table (every rank gets it; rank 0 writes ``out_path``)."""
    pairs = read_pair_list(pair_list_path)
    rank, world = _rank_world(group)
    lo, hi = shard_range(len(pairs), rank, world)
    cache = {}

    def cloud(i):   # one array object per scan id: the native call recognises a scan by it
        if i not in cache:
            cache[i] = read_bin_velodyne(os.path.join(cloud_root, f"{i}.bin"))
        return cache[i]

    global_init = None
    if isinstance(init, str):
        if init != "global":
            raise ValueError("init must be a dict, None or 'global'")
        global_init = {"voxel_size": voxel_size, "seed": seed}
        init = None
    work = [(None, None, None)] * len(pairs)   # other ranks' pairs are placeholders: only the local share is touched
    for i in range(lo, hi):
        trg, src = pairs[i]
        work[i] = (cloud(src), cloud(trg), None if init is None else init.get((trg, src)))
    results = register_batch(work, streams=streams, mode=mode, global_init=global_init, group=group, **kw)
    table = np.zeros((len(pairs), 9))
    for row, (trg, src), res in zip(table, pairs, results):
        row[:2] = (trg, src)
        row[2:] = homo2tq(res["T"])
    if rank == 0:
        write_reg_result(out_path, table)
    return table
