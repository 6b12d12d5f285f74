"""GPU: the work queue's failure path (ADVICE r2 / VERDICT r3).  In the one-launch ICP pass (Registration/main.py:107-154 is one
iteration) idle waves wait at queue slots for items the slower tiles publish; a waiter gives up after PCR_PASS_SPIN_LIMIT polls.
With the product's limit (2^21) that never happens, so the code behind it never ran.  __graft_entry__.build() also links
point-cloud-process_amd/libpcr_giveup.so from the same sources with -DPCR_PASS_SPIN_LIMIT=2: almost every waiter gives up, and the
launch's last wave has to find and serve what they left.  The registration must come out BIT FOR BIT as with the product library
(what a wave adds to the fixed-point accumulators does not depend on who serves an item), never fail, and leave the context usable."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "point-cloud-process_amd", "libpcr_giveup.so")

CHILD = r"""
import importlib, os, sys, ctypes as C
import numpy as np
sys.path.insert(0, sys.argv[1])
pkg = importlib.import_module("point-cloud-process_amd")
L = pkg._lib
out = {}
ctx = pkg.Context(0)
for n, seed in ((20000, 3), (120000, 0)):
    src, tgt, _ = pkg.synthetic.perturbed_pair(n, seed=seed)
    index = pkg.TargetIndex(pkg.DeviceCloud.upload(tgt, ctx), ctx=ctx)
    for tag, kw in (("compat", dict(mode="compat")), ("total", dict(mode="total", max_iter=12, r_thres=-1.0, t_thres=-1.0, min_iter=12))):
        for rep in range(2):                      # the context must stay usable call after call
            sd = pkg.DeviceCloud.upload(src, ctx)
            r = pkg.icp_device(sd, index, np.eye(4), **kw)
            moved = sd.download()
            sd.free()
        out[f"{n}_{tag}_T"] = r["T"]; out[f"{n}_{tag}_Tt"] = r["T_total"]
        out[f"{n}_{tag}_meta"] = np.array([r["iters"], r["n_assoc"], r["status"]], dtype=np.float64)
        out[f"{n}_{tag}_src"] = moved
    index.free()
buf = np.zeros(1 << 19, dtype=np.uint64)
L.check(L.lib().pcr_debug_read(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size))
out["giveup_passes"] = np.array([int(buf[(1 << 19) - 16])], dtype=np.float64)
np.savez(sys.argv[2], **out)
"""


def _run(lib, path):
    env = dict(os.environ, PCR_PASS_INLINE="1", PCR_DEBUG_STAMPS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if lib:
        env["PCR_LIB_PATH"] = lib
    else:
        env.pop("PCR_LIB_PATH", None)
    subprocess.run([sys.executable, "-c", CHILD, ROOT, path], check=True, env=env, timeout=600)
    return np.load(path)


def test_waiters_that_give_up_cost_nothing_but_time(tmp_path):
    if not os.path.exists(VARIANT):
        pytest.skip("libpcr_giveup.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    ref = _run(None, str(tmp_path / "ref.npz"))
    got = _run(VARIANT, str(tmp_path / "got.npz"))
    assert ref["giveup_passes"][0] == 0                    # the product library: nobody ever gives up
    assert got["giveup_passes"][0] > 0, "the variant never exercised the give-up path"
    for k in ref.files:
        if k == "giveup_passes":
            continue
        assert np.array_equal(ref[k], got[k]), k            # transforms, counts, status and the transformed source itself
    for n in (20000, 120000):
        assert got[f"{n}_compat_meta"][2] == 0 and got[f"{n}_total_meta"][2] == 0
