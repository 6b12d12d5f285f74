"""Per-launch durations of grid_pass_kernel from a rocprofv3 --kernel-trace CSV (argv[1]): the trend over the passes of each ICP call.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/ab_pass.py ; python3 scripts/pass_trace.py DIR/**/*kernel_trace.csv"""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("grid_pass_kernel")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
s = np.array([int(r["Start_Timestamp"]) for r in rows]); e = np.array([int(r["End_Timestamp"]) for r in rows])
gap = (s[1:] - e[:-1]) / 1e3
it = int(sys.argv[2]) if len(sys.argv) > 2 else 25
n = len(d) // it
D = d[: n * it].reshape(n, it)
print("launches", len(d), "calls", n, "passes per call", it)
print("median duration by pass (us):", " ".join("%.1f" % v for v in np.median(D[2:], axis=0)))
g = np.concatenate([gap, [0]])[: n * it].reshape(n, it)
print("median gap to the next launch by pass (us):", " ".join("%.1f" % v for v in np.median(g[2:], axis=0)))
print("whole: median duration %.2f  mean %.2f | median gap %.2f" % (np.median(d), d.mean(), np.median(gap)))
