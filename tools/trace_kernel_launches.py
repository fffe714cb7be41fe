"""Every launch of the kernels whose name contains PATTERN inside ONE proof of a rocprofv3 kernel trace of `mzk_prove` (cut as
tools/trace_proof.py cuts it): start offset, duration, grid.
    python tools/trace_kernel_launches.py out/t_kernel_trace.csv nttx_pass"""
import csv
import gzip
import sys

path, pat = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(gzip.open(path, "rt") if path.endswith(".gz") else open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])),
              int(r.get("Grid_Size_Y", 1) or 1)) for r in rows))
idx = [i for i, e in enumerate(ev) if "poly_degree_kernel" in e[2]]
seg = ev[idx[-3]:idx[-2]]
t0 = seg[0][0]
tot = 0
for s, e, n, gx, gy in seg:
    if pat in n:
        tot += e - s
        print("%9.3f ms  %8.1f us  grid %6d x %d  %s" % ((s - t0) / 1e6, (e - s) / 1e3, gx, gy, n.split("(")[0][-60:]))
print("total %.3f ms" % (tot / 1e6))
