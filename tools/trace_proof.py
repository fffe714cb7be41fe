"""Per-proof GPU timeline from a rocprofv3 kernel trace of `mzk_prove` (or any run that proves repeatedly): cuts the trace at the
once-per-proof poly_degree_kernel, takes the last-but-one proof, and prints span / busy / idle, the largest gaps and the time per kernel.
    rocprofv3 --kernel-trace --output-format csv -d out -o t -- mpc-jellyfish_amd/mzk_prove 0 turbo 32768 6
    python tools/trace_proof.py out/t_kernel_trace.csv[.gz]"""
import collections
import csv
import gzip
import sys

path = sys.argv[1]
rows = list(csv.DictReader(gzip.open(path, "rt") if path.endswith(".gz") else open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
idx = [i for i, e in enumerate(ev) if "poly_degree_kernel" in e[2]]
print("proofs in trace: %d; ms between their degree checks: %s" % (len(idx), ["%.2f" % ((ev[b][0] - ev[a][0]) / 1e6) for a, b in zip(idx, idx[1:])]))
seg = ev[idx[-3]:idx[-2]]
span = seg[-1][1] - seg[0][0]
busy, end, gaps, prev = 0, seg[0][0], [], None
for s, e, n in seg:
    if s > end:
        gaps.append((s - end, prev, n))
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    end, prev = max(end, e), n
short = lambda n: n.split("(")[0].split("<")[0][-44:]
print("one proof: span %.3f ms, busy %.3f ms, idle %.3f ms, %d kernels" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(seg)))
pairs = collections.Counter()
for g, p, n in gaps:
    pairs[(short(p), short(n))] += g
print("idle by (kernel before, kernel after):")
for k, v in pairs.most_common(12):
    print("  %8.1f us  %s -> %s" % (v / 1e3, k[0], k[1]))
kt, kc = collections.Counter(), collections.Counter()
for s, e, n in seg:
    kt[short(n)] += e - s
    kc[short(n)] += 1
print("kernel time:")
for k, v in kt.most_common(30):
    print("  %8.3f ms %4d  %s" % (v / 1e6, kc[k], k))
