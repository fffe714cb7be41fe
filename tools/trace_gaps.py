"""GPU idle time inside proofs: python tools/trace_gaps.py <kernel_trace.csv> -- sums the gaps between consecutive kernels
(on the whole device) over the last 60 % of the trace (the timed repetitions), and lists the largest gaps with their neighbours."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows))
t0, t1 = ev[0][0], ev[-1][1]
cut = t0 + int((t1 - t0) * 0.4)
ev = [e for e in ev if e[0] >= cut]
busy, gaps, end = 0, [], ev[0][0]
for s, e, name in ev:
    if s > end:
        gaps.append((s - end, name))
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    end = max(end, e)
span = ev[-1][1] - ev[0][0]
print("span %.2f ms, busy %.2f ms (%.1f %%), idle %.2f ms in %d gaps" % (span / 1e6, busy / 1e6, 100 * busy / span, (span - busy) / 1e6, len(gaps)))
hist = {}
for g, name in gaps:
    hist.setdefault(name, [0, 0])
    hist[name][0] += g
    hist[name][1] += 1
for name, (tot, cnt) in sorted(hist.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  idle before %-60s %8.3f ms in %4d gaps" % (name, tot / 1e6, cnt))
