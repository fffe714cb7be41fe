"""Timeline of the LAST `span_ms` milliseconds of a rocprofv3 --kernel-trace csv: start offset, duration, queue, kernel (short name).
python tools/trace_timeline.py <kernel_trace.csv> [span_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
span = float(sys.argv[2]) if len(sys.argv) > 2 else 4.5
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows))
end = max(k[1] for k in ks)
t0 = end - int(span * 1e6)
for s, e, q, nm in ks:
    if s < t0: continue
    nm = nm.split("(")[0].replace("void mzk::", "").replace("mzk::", "")
    print("%9.1f us  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, nm[:70]))
