#!/usr/bin/env python3
"""tools/pmc_collect_profiles.py <gpurun_out/TAG> <profiles/TAG> -- copies what tools/pmc_passes.sh wrote into the tracked tree:
per pass the raw counter_collection.csv and a per-kernel summary (kernel, counter, launches, mean per launch), plus srchash.txt."""
import csv, glob, os, shutil, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for p in ("p1", "p2", "p3", "q1", "q2", "q3"):
    files = glob.glob(f"{src}/{p}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    shutil.copy(files[0], f"{dst}/{p}_counter_collection.csv")
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(files[0])):
        a = acc[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    with open(f"{dst}/{p}_counters_by_kernel.csv", "w") as f:
        f.write("kernel,counter,launches,mean_per_launch\n")
        for (k, c), (tot, cnt) in sorted(acc.items()):
            f.write("%s,%s,%d,%.1f\n" % (k.replace(",", ";"), c, cnt, tot / cnt))
shutil.copy(f"{src}/srchash.txt", f"{dst}/srchash.txt")
print("ok", sorted(os.listdir(dst)))
