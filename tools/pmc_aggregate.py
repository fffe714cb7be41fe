#!/usr/bin/env python3
"""tools/pmc_aggregate.py <gpurun_out/TAG> <profiles/OUT.json> -- per-kernel means of the PMC passes written by
tools/pmc_passes.sh (p1: SQ group, p2: FETCH_SIZE, p3: WRITE_SIZE; one counter group per pass as MI355X_MICROARCH.md
prescribes).  HBM bytes per launch = FETCH_SIZE + WRITE_SIZE (KB -> bytes); for wide coalesced streaming reads gfx950's
FETCH_SIZE reports half the bytes (the guide's correction), applied to the NTT passes only and stated in the output."""
import csv
import glob
import json
import sys
from collections import defaultdict

tag_dir, out_path = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for p in ("p1", "p2", "p3"):
    for f in glob.glob(f"{tag_dir}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc[name][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
kernels = {}
for name, ctrs in sorted(acc.items()):
    if not name.startswith("mzk::"):
        continue
    d = {"launches": max(v[1] for v in ctrs.values())}
    for cn, (tot, cnt) in sorted(ctrs.items()):
        key = {"FETCH_SIZE": "FETCH_SIZE_KB_raw", "WRITE_SIZE": "WRITE_SIZE_KB"}.get(cn, cn)
        d[key] = round(tot / cnt, 1)
    kernels[name] = d


def hbm(name, double_fetch):
    k = kernels[name]
    return int((k.get("FETCH_SIZE_KB_raw", 0) * (2 if double_fetch else 1) + k.get("WRITE_SIZE_KB", 0)) * 1024)


acc_name = next(k for k in kernels if "msm_accumulate_kernel" in k)
ntt_name = next(k for k in kernels if "nttx_pass_kernel" in k)
out = {"source": "tools/pmc_passes.sh + tools/pmc_aggregate.py: rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 bench.py "
                 "--steps 2 --warmup 1 --no-cpu-baseline --no-plonk; three separate passes (SQ group, FETCH_SIZE, WRITE_SIZE)",
       "units": "FETCH_SIZE / WRITE_SIZE in KB per launch (mean over launches).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the "
                "bytes of a wide coalesced streaming read (NTT passes: 16 B/lane) -- doubled below for the NTT; the MSM gather (16-B loads at "
                "random rows of the precomputed table) is uncalibrated and taken as reported.",
       "msm_accumulate_kernel": acc_name, "msm_accumulate_hbm_bytes_per_launch": hbm(acc_name, False),
       "ntt_pass_kernel": ntt_name, "ntt_pass_hbm_bytes_per_launch": hbm(ntt_name, True), "kernels": kernels}
json.dump(out, open(out_path, "w"), indent=1)
print(out_path, out["msm_accumulate_hbm_bytes_per_launch"], out["ntt_pass_hbm_bytes_per_launch"])
