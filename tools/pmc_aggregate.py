#!/usr/bin/env python3
"""tools/pmc_aggregate.py <gpurun_out/TAG> <profiles/OUT.json> -- per-kernel means of the PMC passes written by
tools/pmc_passes.sh (p1: SQ group, p2: FETCH_SIZE, p3: WRITE_SIZE on the fixed-base table path; q1..q3 the same with the table
off; one counter group per pass as MI355X_MICROARCH.md prescribes).  HBM bytes per launch = FETCH_SIZE + WRITE_SIZE (KB -> bytes);
for wide coalesced streaming reads gfx950's FETCH_SIZE reports half the bytes (the guide's correction), applied to the NTT passes
only and stated in the output.  The output carries the hash of the kernel sources it was collected on (tools/srchash.py): bench.py
refuses to quote it once they change."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import srchash  # noqa: E402

tag_dir, out_path = sys.argv[1], sys.argv[2]


def collect(prefixes):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for p in prefixes:
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
    return kernels


def hbm(k, double_fetch):
    return int((k.get("FETCH_SIZE_KB_raw", 0) * (2 if double_fetch else 1) + k.get("WRITE_SIZE_KB", 0)) * 1024)


# the hash written on the GPU box when the counters were collected (tools/pmc_passes.sh); an older run without it is stamped
# with the tree as it stands, which is only right if nothing changed in between
try:
    w = open(os.path.join(tag_dir, "srchash.txt")).read().split()
    stamp = {w[0]: w[1], w[2]: w[3]}
except Exception:
    stamp = {"msm": srchash.sha16(srchash.MSM_SOURCES), "ntt": srchash.sha16(srchash.NTT_SOURCES)}
table, plain = collect(("p1", "p2", "p3")), collect(("q1", "q2", "q3"))
acc_name = next(k for k in table if "msm_accumulate_kernel" in k)
ntt_name = next(k for k in table if "nttx_pass_kernel" in k)
out = {"source": "tools/pmc_passes.sh + tools/pmc_aggregate.py: rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 bench.py "
                 "--steps 2 --warmup 1 --no-cpu-baseline --no-plonk --no-batch; separate passes (SQ group, FETCH_SIZE, WRITE_SIZE), "
                 "p* = headline (variable base) + fixed_base leg + NTT, q* with MZK_BENCH_TABLE=0 = the variable-base headline alone",
       "source_sha16": stamp,
       "units": "FETCH_SIZE / WRITE_SIZE in KB per launch (mean over launches).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the "
                "bytes of a wide coalesced streaming read (NTT passes: 16 B/lane) -- doubled below for the NTT; the MSM gather (16-B loads at "
                "random rows of the SRS / its table) is uncalibrated and taken as reported.",
       "msm_accumulate_kernel": acc_name, "msm_accumulate_hbm_bytes_per_launch": hbm(table[acc_name], False),
       "msm_accumulate_SQ_INSTS_VALU": table[acc_name].get("SQ_INSTS_VALU"),
       "ntt_pass_kernel": ntt_name, "ntt_pass_hbm_bytes_per_launch": hbm(table[ntt_name], True),
       "ntt_pass_SQ_INSTS_VALU": table[ntt_name].get("SQ_INSTS_VALU"), "kernels": table}
# table off: 16 windows of 2^15 buckets at 2^20 pairs -- mean load 32: whole-bucket threads for the first 7/8 of the ranks, the last eighth
# split four ways (msm_accumulate_split_kernel<EC, false> + msm_split_combine_kernel); the accumulation launch is the one quoted
pname = next((k for k in plain if "msm_accumulate" in k), None)
if pname:
    out.update({"msm_accumulate_plain_kernel": pname, "msm_accumulate_plain_hbm_bytes_per_launch": hbm(plain[pname], False),
                "msm_accumulate_plain_SQ_INSTS_VALU": plain[pname].get("SQ_INSTS_VALU"), "kernels_table_off": plain})
json.dump(out, open(out_path, "w"), indent=1)
print(out_path, out["msm_accumulate_hbm_bytes_per_launch"], out.get("msm_accumulate_plain_hbm_bytes_per_launch"), out["ntt_pass_hbm_bytes_per_launch"])
