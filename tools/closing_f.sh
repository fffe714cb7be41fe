# tools/closing_f.sh -- after bench.py's legs were reordered (fixed-base leg before the headline): the bench-contract tests, the driver's
# command, the same command's MSM legs under rocprofv3 (kernel stats + the line printed under the profiler), the two-rank rehearsal.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r05_f
timeout -k 10 500 python3 -m pytest tests/test_bench_contract_gpu.py -x -q -m gpu > gpurun_out/${T}_contract.log 2>&1 || { tail -30 gpurun_out/${T}_contract.log; exit 1; }
echo "contract tests done"; tail -2 gpurun_out/${T}_contract.log
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail -20 gpurun_out/${T}_bench.err; exit 1; }
echo "bench done"
mkdir -p gpurun_out/${T}_h
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_h -- python3 bench.py --steps 20 --warmup 5 --no-plonk --no-ntt --no-cpu-baseline --no-batch > gpurun_out/${T}_bench_headline_under_rocprof.json 2> gpurun_out/${T}_h/err.txt || { tail -20 gpurun_out/${T}_h/err.txt; exit 1; }
cp "$(find gpurun_out/${T}_h -name '*kernel_stats.csv' | head -1)" gpurun_out/${T}_bench_headline_kernel_stats.csv
find gpurun_out/${T}_h -name '*kernel_trace.csv' -delete
echo "headline profile done"
python3 - <<P
import json
for f in ("gpurun_out/${T}_bench.json", "gpurun_out/${T}_bench_headline_under_rocprof.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "step %.3f ms" % d["ms_per_step"], "acc", d["roofline"]["avg_launch_ms"], d["phases_ms"], "| fixed-base %.3f" % d["fixed_base"]["ms_per_step"], d["config"]["leg_order"][:40])
P
bash tools/rehearse_two_ranks.sh $T
