# tools/final_measure.sh -- the round's closing measurements on one box: the bench line, the headline's launches alone under rocprofv3
# (kernel stats + the line printed under the profiler), whole proofs under rocprofv3 (tools/prof_prove.sh), the scale model.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_b_bench.json 2> gpurun_out/r05_b_bench.err
echo "bench done"
mkdir -p gpurun_out/r05h
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05h -- python3 bench.py --steps 20 --warmup 5 --no-plonk --no-ntt --no-cpu-baseline --no-fixed-base --no-batch > gpurun_out/r05_b_bench_headline_under_rocprof.json 2> gpurun_out/r05h/err.txt
cp "$(find gpurun_out/r05h -name '*kernel_stats.csv' | head -1)" gpurun_out/r05_b_bench_headline_kernel_stats.csv
find gpurun_out/r05h -name '*kernel_trace.csv' -delete
echo "headline profile done"
bash tools/prof_prove.sh > gpurun_out/r05_prof_prove.log 2>&1
echo "prove profiles done"
python3 tools/scale_model.py > gpurun_out/r05_scale_model.json 2> gpurun_out/r05_scale_model.err
echo "scale model done"
tail -c 600 gpurun_out/r05_scale_model.json
