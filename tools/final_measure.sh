# tools/final_measure.sh -- the round's closing measurements on one box: the bench line, the headline's launches alone under rocprofv3
# (kernel stats + the line printed under the profiler), whole proofs under rocprofv3 (tools/prof_prove.sh), the scale model.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05_c}          # tag of the output files: gpurun_out/${T}_* -> profiles/${T}_*
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench done"
mkdir -p gpurun_out/${T}_h
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_h -- python3 bench.py --steps 20 --warmup 5 --no-plonk --no-ntt --no-cpu-baseline --no-fixed-base --no-batch > gpurun_out/${T}_bench_headline_under_rocprof.json 2> gpurun_out/${T}_h/err.txt
cp "$(find gpurun_out/${T}_h -name '*kernel_stats.csv' | head -1)" gpurun_out/${T}_bench_headline_kernel_stats.csv
find gpurun_out/${T}_h -name '*kernel_trace.csv' -delete
echo "headline profile done"
bash tools/prof_prove.sh $T > gpurun_out/${T}_prof_prove.log 2>&1
echo "prove profiles done"
python3 tools/scale_model.py > gpurun_out/${T}_scale_model.json 2> gpurun_out/${T}_scale_model.err
echo "scale model done"
tail -c 600 gpurun_out/${T}_scale_model.json
bash tools/small_proofs.sh > gpurun_out/${T}_small_proofs.txt 2>&1
echo "small proofs done"
