cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05_c}
mkdir -p gpurun_out/${T}_p20 gpurun_out/${T}_p15
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_p20 -- ./mpc-jellyfish_amd/mzk_prove 0 turbo 1048576 10 > gpurun_out/${T}_p20/line.json 2> gpurun_out/${T}_p20/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_p15 -- ./mpc-jellyfish_amd/mzk_prove 0 turbo 32768 20 > gpurun_out/${T}_p15/line.json 2> gpurun_out/${T}_p15/err.txt
for d in ${T}_p20 ${T}_p15; do
  t=$(find gpurun_out/$d -name '*kernel_trace.csv' | head -1)
  s=$(find gpurun_out/$d -name '*kernel_stats.csv' | head -1)
  cp "$s" gpurun_out/${d}_kernel_stats.csv
  python tools/trace_gaps.py "$t" > gpurun_out/${d}_gaps.txt
  python tools/trace_proof.py "$t" > gpurun_out/${d}_proof.txt 2>&1 || true
  cut -c1-400 gpurun_out/$d/line.json > gpurun_out/${d}_line_head.txt
  rm -f "$t"   # large
done
find gpurun_out/${T}_p20 gpurun_out/${T}_p15 -name '*.csv' -size +2M -delete
cat gpurun_out/${T}_p20_gaps.txt gpurun_out/${T}_p15_gaps.txt
