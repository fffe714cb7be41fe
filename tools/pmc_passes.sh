#!/bin/bash
# tools/pmc_passes.sh -- PMC counters for the bench kernels, one counter group per pass
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains beside --pmc).
# Run on the GPU box from the repo root:  bash tools/pmc_passes.sh [tag]
# p1..p3: the headline (variable-base) + the fixed_base leg + the NTT in one run (kernels shared by both MSM paths -- fold, order, long --
# are averaged over both there); q1..q3: MZK_BENCH_TABLE=0, the variable-base headline alone.
export TMPDIR=/tmp; R=$PWD; TAG=${1:-pmc}
mkdir -p gpurun_out/$TAG
python3 tools/srchash.py > gpurun_out/$TAG/srchash.txt      # the kernel sources these counters are collected on
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-plonk --no-batch"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/$TAG/p1 -- python3 bench.py $ARGS > gpurun_out/$TAG/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/p2 -- python3 bench.py $ARGS > gpurun_out/$TAG/p2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/p3 -- python3 bench.py $ARGS > gpurun_out/$TAG/p3.log 2>&1 &&
export MZK_BENCH_TABLE=0 &&
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/$TAG/q1 -- python3 bench.py $ARGS --no-ntt > gpurun_out/$TAG/q1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/q2 -- python3 bench.py $ARGS --no-ntt > gpurun_out/$TAG/q2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/q3 -- python3 bench.py $ARGS --no-ntt > gpurun_out/$TAG/q3.log 2>&1 &&
unset MZK_BENCH_TABLE &&
find gpurun_out/$TAG -name "*.csv" | wc -l
