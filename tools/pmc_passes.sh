#!/bin/bash
# tools/pmc_passes.sh -- PMC counters for the bench kernels, one counter group per pass
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains beside --pmc).
# Run on the GPU box from the repo root:  bash tools/pmc_passes.sh [tag]
export TMPDIR=/tmp; R=$PWD; TAG=${1:-pmc}
mkdir -p gpurun_out/$TAG
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-plonk"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/p1 -- python3 bench.py $ARGS > gpurun_out/$TAG/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/p2 -- python3 bench.py $ARGS > gpurun_out/$TAG/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/p3 -- python3 bench.py $ARGS > gpurun_out/$TAG/p3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/stats.json 2> gpurun_out/$TAG/stats.log
# bench.py spawns mzk_prove (the C++ host) for its prove legs: the stats directory holds one *_kernel_stats.csv per process, the
# first (lowest pid) is bench.py itself
find gpurun_out/$TAG -name "*.csv" | wc -l
