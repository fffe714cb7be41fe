#!/bin/bash
# tools/ntt_prof.sh [tag] -- rocprofv3 passes over tools/ntt_prof.py (run on the GPU box from the repo root)
export TMPDIR=/tmp; R=$PWD; TAG=${1:-nttprof}
mkdir -p gpurun_out/$TAG
rocprofv3 -L > gpurun_out/$TAG/counters.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 tools/ntt_prof.py > gpurun_out/$TAG/trace.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/$TAG/p1 -- python3 tools/ntt_prof.py > gpurun_out/$TAG/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/$TAG/p2 -- python3 tools/ntt_prof.py > gpurun_out/$TAG/p2.log 2>&1
find gpurun_out/$TAG -name "*.csv" | head -20
tail -3 gpurun_out/$TAG/p2.log
