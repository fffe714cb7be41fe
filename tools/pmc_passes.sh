export TMPDIR=/tmp; R=$PWD
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc/p1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/p2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/p3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/p3.log 2>&1
find gpurun_out/pmc -name "*.csv" | head -20
