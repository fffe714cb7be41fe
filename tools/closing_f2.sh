# tools/closing_f2.sh -- the final tree on a fresh box: every -m gpu test, smoke(), a fifth soak run (new seed).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 800 python3 -m pytest tests -x -q -m gpu > gpurun_out/r05_f_gputests.log 2>&1 || { tail -30 gpurun_out/r05_f_gputests.log; exit 1; }
tail -2 gpurun_out/r05_f_gputests.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_f_smoke.log 2>&1 || { tail -20 gpurun_out/r05_f_smoke.log; exit 1; }
tail -1 gpurun_out/r05_f_smoke.log
timeout -k 10 600 python3 tools/soak.py --seconds 400 --seed 5 --max-log-n 19 --out gpurun_out/soak_f.json > gpurun_out/soak_f.log 2>&1 || { tail -20 gpurun_out/soak_f.log; exit 1; }
tail -3 gpurun_out/soak_f.log
