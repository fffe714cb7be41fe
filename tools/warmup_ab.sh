# tools/warmup_ab.sh -- does the number of untimed warm-up steps move the headline?  Same box, alternating; headline only.
cd $GRAFT_REPO_ROOT
F="--no-plonk --no-ntt --no-cpu-baseline --no-fixed-base --no-batch"
for rep in 1 2 3; do
  for wk in "2 10" "5 20" "50 20" "200 20"; do
    set -- $wk
    python3 bench.py --warmup $1 --steps $2 $F 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep warmup $1 steps $2: step %.3f ms acc %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']), d['phases_ms'])"
  done
done
