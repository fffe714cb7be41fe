# tools/headline_split_ab.sh -- the headline step (variable base) and the fixed-base leg of bench.py under the accumulation's scheduling switches
# of csrc/msm.hip (MZK_MSM_NO_TAIL_SPLIT, MZK_MSM_TAIL_FRAC_LOG, MZK_MSM_TAIL_SPLIT, MZK_MSM_TOP_FIRST; MZK_MSM_FORCE_SPLIT=k forces every bucket
# over 2^k threads), alternating, one box.
cd $GRAFT_REPO_ROOT
A="--steps 20 --warmup 5 --no-plonk --no-ntt --no-cpu-baseline --no-batch"
get() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['phases_ms']; f=d['fixed_base']['phases_ms']; print('step', round(d['ms_per_step'],3), 'acc', p['accumulate'], 'comb', p['split_combine'], 'dev', p['device_total'], '| fixed-base step', round(d['fixed_base']['ms_per_step'],3), 'acc', f['accumulate'], 'comb', f['split_combine'], 'dev', f['device_total'])"; }
run() { echo "rep $rep [$*]: $(env "$@" python3 bench.py $A 2>/dev/null | get)"; }
for rep in 1 2 3; do
  run MZK_X=0
  run MZK_MSM_NO_TAIL_SPLIT=1
  run MZK_MSM_TOP_FIRST=1
  run MZK_MSM_TOP_FIRST=1 MZK_MSM_NO_TAIL_SPLIT=1
  run MZK_MSM_TOP_FIRST=1 MZK_MSM_FORCE_SPLIT=0
  run MZK_MSM_TOP_FIRST=1 MZK_MSM_TAIL_FRAC_LOG=4
  run MZK_MSM_TAIL_FRAC_LOG=2
  run MZK_MSM_TAIL_FRAC_LOG=4
done
