# tools/ab_env.sh VAR=VALUE [reps] -- same-box A/B of one environment switch of the library on the compiled host's small proofs: the default
# build against the same build with the switch set, alternating (boxes differ by 3-5 %; one proof in a few dozen takes milliseconds longer).
#   bash tools/ab_env.sh MZK_MSM_DIGITS_PER_MSM=1 > gpurun_out/ab_env.txt
cd $GRAFT_REPO_ROOT/mpc-jellyfish_amd
SW=$1; REPS=${2:-3}
ms() { "$@" 2>/dev/null | grep -o '"prove_ms": [0-9.]*' | grep -o '[0-9.]*$'; }
for rep in $(seq 1 $REPS); do
  for cfg in "0 turbo 32768 30" "0 turbo 1024 30" "1 ultra 32768 30" "0 turbo 131072 20"; do
    echo "rep $rep [curve / system / gates / proofs: $cfg] default $(ms ./mzk_prove $cfg) ms, $SW $(ms env $SW ./mzk_prove $cfg) ms"
  done
done
