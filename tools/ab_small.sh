# tools/ab_small.sh -- same-box A/B of two builds of the library + compiled host on the small proofs (boxes of the pool differ by 3-5 %, and
# one proof in a few dozen takes a few ms longer: compare builds on ONE box, several alternating repetitions).  Put the two builds
# (libmi355zk.so + mzk_prove each) into ab_old/ and ab_new/ at the repo root (git-ignored; they travel with gpurun), then on the GPU box:
#   bash tools/ab_small.sh > gpurun_out/ab.txt
cd $GRAFT_REPO_ROOT
run() { d=$1; shift; (cd $d && ./mzk_prove "$@" 2>/dev/null | grep -o '"prove_ms": [0-9.]*'); }
for rep in 1 2 3; do
  for cfg in "1 turbo 32768 20" "1 ultra 32768 20" "1 ultra 1024 20" "0 turbo 32768 20" "0 turbo 1024 20"; do
    echo "rep $rep [curve / system / gates / reps: $cfg] old $(run ab_old $cfg) new $(run ab_new $cfg)"
  done
done
