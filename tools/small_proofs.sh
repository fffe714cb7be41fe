# tools/small_proofs.sh -- the compiled host's proofs at the small sizes (2^10 .. 2^18 gates; the reference publishes 2^15: bench.rs:26), with the
# round-5 latency paths on (default) and off, one JSON line each cut to its timings.  Run from the repo root on the GPU box.
cd $GRAFT_REPO_ROOT/mpc-jellyfish_amd
cut_line() { cut -c1-900 | sed 's/.*"lagrange_round1"/"lagrange_round1"/; s/, "vk_hex.*//'; }
for g in 1024 8192 32768 131072 262144; do
  echo "== TurboPlonk / BLS12-381, $g gates (default)"; ./mzk_prove 0 turbo $g 20 2>/dev/null | cut_line
  echo "== ... every class its own launches (MZK_QUOTIENT_NO_CLASS_BATCH=1)"; MZK_QUOTIENT_NO_CLASS_BATCH=1 ./mzk_prove 0 turbo $g 20 2>/dev/null | cut_line
  echo "== ... round 1 over the Lagrange basis (--lagrange)"; ./mzk_prove 0 turbo $g 20 8 --lagrange 2>/dev/null | cut_line
done
for g in 1024 32768; do
  echo "== UltraPlonk / BN254, $g gates (default)"; ./mzk_prove 1 ultra $g 20 2>/dev/null | cut_line
  echo "== ... every class its own launches"; MZK_QUOTIENT_NO_CLASS_BATCH=1 ./mzk_prove 1 ultra $g 20 2>/dev/null | cut_line
done
echo "== UltraPlonk / BLS12-381, 32768 gates (default)"; ./mzk_prove 0 ultra 32768 20 2>/dev/null | cut_line
echo "== TurboPlonk / BN254, 32768 gates (default)"; ./mzk_prove 1 turbo 32768 20 2>/dev/null | cut_line
