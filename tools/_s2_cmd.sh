python -m pytest tests/test_poly_gpu.py tests/test_golden_proofs_gpu.py tests/test_native_prover_gpu.py tests/test_plonk_gpu.py tests/test_verifier_gpu.py -m gpu -x -q 2>&1 | tail -3
cd mpc-jellyfish_amd
for g in 1024 32768 1048576; do echo "== gates $g"; ./mzk_prove 0 turbo $g 20 2>/dev/null | cut -c1-560 | sed 's/.*"prove_ms"/"prove_ms"/'; done
echo "== ultra 32768"; ./mzk_prove 1 ultra 32768 20 2>/dev/null | cut -c1-700 | sed 's/.*"prove_ms"/"prove_ms"/'
