"""GPU: the C++ host layer (mpc-jellyfish_amd/host/: bench circuit, preprocess, prove, transcript, proof bytes above the C ABI,
built by g++) must emit byte for byte the proof of the Python mirror for the same circuit, SRS trapdoor and `test_rng` stream."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mpc-jellyfish_amd", "mzk_prove")


@pytest.mark.parametrize("curve_id,plonk_type,num_gates,range_bits", [(0, "TurboPlonk", 32, 8), (1, "TurboPlonk", 1 << 12, 8), (1, "UltraPlonk", 32, 3),
                                                                      (0, "UltraPlonk", 1 << 11, 8)])
def test_cpp_host_prove_matches_python_mirror(gpu, mj, curve_id, plonk_type, num_gates, range_bits):
    if not os.path.exists(BIN):                                        # normally built by __graft_entry__.build(); g++ is in the image
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mpc-jellyfish_amd", "host"), "-s"])
    assert os.path.exists(BIN)
    out = subprocess.run([BIN, str(curve_id), "ultra" if plonk_type == "UltraPlonk" else "turbo", str(num_gates), "0", str(range_bits)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    c = mj.params.CURVES[curve_id]
    cs = mj.snark.gen_circuit_for_bench(c, num_gates, plonk_type, range_bit_len=range_bits)
    rng = mj.rng.test_rng()
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), cs.n + 2)
    pk = mj.snark.preprocess(ck, cs)
    _, proof_bytes = mj.snark.prove(rng, cs, pk)
    assert got["log_n"] == cs.n.bit_length() - 1
    assert got["proof_hex"] == proof_bytes.hex()
    pk.release()
    ck.release()
