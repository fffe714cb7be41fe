import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
un = 1 << 20
bls = mj.params.BLS12_381
ck0 = mj.UnivariateProverParam.gen_srs_for_testing(bls, 12345, un + 2)
cs0 = mj.snark.gen_circuit_for_bench(bls, un, "TurboPlonk")
pk0 = mj.snark.preprocess(ck0, cs0)
rng = mj.rng.test_rng()
for _ in range(2):
    mj.snark.prove(rng, cs0, pk0)
pk0.release(); del cs0
c = mj.params.BN254
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, un + 2)
cs = mj.snark.gen_circuit_for_bench(c, un, "UltraPlonk")
pk = mj.snark.preprocess(ck, cs)
mj.snark.prove(rng, cs, pk)
mj.snark.prove(rng, cs, pk)
orig = mj.poly.evaluate
def timed(curve, t, x, length=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(curve, t, x, length)
    print("evaluate", tuple(t.shape), round((time.perf_counter() - t0) * 1e3, 3), "ms")
    return r
mj.prover.poly.evaluate = timed
orig_sq = mj.prover.TranscriptChallenges.after_round3
def t3(self, comms):
    t0 = time.perf_counter(); r = orig_sq(self, comms); print("after_round3", round((time.perf_counter() - t0) * 1e3, 3)); return r
mj.prover.TranscriptChallenges.after_round3 = t3
import torch.cuda
orig_idx = torch.Tensor.__getitem__
core, b = mj.snark.prove(rng, cs, pk, profile=True)
print(core.timings_ms)
print(torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9)
