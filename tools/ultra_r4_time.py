import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
c = mj.params.BN254
un = 1 << int(sys.argv[1])
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, un + 2)
cs = mj.snark.gen_circuit_for_bench(c, un, "UltraPlonk")
pk = mj.snark.preprocess(ck, cs)
rng = mj.rng.test_rng()
mj.snark.prove(rng, cs, pk)
orig = mj.poly.evaluate
def timed(curve, t, x, length=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(curve, t, x, length)
    print("evaluate", tuple(t.shape), round((time.perf_counter() - t0) * 1e3, 3), "ms")
    return r
mj.poly.evaluate = timed
mj.prover.poly.evaluate = timed
orig_sq = mj.prover.TranscriptChallenges.after_round3
def t3(self, comms):
    t0 = time.perf_counter(); r = orig_sq(self, comms); print("after_round3", round((time.perf_counter() - t0) * 1e3, 3)); return r
mj.prover.TranscriptChallenges.after_round3 = t3
core, b = mj.snark.prove(rng, cs, pk, profile=True)
print(core.timings_ms)
