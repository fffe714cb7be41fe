"""Where the Python host spends its time inside one 2^20-gate proof (cProfile, GPU time excluded only in so far as calls are
asynchronous):  python tools/prove_host_profile.py [log_n]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mpc_jellyfish_amd as mj
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c = mj.params.BLS12_381
cs = mj.snark.gen_circuit_for_bench(c, 1 << lg, "TurboPlonk")
rng = mj.rng.test_rng()
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), cs.n + 2)
pk = mj.snark.preprocess(ck, cs)
for _ in range(3):
    mj.snark.prove(rng, cs, pk)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    mj.snark.prove(rng, cs, pk)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
