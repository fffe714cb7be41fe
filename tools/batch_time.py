"""PlonkKzgSnark::batch_prove of two 2^log_n-row TurboPlonk bench circuits vs the two proofs one after another:
python tools/batch_time.py [log_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mpc_jellyfish_amd as mj
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c = mj.params.BLS12_381
circuits = [mj.snark.gen_circuit_for_bench(c, (1 << lg) - g, "TurboPlonk") for g in (576, 76)]
rng = mj.rng.test_rng()
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), circuits[0].n + 2)
pks = [mj.snark.preprocess(ck, cs) for cs in circuits]
for _ in range(2):
    mj.snark.batch_prove(rng, circuits, pks)
    for cs, pk in zip(circuits, pks):
        mj.snark.prove(rng, cs, pk)
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mj.snark.batch_prove(rng, circuits, pks)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for cs, pk in zip(circuits, pks):
        mj.snark.prove(rng, cs, pk)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("batch of 2: %.2f ms   two single proofs: %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
