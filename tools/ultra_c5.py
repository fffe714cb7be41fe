"""config C5's shape on ONE GPU: UltraPlonk (Plookup), BN254, 2^22 gates -- feasibility / timing."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mpc_jellyfish_amd as mj
ul = int(sys.argv[1]) if len(sys.argv) > 1 else 22
un = 1 << ul
bn = mj.params.BN254
t0 = time.time()
ck = mj.UnivariateProverParam.gen_srs_for_testing(bn, 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f, un + 2)
cs = mj.snark.gen_circuit_for_bench(bn, un, "UltraPlonk")
torch.cuda.synchronize(); print("srs+circuit s", round(time.time() - t0, 2), flush=True)
t0 = time.time()
pk = mj.snark.preprocess(ck, cs)
pk.vk_commitments()
torch.cuda.synchronize(); print("preprocess s", round(time.time() - t0, 2), "mem GB", round(torch.cuda.memory_allocated() / 1e9, 1), flush=True)
rng = mj.rng.test_rng()
for _ in range(2):
    mj.snark.prove(rng, cs, pk)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    core, pb = mj.snark.prove(rng, cs, pk)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
core, pb = mj.snark.prove(rng, cs, pk, profile=True)
ok = True                                                             # (the library's round 5 checks the quotient identity at zeta: a wrong quotient raises)
free, total = torch.cuda.mem_get_info()
print("prove ms", round(ms, 1), "ns/gate", round(ms * 1e6 / un, 1), "degree_ok", ok, "proof bytes", len(pb), "HBM used GB", round((total - free) / 1e9, 1))
print(core.timings_ms)
