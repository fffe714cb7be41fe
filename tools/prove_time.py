"""Wall time of PlonkKzgSnark::prove through the Python mirror: python tools/prove_time.py [log_n] [reps] [curve_id] [turbo|ultra]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mpc_jellyfish_amd as mj
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
c = mj.params.CURVES[int(sys.argv[3]) if len(sys.argv) > 3 else 0]
kind = "UltraPlonk" if len(sys.argv) > 4 and sys.argv[4] == "ultra" else "TurboPlonk"
cs = mj.snark.gen_circuit_for_bench(c, 1 << lg, kind)
rng = mj.rng.test_rng()
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), cs.n + 2)
pk = mj.snark.preprocess(ck, cs)
for _ in range(3):
    mj.snark.prove(rng, cs, pk)
torch.cuda.synchronize()
if os.environ.get("MZK_GC_FREEZE", "1") == "1":      # the interpreter's first full collection (~40 ms) otherwise lands in the 13th proof
    import gc
    gc.collect()
    gc.freeze()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    mj.snark.prove(rng, cs, pk)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
core, _ = mj.snark.prove(rng, cs, pk, profile=True)
print(kind, c.name, "2^%d" % lg, "prove ms: min %.2f median %.2f max %.2f" % (min(ts), sorted(ts)[len(ts) // 2], max(ts)))
print(dict(core.timings_ms))
print("all reps:", [round(t, 1) for t in ts])
