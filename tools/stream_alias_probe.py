"""tools/stream_alias_probe.py <dense_only|bench_then_dense|bench_batch_then_dense> -- proofs after / without a batch of five MSMs in the same process:\nthe probe behind profiles/r05_stream_queue_aliasing.txt (a prover stream sharing a hardware queue with a sort stream)."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import mpc_jellyfish_amd as mj
from importlib import import_module
mlib = import_module("mpc-jellyfish_amd.lib")
L = mlib.init(0)
c = mj.params.BLS12_381
n = 1 << 20
def timed(cs, pk, rng, k=6):
    for _ in range(3): mj.snark.prove(rng, cs, pk)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): mj.snark.prove(rng, cs, pk)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / k * 1e3
    core, _ = mj.snark.prove(rng, cs, pk, profile=True)
    return round(ms, 2), {k_: round(v, 2) for k_, v in core.timings_ms.items()}
rng = mj.rng.test_rng()
ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), n + 2)
mode = sys.argv[1]
if mode.startswith("extra_streams"):                     # a process that already holds other streams (host-pointer I/O slots, torch side streams ..)
    sc = torch.from_numpy(mj.params.random_fr_mont(c, n, seed=1).view(np.int64)).cuda()
    for _ in range(2): mj.msm_bigint_batch(ck, [sc] * 5, scalars_are_mont=True)      # the two sort streams exist
    extra = [torch.cuda.Stream() for _ in range(int(mode.split(":")[1]))]
    for st_ in extra:
        with torch.cuda.stream(st_):
            torch.zeros(8, device="cuda")
    torch.cuda.synchronize()
if mode in ("bench_then_dense", "bench_batch_then_dense"):
    if mode == "bench_batch_then_dense":
        sc = torch.from_numpy(mj.params.random_fr_mont(c, n, seed=1).view(np.int64)).cuda()
        for _ in range(3): mj.msm_bigint_batch(ck, [sc] * 5, scalars_are_mont=True)
    cs = mj.snark.gen_circuit_for_bench(c, n, "TurboPlonk")
    pk = mj.snark.preprocess(ck, cs)
    print(mode, "bench circuit", timed(cs, pk, rng), flush=True)
    pk.release(); del cs
dense = mj.snark.gen_circuit_for_bench(c, n, "TurboPlonk", dense_seed=77)
pk = mj.snark.preprocess(ck, dense)
print(mode, "dense", timed(dense, pk, rng), flush=True)
