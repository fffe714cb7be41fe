#!/usr/bin/env python3
"""tools/collective_time.py [--ranks 2] [--k 5] [--reps 2000] -- the collective path of ONE commit group (sharding.gather_partials + the
local EC sums: what `ShardedCommitter.commit_jacobian` does after its MSMs), timed between `ranks` gloo processes on this host.
tools/scale_model.py charged 40 us per small collective as an assumed constant (VERDICT r4 #6b); this is the measured figure of the
HOST side of that path -- tensor set-up, all_gather_into_tensor of k x 144 B per rank, one transfer back, k x (ranks - 1) Jacobian
additions through mzk_g1_sum_jacobian.  Over RCCL / xGMI the wire time differs (no multi-GPU node is available to measure it) but the
host side is the same.  Prints one JSON line (rank 0).  Needs no GPU."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, k, reps, port, q):
    import numpy as np
    import torch.distributed as dist
    import mpc_jellyfish_amd as mj
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = mj.params.BLS12_381
    # k valid Jacobian points per rank: multiples of the generator (host arithmetic only)
    from importlib import import_module
    sh = import_module("mpc-jellyfish_amd.sharding")
    gx, gy = mj.params.fq_to_mont(c, [c.gx]), mj.params.fq_to_mont(c, [c.gy])
    one = mj.params.fq_to_mont(c, [1])
    pt = np.concatenate([gx, gy, one]).reshape(1, 3, c.fq_limbs)
    part = np.repeat(pt, k, axis=0)
    for _ in range(50):
        st = sh.gather_partials(part)
        [sh.sum_jacobian(c, st[:, i]) for i in range(k)]
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        st = sh.gather_partials(part)
        out = [sh.sum_jacobian(c, st[:, i]) for i in range(k)]
    el = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    for _ in range(reps):
        out = [sh.sum_jacobian(c, st[:, i]) for i in range(k)]
    sum_us = (time.perf_counter() - t0) / reps * 1e6
    dist.barrier()
    if rank == 0:
        q.put({"ranks": world, "k": k, "bytes_per_rank": int(part.nbytes), "commit_group_collective_us": round(el, 1), "of_which_host_ec_sums_us": round(sum_us, 1),
               "backend": "gloo (loopback, this host's CPU)", "reps": reps})
    dist.destroy_process_group()


if __name__ == "__main__":
    import multiprocessing as mp
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--reps", type=int, default=2000)
    ap.add_argument("--port", type=int, default=29671)
    a = ap.parse_args()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, a.ranks, a.k, a.reps, a.port, q)) for r in range(a.ranks)]
    for p in ps:
        p.start()
    res = q.get(timeout=300)
    for p in ps:
        p.join()
    print(json.dumps(res))
