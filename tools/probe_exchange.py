"""tools/probe_exchange.py -- two gloo ranks on ONE card, sharding.TorchComm with CUDA payloads (the RCCL form): every class exchange is
done on the device path and compared with the host path's result; prints where they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import ctypes as C
    import numpy as np
    import torch
    import torch.distributed as dist
    import mpc_jellyfish_amd as mj
    from importlib import import_module
    _lib = import_module("mpc-jellyfish_amd.lib")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = mj.params.CURVES[0]
    cs = mj.snark.gen_circuit_for_bench(c, 1 << 12, "TurboPlonk")
    rng = mj.rng.test_rng()
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), cs.n + 2)
    comm = mj.sharding.TorchComm(device=torch.device("cuda", 0))
    inner = comm._exchange.__func__

    def spy(self, ctx, d_rem, class_bytes, first_own, n_own, n_classes):
        L = _lib.load()
        total = n_classes * class_bytes
        per = -(-n_classes // self.world)
        before = np.empty(total, dtype=np.uint8)
        _lib.check(L.mzk_dev_download(C.c_void_p(before.ctypes.data), C.c_void_p(d_rem), total), "dl")
        mine = torch.zeros(per * class_bytes, dtype=torch.uint8)
        mine[:n_own * class_bytes] = torch.from_numpy(before[first_own * class_bytes:(first_own + n_own) * class_bytes].copy())
        want = torch.empty(self.world * per * class_bytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(want, mine, group=self.group)
        want = want.numpy()[:total]
        rc = inner(self, ctx, d_rem, class_bytes, first_own, n_own, n_classes)
        after = np.empty(total, dtype=np.uint8)
        _lib.check(L.mzk_dev_download(C.c_void_p(after.ctypes.data), C.c_void_p(d_rem), total), "dl")
        bad = np.nonzero(after != want)[0]
        print(f"rank {self.rank}: exchange rc={rc} class_bytes={class_bytes} first={first_own} n_own={n_own} n_classes={n_classes} per={per}; "
              f"mismatching bytes {bad.size}" + (f" first at {bad[0]} (class {bad[0] // class_bytes}) last at {bad[-1]}" if bad.size else ""), flush=True)
        return rc

    inner_ag = comm._all_gather.__func__
    calls = [0]

    def spy_ag(self, ctx, send, nbytes, recv):
        src = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone() if nbytes else torch.empty(0, dtype=torch.uint8)
        want = torch.empty(self.world * nbytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(want, src, group=self.group)
        rc = inner_ag(self, ctx, send, nbytes, recv)
        got = np.frombuffer((C.c_uint8 * (self.world * nbytes)).from_address(recv), dtype=np.uint8) if nbytes else np.empty(0, np.uint8)
        bad = np.nonzero(got != want.numpy())[0]
        calls[0] += 1
        print(f"rank {self.rank}: all_gather #{calls[0]} rc={rc} nbytes={nbytes} mismatching {bad.size}" + (f" first at {bad[0]}" if bad.size else ""), flush=True)
        return rc

    AG = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
    cb_ag = AG(lambda *a: spy_ag(comm, *a))
    comm._struct.all_gather = cb_ag
    EX = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32)
    cb = EX(lambda *a: spy(comm, *a))
    comm._cb = (cb_ag, comm._cb[1], cb)
    comm._struct.exchange_classes = cb
    pk = mj.snark.preprocess(ck, cs, comm=comm)
    g1 = mj.rng.test_rng()
    mj.rng.fr_rand(c, g1)
    try:
        _, proof_bytes = mj.snark.prove(g1, cs, pk)
        print(f"rank {rank}: proof ok, {len(proof_bytes)} bytes", flush=True)
    except Exception as e:
        print(f"rank {rank}: prove failed: {str(e)[:150]}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, 29951), nprocs=2, join=True)
