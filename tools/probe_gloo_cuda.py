import os, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29871", RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    t = torch.full((5,), rank + 1, dtype=torch.uint8, device=dev)
    out = torch.empty(world * 5, dtype=torch.uint8, device=dev)
    try:
        dist.all_gather_into_tensor(out, t)
        print(rank, "all_gather_into_tensor cuda ok", out.cpu().tolist(), flush=True)
    except Exception as e:
        print(rank, "all_gather_into_tensor cuda FAILED", repr(e)[:200], flush=True)
    x = torch.tensor([float(rank)], dtype=torch.float64, device=dev)
    try:
        dist.all_reduce(x, op=dist.ReduceOp.MAX); print(rank, "all_reduce cuda ok", x.item(), flush=True)
    except Exception as e:
        print(rank, "all_reduce cuda FAILED", repr(e)[:200], flush=True)
    dist.barrier()
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2,), nprocs=2, join=True)
