#!/usr/bin/env python
"""Can two processes on this box map each other's device buffers (hipIpc through torch's CUDA-tensor sharing)?  Round-5 probe
for the one-shot exchange of SURVEY section 8 e.  Rank r fills its buffer with r + 1, exports it, maps the peer's, reads it,
writes a flag INTO the peer's buffer, and checks that the peer's write arrived in its own."""
import os, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.multiprocessing.reductions import reduce_tensor


def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = torch.full((1024,), float(rank + 1), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    fn, args = reduce_tensor(mine)
    gathered = [None] * world
    dist.all_gather_object(gathered, (fn, args))
    peers = []
    for r, (f, a) in enumerate(gathered):
        peers.append(mine if r == rank else f(*a))
    dist.barrier()
    seen = [float(p[0].item()) for p in peers]
    for r, p in enumerate(peers):
        if r != rank:
            p[100 + rank] = 1000.0 + rank            # write into the peer's buffer
    torch.cuda.synchronize()
    dist.barrier()
    got = [float(mine[100 + r].item()) for r in range(world) if r != rank]
    print("rank %d: peers' first elements %s; peers' writes into mine %s" % (rank, seen, got), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(int(sys.argv[1]) if len(sys.argv) > 1 else 2, 29533), nprocs=int(sys.argv[1]) if len(sys.argv) > 1 else 2, join=True)
