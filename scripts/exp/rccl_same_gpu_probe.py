"""Can two RCCL ranks share ONE GPU on this stack?  (No: RCCL refuses duplicate devices - recorded for DESIGN.md.)"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        x = torch.full((1024,), float(rank + 1), device="cuda:0")
        dist.all_reduce(x)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce over two ranks on one GPU -> {x[0].item()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print(f"rank {rank}: {type(e).__name__}: {str(e)[:300]}", flush=True)


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29611), nprocs=2, join=True)
