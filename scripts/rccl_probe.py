"""Can RCCL (backend "nccl") run on THIS box at all?  world 1 on cuda:0, then two ranks sharing cuda:0 (RCCL normally refuses a
duplicate GPU in one communicator).  usage: python scripts/rccl_probe.py  |  torchrun --nproc-per-node 2 scripts/rccl_probe.py"""
import os, sys, time
import torch, torch.distributed as dist
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world)
    x = torch.full((1 << 20,), float(rank + 1), device="cuda:0")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    g = dist.new_group()
    y = torch.ones(1024, device="cuda:0"); dist.all_reduce(y, group=g); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    print(f"rank {rank}/{world}: nccl all_reduce ok, x[0]={float(x[0]):.1f} y[0]={float(y[0]):.1f}, {(time.perf_counter() - t0) / 20 * 1e6:.0f} us per 4 MB all-reduce", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}/{world}: FAILED {type(e).__name__}: {str(e)[:400]}", flush=True)
