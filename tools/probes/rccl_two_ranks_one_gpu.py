"""Probe: can two RCCL ranks share the one GPU of a development box? (Communication libraries usually refuse duplicate
devices; if this works, the data-parallel path can be exercised with a real exchange between ranks here.)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29591 tools/probes/rccl_two_ranks_one_gpu.py"""
import os

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    import datetime
    dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=40),
                            device_id=torch.device("cuda", 0))
    t = torch.full((1024,), float(rank + 1), device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce ->", float(t[0]), flush=True)
    dist.destroy_process_group()
except Exception as exc:  # noqa: BLE001
    print("rank", rank, "failed:", type(exc).__name__, str(exc)[:300], flush=True)
