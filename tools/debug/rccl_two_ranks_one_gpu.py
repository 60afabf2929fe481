#!/usr/bin/env python3
"""Two torch.distributed ranks over the nccl (= RCCL) backend on ONE GPU: does this pool's RCCL accept two ranks per device?
Each rank all-reduces a small tensor and runs GradReducer over a flat buffer in both exchange formats.  Exit code 0 = works,
3 = RCCL refused the communicator (message on stderr), anything else = a real failure.  Launched by tests/test_fullsize_gpu.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.full((1024,), float(rank + 1), device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
except Exception as exc:          # RCCL: "Duplicate GPU detected" / invalid usage when two ranks name the same device
    print(f"RCCL refused two ranks on one device: {type(exc).__name__}: {str(exc)[:400]}", file=sys.stderr)
    sys.exit(3)
assert float(t[0]) == 3.0, float(t[0])
from mirage_amd.dp import GradReducer
for comm in ("allreduce", "rs_ag"):
    os.environ["MRG_DP_COMM"] = comm
    g = torch.full((1 << 20,), float(rank), device="cuda")
    r = GradReducer(g, bucket_bytes=1 << 20)
    r.reduce()
    torch.cuda.synchronize()
    assert float(g.min()) == float(g.max()) == 0.5, (comm, float(g.min()), float(g.max()))
dist.barrier()
dist.destroy_process_group()
print(f"rank {rank}: RCCL two-ranks-one-GPU ok")
