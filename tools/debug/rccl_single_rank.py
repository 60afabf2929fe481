"""Single-rank RCCL smoke check (world_size 1): process-group init with device_id, ncclAvg probe, bucketed reduce on a side
stream, broadcast, barrier -- the calls bench.py / run_pretraining.py make with several ranks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from mirage_amd.dp import GradReducer, broadcast_params, all_reduce_scalar_mean
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
t0 = time.time()
dist.init_process_group("nccl", device_id=dev)
x = torch.ones(1 << 20, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.AVG); dist.broadcast(x, src=0); dist.barrier()
torch.cuda.synchronize()
print("nccl backend:", dist.get_backend(), "avg ok:", bool((x == 1).all()), "init+collectives %.1f s" % (time.time() - t0))
g = torch.randn(40_000_000, device=dev)
r = GradReducer(g, bucket_bytes=64 << 20)
print("world", r.world, "use_avg", r.use_avg, "buckets", len(r.ranges))
r.reduce(); torch.cuda.synchronize()
dist.destroy_process_group()
print("OK")
