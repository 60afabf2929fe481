"""Phase timestamps of the attention FORWARD kernel (debug build of attention.hip with -DMRG_ATTN_TS; the timestamps overwrite the lse
rows): staging of K / V, the query tiles of wave 0, the last-token tail, per workgroup; and when workgroups start (rounds).
usage: attn_fwd_ts.py [B H N hd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mirage_amd import kernels as K
dev = "cuda:0"; BF16 = torch.bfloat16
def run(B, H, N, hd):
    Dm = H * hd
    qkv = torch.randn(B, N, 3 * Dm, device=dev).to(BF16); st = (N * 3 * Dm, 3 * Dm)
    q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
    for _ in range(3):
        o, lse = K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st, need_lse=True)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        o, lse = K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st, need_lse=True)
    t1.record(); torch.cuda.synchronize()
    raw = lse.cpu().numpy().reshape(B * H, N).view(np.uint32).reshape(B * H, N)[:, :8].copy().view(np.uint64).reshape(B * H, 4).astype(np.int64)
    base = raw[:, 0].min()
    us = (raw - base) * 0.01
    print(f"B={B} H={H} N={N} hd={hd}: {B * H} workgroups, {t0.elapsed_time(t1) * 100:.1f} us per launch (timed over 10); span {us.max():.1f} us")
    for i, n in enumerate(["K/V staging (+ barrier)", "query tiles (wave 0)", "tail + final barrier"]):
        d = us[:, i + 1] - us[:, i]
        print(f"  {n:26s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
    life = us[:, 3] - us[:, 0]
    print(f"  workgroup life             median {np.median(life):6.2f}  p90 {np.percentile(life, 90):6.2f} us")
    s0 = np.sort(us[:, 0]); e0 = np.sort(us[:, 3])
    print("  start deciles (us):", " ".join(f"{s0[int(f * (len(s0) - 1))]:.1f}" for f in np.linspace(0, 1, 11)))
    print("  end   deciles (us):", " ".join(f"{e0[int(f * (len(e0) - 1))]:.1f}" for f in np.linspace(0, 1, 11)))
args = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else None
for cfg in ([args] if args else [(128, 16, 257, 64), (128, 16, 256, 64), (32, 12, 129, 64), (64, 12, 129, 64)]):
    run(*cfg)
