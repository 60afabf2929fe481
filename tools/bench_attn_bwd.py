"""Attention backward in isolation (run under `rocprofv3 --kernel-trace --stats` for the per-kernel split)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K
dev = "cuda:0"; BF16 = torch.bfloat16
def run(B, H, N, hd, tag, iters=20):
    Dm = H * hd
    qkv = torch.randn(B, N, 3 * Dm, device=dev).to(BF16); st = (N * 3 * Dm, 3 * Dm)
    q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
    o, lse = K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st)
    d_o = torch.randn(B, N, Dm, device=dev).to(BF16)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[..., :Dm], dqkv[..., Dm:2 * Dm], dqkv[..., 2 * Dm:]
    gb = None if os.environ.get('NOBIAS') else [torch.zeros(Dm, device=dev) for _ in range(3)]
    def f(): K.attn_bwd(q, k, v, o, lse, d_o, dq, dk, dv, B, H, N, N, hd, hd ** -0.5, st, st, st, st, st, st, bias_grads=gb)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    print(f"{tag} B={B} H={H} N={N} hd={hd}: bwd {s.elapsed_time(e) / iters * 1e3:.1f} us per call (dq + dkv + bias reduce)")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "c2"): run(64, 12, 129, 64, "C2")
if which in ("all", "c4"): run(64, 16, 99, 64, "C4")
if which in ("all", "dec"): run(64, 8, 256, 32, "dec")
if which == "sweep":                      # items per CU: 1, 2, 3 (one workgroup per (batch, head))
    for H in (4, 8, 12): run(64, H, 129, 64, f"sweep H={H}", iters=10)
