import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K
dev = "cuda:0"; BF16 = torch.bfloat16
def run(B, H, N, hd, tag):
    Dm = H * hd
    qkv = torch.randn(B, N, 3 * Dm, device=dev).to(BF16); st = (N * 3 * Dm, 3 * Dm)
    q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
    for _ in range(3): K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st)
    e.record(); torch.cuda.synchronize()
    print(f"{tag} B={B} H={H} N={N}: {s.elapsed_time(e) / 20 * 1e3:.1f} us (MRG_ATTN_CB={os.environ.get('MRG_ATTN_CB', 'auto')})")
run(64, 12, 129, 64, "C2"); run(256, 16, 257, 64, "C5"); run(64, 16, 99, 64, "C4"); run(64, 8, 256, 32, "dec")
