import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K
dev = "cuda:0"; BF16 = torch.bfloat16
B, H, N, hd = 64, 12, 129, 64
Dm = H * hd
qkv = torch.randn(B, N, 3 * Dm, device=dev).to(BF16); st = (N * 3 * Dm, 3 * Dm)
q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
d_o = torch.randn(B, N, Dm, device=dev).to(BF16); dqkv = torch.empty_like(qkv)
for _ in range(5):
    o, lse = K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st)
    K.attn_bwd(q, k, v, o, lse, d_o, dqkv[..., :Dm], dqkv[..., Dm:2 * Dm], dqkv[..., 2 * Dm:], B, H, N, N, hd, hd ** -0.5, st, st, st, st, st, st)
torch.cuda.synchronize()
