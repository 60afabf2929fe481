"""Phase timestamps of the fused attention-backward kernel (debug build: make -C mirage_amd/csrc EXTRA=-DMRG_ATTN_TS).  Prints, per
phase, the median / p90 duration over the workgroups and the distribution of workgroup start times (rounds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
from mirage_amd import kernels as K
dev = "cuda:0"; BF16 = torch.bfloat16
def run(B, H, N, hd, tag):
    Dm = H * hd
    qkv = torch.randn(B, N, 3 * Dm, device=dev).to(BF16); st = (N * 3 * Dm, 3 * Dm)
    q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
    o, lse = K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st)
    d_o = torch.randn(B, N, Dm, device=dev).to(BF16)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[..., :Dm], dqkv[..., Dm:2 * Dm], dqkv[..., 2 * Dm:]
    # same as K.attn_bwd but keeping the dsum workspace
    d = K._attn_desc(q, k, v, o, lse, B, H, N, N, hd, hd ** -0.5, (st, st, st, (N * Dm, Dm)))
    dsum = torch.zeros((B, H, N), dtype=torch.float32, device=dev)
    d.d_o, d.do_bs, d.do_rs = d_o.data_ptr(), d_o.stride(0), d_o.stride(1)
    d.dsum = dsum.data_ptr()
    d.dq, d.dk, d.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    (d.dq_bs, d.dq_rs), (d.dk_bs, d.dk_rs), (d.dv_bs, d.dv_rs) = st, st, st
    for _ in range(3):
        K.check(K.lib.mrg_attn_bwd(C.byref(d), K._s()), "bwd")
    torch.cuda.synchronize()
    raw = dsum.cpu().numpy().reshape(-1).view(np.uint64)[:B * H * 8].reshape(B * H, 8).astype(np.int64)   # [WG, 8]
    t0 = raw[:, 0].min()
    us = (raw - t0) * 0.01
    names = ["loads issued", "staged (barrier)", "phase A", "barrier", "dv dk flushed", "phase B", "dq flushed"]
    print(f"{tag}: {B * H} workgroups; kernel span {us.max():.1f} us")
    for i, n in enumerate(names):
        dlt = us[:, i + 1] - us[:, i]
        print(f"  {n:18s} median {np.median(dlt):6.2f}  p90 {np.percentile(dlt, 90):6.2f}  max {dlt.max():6.2f} us")
    life = us[:, 7] - us[:, 0]
    print(f"  workgroup life     median {np.median(life):6.2f}  p90 {np.percentile(life, 90):6.2f}")
    st_ = np.sort(us[:, 0])
    print("  start times (us) deciles:", " ".join(f"{st_[int(q * (len(st_) - 1))]:.1f}" for q in np.linspace(0, 1, 11)))
    en = np.sort(us[:, 7])
    print("  end times   (us) deciles:", " ".join(f"{en[int(q * (len(en) - 1))]:.1f}" for q in np.linspace(0, 1, 11)))
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "c2"): run(64, 12, 129, 64, "C2")
if which in ("all", "c4"): run(64, 16, 99, 64, "C4")
