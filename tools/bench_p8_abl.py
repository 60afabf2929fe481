#!/usr/bin/env python3
"""Ablation of the 8-phase kernel's main loop on one shape (MRG_GEMM_ABLATE bits: 1 no DMA, 2 no MFMA, 4 no LDS reads, 8 no setprio,
16 no stagger).  Each setting runs in its own process (the flag is read once)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from mirage_amd import kernels as K, _lib as L
    M, N, Kd = (int(x) for x in sys.argv[2:5])
    A = torch.randn(M, Kd, device="cuda").bfloat16(); B = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    K.TUNE["force"] = L.MRG_GEMM_P8; K.TUNE["enabled"] = False
    f = lambda: K.gemm(A, B, M, N, Kd, out=out)
    for _ in range(5): f()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 100)
    t = sorted(ts)[2]
    print(f"ablate={os.environ.get('MRG_GEMM_ABLATE', '0'):>2s} {M}x{N}x{Kd}: {t:8.1f} us  {2.0 * M * N * Kd / t / 1e6:7.0f} TF-equivalent")
else:
    shape = sys.argv[1:4] if len(sys.argv) > 3 else ["4096", "4096", "4096"]
    for abl in (0, 8, 16, 1, 4, 5, 2, 3, 6, 7):
        subprocess.run([sys.executable, __file__, "child", *shape], env=dict(os.environ, MRG_GEMM_ABLATE=str(abl)))
