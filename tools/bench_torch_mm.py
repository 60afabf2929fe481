"""Calibration only (not used by the product path): what the vendor BLAS reaches on the same GEMM shapes."""
import torch, sys
dev = "cuda"
shapes = [("qkv fwd", 8256, 2304, 768), ("proj fwd", 8256, 768, 768), ("fc1 fwd", 8256, 3072, 768), ("fc2 fwd", 8256, 768, 3072),
          ("fc1 wgrad", 3072, 768, 8256), ("proj wgrad", 768, 768, 8256),
          ("L qkv", 6336, 3072, 1024), ("L proj", 6336, 1024, 1024), ("L fc1", 6336, 4096, 1024), ("L fc2", 6336, 1024, 4096),
          ("L fc1 wgrad", 4096, 1024, 6336), ("C5 fc1", 65792, 4096, 1024), ("C5 fc2", 65792, 1024, 4096), ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(20):
        c = a @ b.t()
    en.record(); en.synchronize()
    us = st.elapsed_time(en) / 20 * 1e3
    print(f"{name:12s} {M}x{N}x{K}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.0f} TF/s", flush=True)
