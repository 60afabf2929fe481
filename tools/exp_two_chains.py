#!/usr/bin/env python3
"""Experiment: does the encoder forward gain from running two half-batch chains concurrently (two streams inside one hipGraph)
instead of one full-batch chain?  The forward pass is a serial chain of 84 kernels that alternate between MFMA-bound (qkv, fc1),
HBM-bound (LayerNorm, residual epilogues) and latency-bound (attention) launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from mirage_amd import builder, engine as E

dev = torch.device("cuda", 0)
model_name, doms, B, Kk, _ = Bn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
args = builder.make_args(doms, ())
torch.manual_seed(0)
model = builder.get_model(args, model_name, with_decoders=False).to(dev).eval()
N, D = Kk + 1, model.global_tokens.shape[-1]
blocks = list(model.encoder)


def chain(x32):
    with torch.no_grad():
        for blk in blocks:
            x32, _, _ = E.block_fwd(blk, x32, save=True)      # save=True: the training forward (LSE, GELU' side output)
    return x32


def bench(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


x_full = torch.randn(B, N, D, device=dev)
halves = [x_full[:B // 2].contiguous(), x_full[B // 2:].contiguous()]
s2 = torch.cuda.Stream()


def one():
    chain(x_full)


def two():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    s2.wait_event(ev)
    with torch.cuda.stream(s2):
        chain(halves[1])
    chain(halves[0])
    main.wait_stream(s2)


def two_serial():
    chain(halves[0]); chain(halves[1])


t1 = bench(one); t2 = bench(two); t3 = bench(two_serial)
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'c2'} encoder forward (training mode), B={B}: one chain {t1:.3f} ms | two half-batch chains on two streams {t2:.3f} ms | "
      f"two half-batch chains back to back {t3:.3f} ms")
