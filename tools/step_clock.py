#!/usr/bin/env python3
"""Shader clock over a graph-replayed training step: a one-wave monitor kernel (mrg_debug_clock_monitor) samples {wall clock, shader
clock} every 10 us on its own stream while the step replays; the step's own phase timestamps (MRG_TS) are on the same wall clock, so
every phase gets the clock the chip ran at.  usage: step_clock.py [c2|c4|c5]"""
import os, sys
os.environ["MRG_TS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as Bn
from mirage_amd import builder, kernels as K, optim, _lib as L
from mirage_amd.trainer import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
model_name, doms, B, Kk, train = Bn.CONFIGS[cfg]
dev = torch.device("cuda", 0)
args = builder.make_args(doms, doms)
torch.manual_seed(0)
model = builder.get_model(args, model_name).to(dev).train()
pool = Bn.synth_batches(doms, B, 2, 1234, dev)
opt = optim.FusedAdamW(model, [1e-4] * 4096, [0.05] * 4096, betas=(0.9, 0.95), eps=1e-8)
ts = TrainStep(model, builder.get_losses(args), opt, pool[0], Kk, use_graph=True)
ts.load(pool[0]); ts.warmup_and_capture(1)
for _ in range(8):
    ts.step(pool[0])
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(8):
    ts.step(pool[0])
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 8
PERIOD = 1000                                            # 10 us apart
NS = int(ms * 100) - 40                                  # the monitor is a branch of the captured step and ends just before it
buf = torch.zeros(2 * NS, dtype=torch.int64, device=dev)
mon = torch.cuda.Stream(device=dev)
del ts
K.TS["names"].clear()
ts = TrainStep(model, builder.get_losses(args), opt, pool[0], Kk, use_graph=True)
plain = ts._fwd_bwd


def with_monitor():
    # a stream beside a replaying graph is serialised behind it on this runtime, so the monitor is captured INTO the step: forked at
    # its start, joined at its end
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    mon.wait_event(ev)
    with torch.cuda.stream(mon):
        L.check(L.lib.mrg_debug_clock_monitor(buf.data_ptr(), NS, PERIOD, mon.cuda_stream), "mrg_debug_clock_monitor")
    plain()
    cur.wait_stream(mon)


ts._fwd_bwd = with_monitor
ts.load(pool[0]); ts.warmup_and_capture(1)
for _ in range(6):
    ts.step(pool[0])
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(-1, 2).astype(np.int64)
s = s[s[:, 0] > 0]
wall, clk = s[:, 0], s[:, 1]
mhz = np.diff(clk) / np.diff(wall).astype(np.float64) * 100.0
mid = (wall[1:] + wall[:-1]) // 2
names = list(K.TS["names"])
t = K.TS["buf"].cpu().numpy()[:len(names)].astype(np.int64)
order = np.argsort(t)
start = t[names.index("step_start")]
end = t.max()
print(f"{cfg}: {ms:.3f} ms/step un-monitored; monitored replay {(end - start) / 100:.0f} us from step_start to the last phase stamp; "
      f"{len(mhz)} clock samples")
print(f"monitor samples cover {(wall[0] - start) / 100:.0f} .. {(wall[-1] - start) / 100:.0f} us relative to the last replay's step_start")
sel = (mid >= start) & (mid <= end)
if not sel.any():
    sys.exit("the monitor did not run beside the last replay")
print(f"whole step: mean {mhz[sel].mean():.0f} MHz, median {np.median(mhz[sel]):.0f}, min {mhz[sel].min():.0f}, max {mhz[sel].max():.0f}; "
      f"time below 1.6 GHz {100.0 * (mhz[sel] < 1600).mean():.0f} %, below 2.0 GHz {100.0 * (mhz[sel] < 2000).mean():.0f} %")
main = [n for n in ("step_start", "enc_fwd_start", "enc_fwd_end", "enc_bwd_start", "enc_bwd_chain_end", "opt_start", "step_end") if n in names]
print("| from | to | us | mean MHz | min | max |")
print("|---|---|---|---|---|---|")
for a, b in zip(main[:-1], main[1:]):
    ta, tb = t[names.index(a)], t[names.index(b)]
    m = mhz[(mid >= ta) & (mid < tb)]
    if len(m):
        print(f"| {a} | {b} | {(tb - ta) / 100:.0f} | {m.mean():.0f} | {m.min():.0f} | {m.max():.0f} |")
print("\nside-stream groups (start -> weight gradients done):")
for n in names:
    if n.endswith(".start") and n.startswith("side_flush"):
        e = n.replace(".start", ".wgrad_end")
        if e in names:
            ta, tb = t[names.index(n)], t[names.index(e)]
            m = mhz[(mid >= ta) & (mid < tb)]
            if len(m):
                print(f"  {n[:-6]}: {(ta - start) / 100:8.0f} us + {(tb - ta) / 100:6.0f} us  mean {m.mean():.0f} MHz (min {m.min():.0f})")
print("\ntimeline (250 us bins from step_start): " + " ".join(f"{mhz[(mid >= start + i * 25000) & (mid < start + (i + 1) * 25000)].mean():.0f}"
                                                              for i in range(int((end - start) // 25000) + 1)
                                                              if ((mid >= start + i * 25000) & (mid < start + (i + 1) * 25000)).any()))
