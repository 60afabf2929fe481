#!/usr/bin/env python3
"""Re-tune selected entries of mirage_amd/gemm_tune_gfx950.json in place (isolated launches, the autotuner's own protocol) after
a new kernel variant became a candidate.  Usage: retune_shapes.py c5 [--save]   (c5: the encoder-only Large forward, M = 65792)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from mirage_amd import builder, kernels as K

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
model_name, doms, B, Kk, train = Bn.CONFIGS[cfg]
assert not train, "forward-only configurations (use tools/tune_in_step.py for training steps)"
dev = torch.device("cuda", 0)
args = builder.make_args(doms, ())
torch.manual_seed(0)
model = builder.get_model(args, model_name, with_decoders=False).to(dev).eval()
x = Bn.synth_batches(doms, B, 1, 1234, dev)[0]
K.TUNE["hits"] = set()
with torch.no_grad():
    model(x, mask_inputs=False)
torch.cuda.synchronize()
hits = [k for k in K.TUNE["hits"] if 2.0 * k[2] * k[3] * k[4] > 5e9]
K.TUNE["hits"] = None
old = {k: K._tune_cache.get(k) for k in hits}
for k in hits:
    K._tune_cache.pop(k, None)
K.TUNE["log"].clear()
with torch.no_grad():
    model(x, mask_inputs=False)
torch.cuda.synchronize()
for key, best, us in K.TUNE["log"]:
    print(f"{key[:5]} act={key[5]}: {old.get(key)} -> {best}  {us:.1f} us  {2.0 * key[2] * key[3] * key[4] / us / 1e6:.0f} TF/s")
if "--save" in sys.argv:
    K.save_tune_cache()
    print("saved")
