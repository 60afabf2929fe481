import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, numpy as np
from conftest import load_golden
from oracle import mirage_oracle as O
from mirage_amd import engine as E, kernels as K
import test_model_gpu as T

def rr(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

name = sys.argv[1] if len(sys.argv) > 1 else "tiny3"
fx, meta = load_golden(name)
cfg, B, Kk = meta["cfg"], meta["B"], meta["K"]
model, args, sd = T._build(cfg)
dev = "cuda:0"
sdg = {k: v.to(dev) for k, v in sd.items()}
x = {d: t.to(dev) for d, t in O.synthetic_batch(cfg, B, seed=meta["data_seed"]).items()}
doms = cfg["in_domains"]
ids_keep = torch.from_numpy(fx["ids_keep"]).to(dev); ids_restore = torch.from_numpy(fx["ids_restore"]).to(dev)
model._ensure_arena(torch.device(dev))
with torch.no_grad():
    toks = O.input_tokens(cfg, sdg, x)
    enc_in = O.select_and_append_global(cfg, sdg, toks, ids_keep)
    x0, _ = E.embed_fwd(model, x, ids_keep, False)
    print("embed", rr(x0, enc_in))
    enc_ref = O.encoder(cfg, sdg, enc_in)
    cur = x0
    for i, blk in enumerate(model.encoder):
        cur, cur16, _ = E.block_fwd(blk, cur, False, want_bf16_out=True)
    print("enc", rr(cur, enc_ref))
    dom_start = model._dom_start(doms)
    for d in cfg["out_domains"]:
        ad = model.output_adapters[d]
        img, sv = E.decoder_fwd(ad, enc_ref.contiguous(), E.to_bf16(enc_ref.contiguous()), ids_keep, ids_restore, doms, dom_start, 1, True)
        ref = O.output_adapter(cfg, sdg, d, enc_ref, ids_keep, ids_restore)
        (desc, task_embs, queries, context, qn, mq, rq, cn, mc, rc, qp, kv, o, lse, x1, on, mo, ro, h_pre, h_act, x2, blk_saved, x16, enc16, dims) = sv
        # oracle intermediates
        pre = f"output_adapters.{d}."
        ctx = enc_ref @ sdg[pre + "proj_context.weight"].t() + sdg[pre + "proj_context.bias"]
        print(d, "pred", rr(img, ref), "x2", float(x2.abs().mean()), "x16 rms", float(x16.float().pow(2).mean().sqrt()))
        # recompute oracle pieces for queries/context
        import torch.nn.functional as F
        layout, Ttot = O.token_layout(cfg)
        Dd = cfg["decoder_dim"]
        cwm = torch.cat([ctx[:, :-1], sdg[pre + "mask_token"].expand(B, Ttot - Kk, -1)], 1)
        cwm = torch.gather(cwm, 1, ids_restore[:, :, None].expand(-1, -1, Dd))
        pos = O.posemb_tokens(sdg[pre + "pos_emb"])
        embs = [sdg[pre + "task_embeddings." + t].expand(1, pos.shape[0], -1) + pos[None] for t in doms]
        cwm = cwm + torch.cat(embs, 1)
        s, e = layout[d]
        print("   queries", rr(queries, cwm[:, s:e]), "context", rr(context, torch.cat([torch.gather(cwm, 1, ids_keep[:, :, None].expand(-1, -1, Dd)), ctx[:, -1:]], 1)))
        qn_r = O.layer_norm(cwm[:, s:e], sdg[pre + "query_norm.weight"], sdg[pre + "query_norm.bias"], 1e-6)
        print("   qn", rr(qn, qn_r))
        cn_r = O.layer_norm(torch.cat([torch.gather(cwm, 1, ids_keep[:, :, None].expand(-1, -1, Dd)), ctx[:, -1:]], 1), sdg[pre + "context_norm.weight"], sdg[pre + "context_norm.bias"], 1e-6)
        xa = O.cross_attention(qn_r, cn_r, sdg, pre + "decoder.", cfg["decoder_heads"])
        print("   x1(cross-attn)", rr(x1, xa))
        xm = xa + O.mlp(O.layer_norm(xa, sdg[pre + "out_norm.weight"], sdg[pre + "out_norm.bias"], 1e-6), sdg, pre + "mlp.")
        print("   x2(mlp)", rr(x2, xm))
        xb = xm
        for i in range(cfg["decoder_depth"]):
            xb = O.block(xb, sdg, pre + f"decoder_transformer.{i}.", cfg["decoder_heads"], 1e-6)
        print("   blocks", rr(x16, xb))
        # loss
        tm = O.task_masks_from_ids(cfg, ids_restore, Kk)
        P = cfg["domains"][d]["patch"]
        if cfg["domains"][d]["kind"] == "semseg":
            print("   loss", float(O.masked_ce(img.float(), x[d], tm[d], P)), float(O.masked_ce(ref, x[d], tm[d], P)), float(fx[f"loss.{d}"]))
        else:
            print("   loss", float(O.masked_mse(img.float(), x[d], tm[d], P)), float(O.masked_mse(ref, x[d], tm[d], P)), float(O.masked_mse(ref.bfloat16().float(), x[d], tm[d], P)), float(fx[f"loss.{d}"]))
