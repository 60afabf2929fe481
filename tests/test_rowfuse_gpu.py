"""The output adapters' fused row chains (csrc/rowfuse.hip, through the C ABI) against (a) torch fp32 on the same bf16-rounded operands and
(b) the separate launches they replace (mrg_ln_fwd + mrg_gemm epilogues): same side outputs, same contract.  GPU only."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mirage_amd import kernels
    return kernels


def dev():
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0, dtype=F32, seed=0):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def rel_rms(a, b):
    a, b = a.double(), b.double()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def _norm(seed):
    n = types.SimpleNamespace()
    n.weight = (1.0 + 0.2 * rnd(256, seed=seed)).contiguous()
    n.bias = (0.1 * rnd(256, seed=seed + 1)).contiguous()
    n.eps = 1e-6
    return n


def _lin(nout, nin, seed):
    l = types.SimpleNamespace()
    l.w16 = rnd(nout, nin, scale=nin ** -0.5, dtype=BF16, seed=seed)
    l.bias = (0.1 * rnd(nout, seed=seed + 1)).contiguous()
    return l


def _ln_ref(x, n):
    return torch.nn.functional.layer_norm(x.double(), (256,), n.weight.double(), n.bias.double(), n.eps)


@pytest.mark.parametrize("M,NH", [(64, 256), (8256, 512), (16384, 768), (6336, 256), (128, 1024)])
@pytest.mark.parametrize("save", [True, False])
def test_layernorm_plus_projection(K, M, NH, save):
    x = rnd(M, 256, seed=1) * (1.0 + rnd(M, 1, seed=2).abs()) + 0.3
    n, l = _norm(3), _lin(NH, 256, 5)
    y, ln, mean, rstd = K.rowfuse_ln_proj(x, n, l, l.w16, save)
    torch.cuda.synchronize()
    ln_ref = _ln_ref(x, n)
    # (a) torch: the projection of the bf16-rounded LayerNorm output, as the kernel computes it
    y_ref = ln_ref.float().to(BF16).double() @ l.w16.double().t() + l.bias.double()
    assert rel_rms(y.double(), y_ref) < 4e-3
    # (b) the separate launches
    ln2, mean2, rstd2 = K.ln_fwd(x, n.weight, n.bias, n.eps)
    y2 = K.gemm(ln2, l.w16, M, NH, 256, bias=l.bias, out_dtype=BF16)
    assert rel_rms(y.double(), y2.double()) < 3e-3
    if save:
        assert rel_rms(ln.double(), ln_ref) < 4e-3
        torch.testing.assert_close(mean, mean2, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rstd, rstd2, rtol=1e-5, atol=1e-6)
        assert (ln.float() - ln2.float()).abs().max() <= 2.0 ** -6 * ln2.float().abs().max()      # at most one bf16 ulp apart
    else:
        assert ln is None and mean is None


@pytest.mark.parametrize("M", [64, 6336, 16384])
@pytest.mark.parametrize("resid,save,twin", [(True, True, True), (False, True, False), (True, False, True), (False, False, False)])
def test_post_attention_chain(K, M, resid, save, twin):
    from mirage_amd import _lib as L
    o = rnd(M, 256, dtype=BF16, seed=11)
    r = (rnd(M, 256, seed=12) * 2.0).contiguous() if resid else None
    proj, n, fc1, fc2 = _lin(256, 256, 13), _norm(15), _lin(1024, 256, 17), _lin(256, 1024, 19)
    y32, y16, saved = K.rowfuse_mlp(o, proj, proj.w16, r, n, fc1, fc1.w16, fc2, fc2.w16, save, twin)
    torch.cuda.synchronize()
    # (a) torch fp64 with the kernel's rounding points (bf16 LayerNorm output and hidden activations)
    x_ref = o.double() @ proj.w16.double().t() + proj.bias.double() + (r.double() if resid else 0.0)
    ln_ref = _ln_ref(x_ref, n)
    pre = ln_ref.float().to(BF16).double() @ fc1.w16.double().t() + fc1.bias.double()
    h_ref = torch.nn.functional.gelu(pre)
    y_ref = h_ref.float().to(BF16).double() @ fc2.w16.double().t() + fc2.bias.double() + x_ref
    assert rel_rms(y32.double(), y_ref) < 2e-3
    if twin:
        assert rel_rms(y16.double(), y_ref) < 4e-3
    else:
        assert y16 is None
    # (b) the separate launches
    x2 = K.gemm(o, proj.w16, M, 256, 256, bias=proj.bias, resid=r, out_dtype=F32)
    ln2, mean2, rstd2 = K.ln_fwd(x2, n.weight, n.bias, n.eps)
    hg2 = torch.empty((M, 1024), dtype=torch.float16, device=dev())
    h2 = K.gemm(ln2, fc1.w16, M, 1024, 256, bias=fc1.bias, out_dtype=BF16, act=L.MRG_ACT_GELU_SAVE_GRAD, aux_out=hg2)
    y2 = K.gemm(h2, fc2.w16, M, 256, 1024, bias=fc2.bias, resid=x2, out_dtype=F32)
    assert rel_rms(y32.double(), y2.double()) < 2e-3
    if save:
        x, ln, mean, rstd, hg, h = saved
        assert rel_rms(x.double(), x_ref) < 1e-5 * 50
        torch.testing.assert_close(x, x2, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(mean, mean2, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(rstd, rstd2, rtol=1e-4, atol=1e-5)
        assert rel_rms(ln.double(), ln_ref) < 4e-3
        assert rel_rms(h.double(), h_ref) < 6e-3
        gd_ref = 0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5
        assert float((hg.double() - gd_ref).abs().max()) < 2e-2       # fp16 storage of a derivative whose argument moved by bf16 roundoff
        assert rel_rms(hg.double(), hg2.double()) < 5e-3
    else:
        assert saved is None


def test_fused_chain_is_repeatable_and_refuses_ragged_tiles(K):
    M = 8256
    o = rnd(M, 256, dtype=BF16, seed=21)
    r = rnd(M, 256, seed=22)
    proj, n, fc1, fc2 = _lin(256, 256, 23), _norm(25), _lin(1024, 256, 27), _lin(256, 1024, 29)
    first = None
    for _ in range(10):
        y32, y16, saved = K.rowfuse_mlp(o, proj, proj.w16, r, n, fc1, fc1.w16, fc2, fc2.w16, True, True)
        cur = [y32, y16] + list(saved)
        if first is None:
            first = [t.clone() for t in cur]
        else:
            for a, b in zip(first, cur):
                assert torch.equal(a, b)
    assert not K.rowfuse_ok(100, 256, 1024) and not K.rowfuse_ok(128, 192, 768) and K.rowfuse_ok(6336, 256, 768)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        K.rowfuse_ln_proj(rnd(100, 256, seed=1), n, fc1, fc1.w16, False)
