"""The C-ABI library loads (no GPU needed) and exports exactly what include/mirage_hip.h declares; the ctypes binding
covers every entry point.  No compute calls here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mirage_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  (same load order as the product)
    lib = ctypes.CDLL(os.path.join(ROOT, "mirage_amd", "libmirage_hip.so"))
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mirage_hip.h but not exported"


def test_binding_covers_the_header():
    from mirage_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    assert _lib.ABI_VERSION == _lib.EXPECTED_ABI_VERSION == 16
    assert _lib.lib.mrg_last_error_string() is not None


def test_argument_errors_are_reported_not_crashed():
    from mirage_amd import _lib
    # shape validation happens before any launch: callable without a GPU
    rc = _lib.lib.mrg_ln_fwd(None, None, None, None, None, None, None, 4, 768, 1e-6, None)
    assert rc < 0 and b"null pointer" in _lib.lib.mrg_last_error_string()
    d = _lib.GemmDesc()
    rc = _lib.lib.mrg_gemm(ctypes.byref(d), None)
    assert rc < 0 and b"mrg_gemm" in _lib.lib.mrg_last_error_string()
    r = _lib.RowFuseDesc()
    r.M, r.NH = 100, 1024
    rc = _lib.lib.mrg_rowfuse_fwd(ctypes.byref(r), None)
    assert rc < 0 and b"multiple of 64" in _lib.lib.mrg_last_error_string()
    assert _lib.lib.mrg_ln_bwd_nblocks(8256) >= 1 and _lib.lib.mrg_loss_partial_floats(64) == 64 * 64


def test_product_fails_loudly_on_cpu():
    import pytest
    import torch
    from mirage_amd import builder
    args = builder.make_args(("bscan",), ("bscan",), input_size=64, patch_size=8, decoder_dim=64, decoder_depth=1, decoder_num_heads=2)
    model = builder.get_model(args, dim_tokens=128, depth=1, num_heads=2)
    with pytest.raises(RuntimeError, match="GPU"):
        model({"bscan": torch.zeros(1, 1, 64, 64)}, num_encoded_tokens=8)
