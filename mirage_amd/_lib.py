"""ctypes binding of libmirage_hip.so (the C ABI declared in include/mirage_hip.h).

`import torch` happens first on purpose: the library must bind to the HIP runtime torch has already mapped
(same soname), otherwise torch's streams / device pointers would be meaningless to the kernels.
There is no fallback: if the shared library is missing or fails to load, importing this module raises.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: one HIP runtime per process)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmirage_hip.so")
if os.environ.get("MRG_LIB_PATH"):          # A/B of two builds of the same ABI (tools only)
    LIB_PATH = os.environ["MRG_LIB_PATH"]

MRG_DT_BF16, MRG_DT_F32 = 0, 1
MRG_OP_KCONTIG, MRG_OP_KSTRIDED = 0, 1
MRG_ACT_NONE, MRG_ACT_GELU, MRG_ACT_GELU_BWD, MRG_ACT_GELU_SAVE_GRAD, MRG_ACT_MUL_AUX_F16 = 0, 1, 2, 3, 4
MRG_LOSS_MSE, MRG_LOSS_CE = 0, 1
MRG_MAX_DOMAINS = 4
MRG_GEMM_AUTO, MRG_GEMM_GLDS3, MRG_GEMM_GLDS4, MRG_GEMM_PIPE128, MRG_GEMM_WS256, MRG_GEMM_P8 = 0, 1, 2, 3, 7, 11      # every other number is reserved (variants measured and dropped)
MRG_GEMM_P8_GROUP = 100     # not a mrg_gemm variant: the kernel-family code of mrg_wgrad_group in timing records

vp, ci, cl, cf = C.c_void_p, C.c_int, C.c_long, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("M", ci), ("N", ci), ("K", ci), ("lda", ci), ("ldb", ci),
                ("a_mode", ci), ("b_mode", ci), ("a_map", vp), ("b_map", vp), ("m_dev", vp), ("k_dev", vp),
                ("C", vp), ("ldc", ci), ("c_dtype", ci), ("C2", vp), ("ldc2", ci), ("bias", vp),
                ("resid", vp), ("ldr", ci), ("accumulate", ci), ("act", ci),
                ("aux_out", vp), ("aux_in", vp), ("ldaux", ci), ("row_map", vp),
                ("add_tab", vp), ("add_row", vp), ("ldt", ci), ("splitk", ci), ("ws", vp), ("variant", ci), ("ev_start", vp), ("ev_stop", vp), ("row_scale", vp), ("colsum_out", vp), ("defer_reduce", ci), ("splitk_counters", vp),
                ("drop_rng", vp), ("drop_p", cf), ("drop_stream", C.c_uint)]


MRG_WGRAD_GROUP_MAX = 48


class WgradGroupDesc(C.Structure):
    _fields_ = [("count", ci), ("K", ci), ("A", vp * MRG_WGRAD_GROUP_MAX), ("B", vp * MRG_WGRAD_GROUP_MAX), ("C", vp * MRG_WGRAD_GROUP_MAX),
                ("M", ci * MRG_WGRAD_GROUP_MAX), ("N", ci * MRG_WGRAD_GROUP_MAX), ("lda", ci * MRG_WGRAD_GROUP_MAX),
                ("ldb", ci * MRG_WGRAD_GROUP_MAX), ("ldc", ci * MRG_WGRAD_GROUP_MAX), ("ev_start", vp), ("ev_stop", vp),
                ("splitk", ci), ("ws", vp), ("fuse", vp)]


class AdamWFuse(C.Structure):
    _fields_ = [("g0", vp), ("p", vp), ("m", vp), ("v", vp), ("shadow_bf16", vp), ("hp", vp), ("n_decay", cl),
                ("beta1", cf), ("beta2", cf), ("eps", cf), ("ss", vp * MRG_WGRAD_GROUP_MAX)]


class AttnDesc(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("o", vp), ("lse", vp),
                ("q_bs", cl), ("q_rs", cl), ("k_bs", cl), ("k_rs", cl), ("v_bs", cl), ("v_rs", cl), ("o_bs", cl), ("o_rs", cl),
                ("B", ci), ("H", ci), ("Nq", ci), ("Nk", ci), ("head_dim", ci), ("scale", cf),
                ("d_o", vp), ("do_bs", cl), ("do_rs", cl), ("dsum", vp),
                ("dq", vp), ("dk", vp), ("dv", vp),
                ("dq_bs", cl), ("dq_rs", cl), ("dk_bs", cl), ("dk_rs", cl), ("dv_bs", cl), ("dv_rs", cl),
                ("colsum_q", vp), ("colsum_k", vp), ("colsum_v", vp), ("colsum_bs", cl),
                ("drop_rng", vp), ("drop_p", cf), ("drop_stream", C.c_uint)]


class DecAssembleDesc(C.Structure):
    _fields_ = [("ctx", vp), ("mask_token", vp), ("task_emb", vp * MRG_MAX_DOMAINS), ("pos", vp),
                ("ids_keep", vp), ("ids_restore", vp), ("dom_start", ci * (MRG_MAX_DOMAINS + 1)),
                ("ndom", ci), ("qdom", ci), ("q_task_emb", vp),
                ("B", ci), ("N", ci), ("K", ci), ("T", ci), ("Tq", ci), ("Dd", ci),
                ("queries", vp), ("context", vp), ("d_queries", vp), ("d_context", vp),
                ("d_ctx", vp), ("partials", vp), ("d_task_emb", vp * MRG_MAX_DOMAINS), ("d_mask_token", vp)]


class LossDesc(C.Structure):
    _fields_ = [("kind", ci), ("pred", vp), ("pred_dtype", ci), ("target", vp), ("mask", vp),
                ("B", ci), ("C", ci), ("H", ci), ("W", ci), ("P", ci),
                ("partial", vp), ("loss", vp), ("coef", vp), ("per_sample", vp), ("gout", vp), ("dpred", vp),
                ("target_slot", vp), ("pred_tokens", ci), ("label_smoothing", cf), ("norm_pix", ci), ("stats", vp), ("nvalid", ci), ("mask_bs", cl)]


class RowFuseDesc(C.Structure):
    _fields_ = [("M", ci), ("NH", ci), ("eps", cf), ("a0", vp), ("w0", vp), ("b0", vp), ("resid", vp), ("x_out", vp),
                ("gamma", vp), ("beta", vp), ("mean", vp), ("rstd", vp), ("ln_out", vp),
                ("w1", vp), ("b1", vp), ("h_out", vp), ("hgrad_out", vp), ("w2", vp), ("b2", vp), ("y32", vp), ("y16", vp)]


class AdamWDesc(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("shadow_bf16", vp), ("n", cl), ("n_decay", cl),
                ("lr_table", vp), ("wd_table", vp), ("step", vp), ("table_len", ci),
                ("lr_scale", cf), ("beta1", cf), ("beta2", cf), ("eps", cf),
                ("grad_norm", vp), ("clip_norm", cf), ("skip_norm", cf), ("sumsq_out", vp)]


# name -> (restype, argtypes); kept in sync with include/mirage_hip.h (tests/test_abi.py checks the exports)
SIGNATURES = {
    "mrg_version": (ci, []),
    "mrg_posemb_grad": (ci, [vp, cl, ci, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp]),
    "mrg_debug_timestamp": (ci, [vp, vp]),
    "mrg_debug_clock_monitor": (ci, [vp, ci, ci, vp]),
    "mrg_last_error_string": (C.c_char_p, []),
    "mrg_gemm": (ci, [C.POINTER(GemmDesc), vp]),
    "mrg_wgrad_group": (ci, [C.POINTER(WgradGroupDesc), vp]),
    "mrg_ln_fwd": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, vp]),
    "mrg_ln_bwd_nblocks": (ci, [ci]),
    "mrg_ln_bwd": (ci, [vp] * 13 + [ci, ci, vp]),
    "mrg_colsum_nblocks": (ci, [ci]),
    "mrg_colsum": (ci, [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]),
    "mrg_attn_fwd": (ci, [C.POINTER(AttnDesc), vp]),
    "mrg_attn_bwd": (ci, [C.POINTER(AttnDesc), vp]),
    "mrg_embed_plan": (ci, [vp, ci, ci, ci, ci, C.POINTER(ci), vp, vp, vp, vp, vp]),
    "mrg_gather_rows_bf16": (ci, [vp, cl, vp, vp, ci, ci, vp, vp]),
    "mrg_set_slots": (ci, [vp, C.POINTER(vp), ci, vp]),
    "mrg_im2col_image": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]),
    "mrg_im2col_layermap": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp]),
    "mrg_class_emb_grad_slices": (ci, []),
    "mrg_class_emb_grad": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp]),
    "mrg_fill_global": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
    "mrg_dec_assemble_fwd": (ci, [C.POINTER(DecAssembleDesc), vp]),
    "mrg_dec_assemble_bwd_nblocks": (ci, [ci, ci, ci]),
    "mrg_dec_assemble_bwd": (ci, [C.POINTER(DecAssembleDesc), vp]),
    "mrg_gemm_multi": (ci, [vp, ci, vp]),
    "mrg_loss_partial_floats": (ci, [ci]),
    "mrg_masked_loss_fwd": (ci, [C.POINTER(LossDesc), vp]),
    "mrg_masked_loss_bwd": (ci, [C.POINTER(LossDesc), vp]),
    "mrg_loss_coef": (ci, [vp, cl, ci, ci, ci, vp, vp]),
    "mrg_masked_loss_fused": (ci, [C.POINTER(LossDesc), vp]),
    "mrg_masked_loss_finish": (ci, [C.POINTER(LossDesc), vp]),
    "mrg_patch_rearrange": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "mrg_maskgen": (ci, [vp, vp, vp, ci, ci, ci, ci, C.POINTER(ci), vp, vp, vp, vp]),
    "mrg_maskgen_rng": (ci, [vp, vp, ci, ci, ci, ci, C.POINTER(ci), vp, vp, vp, vp, vp, vp, vp]),
    "mrg_sumsq_nblocks": (ci, [cl]),
    "mrg_grad_norm": (ci, [vp, cl, vp, vp, vp]),
    "mrg_adamw": (ci, [C.POINTER(AdamWDesc), vp]),
    "mrg_adamw_hp": (ci, [C.POINTER(AdamWDesc), vp, vp]),
    "mrg_adamw_nblocks": (ci, [cl]),
    "mrg_sumsq_finish": (ci, [vp, ci, vp, vp]),
    "mrg_step_inc": (ci, [vp, vp, C.c_float, C.c_float, vp]),
    "mrg_cast_bf16": (ci, [vp, vp, cl, vp]),
    "mrg_row_scale_bf16": (ci, [vp, vp, vp, cl, ci, vp]),
    "mrg_dropout_bwd_bf16": (ci, [vp, vp, vp, cl, ci, vp, cf, C.c_uint, vp]),
    "mrg_ln_bwd_partial_rows": (ci, [ci, ci]),
    "mrg_reduce_batch": (ci, [vp, ci, vp]),
    "mrg_rowfuse_fwd": (ci, [C.POINTER(RowFuseDesc), vp]),
    "mrg_augment_image": (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),
    "mrg_augment_label": (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),
    "mrg_resize_image": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
    "mrg_resize_label": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  mirage_amd has no CPU or eager fallback.")

lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

ABI_VERSION = lib.mrg_version()
EXPECTED_ABI_VERSION = 16      # must match MRG_ABI_VERSION in include/mirage_hip.h: the ctypes structs above mirror that layout
if ABI_VERSION != EXPECTED_ABI_VERSION:
    raise ImportError(f"{LIB_PATH} reports ABI version {ABI_VERSION}, the Python bindings expect {EXPECTED_ABI_VERSION}: rebuild it "
                      "(python -c 'import __graft_entry__ as g; g.build()')")


_SYNC_DEBUG = os.environ.get("MRG_SYNC_DEBUG", "0") == "1"


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.mrg_last_error_string()
        raise RuntimeError(f"libmirage_hip {what} failed (rc={rc}): {msg.decode() if msg else ''}")
    if _SYNC_DEBUG:   # debugging aid: surface asynchronous faults at the launch that caused them
        print(f"[mrg] {what}", flush=True)
        torch.cuda.synchronize()


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
