/* libmirage_hip.so: C ABI of the MI355X (gfx950) kernels behind the MIRAGE MultiMAE pretraining path.
 *
 * The reference (j-morano MIRAGE) has no native code and no FFI: its hot path is a chain of aten dispatches
 * issued from Python nn.Modules.  Each entry point below replaces one group of those dispatches; the
 * reference call site it stands in for is cited as file:line (paths relative to the reference tree).
 * Host code (the mirage_amd python package, via ctypes) is the only caller; see INTEGRATION.md for the binding.
 *
 * Conventions
 *   - every function returns int: 0 ok, <0 argument/shape error (mrg_last_error_string() has the text),
 *     >0 a hipError_t from the launch.  Kernels are asynchronous on `stream` (a hipStream_t).
 *   - all pointers are DEVICE pointers unless a parameter says "host".  The library owns no memory and
 *     keeps no state besides the last error string: callers pass workspaces.
 *   - "bf16" = raw bfloat16 bits (uint16_t), row-major, explicit leading dimensions in ELEMENTS.
 *   - the library never calls hipSetDevice / hipDeviceSynchronize.
 */
#ifndef MIRAGE_HIP_H
#define MIRAGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define MRG_ABI_VERSION 16   /* 16: the GEMM variants that the committed table never selected for a production shape are gone (two-stage 256 x 128 = 4, 64-row tile = 8, single-stage 256 x 128 = 10, 128 x 256 two-workgroup kernel = 13: mrg_gemm_desc.variant refuses them like the other reserved numbers; round-5 review item 6); 15: mrg_rowfuse_fwd (the output adapters' row-local chains in one launch, csrc/rowfuse.hip); 14: mrg_gemm_multi; mrg_loss_coef, mrg_masked_loss_fused, mrg_masked_loss_finish (loss forward + gradient in one pass), mrg_loss_desc.nvalid / mask_bs; 13: the stream-K schedule of round 4 is gone (measured slower everywhere it was meant to win and selected nowhere: profiles/r04_streamk.md; the code is in the history of csrc/gemm8p.hip, round 4) -- variant 12 reserved, mrg_gemm_desc.sk_grid, mrg_wgrad_group_desc.streamk / counters / sk_grid, mrg_streamk_ws_bytes and mrg_debug_sk_trace removed; 12: MRG_GEMM_DUO (mrg_gemm_desc.variant 13: the 128 x 256 two-workgroups-per-CU kernel, csrc/gemm_duo.hip); 11: mrg_attn_desc.drop_rng / drop_p / drop_stream (dropout on the attention probabilities), mrg_posemb_grad, mrg_loss_desc.label_smoothing / norm_pix / stats, mrg_wgrad_group_desc.fuse + mrg_adamw_hp (AdamW in the weight-gradient epilogue), mrg_debug_clock_monitor; 10: MRG_GEMM_P8SK (stream-K schedule of the 8-phase kernel), mrg_gemm_desc.sk_grid, mrg_wgrad_group_desc.streamk / counters / sk_grid, mrg_streamk_ws_bytes; 9: mrg_maskgen_rng, input slots (mrg_set_slots, *_slot parameters, mrg_loss_desc.target_slot); 8: mrg_gemm_desc.drop_rng / drop_p / drop_stream, mrg_dropout_bwd_bf16; 7: mrg_wgrad_group, mrg_debug_timestamp; 2: mrg_gemm_desc grew row_scale / defer_reduce, act codes 3, 4, reduce_batch / augment / row_scale entry points; 3: colsum_out, attention colsum_*; 4: step = int[2], mrg_step_inc takes the skip decision; 5: splitk_counters, MRG_GEMM_P8, resize entry points; 6: mrg_adamw_desc.sumsq_out, mrg_sumsq_finish */

enum { MRG_DT_BF16 = 0, MRG_DT_F32 = 1 };
enum { MRG_OP_KCONTIG = 0, MRG_OP_KSTRIDED = 1 };
enum { MRG_ACT_NONE = 0, MRG_ACT_GELU = 1, MRG_ACT_GELU_BWD = 2,
       MRG_ACT_GELU_SAVE_GRAD = 3,   /* out = gelu(v); aux_out (IEEE fp16, not bf16) = gelu'(v): the derivative is in [-0.13, 1.13],
                                        so fp16 keeps 11 significant bits and the backward GEMM needs no transcendental math */
       MRG_ACT_MUL_AUX_F16 = 4 };    /* v *= aux_in (IEEE fp16): backward of MRG_ACT_GELU_SAVE_GRAD */
enum { MRG_LOSS_MSE = 0, MRG_LOSS_CE = 1 };
#define MRG_MAX_DOMAINS 4

int mrg_version(void);
const char* mrg_last_error_string(void);

/* ---------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[M,N] = A(MxK) * B(NxK)^T  (+bias) (act) (+add_tab) (+resid) ...
 * Replaces nn.Linear forward (mirage/utils.py:155-157,177,186,209-212,221; output_adapters.py:272,288),
 * the patch-embedding Conv2d as an im2col GEMM (mirage/input_adapters.py:101,229) and autograd's
 * dgrad / wgrad `mm`s of all of them.
 *   a_mode/b_mode  MRG_OP_KCONTIG: element (row,k) at P[row*ld + k];  MRG_OP_KSTRIDED: at P[k*ld + row].
 *     forward  y = x W^T      : A = x  [M,K]  KCONTIG,  B = W [N,K] KCONTIG
 *     dgrad    dx = dy W      : A = dy [M,N'] KCONTIG,  B = W [N',K'] KSTRIDED (ldb = K')
 *     wgrad    dW = dy^T x    : A = dy [m,N'] KSTRIDED, B = x [m,K'] KSTRIDED, K = m (token count)
 *   a_map   optional int32 gather: KCONTIG -> source row = a_map[row]; KSTRIDED -> source k-row = a_map[k]
 *   b_map   same for B (KSTRIDED only)
 *   m_dev   optional device int: effective M = min(M, *m_dev) (data-dependent per-domain token counts)
 *   k_dev   optional device int: effective K = min(K, *k_dev)
 *   C       bf16 or fp32 [M,N]; C2 optional bf16 copy; accumulate: fp32 C += result
 *   bias    fp32 [N]; resid fp32 [M,N] added; act: GELU stores gelu(v) (and pre-activation v to aux_out),
 *           GELU_BWD multiplies by gelu'(aux_in)
 *   row_map optional int32 scatter of output rows; add_tab/add_row: v += add_tab[add_row[row]*ldt + col]
 *           (positional-embedding add of the input adapters, mirage/input_adapters.py:104-108)
 *   splitk  >1: partial sums to ws [splitk][M][N] fp32, then reduced (bias/accumulate epilogue only)
 *   variant kernel variant for aligned, map-free problems: single-stage LDS-DMA 128 x 128 kernel at 3 or 4 workgroups/CU, two-stage pipelined
 *           128 x 128 kernel, warp-specialised 256 x 128 kernel (8 compute + 4 loader waves, 3-stage ring), MRG_GEMM_P8 = the 256 x 256 8-phase
 *           kernel, one tile per workgroup; 0 picks by shape.  Every other number is reserved (kernels measured and dropped) and refused.
 */
typedef struct mrg_gemm_desc {
    const void* A; const void* B;
    int M, N, K, lda, ldb;
    int a_mode, b_mode;
    const int* a_map; const int* b_map;
    const int* m_dev; const int* k_dev;
    void* C; int ldc; int c_dtype;
    void* C2; int ldc2;
    const float* bias;
    const float* resid; int ldr;
    int accumulate;
    int act;
    void* aux_out; const void* aux_in; int ldaux;
    const int* row_map;
    const float* add_tab; const int* add_row; int ldt;
    int splitk; float* ws;
    int variant;   /* 0 = heuristic choice; 1..11 force a kernel variant (see MRG_GEMM_* below); used by the host-side autotuner */
    void* ev_start; void* ev_stop;   /* optional hipEvent_t pair recorded on `stream` around the main GEMM kernel only (not the
                                        split-K finishing kernel): lets a benchmark time exactly what rocprofv3 reports per kernel */
    const float* row_scale;          /* optional [M]: (acc + bias -> act) is multiplied by row_scale[row] before add_tab / resid are
                                        added -- stochastic depth, x + drop_path(f(x)) (mirage/utils.py:103-118,259-262);
                                        not combinable with splitk > 1 */
    float* colsum_out;               /* optional fp32 [ceil(M/128)][N]: per 128-row tile, the column sums of the values written to C (after
                                        every epilogue step): summed over the tiles (mrg_reduce_batch) they are e.g. the bias gradient
                                        of the layer whose output gradient this GEMM produces; forces the single-stage 128x128 kernel */
    int defer_reduce;                /* splitk > 1 only: leave the split-K slabs ws[splitk][M*N] unreduced -- the caller sums them
                                        later with mrg_reduce_batch (C must be contiguous fp32, no bias / accumulate / C2) */
    int* splitk_counters;            /* splitk > 1 only, optional: ceil(M/128) * ceil(N/128) ints, ZERO on entry and left zero on exit.
                                        With them the split is finished inside the launch: every k-slice workgroup stores its slab,
                                        takes a ticket on its tile's counter, and the last arriver sums the slabs in slice order
                                        (deterministic) and applies the epilogue -- no second kernel, the slabs are consumed while
                                        they are still in L2 / Infinity Cache.  Kernel variants without the protocol fall back to
                                        the separate reduction kernel */
    const void* drop_rng;            /* optional device uint64[2] = {seed, step}: with drop_p > 0 the value (acc + bias) is multiplied by a
                                        Philox4x32-10 keep mask / (1 - drop_p) before row_scale and resid are applied -- nn.Dropout after
                                        attn.proj / mlp.fc2 (mirage/utils.py:152,158,172,186-187) fused into the producing GEMM.  Element
                                        (row, col) uses counter ((row * N + col) >> 2, drop_stream, step), key = seed, draw col & 3; it is
                                        kept when the draw >= floor(drop_p * 2^32).  act must be NONE, N %% 4 == 0, no split-K.
                                        mrg_dropout_bwd_bf16 regenerates the same mask for the backward pass. */
    float drop_p; unsigned drop_stream;
} mrg_gemm_desc;
enum { MRG_GEMM_AUTO = 0, MRG_GEMM_GLDS3 = 1, MRG_GEMM_GLDS4 = 2, MRG_GEMM_PIPE128 = 3, MRG_GEMM_WS256 = 7, MRG_GEMM_P8 = 11 };   /* 4, 5, 6, 8, 9, 10, 12, 13 reserved */

int mrg_gemm(const mrg_gemm_desc* d, hipStream_t stream);
/* n <= MRG_MAX_DOMAINS independent forward-layout products (descriptors ds[0 .. n-1], results identical to n mrg_gemm calls) in ONE launch:
 * the patch projections of a step's modalities (mirage/input_adapters.py:101,229 run once per modality) are each a chain of k-tiles on fewer
 * tiles than the chip has CUs; side by side their chains overlap.  Accepted: both operands k-contiguous and 16-byte aligned, K % 64 == 0,
 * N % 4 == 0, act none, optional bias, m_dev, row_map + add_tab / add_row, fp32 or bf16 C; anything else is refused (MRG_ERR_UNSUPPORTED). */
int mrg_gemm_multi(const mrg_gemm_desc* ds, int n, hipStream_t stream);

/* Grouped weight gradients: C_p[M_p, N_p] (fp32, row stride ldc_p, OVERWRITTEN) = A_p^T B_p for p < count, with A_p = dY bf16 [K, M_p]
 * (row stride lda_p) and B_p = X bf16 [K, N_p] (row stride ldb_p): both operands contracted over their leading (token) axis, i.e.
 * the `grad_weight = grad_output^T @ input` mm of nn.Linear's backward (mirage/utils.py:155-157,177,186 under autograd) for several
 * Linears / transformer blocks in ONE launch without split-K.  K (tokens) is common and a multiple of 64; operands 16-B aligned,
 * M_p, N_p, lda_p, ldb_p multiples of 8.  ev_start / ev_stop: optional hipEvent_t recorded on `stream` around the launch.
 * splitk > 1 (groups with too few tiles for the chip): every tile's k-range is cut into `splitk` parts that write fp32 slabs instead of
 * C: problem p's slabs [splitk][M_p * N_p] start at ws + splitk * sum_{q<p} M_q N_q floats (C_p must be contiguous, ldc_p == N_p) and the
 * CALLER sums them into C_p (mrg_reduce_batch). */
#define MRG_WGRAD_GROUP_MAX 48
typedef struct mrg_wgrad_group_desc {
    int count, K;
    const void* A[MRG_WGRAD_GROUP_MAX]; const void* B[MRG_WGRAD_GROUP_MAX]; void* C[MRG_WGRAD_GROUP_MAX];
    int M[MRG_WGRAD_GROUP_MAX], N[MRG_WGRAD_GROUP_MAX], lda[MRG_WGRAD_GROUP_MAX], ldb[MRG_WGRAD_GROUP_MAX], ldc[MRG_WGRAD_GROUP_MAX];
    void* ev_start; void* ev_stop;
    int splitk; float* ws;
    /* optional: AdamW applied by the launch itself.  A weight gradient produced here is final (the tile holds the whole contraction), so
     * the epilogue of every tile also performs torch.optim.AdamW's update (mutils/optim_factory.py:180) of the parameter elements it
     * covers: C_p must then point INTO the gradient arena fuse->g0 (contiguous, ldc_p == N_p), and element i of the arenas p / m / v /
     * shadow_bf16 is updated from element i of g0 exactly as mrg_adamw would (same rounding: csrc/common.h adamw_elem); the gradient is
     * still stored.  hp: device fp32 [4] from mrg_adamw_hp (this step); n_decay: elements [0, n_decay) of the arenas take weight decay;
     * ss[p]: fp32 [8 * tiles of problem p] partial sums of g^2 (per tile and wave) for the gradient norm (mrg_sumsq_finish).  Not with
     * splitk > 1.  NULL: gradients only. */
    const struct mrg_adamw_fuse* fuse;
} mrg_wgrad_group_desc;
typedef struct mrg_adamw_fuse {
    const float* g0; float* p; float* m; float* v; void* shadow_bf16;
    const float* hp; long n_decay; float beta1, beta2, eps;
    float* ss[MRG_WGRAD_GROUP_MAX];
} mrg_adamw_fuse;
int mrg_wgrad_group(const mrg_wgrad_group_desc* d, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * LayerNorm (eps passed by caller; the path uses 1e-6: mirage/model.py:57).  Replaces aten native_layer_norm
 * and its backward (mirage/utils.py:260-261, mirage/output_adapters.py:279-280).
 *   fwd: x fp32 [M,D] -> y bf16 and/or fp32, mean/rstd fp32 [M].  D % 4 == 0, D <= 1024.
 *   bwd: dx = dres + LN'(dy); writes dx fp32 and/or bf16; dgamma/dbeta fp32 [D]; dres_colsum (optional) = column
 *        sum of dres = bias gradient of the Linear whose output entered the residual stream.
 *        partials: workspace fp32 [mrg_ln_bwd_partial_rows(M, D)][3][D] (one row of partial sums per workgroup of the launch).
 */
int mrg_ln_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32, float* mean, float* rstd,
               int M, int D, float eps, hipStream_t stream);
int mrg_ln_bwd_nblocks(int M);
int mrg_ln_bwd_partial_rows(int M, int D);   /* rows of `partials` mrg_ln_bwd writes = its workgroup count (<= mrg_ln_bwd_nblocks(M)) */
/* Batched slice reduction: for every task dst[i] = sum_{s < nslices} src[s * stride + i], i < n (fp32; n and stride multiples
 * of 4, 16-byte aligned).  One or two launches for the whole table: it finishes the split-K slabs of deferred weight-gradient
 * GEMMs (mrg_gemm defer_reduce) and the partial column sums mrg_ln_bwd / mrg_colsum leave behind when called with null
 * outputs (partials [nb][3][D] resp. [nb][N]).  `tasks` is host memory and is copied into the kernel arguments. */
typedef struct mrg_reduce_task { const float* src; float* dst; long n; long stride; int nslices; int reserved; } mrg_reduce_task;
int mrg_reduce_batch(const mrg_reduce_task* tasks, int ntasks, hipStream_t stream);
int mrg_ln_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const float* mean, const float* rstd,
               const float* gamma, const float* dres, float* dx_f32, void* dx_bf16, float* partials, float* dgamma,
               float* dbeta, float* dres_colsum, int M, int D, hipStream_t stream);
/* out[N] = column sums of x [M,N] (bf16 or fp32), optional row gather and device row count;
 * partials: workspace fp32 [mrg_colsum_nblocks(M)][N].  Bias gradients of Linear layers. */
int mrg_colsum_nblocks(int M);
int mrg_colsum(const void* x, int x_dtype, int M, int N, int ld, const int* row_map, const int* m_dev, float* partials,
               float* out, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Row-local chains of the 256-wide output adapters in ONE launch (csrc/rowfuse.hip).  Rows are 256 wide, M % 64 == 0, every
 * operand contiguous and 16-byte aligned, weights bf16 [out][in] as nn.Linear stores them, biases / LayerNorm vectors fp32.
 *   a0 != NULL ("post-attention"):  x = a0 W0^T + b0 (+ resid);  ln = LayerNorm(x; gamma, beta, eps);  h = gelu(ln W1^T + b1);
 *        y = h W2^T + b2 + x.   a0 bf16 [M,256] (attention output), W0 [256,256], W1 [NH,256], W2 [256,NH], NH = 1024:
 *        attn.proj (+ residual), norm2 / out_norm and the Mlp of a decoder Block or of the adapter's cross-attention stage
 *        (mirage/utils.py:155-158,186-187,259-262; mirage/output_adapters.py:279-280).  y32 fp32 [M,256] (+ optional bf16 copy y16).
 *   a0 == NULL ("norm + projection"): x = resid (fp32 [M,256]);  h_out = LayerNorm(x) W1^T + b1, bf16 [M,NH], NH = 256 / 512 / 768:
 *        query_norm + q, context_norm + kv (mirage/output_adapters.py:275-276), norm1 + qkv (mirage/utils.py:177,259).
 *   training side outputs (all or none; ln_out != NULL selects them): x_out fp32 [M,256] (a0 != NULL only), mean / rstd fp32 [M],
 *        ln_out bf16 [M,256], and with a0 != NULL h_out = gelu(.) bf16 [M,NH] and hgrad_out = gelu'(.) IEEE fp16 [M,NH] -- the tensors
 *        mrg_ln_fwd and the MRG_ACT_GELU_SAVE_GRAD epilogue of mrg_gemm leave for the backward pass. */
typedef struct mrg_rowfuse_desc {
    int M, NH; float eps;
    const void* a0; const void* w0; const float* b0; const float* resid; float* x_out;
    const float* gamma; const float* beta; float* mean; float* rstd; void* ln_out;
    const void* w1; const float* b1; void* h_out; void* hgrad_out;
    const void* w2; const float* b2; float* y32; void* y16;
} mrg_rowfuse_desc;
int mrg_rowfuse_fwd(const mrg_rowfuse_desc* d, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused softmax attention.  Replaces F.scaled_dot_product_attention (mirage/utils.py:181 self, :216 cross;
 * scale = head_dim^-0.5, no mask, dropout 0) and its backward.  All tensors bf16; *_bs / *_rs are batch / row
 * strides in elements, the head stride is head_dim (heads are adjacent inside a row), so q/k/v may point into the
 * fused qkv / kv projection outputs and o / d* use the merged-head [B, N, H*hd] layout.
 *   fwd: o, lse [B,H,Nq] fp32 (lse may be NULL for inference)
 *   bwd: needs o, d_o, lse; dsum [B,H,Nq] fp32 workspace; writes dq, dk, dv.
 */
typedef struct mrg_attn_desc {
    const void* q; const void* k; const void* v; void* o; float* lse;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
    int B, H, Nq, Nk, head_dim;
    float scale;
    const void* d_o; long do_bs, do_rs;
    float* dsum;
    void* dq; void* dk; void* dv;
    long dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    /* optional (backward): per-batch column sums of dQ / dK / dV, fp32 [B][..] with row stride colsum_bs floats, entry
     * [b * colsum_bs + h * head_dim + c] -- summed over b they are the bias gradients of the q / k / v projections
     * (colsum_k and colsum_v are used together) */
    float* colsum_q; float* colsum_k; float* colsum_v; long colsum_bs;
    /* optional: dropout on the attention probabilities (mirage/utils.py:181-185, 216-220: SDPA's dropout_p = attn_drop).  drop_rng:
     * device [2] = {seed, step} as in mrg_gemm_desc; drop_p in (0, 1); drop_stream tells the launches of one step apart.  The
     * probabilities of one launch form the matrix [B H Nq, Nk4] (Nk4 = Nk rounded up to 4): element (r, j) takes draw j & 3 of
     * Philox4x32-10(counter = (r Nk4 / 4 + j / 4 as 64 bits, drop_stream, step), key = seed) and is kept (x 1 / (1 - p)) when
     * draw >= floor(p 2^32); forward and backward regenerate the same mask.  This path is plain fp32 FMA code (no reference script
     * sets attn_drop); colsum_* must be NULL with it.  drop_rng = NULL or drop_p = 0: no dropout. */
    const unsigned long long* drop_rng; float drop_p; unsigned int drop_stream;
} mrg_attn_desc;
int mrg_attn_fwd(const mrg_attn_desc* d, hipStream_t stream);
int mrg_attn_bwd(const mrg_attn_desc* d, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Gather-then-embed input adapters (mirage/input_adapters.py:87-110 PatchedInputAdapter.forward, :211-238
 * SemSegInputAdapter.forward, mirage/model.py:384-391 token select + global tokens).
 *   mrg_embed_plan      ids_keep int64 [B,K] -> per-modality compact lists (cap = B*K entries each):
 *                       rows (destination row b*N+k), toks (token inside the modality grid), srcb (sample), counts.
 *                       dom_start: HOST array [ndom+1] of token offsets of the modalities.
 *   mrg_im2col_image    fp32 image [B,C,H,W] -> bf16 rows [cap, C*P*P] for the kept patches of one modality
 *   mrg_im2col_layermap int64 labels [B,Hs,Ws] + class_emb fp32 [ncls,E] -> bf16 rows [cap, E*P*P]
 *   mrg_class_emb_grad  d class_emb [ncls,E] from dA bf16 [cap, E*P*P] (dgrad of the layer-map patch GEMM)
 *   mrg_fill_global     x[b*N + K + g] = global_tokens[g]
 * The projection itself is mrg_gemm with m_dev = &counts[d], row_map = rows, add_tab/add_row = posemb/toks.
 */
/* Input slots.  The kernels that read a raw input batch -- the patch packers, the class-embedding gradient and the masked losses (their
 * targets ARE the inputs) -- take, next to the plain pointer, an optional SLOT: a device location holding the batch's base address,
 * read by the kernel when it runs.  A captured step (hipGraph) thereby consumes each new batch where the data pipeline left it: the
 * host stores the addresses with mrg_set_slots (one launch, values passed by value) and replays; no copy into a staging buffer.
 * With a slot the plain pointer may be NULL; the address stored must satisfy the plain pointer's alignment rule (16 B). */
int mrg_set_slots(const void** slots_dev, const void* const* values_host, int n, hipStream_t stream);   /* n <= 8 */
int mrg_embed_plan(const long* ids_keep, int B, int K, int N, int ndom, const int* dom_start_host, int* rows, int* toks,
                   int* srcb, int* counts, hipStream_t stream);
/* dst bf16 [cap, D] = rows map[0 .. *count_dev) of src bf16 (row stride src_ld elements), zero beyond the count: the gradient rows of
 * one modality's kept tokens, compacted, so that the patch-embedding weight gradient (torch conv2d backward-weight of
 * mirage/input_adapters.py:101,229) is a plain k-strided GEMM over contiguous rows. */
int mrg_gather_rows_bf16(const void* src_bf16, long src_ld, const int* map, const int* count_dev, int cap, int D, void* dst_bf16,
                         hipStream_t stream);
/* d pos_emb of a learnable positional embedding (mirage/input_adapters.py:72,104-108 / :185,232-236): out fp32 [D][T] (the parameter's
 * [1, D, h, w] layout) = for every token t the sum, over the samples that kept it, of the embedding-gradient row dx[row][0 .. D) (row
 * stride ld floats).  rows / toks / srcb / count_dev: one modality's compact lists from mrg_embed_plan; inv_ws: workspace int [B * T].
 * rows = NULL: every sample holds every token at row b * T + t (the adapter's stand-alone forward).  Sample order, no atomics. */
int mrg_posemb_grad(const float* dx, long ld, int D, const int* rows, const int* toks, const int* srcb, const int* count_dev,
                    int cap, int B, int T, int* inv_ws, float* out, hipStream_t stream);
int mrg_im2col_image(const float* img, const int* srcb, const int* toks, const int* count_dev, int cap, int C, int H, int W,
                     int P, void* out_bf16, const void* const* img_slot, hipStream_t stream);
int mrg_im2col_layermap(const long* labels, const float* class_emb, const int* srcb, const int* toks, const int* count_dev,
                        int cap, int ncls, int E, int Hs, int Ws, int P, void* out_bf16, const void* const* labels_slot,
                        hipStream_t stream);
int mrg_class_emb_grad_slices(void);   /* rows of the `partials` workspace: fp32 [slices][ncls][E] */
int mrg_class_emb_grad(const void* dA_bf16, const long* labels, const int* srcb, const int* toks, const int* count_dev,
                       int ncls, int E, int Hs, int Ws, int P, float* partials, float* out, const void* const* labels_slot,
                       hipStream_t stream);
int mrg_fill_global(float* x, const float* global_tokens, int B, int N, int K, int ng, int D, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Decoder query/context assembly (mirage/output_adapters.py:164-246 generate_context_embeddings +
 * get_queries_and_context) and its backward.  ctx = proj_context(encoder tokens) fp32 [B,N,Dd].
 * dom_start is a HOST array; task_emb[i] / d_task_emb[i] device pointers per input modality (may be NULL).
 * partials: fp32 [mrg_dec_assemble_bwd_nblocks()][MRG_MAX_DOMAINS+1][Dd].
 */
typedef struct mrg_dec_assemble_desc {
    const float* ctx; const float* mask_token; const float* task_emb[MRG_MAX_DOMAINS]; const float* pos;
    const long* ids_keep; const long* ids_restore;
    int dom_start[MRG_MAX_DOMAINS + 1];
    int ndom, qdom;
    const float* q_task_emb;
    int B, N, K, T, Tq, Dd;
    float* queries; float* context;
    const float* d_queries; const float* d_context;
    float* d_ctx; float* partials;
    float* d_task_emb[MRG_MAX_DOMAINS]; float* d_mask_token;
} mrg_dec_assemble_desc;
int mrg_dec_assemble_fwd(const mrg_dec_assemble_desc* d, hipStream_t stream);
int mrg_dec_assemble_bwd_nblocks(int B, int N, int Tq);
int mrg_dec_assemble_bwd(const mrg_dec_assemble_desc* d, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Masked losses (mirage/criterion.py:31-51 CE, :87-117 MSE) on image-layout predictions [B,C,H,W] (bf16 or fp32),
 * token mask int64 [B,(H/P)*(W/P)] (1 = masked = counted; NULL = plain mean), target fp32 [B,C,H,W] (MSE) or
 * int64 [B,H,W] (CE).  fwd: loss[1], coef[B] (backward coefficients), per_sample[B] optional.
 * bwd: dpred (same dtype as pred) = gout * dloss/dpred.  partial: fp32 [mrg_loss_partial_floats(B)].
 * mrg_patch_rearrange: tokens [B,g*g,C*P*P] <-> image [B,C,g*P,g*P] (mirage/output_adapters.py:291-294).
 */
typedef struct mrg_loss_desc {
    int kind; const void* pred; int pred_dtype; const void* target; const long* mask;
    int B, C, H, W, P;
    float* partial; float* loss; float* coef; float* per_sample;
    const float* gout; void* dpred;
    const void* const* target_slot;   /* input slot of the target (see "Input slots"); NULL: `target` is read */
    int pred_tokens;                  /* != 0: pred / dpred are in the decoder's token layout [B,(H/P)*(W/P),C*P*P] (inner order c, ph, pw):
                                         the loss reads the out_proj output as it is and mrg_patch_rearrange is not needed */
    float label_smoothing;            /* CE only: F.cross_entropy(label_smoothing=) of MaskedCrossEntropyLoss (criterion.py:22,33); in [0, 1) */
    int norm_pix; float* stats;       /* MSE only: MaskedMSELoss(norm_pix=True) (criterion.py:92-98): the target is normalised per patch over its
                                         C*P*P elements ((t - mean) / sqrt(unbiased var + 1e-6)); stats = workspace fp32 [B*(H/P)*(W/P)*2] that the
                                         forward call fills ({mean, 1/sqrt(var + eps)} per patch) and the backward call reads */
    int nvalid;                       /* mrg_masked_loss_fused with coef == NULL: the caller's GUARANTEE that exactly nvalid samples have masked pixels (the
                                         training step: all B, because fewer tokens are kept than a modality has); the launch then forms each sample's
                                         coefficient 1 / (its masked pixels x nvalid) from its own mask row and no mrg_loss_coef launch is needed */
    long mask_bs;                     /* elements between the mask rows of consecutive samples (0 = tokens per sample: a contiguous mask); lets a
                                         task's columns of the step's [B, all tokens] mask be read in place */
} mrg_loss_desc;
int mrg_loss_partial_floats(int B);
int mrg_masked_loss_fwd(const mrg_loss_desc* d, hipStream_t stream);
int mrg_masked_loss_bwd(const mrg_loss_desc* d, hipStream_t stream);
/* The training step's form of the pair above (the reference differentiates loss.backward() through criterion.py:31-51 / :87-117 after the
 * forward pass; the gradient of a masked mean does not depend on the loss VALUE, only on the mask's pixel counts):
 * mrg_loss_coef forms coef[B] = 1 / (masked pixels of sample b * samples with any) from the token mask alone (ntok tokens of ppt pixels per
 * sample; mask NULL = every pixel) -- the numbers mrg_masked_loss_fwd derives from its pixel counts -- as soon as the masks exist;
 * mrg_masked_loss_fused then reads coef (and gout, NULL = 1) and writes partial AND dpred in ONE pass over the predictions;
 * mrg_masked_loss_finish turns partial into loss / per_sample whenever the caller wants the VALUE (nothing in a training step waits for it:
 * the engine issues it on its side stream).  Bit-identical to mrg_masked_loss_fwd + mrg_masked_loss_bwd. */
int mrg_loss_coef(const long* mask, long mask_bs, int B, int ntok, int ppt, float* coef, hipStream_t stream);
int mrg_masked_loss_fused(const mrg_loss_desc* d, hipStream_t stream);
int mrg_masked_loss_finish(const mrg_loss_desc* d, hipStream_t stream);
int mrg_patch_rearrange(const void* src, void* dst, int dtype, int to_image, int B, int C, int g, int P, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Mask generation (mirage/model.py:168-239) from explicit noise: p fp32 [B,ndom] Dirichlet shares, u fp32 [B,T]
 * per-modality noise (modalities concatenated), g fp32 [B,T].  Outputs int64 like the reference.
 */
int mrg_maskgen(const float* p, const float* u, const float* g, int B, int T, int K, int ndom, const int* dom_start_host,
                long* task_mask, long* ids_keep, long* ids_restore, hipStream_t stream);
/* Same masks with the noise drawn inside the launch (the production path: replaces the reference's Dirichlet draw + two torch.rand
 * calls, mirage/model.py:205-207,216,225): conc fp32 [B,ndom] Dirichlet concentrations; rng = device uint64[3] {seed, step, 0}: the
 * launch reads seed and step, and advances step by one when its last workgroup retires (hipGraph-replayable: every replay draws new
 * masks).  u, g = 24-bit uniforms of Philox4x32-10 streams 0 / 1 over (sample, token quad), p = normalised Gamma(conc) draws of
 * stream 2.  p_out [B,ndom] / u_out / g_out [B,T] (each may be NULL) receive the noise that was drawn: mrg_maskgen fed with them
 * returns the same masks. */
int mrg_maskgen_rng(const float* conc, unsigned long long* rng, int B, int T, int K, int ndom, const int* dom_start_host,
                    long* task_mask, long* ids_keep, long* ids_restore, float* p_out, float* u_out, float* g_out, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Step boundary on flat arenas: global grad L2 norm (mutils/native_scaler.py:46-61), AdamW with decoupled decay on
 * the first n_decay elements (mutils/optim_factory.py:33-92 groups; torch.optim.AdamW update), per-step lr / wd
 * read from device tables (run_pretraining.py:683-688), bf16 shadow weights refreshed in the same pass.
 * step = int[2] on the device: step[0] = iteration counter (schedule index), step[1] = AdamW updates applied so far (torch's
 * per-parameter state['step']: the bias-correction exponent is step[1] + 1).  clip_norm / skip_norm <= 0 disable clipping /
 * skipping; as in native_scaler.py:19-34 clipping takes precedence (skip_norm is ignored beside it) and a skipped iteration
 * (grad_norm >= skip_norm) changes neither weights nor moments.  mrg_step_inc ends the iteration: step[0] += 1 always,
 * step[1] += 1 unless the iteration was skipped (same decision from the same grad_norm / clip_norm / skip_norm).
 */
typedef struct mrg_adamw_desc {
    float* p; const float* g; float* m; float* v; void* shadow_bf16;
    long n, n_decay;
    const float* lr_table; const float* wd_table; const int* step; int table_len;
    float lr_scale, beta1, beta2, eps;
    const float* grad_norm; float clip_norm, skip_norm;
    float* sumsq_out;   /* optional fp32 [mrg_adamw_nblocks(n)]: per-block partial sums of g^2 over this launch's range -- the global
                           gradient norm for logging (native_scaler.py:46-61) then costs no extra pass over the gradients:
                           mrg_sumsq_finish(all partials) = sqrt(sum).  Rejected together with grad_norm and clip_norm > 0 or skip_norm > 0
                           (clipping / skipping need the norm BEFORE the update: use mrg_grad_norm first). */
} mrg_adamw_desc;
int mrg_adamw_nblocks(long n);
int mrg_sumsq_finish(const float* partials, int n, float* out, hipStream_t stream);
int mrg_sumsq_nblocks(long n);
int mrg_grad_norm(const float* g, long n, float* partials, float* out, hipStream_t stream);
int mrg_adamw(const mrg_adamw_desc* d, hipStream_t stream);
/* hp_out fp32 [4] = {lr, wd, 1 / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} of the CURRENT step, as mrg_adamw derives them from the tables and
 * the step counters of `d` (only those fields are read): what mrg_wgrad_group's fused update (mrg_wgrad_group_desc.fuse) consumes. */
int mrg_adamw_hp(const mrg_adamw_desc* d, float* hp_out, hipStream_t stream);
int mrg_step_inc(int* step, const float* grad_norm, float clip_norm, float skip_norm, hipStream_t stream);
int mrg_cast_bf16(const float* src, void* dst_bf16, long n, hipStream_t stream);
/* dst_bf16[r, :] = src[r, :] * row_scale[r] for an [M, D] fp32 matrix (D % 4 == 0): the gradient entering a stochastic-depth
 * branch (backward of drop_path, mirage/utils.py:103-118). */
int mrg_row_scale_bf16(const float* src, const float* row_scale, void* dst_bf16, long M, int D, hipStream_t stream);
/* dst_bf16[r, c] = src[r, c] * keep(r, c) / (1 - drop_p) (* row_scale[r] when given): the gradient entering a branch that ended in
 * nn.Dropout (mirage/utils.py:158,187), with the mask mrg_gemm applied in the forward pass regenerated from (drop_rng = {seed, step},
 * drop_stream) -- nothing is stored between the passes.  D % 4 == 0 and D == the N of that GEMM. */
int mrg_dropout_bwd_bf16(const float* src, const float* row_scale, void* dst_bf16, long M, int D, const void* drop_rng, float drop_p,
                         unsigned drop_stream, hipStream_t stream);

/* ---- F2: GPU-side pretraining augmentation (mutils/datasets_pretrain.py:18-83, DataAugmentationForMIRAGE.__call__) --------------
 * One fused pass over a batch of uint8 samples [B, H, W]: horizontal flip (:41-42), intensity shift + clip to [0,1] (:43-48;
 * image tasks), torchvision TF.affine(BILINEAR, fill=0) (:57-64; label maps take the same bilinear path and are rounded to
 * integers, as the reference does).  params[b] = {t00, t01, t02, t10, t11, t12, shift, flip}: t = the inverse affine matrix of
 * torchvision's _get_inverse_affine_matrix in centred pixel coordinates with row 0 divided by 0.5*W and row 1 by 0.5*H
 * (what _gen_affine_grid feeds its bmm); shift = signed intensity shift (0 = none); flip != 0 mirrors the source along W.
 * do_affine = 0 skips the resampling (args.affine false, :56).  Image output: fp32 [B, H, W] in [0,1]; label output: int64. */
int mrg_augment_image(const uint8_t* src_u8, float* dst, const float* params, int B, int H, int W, int do_affine, hipStream_t stream);
int mrg_augment_label(const uint8_t* src_u8, long* dst, const float* params, int B, int H, int W, int do_affine, hipStream_t stream);
/* TF.resize branch of the same transform (mutils/datasets_pretrain.py:66-81), applied after the affine step when the raw sample
 * size differs from args.input_size[task]: images fp32 [B,Hi,Wi] -> [B,Ho,Wo] bilinear with antialiasing (torchvision resizes
 * tensors with antialias=True: aten _upsample_bilinear2d_aa), label maps int64 nearest (aten upsample_nearest2d). */
int mrg_resize_image(const float* src, float* dst, int B, int Hi, int Wi, int Ho, int Wo, hipStream_t stream);
int mrg_resize_label(const long* src, long* dst, int B, int Hi, int Wi, int Ho, int Wo, hipStream_t stream);

/* Profiling aid: *slot = wall_clock64() (constant 100 MHz counter) when `stream` reaches this point; capturable into a hipGraph. */
int mrg_debug_timestamp(unsigned long long* slot, hipStream_t stream);
/* profiling aid: one wave stores {wall_clock64 (100 MHz), clock64 (shader clock)} pairs into buf[2 * samples] every `period` wall ticks
 * while other streams work: the clock the chip runs at, by phase of a step (tools/step_clock.py) */
int mrg_debug_clock_monitor(unsigned long long* buf, int samples, int period, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MIRAGE_HIP_H */
