// Attention with dropout on the probabilities (mirage/utils.py:181-185, 216-220: F.scaled_dot_product_attention(dropout_p = attn_drop)).
// No script of the reference sets attn_drop (run_pretraining.py builds every model with 0), so this is the plain, general path next to
// the tiled MFMA kernels of attention.hip: fp32 FMA arithmetic on bf16 operands, any Nq / Nk, head_dim 32 or 64, no LDS tiling --
// mrg_attn_fwd / mrg_attn_bwd route here when the descriptor carries a dropout state.
//
//   P = softmax(scale Q K^T),  F = mask / (1 - p),  O = (P . F) V                       (torch SDPA: dropout of the softmax output)
//   dV = (P . F)^T dO,  dP = (dO V^T) . F,  dS = P . (dP - rowsum(dO . O)),  dQ = scale dS K,  dK = scale dS^T Q
//   (rowsum(P . dP) = rowsum((P . F) . (dO V^T)) = rowsum(dO . O): the usual delta survives the mask)
//
// Mask: the probabilities of one launch are the matrix [B H Nq, Nk4] (Nk4 = Nk rounded up to 4, so that a Philox call never straddles
// rows); element (r, j) takes draw j & 3 of Philox4x32-10(counter = (r Nk4 / 4 + j / 4 as 64 bits, stream, step), key = seed) and is
// kept when draw >= floor(p 2^32) -- common.h dropout_apply4's rule, restated by oracle/mirage_oracle.py attention_dropout_factor.
//
// Thread layout of all three kernels: a workgroup of 256 threads owns 16 rows (queries, or keys in the dK / dV kernel); the 16 threads
// of a row stride over the other axis and combine their partial results with 16-lane butterfly shuffles.
#include "common.h"
#include "mirage_hip.h"

namespace {

struct DropAttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
    int B, H, Nq, Nk;
    float scale;
    const bf16_t* d_o; long do_bs, do_rs;
    float* dsum;
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    long dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    MrgDropout drop;
};

template <int HD>
__device__ __forceinline__ void load_row(const bf16_t* p, float (&r)[HD]) {
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
        const uint4 w = *reinterpret_cast<const uint4*>(p + c);
        const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[c + 2 * e] = __uint_as_float(u[e] << 16); r[c + 2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
    }
}
template <int HD>
__device__ __forceinline__ float dot_row(const float (&a)[HD], const bf16_t* p) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
        const uint4 w = *reinterpret_cast<const uint4*>(p + c);
        const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { s = fmaf(a[c + 2 * e], __uint_as_float(u[e] << 16), s); s = fmaf(a[c + 2 * e + 1], __uint_as_float(u[e] & 0xffff0000u), s); }
    }
    return s;
}
template <int HD>
__device__ __forceinline__ void axpy_row(float (&acc)[HD], float w, const bf16_t* p) {
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
        const uint4 q = *reinterpret_cast<const uint4*>(p + c);
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[c + 2 * e] = fmaf(w, __uint_as_float(u[e] << 16), acc[c + 2 * e]); acc[c + 2 * e + 1] = fmaf(w, __uint_as_float(u[e] & 0xffff0000u), acc[c + 2 * e + 1]); }
    }
}
// mask factor of probability (row r of the [B H Nq, Nk4] matrix, key j)
__device__ __forceinline__ float drop_factor(const MrgDropout& d, unsigned long long seed, unsigned long long step, long r, int j, int nk4q) {
    const unsigned long long q = (unsigned long long)r * (unsigned long long)nk4q + (unsigned long long)(j >> 2);
    uint32_t out[4];
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), d.stream, (uint32_t)step, (uint32_t)seed, (uint32_t)(seed >> 32), out);
    const uint32_t draw = (j & 3) == 0 ? out[0] : (j & 3) == 1 ? out[1] : (j & 3) == 2 ? out[2] : out[3];
    return draw >= d.thresh ? d.scale : 0.f;
}
// sum over the 16 threads of a row (lanes l, l ^ 1, l ^ 2, l ^ 4, l ^ 8 share a row)
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int HD>
__device__ __forceinline__ void store_row_slice(bf16_t* p, const float (&acc)[HD], int sl) {
    // after the butterfly every thread of the row holds the full sums; thread `sl` (0..15) stores columns [HD / 16 * sl, + HD / 16)
    constexpr int W = HD / 16;
#pragma unroll
    for (int c = 0; c < HD; ++c)
        if (c / W == sl) p[c] = f2bf(acc[c]);
}

template <int HD>
__global__ __launch_bounds__(256) void attn_drop_fwd_kernel(const DropAttnArgs a) {
    const int h = blockIdx.y, b = blockIdx.z;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sl = threadIdx.x & 15;
    const bool valid = i < a.Nq;
    const int ii = valid ? i : a.Nq - 1;
    const unsigned long long seed = a.drop.rng[0], step = a.drop.rng[1];
    const int nk4q = (a.Nk + 3) >> 2;
    const long r = ((long)b * a.H + h) * a.Nq + ii;
    float qv[HD];
    load_row<HD>(a.q + (long)b * a.q_bs + (long)ii * a.q_rs + h * HD, qv);
    const bf16_t* kp = a.k + (long)b * a.k_bs + h * HD;
    const bf16_t* vp = a.v + (long)b * a.v_bs + h * HD;
    // pass 1: log-sum-exp of the row
    float m = -INFINITY, l = 0.f;
    for (int j = sl; j < a.Nk; j += 16) {
        const float s = a.scale * dot_row<HD>(qv, kp + (long)j * a.k_rs);
        const float mn = fmaxf(m, s);
        l = l * __expf(m - mn) + __expf(s - mn);
        m = mn;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
        const float mn = fmaxf(m, m2);
        const float e1 = (m == -INFINITY) ? 0.f : __expf(m - mn), e2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
        l = l * e1 + l2 * e2;
        m = mn;
    }
    const float lse = m + __logf(l);
    // pass 2: O = (P . F) V
    float acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    for (int j = sl; j < a.Nk; j += 16) {
        const float s = a.scale * dot_row<HD>(qv, kp + (long)j * a.k_rs);
        const float p = __expf(s - lse) * drop_factor(a.drop, seed, step, r, j, nk4q);
        axpy_row<HD>(acc, p, vp + (long)j * a.v_rs);
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = row16_sum(acc[c]);
    if (valid) {
        store_row_slice<HD>(a.o + (long)b * a.o_bs + (long)i * a.o_rs + h * HD, acc, sl);
        if (a.lse && sl == 0) a.lse[((long)b * a.H + h) * a.Nq + i] = lse;
    }
}

// dQ (and dsum[b, h, i] = dO_i . O_i for the dK / dV kernel that follows)
template <int HD>
__global__ __launch_bounds__(256) void attn_drop_bwd_dq_kernel(const DropAttnArgs a) {
    const int h = blockIdx.y, b = blockIdx.z;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sl = threadIdx.x & 15;
    const bool valid = i < a.Nq;
    const int ii = valid ? i : a.Nq - 1;
    const unsigned long long seed = a.drop.rng[0], step = a.drop.rng[1];
    const int nk4q = (a.Nk + 3) >> 2;
    const long r = ((long)b * a.H + h) * a.Nq + ii;
    float qv[HD], dov[HD];
    load_row<HD>(a.q + (long)b * a.q_bs + (long)ii * a.q_rs + h * HD, qv);
    load_row<HD>(a.d_o + (long)b * a.do_bs + (long)ii * a.do_rs + h * HD, dov);
    const float dsum = dot_row<HD>(dov, a.o + (long)b * a.o_bs + (long)ii * a.o_rs + h * HD);
    const float lse = a.lse[r];
    const bf16_t* kp = a.k + (long)b * a.k_bs + h * HD;
    const bf16_t* vp = a.v + (long)b * a.v_bs + h * HD;
    float acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    for (int j = sl; j < a.Nk; j += 16) {
        const bf16_t* kr = kp + (long)j * a.k_rs;
        const float p = __expf(a.scale * dot_row<HD>(qv, kr) - lse);
        const float dp = dot_row<HD>(dov, vp + (long)j * a.v_rs) * drop_factor(a.drop, seed, step, r, j, nk4q);
        axpy_row<HD>(acc, a.scale * p * (dp - dsum), kr);
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = row16_sum(acc[c]);
    if (valid) {
        store_row_slice<HD>(a.dq + (long)b * a.dq_bs + (long)i * a.dq_rs + h * HD, acc, sl);
        if (sl == 0) a.dsum[r] = dsum;
    }
}

// dK and dV of 16 keys per workgroup (two sweeps over the queries: the accumulators of both plus both key rows do not fit the registers)
template <int HD>
__global__ __launch_bounds__(256) void attn_drop_bwd_dkv_kernel(const DropAttnArgs a) {
    const int h = blockIdx.y, b = blockIdx.z;
    const int j = blockIdx.x * 16 + (threadIdx.x >> 4), sl = threadIdx.x & 15;
    const bool valid = j < a.Nk;
    const int jj = valid ? j : a.Nk - 1;
    const unsigned long long seed = a.drop.rng[0], step = a.drop.rng[1];
    const int nk4q = (a.Nk + 3) >> 2;
    const long r0 = ((long)b * a.H + h) * a.Nq;
    const bf16_t* qp = a.q + (long)b * a.q_bs + h * HD;
    const bf16_t* dop = a.d_o + (long)b * a.do_bs + h * HD;
    float kv[HD], acc[HD];
    load_row<HD>(a.k + (long)b * a.k_bs + (long)jj * a.k_rs + h * HD, kv);
    // dV_j = sum_i P_ij F_ij dO_i
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    for (int i = sl; i < a.Nq; i += 16) {
        const float p = __expf(a.scale * dot_row<HD>(kv, qp + (long)i * a.q_rs) - a.lse[r0 + i]);
        axpy_row<HD>(acc, p * drop_factor(a.drop, seed, step, r0 + i, jj, nk4q), dop + (long)i * a.do_rs);
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = row16_sum(acc[c]);
    if (valid) store_row_slice<HD>(a.dv + (long)b * a.dv_bs + (long)j * a.dv_rs + h * HD, acc, sl);
    // dK_j = scale sum_i P_ij (F_ij dO_i . V_j - dsum_i) Q_i
    float vv[HD];
    load_row<HD>(a.v + (long)b * a.v_bs + (long)jj * a.v_rs + h * HD, vv);
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    for (int i = sl; i < a.Nq; i += 16) {
        const bf16_t* qr = qp + (long)i * a.q_rs;
        const float p = __expf(a.scale * dot_row<HD>(kv, qr) - a.lse[r0 + i]);
        const float dp = dot_row<HD>(vv, dop + (long)i * a.do_rs) * drop_factor(a.drop, seed, step, r0 + i, jj, nk4q);
        axpy_row<HD>(acc, a.scale * p * (dp - a.dsum[r0 + i]), qr);
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = row16_sum(acc[c]);
    if (valid) store_row_slice<HD>(a.dk + (long)b * a.dk_bs + (long)j * a.dk_rs + h * HD, acc, sl);
}

void fill(DropAttnArgs& a, const mrg_attn_desc* d) {
    a.q = (const bf16_t*)d->q; a.k = (const bf16_t*)d->k; a.v = (const bf16_t*)d->v; a.o = (bf16_t*)d->o; a.lse = d->lse;
    a.q_bs = d->q_bs; a.q_rs = d->q_rs; a.k_bs = d->k_bs; a.k_rs = d->k_rs; a.v_bs = d->v_bs; a.v_rs = d->v_rs;
    a.o_bs = d->o_bs; a.o_rs = d->o_rs;
    a.B = d->B; a.H = d->H; a.Nq = d->Nq; a.Nk = d->Nk; a.scale = d->scale;
    a.d_o = (const bf16_t*)d->d_o; a.do_bs = d->do_bs; a.do_rs = d->do_rs; a.dsum = d->dsum;
    a.dq = (bf16_t*)d->dq; a.dk = (bf16_t*)d->dk; a.dv = (bf16_t*)d->dv;
    a.dq_bs = d->dq_bs; a.dq_rs = d->dq_rs; a.dk_bs = d->dk_bs; a.dk_rs = d->dk_rs; a.dv_bs = d->dv_bs; a.dv_rs = d->dv_rs;
    a.drop.rng = d->drop_rng;
    double t = (double)d->drop_p * 4294967296.0;
    a.drop.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    a.drop.scale = 1.0f / (1.0f - d->drop_p);
    a.drop.stream = d->drop_stream;
}

}  // namespace

namespace mrg_detail {

int attn_drop_fwd(const mrg_attn_desc* d, hipStream_t stream) {
    MRG_REQUIRE(d->drop_p > 0.f && d->drop_p < 1.f, MRG_ERR_ARG, "mrg_attn_fwd: drop_p = %g must be in (0, 1)", (double)d->drop_p);
    DropAttnArgs a;
    fill(a, d);
    dim3 grid(cdiv(a.Nq, 16), a.H, a.B), block(256);
    if (d->head_dim == 64) hipLaunchKernelGGL(attn_drop_fwd_kernel<64>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(attn_drop_fwd_kernel<32>, grid, block, 0, stream, a);
    return mrg_check_launch("mrg_attn_fwd/dropout");
}

int attn_drop_bwd(const mrg_attn_desc* d, hipStream_t stream) {
    MRG_REQUIRE(d->drop_p > 0.f && d->drop_p < 1.f, MRG_ERR_ARG, "mrg_attn_bwd: drop_p = %g must be in (0, 1)", (double)d->drop_p);
    MRG_REQUIRE(d->colsum_q == nullptr && d->colsum_k == nullptr && d->colsum_v == nullptr, MRG_ERR_UNSUPPORTED,
                "mrg_attn_bwd: the dropout path does not form the bias column sums (pass colsum_* = NULL)");
    DropAttnArgs a;
    fill(a, d);
    dim3 block(256);
    {
        dim3 grid(cdiv(a.Nq, 16), a.H, a.B);
        if (d->head_dim == 64) hipLaunchKernelGGL(attn_drop_bwd_dq_kernel<64>, grid, block, 0, stream, a);
        else hipLaunchKernelGGL(attn_drop_bwd_dq_kernel<32>, grid, block, 0, stream, a);
        const int rc = mrg_check_launch("mrg_attn_bwd/dropout dq");
        if (rc) return rc;
    }
    dim3 grid(cdiv(a.Nk, 16), a.H, a.B);
    if (d->head_dim == 64) hipLaunchKernelGGL(attn_drop_bwd_dkv_kernel<64>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(attn_drop_bwd_dkv_kernel<32>, grid, block, 0, stream, a);
    return mrg_check_launch("mrg_attn_bwd/dropout dkv");
}

}  // namespace mrg_detail
