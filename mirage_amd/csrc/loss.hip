// Masked reconstruction losses (mirage/criterion.py:31-51 MaskedCrossEntropyLoss, :87-117 MaskedMSELoss) and the
// token -> image rearrangement of the decoder output (mirage/output_adapters.py:291-294), forward + backward.
//
// Reference semantics: per-pixel loss (MSE averaged over channels / 13-way softmax CE) times the token mask
// nearest-upsampled to pixels, per-sample sum / per-sample masked-pixel count, then nanmean over samples; a mask
// with no masked token at all gives 0.  The mask is per token, so no upsampled mask or full-size temporaries are
// ever built: stage 1 reduces each (sample, row-chunk) deterministically, stage 2 finishes the batch statistics and
// emits the per-sample backward coefficient, the backward kernel writes dpred in one pass.
#include "common.h"
#include "mirage_hip.h"

namespace {

constexpr int LOSS_CHUNKS = 32;  // partial sums per sample

template <typename T> __device__ __forceinline__ float ldf(const T* p, long i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, long i) { return bf2f(p[i]); }
template <typename T> __device__ __forceinline__ void stf(T* p, long i, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, long i, float v) { p[i] = f2bf(v); }

// element (b, c, y, x) of the predictions: image layout [B, C, H, W], or (tok != 0) the decoder's own token layout
// [B, (H/P)*(W/P), C*P*P] with (c, ph, pw) inside a token (mirage/output_adapters.py:291-294 is then never materialised)
__device__ __forceinline__ long pred_index(int tok, int b, int c, int y, int x, int C, int H, int W, int P) {
    if (!tok) return (((long)b * C + c) * H + y) * W + x;
    const int nw = W / P, ty = y / P, tx = x / P;
    return (((long)b * (H / P) + ty) * nw + tx) * ((long)C * P * P) + (long)c * P * P + (y - ty * P) * P + (x - tx * P);
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0) for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    __syncthreads();
    return s;  // valid in thread 0
}

// norm_pix (criterion.py:92-98): the MSE target is normalised per patch over its (ph pw c) elements -- (t - mean) / sqrt(var + 1e-6) with
// torch.var's unbiased estimate.  One workgroup per (token, sample) writes stats[(b ntok + tok) 2] = {mean, 1 / sqrt(var + eps)}.
__global__ __launch_bounds__(256) void patch_stats_kernel(const float* __restrict__ target, const void* const* __restrict__ tslot, int C, int H, int W,
                                                          int P, float* __restrict__ stats) {
    if (tslot) target = reinterpret_cast<const float*>(*tslot);
    __shared__ float red[4];
    __shared__ float s_mean;
    const int tk = blockIdx.x, b = blockIdx.y, nw = W / P, ntok = (H / P) * nw;
    const int ty = tk / nw, tx = tk - ty * nw;
    const int n = C * P * P;
    const long HW = (long)H * W;
    auto at = [&](int e) { const int c = e / (P * P), r = e - c * P * P, py = r / P, px = r - py * P;
                           return target[((long)b * C + c) * HW + (long)(ty * P + py) * W + tx * P + px]; };
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) s += at(e);
    const float ts = block_sum(s, red);
    if (threadIdx.x == 0) s_mean = ts / (float)n;
    __syncthreads();
    const float mean = s_mean;
    float q = 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) { const float d = at(e) - mean; q += d * d; }
    const float tq = block_sum(q, red);
    if (threadIdx.x == 0) {
        stats[((long)b * ntok + tk) * 2 + 0] = mean;
        stats[((long)b * ntok + tk) * 2 + 1] = rsqrtf(tq / (float)(n - 1) + 1e-6f);
    }
}

// grid (LOSS_CHUNKS, B); partial [B][LOSS_CHUNKS][2] = (sum of masked per-pixel loss, masked pixel count)
template <typename T, bool CE>
__global__ __launch_bounds__(256) void loss_fwd_kernel(const T* __restrict__ pred, const void* __restrict__ target,
                                                       const void* const* __restrict__ tslot,
                                                       const long* __restrict__ mask, int B, int C, int H, int W, int P, int tok,
                                                       float smooth, const float* __restrict__ stats, float* __restrict__ partial) {
    if (tslot) target = *tslot;                              // input slot: the batch is read where it lies (mirage_hip.h, "Input slots")
    __shared__ float red[4];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int nw = W / P;
    const long HW = (long)H * W;
    float s = 0.f, n = 0.f;
    for (long px = (long)chunk * 256 + threadIdx.x; px < HW; px += (long)LOSS_CHUNKS * 256) {
        const int y = (int)(px / W), x = (int)(px - (long)y * W);
        const float m = mask ? (float)mask[(long)b * (H / P) * nw + (y / P) * nw + x / P] : 1.f;
        if (m == 0.f) continue;
        float l;
        if (CE) {
            const int t = (int)reinterpret_cast<const long*>(target)[(long)b * HW + px];
            float mx = -INFINITY, zt = 0.f, zs = 0.f;
            for (int c = 0; c < C; ++c) { const float z = ldf<T>(pred, pred_index(tok, b, c, y, x, C, H, W, P)); mx = fmaxf(mx, z); zt = (c == t) ? z : zt; zs += z; }
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += __expf(ldf<T>(pred, pred_index(tok, b, c, y, x, C, H, W, P)) - mx);
            // F.cross_entropy(label_smoothing = e) = (1 - e) nll + e mean_c(-log p_c) = lse - (1 - e) z_t - e mean_c z_c  (criterion.py:33)
            l = mx + __logf(se) - (1.f - smooth) * zt - smooth * zs / (float)C;
        } else {
            float q = 0.f;
            const float* st = stats ? stats + ((long)b * (H / P) * nw + (y / P) * nw + x / P) * 2 : nullptr;
            for (int c = 0; c < C; ++c) {
                float tv = reinterpret_cast<const float*>(target)[((long)b * C + c) * HW + px];
                if (st) tv = (tv - st[0]) * st[1];
                const float d = ldf<T>(pred, pred_index(tok, b, c, y, x, C, H, W, P)) - tv;
                q += d * d;
            }
            l = q / (float)C;
        }
        s += l * m;
        n += m;
    }
    const float ts = block_sum(s, red);
    const float tn = block_sum(n, red);
    if (threadIdx.x == 0) {
        partial[((long)b * LOSS_CHUNKS + chunk) * 2 + 0] = ts;
        partial[((long)b * LOSS_CHUNKS + chunk) * 2 + 1] = tn;
    }
}

// single block: per-sample loss, nanmean, backward coefficients coef[b] = 1 / (count_b * n_valid)
__global__ void loss_finish_kernel(const float* __restrict__ partial, int B, float* __restrict__ loss, float* __restrict__ coef,
                                   float* __restrict__ per_sample) {
    __shared__ float sl[1024], sn[1024];
    const int t = threadIdx.x;
    float tot = 0.f;
    int nvalid = 0;
    // B <= 1024 per pass; loop passes for larger B
    float total_cnt = 0.f;
    for (int base = 0; base < B; base += 1024) {
        const int b = base + t;
        float s = 0.f, n = 0.f;
        if (b < B)
            for (int c = 0; c < LOSS_CHUNKS; ++c) { s += partial[((long)b * LOSS_CHUNKS + c) * 2]; n += partial[((long)b * LOSS_CHUNKS + c) * 2 + 1]; }
        sl[t] = s; sn[t] = n;
        __syncthreads();
        if (t == 0) {
            for (int i = 0; i < min(1024, B - base); ++i) {
                total_cnt += sn[i];
                if (sn[i] > 0.f) { tot += sl[i] / sn[i]; ++nvalid; }
            }
        }
        __syncthreads();
    }
    __shared__ float s_nvalid;
    if (t == 0) {
        // criterion.py:36,103: mask.sum() == 0 -> 0 ; otherwise nanmean over samples (0/0 samples dropped)
        loss[0] = (total_cnt > 0.f && nvalid > 0) ? tot / (float)nvalid : 0.f;
        s_nvalid = (float)nvalid;
    }
    __syncthreads();
    for (int b = t; b < B; b += blockDim.x) {
        float n = 0.f, s = 0.f;
        for (int c = 0; c < LOSS_CHUNKS; ++c) { s += partial[((long)b * LOSS_CHUNKS + c) * 2]; n += partial[((long)b * LOSS_CHUNKS + c) * 2 + 1]; }
        coef[b] = (n > 0.f && s_nvalid > 0.f) ? 1.f / (n * s_nvalid) : 0.f;
        if (per_sample) per_sample[b] = n > 0.f ? s / n : NAN;
    }
}

template <typename T, bool CE>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const T* __restrict__ pred, const void* __restrict__ target,
                                                       const void* const* __restrict__ tslot,
                                                       const long* __restrict__ mask, const float* __restrict__ coef,
                                                       const float* __restrict__ gout, int B, int C, int H, int W, int P, int tok,
                                                       float smooth, const float* __restrict__ stats, T* __restrict__ dpred) {
    if (tslot) target = *tslot;
    const long HW = (long)H * W;
    const long total = (long)B * HW;
    const int nw = W / P;
    const float go = gout ? gout[0] : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const long px = i - (long)b * HW;
        const int y = (int)(px / W), x = (int)(px - (long)y * W);
        const float m = mask ? (float)mask[(long)b * (H / P) * nw + (y / P) * nw + x / P] : 1.f;
        const float k = go * coef[b] * m;
        if (CE) {
            if (k == 0.f) { for (int c = 0; c < C; ++c) stf<T>(dpred, pred_index(tok, b, c, y, x, C, H, W, P), 0.f); continue; }
            const int t = (int)reinterpret_cast<const long*>(target)[(long)b * HW + px];
            float mx = -INFINITY;
            for (int c = 0; c < C; ++c) mx = fmaxf(mx, ldf<T>(pred, pred_index(tok, b, c, y, x, C, H, W, P)));
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += __expf(ldf<T>(pred, pred_index(tok, b, c, y, x, C, H, W, P)) - mx);
            const float inv = 1.f / se;
            for (int c = 0; c < C; ++c) {
                const long j = pred_index(tok, b, c, y, x, C, H, W, P);
                const float p = __expf(ldf<T>(pred, j) - mx) * inv;
                stf<T>(dpred, j, k * (p - (c == t ? 1.f - smooth : 0.f) - smooth / (float)C));
            }
        } else {
            const float kk = k * 2.f / (float)C;
            const float* st = stats ? stats + ((long)b * (H / P) * nw + (y / P) * nw + x / P) * 2 : nullptr;
            for (int c = 0; c < C; ++c) {
                const long j = ((long)b * C + c) * HW + px, jp = pred_index(tok, b, c, y, x, C, H, W, P);
                float tv = reinterpret_cast<const float*>(target)[j];
                if (st) tv = (tv - st[0]) * st[1];
                stf<T>(dpred, jp, kk * (ldf<T>(pred, jp) - tv));
            }
        }
    }
}

// ---- single-channel bf16 MSE fast path: 8 consecutive pixels (16 B of predictions, 32 B of targets) per thread
__global__ __launch_bounds__(256) void mse1_fwd_vec_kernel(const bf16_t* __restrict__ pred, const float* __restrict__ target,
                                                           const void* const* __restrict__ tslot,
                                                           const long* __restrict__ mask, int B, int H, int W, int P, int tok,
                                                           float* __restrict__ partial) {
    if (tslot) target = static_cast<const float*>(*tslot);
    __shared__ float red[4];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int nw = W / P;
    const long HW8 = (long)H * W / 8;
    float s = 0.f, n = 0.f;
    for (long v = (long)chunk * 256 + threadIdx.x; v < HW8; v += (long)LOSS_CHUNKS * 256) {
        const long px = v * 8;
        const int y = (int)(px / W), x = (int)(px - (long)y * W);
        const float m = mask ? (float)mask[(long)b * (H / P) * nw + (y / P) * nw + x / P] : 1.f;
        if (m == 0.f) continue;
        const uint4 p = *reinterpret_cast<const uint4*>(pred + pred_index(tok, b, 0, y, x, 1, H, W, P));     // P % 8 == 0: 8 pixels stay inside a patch row
        const float4 t0 = *reinterpret_cast<const float4*>(target + (long)b * H * W + px);
        const float4 t1 = *reinterpret_cast<const float4*>(target + (long)b * H * W + px + 4);
        const float d0 = __uint_as_float(p.x << 16) - t0.x, d1 = __uint_as_float(p.x & 0xffff0000u) - t0.y;
        const float d2 = __uint_as_float(p.y << 16) - t0.z, d3 = __uint_as_float(p.y & 0xffff0000u) - t0.w;
        const float d4 = __uint_as_float(p.z << 16) - t1.x, d5 = __uint_as_float(p.z & 0xffff0000u) - t1.y;
        const float d6 = __uint_as_float(p.w << 16) - t1.z, d7 = __uint_as_float(p.w & 0xffff0000u) - t1.w;
        s += ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
        n += 8.f;
    }
    const float ts = block_sum(s, red);
    const float tn = block_sum(n, red);
    if (threadIdx.x == 0) {
        partial[((long)b * LOSS_CHUNKS + chunk) * 2 + 0] = ts;
        partial[((long)b * LOSS_CHUNKS + chunk) * 2 + 1] = tn;
    }
}

__global__ __launch_bounds__(256) void mse1_bwd_vec_kernel(const bf16_t* __restrict__ pred, const float* __restrict__ target,
                                                           const void* const* __restrict__ tslot,
                                                           const long* __restrict__ mask, const float* __restrict__ coef,
                                                           const float* __restrict__ gout, int B, int H, int W, int P, int tok,
                                                           bf16_t* __restrict__ dpred) {
    if (tslot) target = static_cast<const float*>(*tslot);
    const long HW = (long)H * W;
    const long total8 = (long)B * HW / 8;
    const int nw = W / P;
    const float go = gout ? gout[0] : 1.f;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < total8; v += (long)gridDim.x * blockDim.x) {
        const long i = v * 8;
        const int b = (int)(i / HW);
        const long px = i - (long)b * HW;
        const int y = (int)(px / W), x = (int)(px - (long)y * W);
        const float m = mask ? (float)mask[(long)b * (H / P) * nw + (y / P) * nw + x / P] : 1.f;
        const float k = 2.f * go * coef[b] * m;
        uint4 o = make_uint4(0, 0, 0, 0);
        const long ip = pred_index(tok, b, 0, y, x, 1, H, W, P);
        if (k != 0.f) {
            const uint4 p = *reinterpret_cast<const uint4*>(pred + ip);
            const float4 t0 = *reinterpret_cast<const float4*>(target + i);
            const float4 t1 = *reinterpret_cast<const float4*>(target + i + 4);
            o.x = pack_bf2(k * (__uint_as_float(p.x << 16) - t0.x), k * (__uint_as_float(p.x & 0xffff0000u) - t0.y));
            o.y = pack_bf2(k * (__uint_as_float(p.y << 16) - t0.z), k * (__uint_as_float(p.y & 0xffff0000u) - t0.w));
            o.z = pack_bf2(k * (__uint_as_float(p.z << 16) - t1.x), k * (__uint_as_float(p.z & 0xffff0000u) - t1.y));
            o.w = pack_bf2(k * (__uint_as_float(p.w << 16) - t1.z), k * (__uint_as_float(p.w & 0xffff0000u) - t1.w));
        }
        *reinterpret_cast<uint4*>(dpred + ip) = o;
    }
}

// tokens [B, g*g, C*P*P] <-> image [B, C, g*P, g*P]   (inner token order c, ph, pw).  V consecutive x per thread:
// both layouts are contiguous along pw, so every access is a V-element vector (16 B when P % V == 0).
template <typename T, bool TO_IMAGE, int V>
__global__ void patch_rearrange_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int C, int g, int P) {
    const int S = g * P;
    const int SV = S / V;
    const long total = (long)B * C * S * SV;
    struct alignas(sizeof(T) * V) Vec { T v[V]; };
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % SV) * V;
        const int y = (int)((i / SV) % S);
        const int c = (int)((i / ((long)SV * S)) % C);
        const int b = (int)(i / ((long)SV * S * C));
        const long img = (((long)b * C + c) * S + y) * S + x;
        const long tok = ((long)b * g * g + (y / P) * g + x / P) * ((long)C * P * P) + ((long)c * P + y % P) * P + x % P;
        if (TO_IMAGE) *reinterpret_cast<Vec*>(dst + img) = *reinterpret_cast<const Vec*>(src + tok);
        else *reinterpret_cast<Vec*>(dst + tok) = *reinterpret_cast<const Vec*>(src + img);
    }
}

int check(const mrg_loss_desc* d, const char* who) {
    MRG_REQUIRE(d != nullptr, MRG_ERR_ARG, "%s: null descriptor", who);
    MRG_REQUIRE(d->pred && (d->target || d->target_slot) && d->partial && d->coef, MRG_ERR_ARG, "%s: null pointer", who);
    MRG_REQUIRE(d->kind == MRG_LOSS_MSE || d->kind == MRG_LOSS_CE, MRG_ERR_ARG, "%s: bad kind", who);
    MRG_REQUIRE(d->B > 0 && d->C > 0 && d->P > 0 && d->H % d->P == 0 && d->W % d->P == 0, MRG_ERR_SHAPE, "%s: bad geometry", who);
    MRG_REQUIRE(d->pred_dtype == MRG_DT_BF16 || d->pred_dtype == MRG_DT_F32, MRG_ERR_ARG, "%s: bad dtype", who);
    return 0;
}

}  // namespace

extern "C" int mrg_loss_partial_floats(int B) { return B * LOSS_CHUNKS * 2; }

extern "C" int mrg_masked_loss_fwd(const mrg_loss_desc* d, hipStream_t stream) {
    int rc = check(d, "mrg_masked_loss_fwd");
    if (rc) return rc;
    MRG_REQUIRE(d->loss, MRG_ERR_ARG, "mrg_masked_loss_fwd: null loss pointer");
    dim3 grid(LOSS_CHUNKS, d->B), block(256);
    const bool vec_mse = d->kind == MRG_LOSS_MSE && d->C == 1 && d->pred_dtype == MRG_DT_BF16 && d->P % 8 == 0 && d->W % 8 == 0 && !d->norm_pix &&
                         (((uintptr_t)d->pred | (uintptr_t)d->target) & 15) == 0;
    if (d->norm_pix) {
        MRG_REQUIRE(d->kind == MRG_LOSS_MSE && d->stats != nullptr && d->C * d->P * d->P > 1, MRG_ERR_ARG,
                    "mrg_masked_loss_fwd: norm_pix needs the MSE kind, a stats workspace and more than one element per patch");
        hipLaunchKernelGGL(patch_stats_kernel, dim3((d->H / d->P) * (d->W / d->P), d->B), dim3(256), 0, stream, (const float*)d->target, d->target_slot,
                           d->C, d->H, d->W, d->P, d->stats);
        rc = mrg_check_launch("mrg_masked_loss_fwd/norm_pix");
        if (rc) return rc;
    }
    if (vec_mse) {
        hipLaunchKernelGGL(mse1_fwd_vec_kernel, grid, block, 0, stream, (const bf16_t*)d->pred, (const float*)d->target, d->target_slot, d->mask, d->B, d->H,
                           d->W, d->P, d->pred_tokens, d->partial);
    } else {
#define MRG_L(T, CE) hipLaunchKernelGGL((loss_fwd_kernel<T, CE>), grid, block, 0, stream, (const T*)d->pred, d->target, d->target_slot, d->mask, d->B, d->C, d->H, d->W, d->P, d->pred_tokens, d->label_smoothing, d->norm_pix ? d->stats : nullptr, d->partial)
    if (d->pred_dtype == MRG_DT_BF16) { if (d->kind == MRG_LOSS_CE) MRG_L(bf16_t, true); else MRG_L(bf16_t, false); }
    else { if (d->kind == MRG_LOSS_CE) MRG_L(float, true); else MRG_L(float, false); }
#undef MRG_L
    }
    rc = mrg_check_launch("mrg_masked_loss_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, stream, d->partial, d->B, d->loss, d->coef, d->per_sample);
    return mrg_check_launch("mrg_masked_loss_fwd/finish");
}

extern "C" int mrg_masked_loss_bwd(const mrg_loss_desc* d, hipStream_t stream) {
    int rc = check(d, "mrg_masked_loss_bwd");
    if (rc) return rc;
    MRG_REQUIRE(d->dpred, MRG_ERR_ARG, "mrg_masked_loss_bwd: null dpred");
    const long total = (long)d->B * d->H * d->W;
    dim3 grid((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), block(256);
    const bool vec_mse = d->kind == MRG_LOSS_MSE && d->C == 1 && d->pred_dtype == MRG_DT_BF16 && d->P % 8 == 0 && d->W % 8 == 0 && !d->norm_pix &&
                         (((uintptr_t)d->pred | (uintptr_t)d->target | (uintptr_t)d->dpred) & 15) == 0;
    MRG_REQUIRE(!d->norm_pix || d->stats != nullptr, MRG_ERR_ARG, "mrg_masked_loss_bwd: norm_pix needs the stats the forward call filled");
    if (vec_mse) {
        hipLaunchKernelGGL(mse1_bwd_vec_kernel, dim3((unsigned)((total / 8 + 255) / 256 < 4096 ? (total / 8 + 255) / 256 : 4096)), block, 0, stream,
                           (const bf16_t*)d->pred, (const float*)d->target, d->target_slot, d->mask, d->coef, d->gout, d->B, d->H, d->W, d->P, d->pred_tokens, (bf16_t*)d->dpred);
        return mrg_check_launch("mrg_masked_loss_bwd");
    }
#define MRG_L(T, CE) hipLaunchKernelGGL((loss_bwd_kernel<T, CE>), grid, block, 0, stream, (const T*)d->pred, d->target, d->target_slot, d->mask, d->coef, d->gout, d->B, d->C, d->H, d->W, d->P, d->pred_tokens, d->label_smoothing, d->norm_pix ? d->stats : nullptr, (T*)d->dpred)
    if (d->pred_dtype == MRG_DT_BF16) { if (d->kind == MRG_LOSS_CE) MRG_L(bf16_t, true); else MRG_L(bf16_t, false); }
    else { if (d->kind == MRG_LOSS_CE) MRG_L(float, true); else MRG_L(float, false); }
#undef MRG_L
    return mrg_check_launch("mrg_masked_loss_bwd");
}

extern "C" int mrg_patch_rearrange(const void* src, void* dst, int dtype, int to_image, int B, int C, int g, int P, hipStream_t stream) {
    MRG_REQUIRE(src && dst, MRG_ERR_ARG, "mrg_patch_rearrange: null pointer");
    MRG_REQUIRE(B > 0 && C > 0 && g > 0 && P > 0, MRG_ERR_SHAPE, "mrg_patch_rearrange: bad shape");
    const int esz = dtype == MRG_DT_BF16 ? 2 : 4;
    int V = 16 / esz;                                  // 16-byte vectors when the patch width allows it
    while (V > 1 && (P % V != 0 || (((uintptr_t)src | (uintptr_t)dst) % (V * esz)) != 0)) V >>= 1;
    const long total = (long)B * C * g * P * g * P / V;
    dim3 grid((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), block(256);
#define MRG_PR(T, V_) do { if (to_image) hipLaunchKernelGGL((patch_rearrange_kernel<T, true, V_>), grid, block, 0, stream, (const T*)src, (T*)dst, B, C, g, P); \
                           else hipLaunchKernelGGL((patch_rearrange_kernel<T, false, V_>), grid, block, 0, stream, (const T*)src, (T*)dst, B, C, g, P); } while (0)
    if (dtype == MRG_DT_BF16) { if (V == 8) MRG_PR(bf16_t, 8); else if (V == 4) MRG_PR(bf16_t, 4); else if (V == 2) MRG_PR(bf16_t, 2); else MRG_PR(bf16_t, 1); }
    else if (dtype == MRG_DT_F32) { if (V == 4) MRG_PR(float, 4); else if (V == 2) MRG_PR(float, 2); else MRG_PR(float, 1); }
    else return mrg_set_error(MRG_ERR_ARG, "mrg_patch_rearrange: bad dtype");
#undef MRG_PR
    return mrg_check_launch("mrg_patch_rearrange");
}
