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

// The batch statistics of a loss: per-sample loss, nanmean, and (coef != nullptr) the backward coefficients coef[b] = 1 / (count_b * n_valid).
// One workgroup of NT threads; sl / sn = NT floats of LDS each.  The sums run in sample order whatever NT is (bit-identical results from
// the finish launch whatever its width).
template <int NT>
__device__ __forceinline__ void loss_finish_block(const float* __restrict__ partial, int B, float* __restrict__ loss, float* __restrict__ coef,
                                                  float* __restrict__ per_sample, float* sl, float* sn, float* s_nvalid) {
    auto ld = [&](long i) -> float { return partial[i]; };
    const int t = threadIdx.x;
    float tot = 0.f;
    int nvalid = 0;
    float total_cnt = 0.f;
    for (int base = 0; base < B; base += NT) {
        const int b = base + t;
        float s = 0.f, n = 0.f;
        if (b < B)
            for (int c = 0; c < LOSS_CHUNKS; ++c) { s += ld(((long)b * LOSS_CHUNKS + c) * 2); n += ld(((long)b * LOSS_CHUNKS + c) * 2 + 1); }
        sl[t] = s; sn[t] = n;
        __syncthreads();
        if (t == 0) {
            for (int i = 0; i < min(NT, B - base); ++i) {
                total_cnt += sn[i];
                if (sn[i] > 0.f) { tot += sl[i] / sn[i]; ++nvalid; }
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        // criterion.py:36,103: mask.sum() == 0 -> 0 ; otherwise nanmean over samples (0/0 samples dropped)
        loss[0] = (total_cnt > 0.f && nvalid > 0) ? tot / (float)nvalid : 0.f;
        *s_nvalid = (float)nvalid;
    }
    __syncthreads();
    if (!coef && !per_sample) return;
    for (int b = t; b < B; b += NT) {
        float n = 0.f, s = 0.f;
        for (int c = 0; c < LOSS_CHUNKS; ++c) { s += ld(((long)b * LOSS_CHUNKS + c) * 2); n += ld(((long)b * LOSS_CHUNKS + c) * 2 + 1); }
        if (coef) coef[b] = (n > 0.f && *s_nvalid > 0.f) ? 1.f / (n * *s_nvalid) : 0.f;
        if (per_sample) per_sample[b] = n > 0.f ? s / n : NAN;
    }
}

// Backward coefficient of sample b inside a fused launch: coef[b] when the caller formed the coefficients (mrg_loss_coef), otherwise
// 1 / (masked pixels of the sample x nvalid) from the sample's own mask row -- the caller guarantees that nvalid samples have masked
// pixels (in the training step: every one, because fewer tokens are kept than a modality has).  Same expression as loss_finish_kernel's.
__device__ __forceinline__ float fused_coef(const float* __restrict__ coef, const long* __restrict__ mask, long mbs, int b, int ntok, int ppt,
                                            int nvalid, float* red) {
    if (coef) return coef[b];
    __shared__ float s_cb;
    float c = 0.f;
    if (mask) for (int t = threadIdx.x; t < ntok; t += blockDim.x) c += (float)mask[(long)b * mbs + t];
    const float tot = block_sum(c, red);
    if (threadIdx.x == 0) {
        const float n = mask ? tot * (float)ppt : (float)ntok * (float)ppt;
        s_cb = (n > 0.f && nvalid > 0) ? 1.f / (n * (float)nvalid) : 0.f;
    }
    __syncthreads();
    return s_cb;
}

// grid (LOSS_CHUNKS, B); partial [B][LOSS_CHUNKS][2] = (sum of masked per-pixel loss, masked pixel count).
// FUSED: the same pass writes dpred = gout * dloss / dpred with the backward coefficients `coef` that mrg_loss_coef formed from the mask
// alone (the expressions of loss_bwd_kernel on the same mx / se: bit-identical to the two-launch path).  The batch statistics stay a launch of
// their own (mrg_masked_loss_finish): a last-workgroup-retires finish needs a device-scope release per workgroup, i.e. an L2 write-back
// on a part whose L2s are per XCD -- measured 4x the whole kernel.
template <typename T, bool CE, bool FUSED>
__global__ __launch_bounds__(256) void loss_fwd_kernel(const T* __restrict__ pred, const void* __restrict__ target,
                                                       const void* const* __restrict__ tslot,
                                                       const long* __restrict__ mask, long mbs, int B, int C, int H, int W, int P, int tok,
                                                       float smooth, const float* __restrict__ stats, float* __restrict__ partial,
                                                       const float* __restrict__ coef, const float* __restrict__ gout, T* __restrict__ dpred, int nvalid) {
    if (tslot) target = *tslot;                              // input slot: the batch is read where it lies (mirage_hip.h, "Input slots")
    __shared__ float red[4];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int nw = W / P;
    const long HW = (long)H * W;
    const float kb = FUSED ? (gout ? gout[0] : 1.f) * fused_coef(coef, mask, mbs, b, (H / P) * nw, P * P, nvalid, red) : 0.f;
    float s = 0.f, n = 0.f;
    for (long px = (long)chunk * 256 + threadIdx.x; px < HW; px += (long)LOSS_CHUNKS * 256) {
        const int y = (int)(px / W), x = (int)(px - (long)y * W);
        const float m = mask ? (float)mask[(long)b * mbs + (y / P) * nw + x / P] : 1.f;
        if (m == 0.f) {
            if (FUSED) for (int c = 0; c < C; ++c) stf<T>(dpred, pred_index(tok, b, c, y, x, C, H, W, P), 0.f);
            continue;
        }
        const float k = kb * m;
        float l;
        if (CE) {
            const int t = (int)reinterpret_cast<const long*>(target)[(long)b * HW + px];
            float mx = -INFINITY, zt = 0.f, zs = 0.f;
            for (int c = 0; c < C; ++c) { const float z = ldf<T>(pred, pred_index(tok, b, c, y, x, C, H, W, P)); mx = fmaxf(mx, z); zt = (c == t) ? z : zt; zs += z; }
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += __expf(ldf<T>(pred, pred_index(tok, b, c, y, x, C, H, W, P)) - mx);
            // F.cross_entropy(label_smoothing = e) = (1 - e) nll + e mean_c(-log p_c) = lse - (1 - e) z_t - e mean_c z_c  (criterion.py:33)
            l = mx + __logf(se) - (1.f - smooth) * zt - smooth * zs / (float)C;
            if (FUSED) {
                if (k == 0.f) { for (int c = 0; c < C; ++c) stf<T>(dpred, pred_index(tok, b, c, y, x, C, H, W, P), 0.f); }
                else {
                    const float inv = 1.f / se;
                    for (int c = 0; c < C; ++c) {
                        const long j = pred_index(tok, b, c, y, x, C, H, W, P);
                        const float p = __expf(ldf<T>(pred, j) - mx) * inv;
                        stf<T>(dpred, j, k * (p - (c == t ? 1.f - smooth : 0.f) - smooth / (float)C));
                    }
                }
            }
        } else {
            float q = 0.f;
            const float* st = stats ? stats + ((long)b * (H / P) * nw + (y / P) * nw + x / P) * 2 : nullptr;
            const float kk = k * 2.f / (float)C;
            for (int c = 0; c < C; ++c) {
                float tv = reinterpret_cast<const float*>(target)[((long)b * C + c) * HW + px];
                if (st) tv = (tv - st[0]) * st[1];
                const long jp = pred_index(tok, b, c, y, x, C, H, W, P);
                const float pv = ldf<T>(pred, jp);
                const float d = pv - tv;
                q += d * d;
                if (FUSED) stf<T>(dpred, jp, kk * (pv - tv));
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
__global__ __launch_bounds__(1024) void loss_finish_kernel(const float* __restrict__ partial, int B, float* __restrict__ loss, float* __restrict__ coef,
                                                           float* __restrict__ per_sample) {
    __shared__ float sl[1024], sn[1024], s_nvalid;
    loss_finish_block<1024>(partial, B, loss, coef, per_sample, sl, sn, &s_nvalid);
}

// The backward coefficients from the mask alone (they do not depend on the predictions): n_b = masked pixels of sample b = ppt * sum of its
// token mask (mask == nullptr: every pixel), coef[b] = 1 / (n_b * #{b: n_b > 0}) -- the values loss_finish_kernel derives from the forward
// pass's pixel counts (both are exact integers in fp32).  One workgroup, one wave per sample at a time.
__global__ __launch_bounds__(1024) void loss_coef_kernel(const long* __restrict__ mask, long mbs, int B, int ntok, int ppt, float* __restrict__ coef) {
    __shared__ int s_nvalid;
    if (threadIdx.x == 0) s_nvalid = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < B; b += 16) {
        float n;
        if (mask) {
            float c = 0.f;
            for (int t = lane; t < ntok; t += 64) c += (float)mask[(long)b * mbs + t];
            n = wave_sum(c) * (float)ppt;
        } else {
            n = (float)ntok * (float)ppt;
        }
        if (lane == 0) { coef[b] = n; if (n > 0.f) atomicAdd(&s_nvalid, 1); }
    }
    __syncthreads();
    const float nv = (float)s_nvalid;
    for (int b = wave; b < B; b += 16)
        if (lane == 0) { const float n = coef[b]; coef[b] = (n > 0.f && nv > 0.f) ? 1.f / (n * nv) : 0.f; }
}

template <typename T, bool CE>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const T* __restrict__ pred, const void* __restrict__ target,
                                                       const void* const* __restrict__ tslot,
                                                       const long* __restrict__ mask, long mbs, const float* __restrict__ coef,
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
        const float m = mask ? (float)mask[(long)b * mbs + (y / P) * nw + x / P] : 1.f;
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

// ---- cross entropy on token-layout bf16 logits with C classes known at compile time: a thread owns 8 consecutive elements of a token's
// [P][P] block (one 16-B chunk per class: the class blocks of a token are contiguous, so 8 lanes cover 128 B and every line is read once),
// keeps the C chunks packed in registers and walks them three times (max / target / sum, sum of exponentials, gradient) -- instead of the
// scalar kernel's three 2-byte loads per class and pixel with their index arithmetic (52 us for the 27 MB of C4's layer-map logits, 5x
// its traffic time).  grid (LOSS_CHUNKS, B) and partial sums like the scalar kernel's; the forward-only and the one-pass launch are the
// same code, so their partial sums agree bit for bit.
template <bool FUSED, int C>
__global__ __launch_bounds__(256) void ce_tok_vec_kernel(const bf16_t* __restrict__ pred, const long* __restrict__ target,
                                                         const void* const* __restrict__ tslot, const long* __restrict__ mask, long mbs,
                                                         int B, int H, int W, int P, float smooth, float* __restrict__ partial,
                                                         const float* __restrict__ coef, const float* __restrict__ gout, bf16_t* __restrict__ dpred, int nvalid) {
    if (tslot) target = static_cast<const long*>(*tslot);
    __shared__ float red[4];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int nw = W / P, PP = P * P, q8 = PP / 8, ntok = (H / P) * nw;
    const long total = (long)ntok * q8;                        // 16-B chunks of one class plane of the sample
    const float kb = FUSED ? (gout ? gout[0] : 1.f) * fused_coef(coef, mask, mbs, b, ntok, PP, nvalid, red) : 0.f;
    const float sm_c = smooth / (float)C, hit = 1.f - smooth;
    auto lo = [](uint32_t w) { return __uint_as_float(w << 16); };
    auto hi = [](uint32_t w) { return __uint_as_float(w & 0xffff0000u); };
    float s = 0.f, n = 0.f;
    for (long v = (long)chunk * 256 + threadIdx.x; v < total; v += (long)LOSS_CHUNKS * 256) {
        const int tk = (int)(v / q8), q = (int)(v - (long)tk * q8);
        const float m = mask ? (float)mask[(long)b * mbs + tk] : 1.f;
        const long base = ((long)b * ntok + tk) * ((long)C * PP) + q * 8;
        if (m == 0.f) {
            if (FUSED) {
#pragma unroll
                for (int c = 0; c < C; ++c) *reinterpret_cast<uint4*>(dpred + base + (long)c * PP) = make_uint4(0, 0, 0, 0);
            }
            continue;
        }
        uint32_t w[C][4];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(pred + base + (long)c * PP);
            w[c][0] = u.x; w[c][1] = u.y; w[c][2] = u.z; w[c][3] = u.w;
        }
        const int ty = tk / nw, tx = tk - ty * nw, py = (q * 8) / P, px = q * 8 - py * P;
        int t[8];
        {
            const long* tp = target + ((long)b * H + ty * P + py) * W + tx * P + px;
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const longlong2 t2 = *reinterpret_cast<const longlong2*>(tp + i);
                t[i] = (int)t2.x; t[i + 1] = (int)t2.y;
            }
        }
        const float k = kb * m;
        uint32_t o[C][4];
        // two pixels (one packed word per class) at a time: the C x 2 target compares stay in scalar registers
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float mx0 = -INFINITY, mx1 = -INFINITY, zt0 = 0.f, zt1 = 0.f, zs0 = 0.f, zs1 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float z0 = lo(w[c][j]), z1 = hi(w[c][j]);
                mx0 = fmaxf(mx0, z0); mx1 = fmaxf(mx1, z1);
                zt0 = (c == t[2 * j]) ? z0 : zt0; zt1 = (c == t[2 * j + 1]) ? z1 : zt1;
                zs0 += z0; zs1 += z1;
            }
            float se0 = 0.f, se1 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) { se0 += __expf(lo(w[c][j]) - mx0); se1 += __expf(hi(w[c][j]) - mx1); }
            // F.cross_entropy(label_smoothing = e) = lse - (1 - e) z_t - e mean_c z_c  (criterion.py:33)
            const float l0 = mx0 + __logf(se0) - hit * zt0 - smooth * zs0 / (float)C;
            const float l1 = mx1 + __logf(se1) - hit * zt1 - smooth * zs1 / (float)C;
            s += l0 * m; n += m;
            s += l1 * m; n += m;
            if (FUSED) {
                const float i0 = 1.f / se0, i1 = 1.f / se1;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float p0 = __expf(lo(w[c][j]) - mx0) * i0, p1 = __expf(hi(w[c][j]) - mx1) * i1;
                    const float g0 = k == 0.f ? 0.f : k * (p0 - (c == t[2 * j] ? hit : 0.f) - sm_c);
                    const float g1 = k == 0.f ? 0.f : k * (p1 - (c == t[2 * j + 1] ? hit : 0.f) - sm_c);
                    o[c][j] = pack_bf2(g0, g1);
                }
            }
        }
        if (FUSED) {
#pragma unroll
            for (int c = 0; c < C; ++c) *reinterpret_cast<uint4*>(dpred + base + (long)c * PP) = make_uint4(o[c][0], o[c][1], o[c][2], o[c][3]);
        }
    }
    const float ts = block_sum(s, red);
    const float tn = block_sum(n, red);
    if (threadIdx.x == 0) {
        partial[((long)b * LOSS_CHUNKS + chunk) * 2 + 0] = ts;
        partial[((long)b * LOSS_CHUNKS + chunk) * 2 + 1] = tn;
    }
}

// ---- single-channel bf16 MSE fast path: 8 consecutive pixels (16 B of predictions, 32 B of targets) per thread
// FUSED: see loss_fwd_kernel (the gradient expressions are mse1_bwd_vec_kernel's)
template <bool FUSED>
__global__ __launch_bounds__(256) void mse1_fwd_vec_kernel(const bf16_t* __restrict__ pred, const float* __restrict__ target,
                                                           const void* const* __restrict__ tslot,
                                                           const long* __restrict__ mask, long mbs, int B, int H, int W, int P, int tok,
                                                           float* __restrict__ partial,
                                                           const float* __restrict__ coef, const float* __restrict__ gout, bf16_t* __restrict__ dpred, int nvalid) {
    if (tslot) target = static_cast<const float*>(*tslot);
    __shared__ float red[4];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int nw = W / P;
    const long HW8 = (long)H * W / 8;
    const float go = FUSED ? (gout ? gout[0] : 1.f) : 0.f;
    const float cb = FUSED ? fused_coef(coef, mask, mbs, b, (H / P) * nw, P * P, nvalid, red) : 0.f;
    float s = 0.f, n = 0.f;
    for (long v = (long)chunk * 256 + threadIdx.x; v < HW8; v += (long)LOSS_CHUNKS * 256) {
        const long px = v * 8;
        const int y = (int)(px / W), x = (int)(px - (long)y * W);
        const float m = mask ? (float)mask[(long)b * mbs + (y / P) * nw + x / P] : 1.f;
        const long ip = pred_index(tok, b, 0, y, x, 1, H, W, P);                                             // P % 8 == 0: 8 pixels stay inside a patch row
        if (m == 0.f) {
            if (FUSED) *reinterpret_cast<uint4*>(dpred + ip) = make_uint4(0, 0, 0, 0);
            continue;
        }
        const uint4 p = *reinterpret_cast<const uint4*>(pred + ip);
        const float4 t0 = *reinterpret_cast<const float4*>(target + (long)b * H * W + px);
        const float4 t1 = *reinterpret_cast<const float4*>(target + (long)b * H * W + px + 4);
        const float d0 = __uint_as_float(p.x << 16) - t0.x, d1 = __uint_as_float(p.x & 0xffff0000u) - t0.y;
        const float d2 = __uint_as_float(p.y << 16) - t0.z, d3 = __uint_as_float(p.y & 0xffff0000u) - t0.w;
        const float d4 = __uint_as_float(p.z << 16) - t1.x, d5 = __uint_as_float(p.z & 0xffff0000u) - t1.y;
        const float d6 = __uint_as_float(p.w << 16) - t1.z, d7 = __uint_as_float(p.w & 0xffff0000u) - t1.w;
        s += ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
        n += 8.f;
        if (FUSED) {
            const float k = 2.f * go * cb * m;
            uint4 o = make_uint4(0, 0, 0, 0);
            if (k != 0.f) {
                o.x = pack_bf2(k * d0, k * d1); o.y = pack_bf2(k * d2, k * d3);
                o.z = pack_bf2(k * d4, k * d5); o.w = pack_bf2(k * d6, k * d7);
            }
            *reinterpret_cast<uint4*>(dpred + ip) = o;
        }
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
                                                           const long* __restrict__ mask, long mbs, const float* __restrict__ coef,
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
        const float m = mask ? (float)mask[(long)b * mbs + (y / P) * nw + x / P] : 1.f;
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

long mask_stride(const mrg_loss_desc* d) { return d->mask_bs > 0 ? d->mask_bs : (long)(d->H / d->P) * (d->W / d->P); }

int check(const mrg_loss_desc* d, const char* who) {
    MRG_REQUIRE(d != nullptr, MRG_ERR_ARG, "%s: null descriptor", who);
    MRG_REQUIRE(d->pred && (d->target || d->target_slot) && d->partial && (d->coef || d->nvalid > 0), MRG_ERR_ARG, "%s: null pointer", who);
    MRG_REQUIRE(d->kind == MRG_LOSS_MSE || d->kind == MRG_LOSS_CE, MRG_ERR_ARG, "%s: bad kind", who);
    MRG_REQUIRE(d->B > 0 && d->C > 0 && d->P > 0 && d->H % d->P == 0 && d->W % d->P == 0, MRG_ERR_SHAPE, "%s: bad geometry", who);
    MRG_REQUIRE(d->pred_dtype == MRG_DT_BF16 || d->pred_dtype == MRG_DT_F32, MRG_ERR_ARG, "%s: bad dtype", who);
    return 0;
}

}  // namespace

extern "C" int mrg_loss_partial_floats(int B) { return B * LOSS_CHUNKS * 2; }

namespace {
// forward launches of a loss; fused: + gradient and batch statistics in the same pass (mrg_masked_loss_fused)
int launch_loss_fwd(const mrg_loss_desc* d, bool fused, hipStream_t stream, const char* who) {
    int rc;
    const long mbs = mask_stride(d);
    dim3 grid(LOSS_CHUNKS, d->B), block(256);
    const bool vec_mse = d->kind == MRG_LOSS_MSE && d->C == 1 && d->pred_dtype == MRG_DT_BF16 && d->P % 8 == 0 && d->W % 8 == 0 && !d->norm_pix &&
                         (((uintptr_t)d->pred | (uintptr_t)d->target | (fused ? (uintptr_t)d->dpred : 0)) & 15) == 0;
    if (d->norm_pix) {
        MRG_REQUIRE(d->kind == MRG_LOSS_MSE && d->stats != nullptr && d->C * d->P * d->P > 1, MRG_ERR_ARG,
                    "%s: norm_pix needs the MSE kind, a stats workspace and more than one element per patch", who);
        hipLaunchKernelGGL(patch_stats_kernel, dim3((d->H / d->P) * (d->W / d->P), d->B), dim3(256), 0, stream, (const float*)d->target, d->target_slot,
                           d->C, d->H, d->W, d->P, d->stats);
        rc = mrg_check_launch("mrg_masked_loss/norm_pix");
        if (rc) return rc;
    }
    const float* coef = fused ? d->coef : nullptr;
    const int nvalid = d->nvalid;
    // the layer maps' 13 classes (mirage/output_adapters.py: num_channels of the semseg adapter); other class counts take the scalar kernel
    const bool vec_ce = d->kind == MRG_LOSS_CE && d->pred_tokens && d->pred_dtype == MRG_DT_BF16 && d->P % 8 == 0 && d->C == 13 && d->W % 8 == 0 &&
                        (((uintptr_t)d->pred | (uintptr_t)d->target | (fused ? (uintptr_t)d->dpred : 0)) & 15) == 0;
    if (vec_ce) {
#define MRG_C(F) hipLaunchKernelGGL((ce_tok_vec_kernel<F, 13>), grid, block, 0, stream, (const bf16_t*)d->pred, (const long*)d->target, d->target_slot, d->mask, mbs, d->B, \
                           d->H, d->W, d->P, d->label_smoothing, d->partial, coef, d->gout, (bf16_t*)d->dpred, nvalid)
        if (fused) MRG_C(true); else MRG_C(false);
#undef MRG_C
    } else if (vec_mse) {
#define MRG_V(F) hipLaunchKernelGGL(mse1_fwd_vec_kernel<F>, grid, block, 0, stream, (const bf16_t*)d->pred, (const float*)d->target, d->target_slot, d->mask, mbs, d->B, d->H, \
                           d->W, d->P, d->pred_tokens, d->partial, coef, d->gout, (bf16_t*)d->dpred, nvalid)
        if (fused) MRG_V(true); else MRG_V(false);
#undef MRG_V
    } else {
#define MRG_L(T, CE, F) hipLaunchKernelGGL((loss_fwd_kernel<T, CE, F>), grid, block, 0, stream, (const T*)d->pred, d->target, d->target_slot, d->mask, mbs, d->B, d->C, d->H, d->W, d->P, d->pred_tokens, d->label_smoothing, d->norm_pix ? d->stats : nullptr, d->partial, coef, d->gout, (T*)d->dpred, nvalid)
#define MRG_LF(T, CE) do { if (fused) MRG_L(T, CE, true); else MRG_L(T, CE, false); } while (0)
    if (d->pred_dtype == MRG_DT_BF16) { if (d->kind == MRG_LOSS_CE) MRG_LF(bf16_t, true); else MRG_LF(bf16_t, false); }
    else { if (d->kind == MRG_LOSS_CE) MRG_LF(float, true); else MRG_LF(float, false); }
#undef MRG_LF
#undef MRG_L
    }
    return mrg_check_launch(who);
}
}  // namespace

extern "C" int mrg_masked_loss_fwd(const mrg_loss_desc* d, hipStream_t stream) {
    int rc = check(d, "mrg_masked_loss_fwd");
    if (rc) return rc;
    MRG_REQUIRE(d->loss, MRG_ERR_ARG, "mrg_masked_loss_fwd: null loss pointer");
    rc = launch_loss_fwd(d, false, stream, "mrg_masked_loss_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, stream, d->partial, d->B, d->loss, d->coef, d->per_sample);
    return mrg_check_launch("mrg_masked_loss_fwd/finish");
}

extern "C" int mrg_loss_coef(const long* mask, long mask_bs, int B, int ntok, int ppt, float* coef, hipStream_t stream) {
    MRG_REQUIRE(coef != nullptr && B > 0 && ntok > 0 && ppt > 0 && mask_bs >= 0, MRG_ERR_ARG, "mrg_loss_coef: bad argument");
    MRG_REQUIRE((double)ntok * ppt < 16777216.0, MRG_ERR_SHAPE, "mrg_loss_coef: %d x %d pixels per sample are not exact in fp32", ntok, ppt);
    hipLaunchKernelGGL(loss_coef_kernel, dim3(1), dim3(1024), 0, stream, mask, mask_bs > 0 ? mask_bs : (long)ntok, B, ntok, ppt, coef);
    return mrg_check_launch("mrg_loss_coef");
}

extern "C" int mrg_masked_loss_fused(const mrg_loss_desc* d, hipStream_t stream) {
    int rc = check(d, "mrg_masked_loss_fused");
    if (rc) return rc;
    MRG_REQUIRE(d->dpred, MRG_ERR_ARG, "mrg_masked_loss_fused: null dpred pointer");
    MRG_REQUIRE(d->coef || (d->nvalid > 0 && d->nvalid <= d->B), MRG_ERR_ARG, "mrg_masked_loss_fused: neither coef nor a sample count 0 < nvalid <= B");
    return launch_loss_fwd(d, true, stream, "mrg_masked_loss_fused");
}

extern "C" int mrg_masked_loss_finish(const mrg_loss_desc* d, hipStream_t stream) {
    MRG_REQUIRE(d != nullptr && d->partial && d->loss && d->B > 0, MRG_ERR_ARG, "mrg_masked_loss_finish: null descriptor / partial / loss pointer");
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, stream, d->partial, d->B, d->loss, (float*)nullptr, d->per_sample);
    return mrg_check_launch("mrg_masked_loss_finish");
}

extern "C" int mrg_masked_loss_bwd(const mrg_loss_desc* d, hipStream_t stream) {
    int rc = check(d, "mrg_masked_loss_bwd");
    if (rc) return rc;
    MRG_REQUIRE(d->dpred && d->coef, MRG_ERR_ARG, "mrg_masked_loss_bwd: null dpred / coef");
    const long mbs = mask_stride(d);
    const long total = (long)d->B * d->H * d->W;
    dim3 grid((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), block(256);
    const bool vec_mse = d->kind == MRG_LOSS_MSE && d->C == 1 && d->pred_dtype == MRG_DT_BF16 && d->P % 8 == 0 && d->W % 8 == 0 && !d->norm_pix &&
                         (((uintptr_t)d->pred | (uintptr_t)d->target | (uintptr_t)d->dpred) & 15) == 0;
    MRG_REQUIRE(!d->norm_pix || d->stats != nullptr, MRG_ERR_ARG, "mrg_masked_loss_bwd: norm_pix needs the stats the forward call filled");
    if (vec_mse) {
        hipLaunchKernelGGL(mse1_bwd_vec_kernel, dim3((unsigned)((total / 8 + 255) / 256 < 4096 ? (total / 8 + 255) / 256 : 4096)), block, 0, stream,
                           (const bf16_t*)d->pred, (const float*)d->target, d->target_slot, d->mask, mbs, d->coef, d->gout, d->B, d->H, d->W, d->P, d->pred_tokens, (bf16_t*)d->dpred);
        return mrg_check_launch("mrg_masked_loss_bwd");
    }
#define MRG_L(T, CE) hipLaunchKernelGGL((loss_bwd_kernel<T, CE>), grid, block, 0, stream, (const T*)d->pred, d->target, d->target_slot, d->mask, mbs, d->coef, d->gout, d->B, d->C, d->H, d->W, d->P, d->pred_tokens, d->label_smoothing, d->norm_pix ? d->stats : nullptr, (T*)d->dpred)
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
