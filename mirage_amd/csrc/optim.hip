// Step boundary: fused flat-arena AdamW (+ bf16 shadow weights for the next step's GEMMs), global L2 gradient
// norm, fp32 -> bf16 cast.  Replaces torch.optim.AdamW's per-tensor kernels (296 Base / 440 Large tensors,
// mutils/optim_factory.py:180), mutils/native_scaler.py:46-61 get_grad_norm_ and the per-Linear autocast weight
// casts.  HBM-bound: 4 fp32 streams read, 3 written, + 2 B/param shadow.
// Hyper-parameters come from device-resident schedule tables indexed by a device step counter, so the whole
// optimizer step is hipGraph-replayable (run_pretraining.py:683-688 sets lr/wd per step from such arrays).
#include "common.h"
#include <math.h>
#include "mirage_hip.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partials) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = g4[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup of 1024 threads, four loads in flight per thread (the launch sits in the serial tail of the step: 39 -> ~6 us for the
// ~30 k partials of a Base model's AdamW launches); fixed summation order, double accumulation
__global__ __launch_bounds__(1024) void sumsq_finish_kernel(const float* __restrict__ partials, int nb, float* __restrict__ out) {
    __shared__ double red[1024];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * 1024 < nb; i += 4 * 1024) {
        const float a = partials[i], b = partials[i + 1024], c = partials[i + 2048], d = partials[i + 3072];
        s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
    }
    for (; i < nb; i += 1024) s0 += (double)partials[i];
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sqrt(red[0]);
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; bf16_t* shadow;
    long n, n_decay;
    const float* lr_table; const float* wd_table; const int* step; int table_len;
    float lr_scale, beta1, beta2, eps;
    const float* grad_norm; float clip_norm, skip_norm;
    float* sumsq_out;      // optional [gridDim.x]: this launch's partial sums of g^2 (the gradient is read here anyway)
};

__global__ __launch_bounds__(256) void adamw_kernel(const AdamArgs a) {
    __shared__ float hp[5];
    if (threadIdx.x == 0) {
        // step[0]: iteration counter = index into the lr / wd schedules (advances every iteration, run_pretraining.py:683-688);
        // step[1]: updates applied so far = AdamW's per-parameter `step` (torch bias correction; does not advance on a skipped step)
        const int st = a.step[0];
        const int ti = st < a.table_len ? st : a.table_len - 1;
        const double t = (double)(a.step[1] + 1);
        hp[0] = a.lr_table[ti] * a.lr_scale;
        hp[1] = a.wd_table[ti];
        hp[2] = (float)(1.0 / (1.0 - pow((double)a.beta1, t)));
        hp[3] = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, t)));
        float gs = 1.f;
        if (a.grad_norm) {
            const float nrm = a.grad_norm[0];
            if (a.clip_norm > 0.f) gs = fminf(1.f, a.clip_norm / (nrm + 1e-6f));   // native_scaler.py:19-23: torch clip_grad_norm_
            else if (a.skip_norm > 0.f && nrm >= a.skip_norm) gs = -1.f;          // :24-29 (`elif`: ignored beside clip_grad): no update
        }
        hp[4] = gs;
    }
    __syncthreads();
    const float lr = hp[0], wd = hp[1], ibc1 = hp[2], isbc2 = hp[3], gs = hp[4];
    if (gs < 0.f) return;
    const float b1 = a.beta1, b2 = a.beta2;
    const long n4 = a.n >> 2;
    float ss = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        float4 g = reinterpret_cast<const float4*>(a.g)[i];
        ss += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        float* pp = &p.x; float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float decay = ((i << 2) + e) < a.n_decay ? wd : 0.f;
            adamw_elem(pp[e], gp[e] * gs, mp[e], vp[e], lr, decay, b1, b2, ibc1, isbc2, a.eps);
        }
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        if (a.shadow) reinterpret_cast<uint2*>(a.shadow)[i] = make_uint2(pack_bf2(p.x, p.y), pack_bf2(p.z, p.w));
    }
    if (a.sumsq_out) {        // deterministic two-stage sum: per-block partial here, mrg_sumsq_finish over all launches' partials later
        __shared__ float red[4];
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        if (threadIdx.x == 0) a.sumsq_out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// the step's hyper-parameters {lr, wd, 1 / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} as adamw_kernel derives them, for the weight-gradient
// epilogue that applies AdamW itself (gemm8p.hip epi_kind 9): written once per step before the backward pass
__global__ void adamw_hp_kernel(const float* lr_table, const float* wd_table, const int* step, int table_len, float lr_scale, float beta1,
                                float beta2, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int st = step[0];
    const int ti = st < table_len ? st : table_len - 1;
    const double t = (double)(step[1] + 1);
    out[0] = lr_table[ti] * lr_scale;
    out[1] = wd_table[ti];
    out[2] = (float)(1.0 / (1.0 - pow((double)beta1, t)));
    out[3] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
}

// end of an iteration: the schedule index always advances, the AdamW step count only when the update was applied
__global__ void step_inc_kernel(int* step, const float* grad_norm, float clip_norm, float skip_norm) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const bool skipped = grad_norm && !(clip_norm > 0.f) && skip_norm > 0.f && grad_norm[0] >= skip_norm;
        step[0] += 1;
        if (!skipped) step[1] += 1;
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<uint2*>(dst)[i] = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = f2bf(src[(n4 << 2) + threadIdx.x]);
}

// one wave per row segment: dst[r, :] = src[r, :] * scale[r]
__global__ __launch_bounds__(256) void row_scale_bf16_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                             bf16_t* __restrict__ dst, long M, int D) {
    const int d4 = D >> 2;
    const long n4 = M * d4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float s = scale[i / d4];
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<uint2*>(dst)[i] = make_uint2(pack_bf2(v.x * s, v.y * s), pack_bf2(v.z * s, v.w * s));
    }
}

// dst[r, c] = src[r, c] * mask(r, c) / (1 - p) (* scale[r]): the gradient entering a dropped-out branch; mask = the Philox keep mask the
// forward GEMM epilogue applied (common.h dropout_apply4: same counter, key and threshold)
__global__ __launch_bounds__(256) void dropout_bwd_bf16_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                               bf16_t* __restrict__ dst, long M, int D, MrgDropout dr) {
    const int d4 = D >> 2;
    const long n4 = M * d4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long row = i / d4;
        const int col = (int)(i - row * d4) << 2;
        const float4 t = reinterpret_cast<const float4*>(src)[i];
        float v[4] = {t.x, t.y, t.z, t.w};
        dropout_apply4(dr, row, col, D, v);
        const float s = scale ? scale[row] : 1.f;
        reinterpret_cast<uint2*>(dst)[i] = make_uint2(pack_bf2(v[0] * s, v[1] * s), pack_bf2(v[2] * s, v[3] * s));
    }
}

}  // namespace

extern "C" int mrg_sumsq_nblocks(long n) {
    long nb = (n / 4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return (int)nb;
}

extern "C" int mrg_grad_norm(const float* g, long n, float* partials, float* out, hipStream_t stream) {
    MRG_REQUIRE(g && partials && out && n > 0, MRG_ERR_ARG, "mrg_grad_norm: bad argument");
    MRG_REQUIRE(((uintptr_t)g & 15) == 0, MRG_ERR_ARG, "mrg_grad_norm: gradient arena must be 16-B aligned");
    const int nb = mrg_sumsq_nblocks(n);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, stream, g, n, partials);
    int rc = mrg_check_launch("mrg_grad_norm");
    if (rc) return rc;
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(1024), 0, stream, partials, nb, out);
    return mrg_check_launch("mrg_grad_norm/finish");
}

extern "C" int mrg_adamw_nblocks(long n) {
    long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    return (int)nb;
}

extern "C" int mrg_sumsq_finish(const float* partials, int n, float* out, hipStream_t stream) {
    MRG_REQUIRE(partials && out && n > 0, MRG_ERR_ARG, "mrg_sumsq_finish: bad argument");
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(1024), 0, stream, partials, n, out);
    return mrg_check_launch("mrg_sumsq_finish");
}

extern "C" int mrg_adamw(const mrg_adamw_desc* d, hipStream_t stream) {
    MRG_REQUIRE(d && d->p && d->g && d->m && d->v && d->lr_table && d->wd_table && d->step, MRG_ERR_ARG, "mrg_adamw: null pointer");
    MRG_REQUIRE(d->n > 0 && d->n % 4 == 0 && d->n_decay >= 0 && d->n_decay <= d->n && d->table_len > 0, MRG_ERR_SHAPE,
                "mrg_adamw: n must be a positive multiple of 4 and 0 <= n_decay <= n");
    MRG_REQUIRE((((uintptr_t)d->p | (uintptr_t)d->g | (uintptr_t)d->m | (uintptr_t)d->v) & 15) == 0, MRG_ERR_ARG,
                "mrg_adamw: arenas must be 16-B aligned");
    AdamArgs a;
    a.p = d->p; a.g = d->g; a.m = d->m; a.v = d->v; a.shadow = (bf16_t*)d->shadow_bf16; a.n = d->n; a.n_decay = d->n_decay;
    a.lr_table = d->lr_table; a.wd_table = d->wd_table; a.step = d->step; a.table_len = d->table_len;
    a.lr_scale = d->lr_scale; a.beta1 = d->beta1; a.beta2 = d->beta2; a.eps = d->eps;
    a.grad_norm = d->grad_norm; a.clip_norm = d->clip_norm; a.skip_norm = d->skip_norm;
    a.sumsq_out = d->sumsq_out;
    MRG_REQUIRE(!(d->sumsq_out && d->grad_norm && (d->skip_norm > 0.f || d->clip_norm > 0.f)), MRG_ERR_UNSUPPORTED,
                "mrg_adamw: sumsq_out with clip_norm / skip_norm (the norm must exist before the update)");
    long nb = mrg_adamw_nblocks(d->n);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a);
    return mrg_check_launch("mrg_adamw");
}

extern "C" int mrg_adamw_hp(const mrg_adamw_desc* d, float* hp_out, hipStream_t stream) {
    MRG_REQUIRE(d && d->lr_table && d->wd_table && d->step && hp_out && d->table_len > 0, MRG_ERR_ARG, "mrg_adamw_hp: bad argument");
    hipLaunchKernelGGL(adamw_hp_kernel, dim3(1), dim3(64), 0, stream, d->lr_table, d->wd_table, d->step, d->table_len, d->lr_scale, d->beta1,
                       d->beta2, hp_out);
    return mrg_check_launch("mrg_adamw_hp");
}

extern "C" int mrg_step_inc(int* step, const float* grad_norm, float clip_norm, float skip_norm, hipStream_t stream) {
    MRG_REQUIRE(step, MRG_ERR_ARG, "mrg_step_inc: null pointer");
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, stream, step, grad_norm, clip_norm, skip_norm);
    return mrg_check_launch("mrg_step_inc");
}

extern "C" int mrg_cast_bf16(const float* src, void* dst_bf16, long n, hipStream_t stream) {
    MRG_REQUIRE(src && dst_bf16 && n > 0, MRG_ERR_ARG, "mrg_cast_bf16: bad argument");
    MRG_REQUIRE((((uintptr_t)src & 15) | ((uintptr_t)dst_bf16 & 7)) == 0, MRG_ERR_ARG, "mrg_cast_bf16: misaligned");
    long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, stream, src, (bf16_t*)dst_bf16, n);
    return mrg_check_launch("mrg_cast_bf16");
}

extern "C" int mrg_row_scale_bf16(const float* src, const float* row_scale, void* dst_bf16, long M, int D, hipStream_t stream) {
    MRG_REQUIRE(src && row_scale && dst_bf16 && M > 0 && D > 0, MRG_ERR_ARG, "mrg_row_scale_bf16: bad argument");
    MRG_REQUIRE(D % 4 == 0, MRG_ERR_SHAPE, "mrg_row_scale_bf16: D must be a multiple of 4");
    MRG_REQUIRE((((uintptr_t)src & 15) | ((uintptr_t)dst_bf16 & 7)) == 0, MRG_ERR_ARG, "mrg_row_scale_bf16: misaligned");
    long nb = (M * (D / 4) + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(row_scale_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, stream, src, row_scale, (bf16_t*)dst_bf16, M, D);
    return mrg_check_launch("mrg_row_scale_bf16");
}

extern "C" int mrg_dropout_bwd_bf16(const float* src, const float* row_scale, void* dst_bf16, long M, int D, const void* drop_rng, float drop_p,
                                    unsigned drop_stream, hipStream_t stream) {
    MRG_REQUIRE(src && dst_bf16 && drop_rng && M > 0 && D > 0, MRG_ERR_ARG, "mrg_dropout_bwd_bf16: bad argument");
    MRG_REQUIRE(D % 4 == 0, MRG_ERR_SHAPE, "mrg_dropout_bwd_bf16: D must be a multiple of 4");
    MRG_REQUIRE(drop_p > 0.f && drop_p < 1.f, MRG_ERR_ARG, "mrg_dropout_bwd_bf16: drop_p = %g must be in (0, 1)", (double)drop_p);
    MRG_REQUIRE((((uintptr_t)src & 15) | ((uintptr_t)dst_bf16 & 7) | ((uintptr_t)drop_rng & 7)) == 0, MRG_ERR_ARG, "mrg_dropout_bwd_bf16: misaligned");
    MrgDropout dr;
    const double t = floor((double)drop_p * 4294967296.0);
    dr.rng = (const unsigned long long*)drop_rng;
    dr.thresh = (uint32_t)(t > 4294967295.0 ? 4294967295.0 : t);
    dr.scale = 1.f / (1.f - drop_p);
    dr.stream = drop_stream;
    long nb = (M * (D / 4) + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(dropout_bwd_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, stream, src, row_scale, (bf16_t*)dst_bf16, M, D, dr);
    return mrg_check_launch("mrg_dropout_bwd_bf16");
}
