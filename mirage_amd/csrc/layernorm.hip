// LayerNorm forward / backward, wave-per-row (replaces aten native_layer_norm x45 per step and its backward;
// reference call sites mirage/utils.py:260-261, mirage/output_adapters.py:279-280; eps = 1e-6, fp32 statistics
// as under the reference's autocast map).
//
// HBM-bound: forward reads 4 B/elem (fp32 residual stream), writes 2 B/elem (bf16 GEMM operand) + 8 B/row stats.
// Backward reads dy (bf16) + x (fp32) + the incoming residual gradient (fp32), writes dx as fp32 and as the bf16
// operand of the next dgrad/wgrad GEMM, and produces per-block partial column sums for dgamma, dbeta and for the
// bias gradient of the residual-producing Linear (= colsum of the incoming residual gradient), finished by
// mrg_colsum_partials: deterministic two-stage reduction, no atomics.
#include "common.h"
#include "mirage_hip.h"

namespace {

constexpr int LN_MAX_V = 4;  // float4 chunks per lane: D <= 64*4*4 = 1024

__device__ __forceinline__ float4 bf4_to_f4(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ y32, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * D);
    float4 v[LN_MAX_V];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_V; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) { v[j] = xr[c]; s += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
        else v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_V; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) {
            const float a = v[j].x - mu, b = v[j].y - mu, c2 = v[j].z - mu, d = v[j].w - mu;
            q += (a * a + b * b) + (c2 * c2 + d * d);
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int j = 0; j < LN_MAX_V; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) {
            const float4 g = g4[c], b = b4[c];
            const float o0 = (v[j].x - mu) * rs * g.x + b.x, o1 = (v[j].y - mu) * rs * g.y + b.y;
            const float o2 = (v[j].z - mu) * rs * g.z + b.z, o3 = (v[j].w - mu) * rs * g.w + b.w;
            if (y) reinterpret_cast<uint2*>(y + (long)row * D)[c] = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
            if (y32) reinterpret_cast<float4*>(y32 + (long)row * D)[c] = make_float4(o0, o1, o2, o3);
        }
    }
}

// dx32 = dres + LN'(dy); dx16 = bf16(dx32).  partials [gridDim.x][3][D]: dgamma, dbeta, colsum(dres)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy16, const float* __restrict__ dy32,
                                                     const float* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ dres, float* __restrict__ dx32,
                                                     bf16_t* __restrict__ dx16, float* __restrict__ partials, int M, int D) {
    __shared__ __attribute__((aligned(16))) float buf[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 2;
    float4 ag[LN_MAX_V], ab[LN_MAX_V], ar[LN_MAX_V], gm[LN_MAX_V];
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
#pragma unroll
    for (int j = 0; j < LN_MAX_V; ++j) {
        ag[j] = ab[j] = ar[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = lane + 64 * j;
        gm[j] = c < nv ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float4 xh[LN_MAX_V], dyv[LN_MAX_V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAX_V; ++j) {
            const int c = lane + 64 * j;
            if (c < nv) {
                const float4 xv = reinterpret_cast<const float4*>(x + (long)row * D)[c];
                const float4 d = dy16 ? bf4_to_f4(reinterpret_cast<const uint2*>(dy16 + (long)row * D)[c])
                                      : reinterpret_cast<const float4*>(dy32 + (long)row * D)[c];
                xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                dyv[j] = d;
                const float w0 = d.x * gm[j].x, w1 = d.y * gm[j].y, w2 = d.z * gm[j].z, w3 = d.w * gm[j].w;
                s1 += (w0 + w1) + (w2 + w3);
                s2 += (w0 * xh[j].x + w1 * xh[j].y) + (w2 * xh[j].z + w3 * xh[j].w);
                ag[j].x += d.x * xh[j].x; ag[j].y += d.y * xh[j].y; ag[j].z += d.z * xh[j].z; ag[j].w += d.w * xh[j].w;
                ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
            }
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int j = 0; j < LN_MAX_V; ++j) {
            const int c = lane + 64 * j;
            if (c < nv) {
                float4 o;
                o.x = rs * (dyv[j].x * gm[j].x - m1 - xh[j].x * m2);
                o.y = rs * (dyv[j].y * gm[j].y - m1 - xh[j].y * m2);
                o.z = rs * (dyv[j].z * gm[j].z - m1 - xh[j].z * m2);
                o.w = rs * (dyv[j].w * gm[j].w - m1 - xh[j].w * m2);
                if (dres) {
                    const float4 r = reinterpret_cast<const float4*>(dres + (long)row * D)[c];
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                    ar[j].x += r.x; ar[j].y += r.y; ar[j].z += r.z; ar[j].w += r.w;
                }
                if (dx32) reinterpret_cast<float4*>(dx32 + (long)row * D)[c] = o;
                if (dx16) reinterpret_cast<uint2*>(dx16 + (long)row * D)[c] = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
            }
        }
    }
    float* outp = partials + (long)blockIdx.x * 3 * D;
#pragma unroll 1
    for (int qn = 0; qn < 3; ++qn) {
#pragma unroll
        for (int j = 0; j < LN_MAX_V; ++j) {
            const int c = lane + 64 * j;
            if (c < nv) reinterpret_cast<float4*>(&buf[wave][0])[c] = qn == 0 ? ag[j] : (qn == 1 ? ab[j] : ar[j]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) outp[qn * D + c] = (buf[0][c] + buf[1][c]) + (buf[2][c] + buf[3][c]);
        __syncthreads();
    }
}

// out_q[c] = sum_b partials[b][q][c]; grid (ceil(D/64), nq), 256 threads: lane = column, 4 waves split the blocks
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ partials, int nblocks, int nq, int D,
                                                              float* o0, float* o1, float* o2) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, q = blockIdx.y;
    float* out = q == 0 ? o0 : (q == 1 ? o1 : o2);
    if (out == nullptr) return;
    float s = 0.f;
    if (c < D)
        for (int b = wave; b < nblocks; b += 4) s += partials[((long)b * nq + q) * D + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < D) out[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// generic column sum of a [M, N] matrix (bf16 or fp32) with optional row gather map and device row count:
// partials [gridDim.y][N]; finish with colsum_partials (nq = 1).  Used for the bias gradients that are not a
// by-product of a LayerNorm backward (qkv / fc1 / kv / q / out_proj / patch-embed biases).
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x16, const float* __restrict__ x32, int M, int N,
                                                     int ld, const int* __restrict__ row_map, const int* __restrict__ m_dev,
                                                     float* __restrict__ partials) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int Meff = m_dev ? min(M, *m_dev) : M;
    float s = 0.f;
    if (c < N)
        for (int r = blockIdx.y * 4 + wave; r < Meff; r += gridDim.y * 4) {
            const long src = (long)(row_map ? row_map[r] : r) * ld + c;
            s += x16 ? bf2f(x16[src]) : x32[src];
        }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < N) partials[(long)blockIdx.y * N + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

}  // namespace

extern "C" int mrg_ln_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32, float* mean,
                          float* rstd, int M, int D, float eps, hipStream_t stream) {
    MRG_REQUIRE(x && gamma && beta && mean && rstd && (y_bf16 || y_f32), MRG_ERR_ARG, "mrg_ln_fwd: null pointer");
    MRG_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, MRG_ERR_SHAPE, "mrg_ln_fwd: need D %% 4 == 0 and D <= 1024 (D=%d)", D);
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, stream, x, gamma, beta, (bf16_t*)y_bf16, y_f32, mean, rstd, M,
                       D, eps);
    return mrg_check_launch("mrg_ln_fwd");
}

extern "C" int mrg_ln_bwd_nblocks(int M) {
    int nb = cdiv(M, 16);  // >= 4 rows per wave
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    return nb;
}

extern "C" int mrg_ln_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const float* mean, const float* rstd,
                          const float* gamma, const float* dres, float* dx_f32, void* dx_bf16, float* partials, float* dgamma,
                          float* dbeta, float* dres_colsum, int M, int D, hipStream_t stream) {
    MRG_REQUIRE((dy_bf16 || dy_f32) && x && mean && rstd && gamma && partials && (dx_f32 || dx_bf16), MRG_ERR_ARG,
                "mrg_ln_bwd: null pointer");
    MRG_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, MRG_ERR_SHAPE, "mrg_ln_bwd: need D %% 4 == 0 and D <= 1024 (D=%d)", D);
    MRG_REQUIRE(!dres_colsum || dres, MRG_ERR_ARG, "mrg_ln_bwd: dres_colsum without dres");
    const int nb = mrg_ln_bwd_nblocks(M);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb), dim3(256), 0, stream, (const bf16_t*)dy_bf16, dy_f32, x, mean, rstd, gamma, dres,
                       dx_f32, (bf16_t*)dx_bf16, partials, M, D);
    int rc = mrg_check_launch("mrg_ln_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_partials_kernel, dim3(cdiv(D, 64), 3), dim3(256), 0, stream, partials, nb, 3, D, dgamma, dbeta,
                       dres_colsum);
    return mrg_check_launch("mrg_ln_bwd/colsum_partials");
}

extern "C" int mrg_colsum_nblocks(int M) {
    int nb = cdiv(M, 64);
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    return nb;
}

extern "C" int mrg_colsum(const void* x, int x_dtype, int M, int N, int ld, const int* row_map, const int* m_dev, float* partials,
                          float* out, hipStream_t stream) {
    MRG_REQUIRE(x && partials && out, MRG_ERR_ARG, "mrg_colsum: null pointer");
    MRG_REQUIRE(M > 0 && N > 0, MRG_ERR_SHAPE, "mrg_colsum: bad shape");
    const int nb = mrg_colsum_nblocks(M);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64), nb), dim3(256), 0, stream, x_dtype == MRG_DT_BF16 ? (const bf16_t*)x : nullptr,
                       x_dtype == MRG_DT_F32 ? (const float*)x : nullptr, M, N, ld, row_map, m_dev, partials);
    int rc = mrg_check_launch("mrg_colsum");
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_partials_kernel, dim3(cdiv(N, 64), 1), dim3(256), 0, stream, partials, nb, 1, N, out, (float*)nullptr,
                       (float*)nullptr);
    return mrg_check_launch("mrg_colsum/colsum_partials");
}
