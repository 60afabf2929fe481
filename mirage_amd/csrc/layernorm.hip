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
#include <stdlib.h>
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
                                                     float* __restrict__ rstd, int M, int D, float eps, int band) {
    const int lane = threadIdx.x & 63;
    // band: workgroups are dispatched round-robin over the 8 XCDs; give XCD x one contiguous band of rows -- the band of A rows the
    // consuming GEMM's workgroups on the same XCD read (gemm_common.h wg_remap), so that they find them in their own L2
    int blk = blockIdx.x;
    if (band) {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blk & 7, idx = blk >> 3;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int row = blk * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * D);
    float4 v[LN_MAX_V], gv[LN_MAX_V], bvv[LN_MAX_V];
    float s = 0.f;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    const float invD = 1.f / (float)D;
#pragma unroll
    for (int j = 0; j < LN_MAX_V; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) { v[j] = xr[c]; gv[j] = g4[c]; bvv[j] = b4[c]; }     // gamma / beta requested with the row, not after the reductions
        else v[j] = gv[j] = bvv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < LN_MAX_V; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_V; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) {
            const float a = v[j].x - mu, b = v[j].y - mu, c2 = v[j].z - mu, d = v[j].w - mu;
            q += (a * a + b * b) + (c2 * c2 + d * d);
        }
    }
    const float rs = rsqrtf(wave_sum(q) * invD + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int j = 0; j < LN_MAX_V; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) {
            const float4 g = gv[j], b = bvv[j];
            const float o0 = (v[j].x - mu) * rs * g.x + b.x, o1 = (v[j].y - mu) * rs * g.y + b.y;
            const float o2 = (v[j].z - mu) * rs * g.z + b.z, o3 = (v[j].w - mu) * rs * g.w + b.w;
            if (y) reinterpret_cast<uint2*>(y + (long)row * D)[c] = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
            if (y32) reinterpret_cast<float4*>(y32 + (long)row * D)[c] = make_float4(o0, o1, o2, o3);
        }
    }
}

// dx32 = dres + LN'(dy); dx16 = bf16(dx32).  partials [gridDim.x][3][D]: dgamma, dbeta, colsum(dres).
// Templated on the row width (NV float4 chunks per lane) and on the optional operands so that the row loop is
// straight-line code: all loads of a row are issued back to back and registers are sized for the real D.
constexpr int LNB_WAVES = 4;
// (Round 6 tried the opposite trade: no look-ahead row, 3 waves per SIMD and a grid on which every wave owns the same number of rows -- the
// BASELINE shapes leave 64 / 192 of the 2048 waves a row more than the rest.  Slower: 25.2 / 27.6 us against 23.6 / 24.6 at 8256 x 768 /
// 6336 x 1024; the look-ahead is worth more than the occupancy.  profiles/r06_experiments.md)
template <int NV, bool DY16, bool HAS_RES>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dyp, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ dx32, bf16_t* __restrict__ dx16,
                                                     float* __restrict__ partials, int M, int D) {
    __shared__ __attribute__((aligned(16))) float buf[LNB_WAVES][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 2;
    const float invD = 1.f / (float)D;
    float4 ag[NV], ab[NV], ar[NV], gm[NV];
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        ag[j] = ab[j] = ar[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = lane + 64 * j;
        gm[j] = c < nv ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bf16_t* dy16 = reinterpret_cast<const bf16_t*>(dyp);
    const float* dy32 = reinterpret_cast<const float*>(dyp);
    // the next row's operands are requested before this row's reductions and stores (a wave owns one row at a time; without the
    // prefetch its load -> reduce -> store chain leaves the memory pipe idle for most of each row)
    float4 xn[NV], dn[NV], rn[NV];
    float mun = 0.f, rsn = 0.f;
    auto fetch = [&](int row) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j;
            const int cc = c < nv ? c : 0;
            xn[j] = reinterpret_cast<const float4*>(x + (long)row * D)[cc];
            if (DY16) { const uint2 u = reinterpret_cast<const uint2*>(dy16 + (long)row * D)[cc]; dn[j].x = __uint_as_float(u.x); dn[j].y = __uint_as_float(u.y); }
            else dn[j] = reinterpret_cast<const float4*>(dy32 + (long)row * D)[cc];
            if (HAS_RES) rn[j] = reinterpret_cast<const float4*>(dres + (long)row * D)[cc];
        }
        mun = mean[row]; rsn = rstd[row];
    };
    const int row0 = blockIdx.x * LNB_WAVES + wave, rstep = gridDim.x * LNB_WAVES;
    if (row0 < M) fetch(row0);
    for (int row = row0; row < M; row += rstep) {
        float4 xv[NV], dyv[NV], rv[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool ok = (lane + 64 * j) < nv;
            xv[j] = xn[j];
            if (DY16) dyv[j] = bf4_to_f4(make_uint2(__float_as_uint(dn[j].x), __float_as_uint(dn[j].y)));
            else dyv[j] = dn[j];
            if (HAS_RES) rv[j] = rn[j];
            if (!ok) { xv[j] = dyv[j] = make_float4(0.f, 0.f, 0.f, 0.f); if (HAS_RES) rv[j] = xv[j]; }
        }
        const float mu_c = mun, rs_c = rsn;
        if (row + rstep < M) fetch(row + rstep);
        const float mu = mu_c, rs = rs_c;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool ok = (lane + 64 * j) < nv;
            float4 xh = make_float4((xv[j].x - mu) * rs, (xv[j].y - mu) * rs, (xv[j].z - mu) * rs, (xv[j].w - mu) * rs);
            if (!ok) xh = make_float4(0.f, 0.f, 0.f, 0.f);
            xv[j] = xh;
            const float4 d = dyv[j];
            const float w0 = d.x * gm[j].x, w1 = d.y * gm[j].y, w2 = d.z * gm[j].z, w3 = d.w * gm[j].w;
            s1 += (w0 + w1) + (w2 + w3);
            s2 += (w0 * xh.x + w1 * xh.y) + (w2 * xh.z + w3 * xh.w);
            ag[j].x += d.x * xh.x; ag[j].y += d.y * xh.y; ag[j].z += d.z * xh.z; ag[j].w += d.w * xh.w;
            ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
        }
        const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j;
            if (c < nv) {
                float4 o;
                o.x = rs * (dyv[j].x * gm[j].x - m1 - xv[j].x * m2);
                o.y = rs * (dyv[j].y * gm[j].y - m1 - xv[j].y * m2);
                o.z = rs * (dyv[j].z * gm[j].z - m1 - xv[j].z * m2);
                o.w = rs * (dyv[j].w * gm[j].w - m1 - xv[j].w * m2);
                if (HAS_RES) {
                    o.x += rv[j].x; o.y += rv[j].y; o.z += rv[j].z; o.w += rv[j].w;
                    ar[j].x += rv[j].x; ar[j].y += rv[j].y; ar[j].z += rv[j].z; ar[j].w += rv[j].w;
                }
                if (dx32) reinterpret_cast<float4*>(dx32 + (long)row * D)[c] = o;
                if (dx16) reinterpret_cast<uint2*>(dx16 + (long)row * D)[c] = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
            }
        }
    }
    float* outp = partials + (long)blockIdx.x * 3 * D;
    // fully unrolled: selecting among the three accumulator arrays with a run-time index would demote them to scratch memory
    // (they then cost a read-modify-write of 48 * NV bytes per lane per row)
#pragma unroll
    for (int qn = 0; qn < 3; ++qn) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j;
            if (c < nv) reinterpret_cast<float4*>(&buf[wave][0])[c] = qn == 0 ? ag[j] : (qn == 1 ? ab[j] : ar[j]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) outp[qn * D + c] = (buf[0][c] + buf[1][c]) + (buf[2][c] + buf[3][c]);
        __syncthreads();
    }
}


// ---- rows of at most 256 elements (the 256-wide output adapters, mirage/output_adapters.py:279-280 and their decoder blocks) -------
// A wave per row leaves 48 of 64 lanes with one float4 and makes every row pay two 6-step cross-lane reductions: 13.9 us for a
// 16384 x 256 forward whose traffic is worth 5 us, and no faster for a third of the rows.  Here 16 lanes own a row (NV float4 each),
// a wave works on 4 rows at once and the reductions are 4 DPP steps inside the 16-lane row (quad swaps, half mirror, row mirror).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                           float* __restrict__ y32, float* __restrict__ mean,
                                                           float* __restrict__ rstd, int M, int D, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15;                       // lane within the row
    const int row = (blockIdx.x * 4 + wave) * 4 + (lane >> 4);
    const int nv = D >> 2;
    const float invD = 1.f / (float)D;
    const int rr = row < M ? row : M - 1;            // clamped: every lane takes part in the DPP reductions
    const float4* xr = reinterpret_cast<const float4*>(x + (long)rr * D);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    float4 v[NV], gv[NV], bv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = sub + 16 * j;
        if (c < nv) { v[j] = xr[c]; gv[j] = g4[c]; bv[j] = b4[c]; }
        else v[j] = gv[j] = bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mu = row16_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (sub + 16 * j < nv) {
            const float a = v[j].x - mu, b = v[j].y - mu, c2 = v[j].z - mu, d = v[j].w - mu;
            q += (a * a + b * b) + (c2 * c2 + d * d);
        }
    }
    const float rs = rsqrtf(row16_sum(q) * invD + eps);
    if (row >= M) return;
    if (sub == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = sub + 16 * j;
        if (c < nv) {
            const float4 g = gv[j], b = bv[j];
            const float o0 = (v[j].x - mu) * rs * g.x + b.x, o1 = (v[j].y - mu) * rs * g.y + b.y;
            const float o2 = (v[j].z - mu) * rs * g.z + b.z, o3 = (v[j].w - mu) * rs * g.w + b.w;
            if (y) reinterpret_cast<uint2*>(y + (long)row * D)[c] = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
            if (y32) reinterpret_cast<float4*>(y32 + (long)row * D)[c] = make_float4(o0, o1, o2, o3);
        }
    }
}

// backward for rows of at most 256 elements: same contract as ln_bwd_kernel (partials [gridDim.x][3][D]); 16 lanes per row, a wave
// walks 4 rows per step with the next step's operands requested before this step's reductions.
template <int NV, bool DY16, bool HAS_RES>
__global__ __launch_bounds__(256) void ln_bwd_small_kernel(const void* __restrict__ dyp, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ dres,
                                                           float* __restrict__ dx32, bf16_t* __restrict__ dx16,
                                                           float* __restrict__ partials, int M, int D) {
    __shared__ __attribute__((aligned(16))) float buf[16][NV * 64];         // [wave * 4 + row group][D <= 256]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, grp = lane >> 4;
    const int nv = D >> 2;
    const float invD = 1.f / (float)D;
    float4 ag[NV], ab[NV], ar[NV], gm[NV];
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        ag[j] = ab[j] = ar[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = sub + 16 * j;
        gm[j] = c < nv ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bf16_t* dy16 = reinterpret_cast<const bf16_t*>(dyp);
    const float* dy32 = reinterpret_cast<const float*>(dyp);
    float4 xn[NV], dn[NV], rn[NV];
    float mun = 0.f, rsn = 0.f;
    auto fetch = [&](int row) {
        const int r = row < M ? row : M - 1;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = sub + 16 * j;
            const int cc = c < nv ? c : 0;
            xn[j] = reinterpret_cast<const float4*>(x + (long)r * D)[cc];
            if (DY16) { const uint2 u = reinterpret_cast<const uint2*>(dy16 + (long)r * D)[cc]; dn[j].x = __uint_as_float(u.x); dn[j].y = __uint_as_float(u.y); }
            else dn[j] = reinterpret_cast<const float4*>(dy32 + (long)r * D)[cc];
            if (HAS_RES) rn[j] = reinterpret_cast<const float4*>(dres + (long)r * D)[cc];
        }
        mun = mean[r]; rsn = rstd[r];
    };
    const int row0 = (blockIdx.x * 4 + wave) * 4 + grp, rstep = gridDim.x * 16;
    const int base0 = (blockIdx.x * 4 + wave) * 4;          // wave-uniform loop bound: all four row groups run the same trip count
    if (base0 < M) fetch(row0);
    for (int base = base0, row = row0; base < M; base += rstep, row += rstep) {
        const bool live = row < M;
        float4 xv[NV], dyv[NV], rv[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool ok = live && (sub + 16 * j) < nv;
            xv[j] = xn[j];
            if (DY16) dyv[j] = bf4_to_f4(make_uint2(__float_as_uint(dn[j].x), __float_as_uint(dn[j].y)));
            else dyv[j] = dn[j];
            if (HAS_RES) rv[j] = rn[j];
            if (!ok) { xv[j] = dyv[j] = make_float4(0.f, 0.f, 0.f, 0.f); if (HAS_RES) rv[j] = xv[j]; }
        }
        const float mu = mun, rs = rsn;
        if (base + rstep < M) fetch(row + rstep);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool ok = live && (sub + 16 * j) < nv;
            float4 xh = make_float4((xv[j].x - mu) * rs, (xv[j].y - mu) * rs, (xv[j].z - mu) * rs, (xv[j].w - mu) * rs);
            if (!ok) xh = make_float4(0.f, 0.f, 0.f, 0.f);
            xv[j] = xh;
            const float4 d = dyv[j];
            const float w0 = d.x * gm[j].x, w1 = d.y * gm[j].y, w2 = d.z * gm[j].z, w3 = d.w * gm[j].w;
            s1 += (w0 + w1) + (w2 + w3);
            s2 += (w0 * xh.x + w1 * xh.y) + (w2 * xh.z + w3 * xh.w);
            ag[j].x += d.x * xh.x; ag[j].y += d.y * xh.y; ag[j].z += d.z * xh.z; ag[j].w += d.w * xh.w;
            ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
        }
        const float m1 = row16_sum(s1) * invD, m2 = row16_sum(s2) * invD;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = sub + 16 * j;
            if (live && c < nv) {
                float4 o;
                o.x = rs * (dyv[j].x * gm[j].x - m1 - xv[j].x * m2);
                o.y = rs * (dyv[j].y * gm[j].y - m1 - xv[j].y * m2);
                o.z = rs * (dyv[j].z * gm[j].z - m1 - xv[j].z * m2);
                o.w = rs * (dyv[j].w * gm[j].w - m1 - xv[j].w * m2);
                if (HAS_RES) {
                    o.x += rv[j].x; o.y += rv[j].y; o.z += rv[j].z; o.w += rv[j].w;
                    ar[j].x += rv[j].x; ar[j].y += rv[j].y; ar[j].z += rv[j].z; ar[j].w += rv[j].w;
                }
                if (dx32) reinterpret_cast<float4*>(dx32 + (long)row * D)[c] = o;
                if (dx16) reinterpret_cast<uint2*>(dx16 + (long)row * D)[c] = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
            }
        }
    }
    float* outp = partials + (long)blockIdx.x * 3 * D;
#pragma unroll
    for (int qn = 0; qn < 3; ++qn) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = sub + 16 * j;
            if (c < nv) reinterpret_cast<float4*>(&buf[wave * 4 + grp][0])[c] = qn == 0 ? ag[j] : (qn == 1 ? ab[j] : ar[j]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) t += buf[w][c];
            outp[qn * D + c] = t;
        }
        __syncthreads();
    }
}

// out_q[c] = sum_b partials[b][q][c]; grid (ceil(D/64), nq), 1024 threads: lane = column, 16 waves split the rows
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ partials, int nblocks, int nq, int D,
                                                               float* o0, float* o1, float* o2) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, q = blockIdx.y;
    float* out = q == 0 ? o0 : (q == 1 ? o1 : o2);
    if (out == nullptr) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < D) {
        int b = wave;
        for (; b + 48 < nblocks; b += 64) {
            s0 += partials[((long)b * nq + q) * D + c];
            s1 += partials[((long)(b + 16) * nq + q) * D + c];
            s2 += partials[((long)(b + 32) * nq + q) * D + c];
            s3 += partials[((long)(b + 48) * nq + q) * D + c];
        }
        for (; b < nblocks; b += 16) s0 += partials[((long)b * nq + q) * D + c];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && c < D) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += red[w][lane];
        out[c] = s;
    }
}

// generic column sum of a [M, N] matrix (bf16 or fp32) with optional row gather map and device row count:
// partials [gridDim.x][N]; finish with colsum_partials (nq = 1).  Used for the bias gradients that are not a
// by-product of a LayerNorm backward (qkv / fc1 / kv / q / out_proj / patch-embed biases).
// Each thread owns 8 consecutive columns (16-B bf16 / 2x16-B fp32 loads): a wave reads 1 KiB+ contiguous per row.
template <bool IS_BF16>
__global__ __launch_bounds__(256) void colsum_wide_kernel(const void* __restrict__ xin, int M, int N, int ld,
                                                          const int* __restrict__ row_map, const int* __restrict__ m_dev,
                                                          float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [rowlanes][ncg*8]
    const int ncg = N >> 3;                                        // column groups of 8
    const int Meff = m_dev ? min(M, *m_dev) : M;
    const int per_row = ncg < 256 ? ncg : 256;                     // threads along columns
    const int rowlanes = 256 / per_row;                            // >= 1
    const int cgl = threadIdx.x % per_row, rl = threadIdx.x / per_row;
    for (int cg0 = 0; cg0 < ncg; cg0 += per_row) {
        const int cg = cg0 + cgl;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (cg < ncg && rl < rowlanes) {
            for (int r = blockIdx.x * rowlanes + rl; r < Meff; r += gridDim.x * rowlanes) {
                const long src = (long)(row_map ? row_map[r] : r) * ld + (cg << 3);
                if (IS_BF16) {
                    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(xin) + src);
                    a[0] += __uint_as_float(u.x << 16); a[1] += __uint_as_float(u.x & 0xffff0000u);
                    a[2] += __uint_as_float(u.y << 16); a[3] += __uint_as_float(u.y & 0xffff0000u);
                    a[4] += __uint_as_float(u.z << 16); a[5] += __uint_as_float(u.z & 0xffff0000u);
                    a[6] += __uint_as_float(u.w << 16); a[7] += __uint_as_float(u.w & 0xffff0000u);
                } else {
                    const float4 v0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xin) + src);
                    const float4 v1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xin) + src + 4);
                    a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w; a[4] += v1.x; a[5] += v1.y; a[6] += v1.z; a[7] += v1.w;
                }
            }
        }
        if (rl < rowlanes && cg < ncg) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(rl * per_row + cgl) * 8 + e] = a[e];
        }
        __syncthreads();
        if (rl == 0 && cg < ncg) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s = 0.f;
                for (int q = 0; q < rowlanes; ++q) s += red[(q * per_row + cgl) * 8 + e];
                partials[(long)blockIdx.x * N + (cg << 3) + e] = s;
            }
        }
        __syncthreads();
    }
}

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

// ---- batched slice reductions: dst[i] = sum_s src[s * stride + i] for a table of independent tasks in ONE launch ----------
// (split-K partials of the weight gradients, LayerNorm / bias column-sum partials: ~165 tiny launches per training step
// otherwise).  Tasks travel by value in the kernel arguments, so a hipGraph node carries them without a staging copy.
constexpr int RB_TASKS = 64;
struct RbTask { const float* src; float* dst; long n4; long stride4; int ns; int first_block; };
struct RbArgs { RbTask t[RB_TASKS]; int ntasks; };

__device__ __forceinline__ int rb_find(const RbArgs& a, int bid) {
    int lo = 0, hi = a.ntasks - 1;
    while (lo < hi) {                       // last task whose first_block <= bid
        const int mid = (lo + hi + 1) >> 1;
        if (a.t[mid].first_block <= bid) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// few slices, many outputs (split-K): one float4 column per thread, slices summed in order
__global__ __launch_bounds__(256) void reduce_batch_wide_kernel(const RbArgs a) {
    const int ti = rb_find(a, blockIdx.x);
    const RbTask t = a.t[ti];
    const long i = (long)(blockIdx.x - t.first_block) * 256 + threadIdx.x;
    if (i >= t.n4) return;
    const float4* s = reinterpret_cast<const float4*>(t.src) + i;
    float4 acc = s[0];
    for (int z = 1; z < t.ns; ++z) {
        const float4 v = s[(long)z * t.stride4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(t.dst)[i] = acc;
}

// many slices, few outputs (column-sum partials): 64 float4 columns per workgroup, 16 waves split the slices
__global__ __launch_bounds__(1024) void reduce_batch_tall_kernel(const RbArgs a) {
    __shared__ float4 red[16][64];
    const int ti = rb_find(a, blockIdx.x);
    const RbTask t = a.t[ti];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i = (long)(blockIdx.x - t.first_block) * 64 + lane;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (i < t.n4) {
        const float4* s = reinterpret_cast<const float4*>(t.src) + i;
        int z = wave;
        for (; z + 16 < t.ns; z += 32) {
            const float4 u = s[(long)z * t.stride4], v = s[(long)(z + 16) * t.stride4];
            s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
            s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        }
        for (; z < t.ns; z += 16) { const float4 u = s[(long)z * t.stride4]; s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w; }
    }
    red[wave][lane] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (wave == 0 && i < t.n4) {
        float4 r = red[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) { const float4 v = red[w][lane]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        reinterpret_cast<float4*>(t.dst)[i] = r;
    }
}

}  // namespace

extern "C" int mrg_ln_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32, float* mean,
                          float* rstd, int M, int D, float eps, hipStream_t stream) {
    MRG_REQUIRE(x && gamma && beta && mean && rstd && (y_bf16 || y_f32), MRG_ERR_ARG, "mrg_ln_fwd: null pointer");
    MRG_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, MRG_ERR_SHAPE, "mrg_ln_fwd: need D %% 4 == 0 and D <= 1024 (D=%d)", D);
    static const int band = [] { const char* e = getenv("MRG_LN_BAND"); return e ? atoi(e) : 1; }();   // same box: C2 9.527 -> 9.500 ms (3 of 3 pairs), C4 within noise
    static const int small = [] { const char* e = getenv("MRG_LN_SMALL"); return e ? atoi(e) : 1; }();  // A/B: 0 = wave per row also for D <= 256
    if (small && D <= 256) {
        const dim3 grid(cdiv(M, 16)), block(256);
        const int nvr = cdiv(D, 64);
#define MRG_LNF(NV_) hipLaunchKernelGGL((ln_fwd_small_kernel<NV_>), grid, block, 0, stream, x, gamma, beta, (bf16_t*)y_bf16, y_f32, mean, rstd, M, D, eps)
        if (nvr == 1) MRG_LNF(1); else if (nvr == 2) MRG_LNF(2); else if (nvr == 3) MRG_LNF(3); else MRG_LNF(4);
#undef MRG_LNF
        return mrg_check_launch("mrg_ln_fwd");
    }
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, stream, x, gamma, beta, (bf16_t*)y_bf16, y_f32, mean, rstd, M,
                       D, eps, band);
    return mrg_check_launch("mrg_ln_fwd");
}

extern "C" int mrg_ln_bwd_nblocks(int M) {
    // two workgroups (of 4 waves) per CU: measured 32.7 us at 512 workgroups against 38.7 at 768 and 44.6 at 1024 on
    // 8256 x 768 (fewer partial rows to write and to reduce afterwards outweigh the extra memory parallelism); callers size the
    // partial buffer with this upper bound
    int nb = cdiv(M, 8);
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    return nb;
}

extern "C" int mrg_ln_bwd_partial_rows(int M, int D) {
    int nb = mrg_ln_bwd_nblocks(M);
    (void)D;
    return nb;
}

extern "C" int mrg_ln_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const float* mean, const float* rstd,
                          const float* gamma, const float* dres, float* dx_f32, void* dx_bf16, float* partials, float* dgamma,
                          float* dbeta, float* dres_colsum, int M, int D, hipStream_t stream) {
    MRG_REQUIRE((dy_bf16 || dy_f32) && x && mean && rstd && gamma && partials && (dx_f32 || dx_bf16), MRG_ERR_ARG,
                "mrg_ln_bwd: null pointer");
    MRG_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, MRG_ERR_SHAPE, "mrg_ln_bwd: need D %% 4 == 0 and D <= 1024 (D=%d)", D);
    MRG_REQUIRE(!dres_colsum || dres, MRG_ERR_ARG, "mrg_ln_bwd: dres_colsum without dres");
    const int nb = mrg_ln_bwd_partial_rows(M, D);
    const void* dyp = dy_bf16 ? dy_bf16 : (const void*)dy_f32;
    const int NVr = (D + 255) / 256;
#define MRG_LNB(NV_, D16, RES) hipLaunchKernelGGL((ln_bwd_kernel<NV_, D16, RES>), dim3(nb), dim3(256), 0, stream, dyp, x, mean, rstd, \
                                                   gamma, dres, dx_f32, (bf16_t*)dx_bf16, partials, M, D)
#define MRG_LNB2(NV_) do { if (dy_bf16) { if (dres) MRG_LNB(NV_, true, true); else MRG_LNB(NV_, true, false); } \
                           else { if (dres) MRG_LNB(NV_, false, true); else MRG_LNB(NV_, false, false); } } while (0)
    static const int small = [] { const char* e = getenv("MRG_LN_SMALL"); return e ? atoi(e) : 1; }();
    if (small && D <= 256) {
        const int nvs = cdiv(D, 64);
#define MRG_LNS(NV_, D16, RES) hipLaunchKernelGGL((ln_bwd_small_kernel<NV_, D16, RES>), dim3(nb), dim3(256), 0, stream, dyp, x, mean, rstd, \
                                                   gamma, dres, dx_f32, (bf16_t*)dx_bf16, partials, M, D)
#define MRG_LNS2(NV_) do { if (dy_bf16) { if (dres) MRG_LNS(NV_, true, true); else MRG_LNS(NV_, true, false); } \
                           else { if (dres) MRG_LNS(NV_, false, true); else MRG_LNS(NV_, false, false); } } while (0)
        if (nvs == 1) MRG_LNS2(1); else if (nvs == 2) MRG_LNS2(2); else if (nvs == 3) MRG_LNS2(3); else MRG_LNS2(4);
#undef MRG_LNS2
#undef MRG_LNS
    } else
    if (NVr == 1) MRG_LNB2(1); else if (NVr == 2) MRG_LNB2(2); else if (NVr == 3) MRG_LNB2(3); else MRG_LNB2(4);
#undef MRG_LNB2
#undef MRG_LNB
    int rc = mrg_check_launch("mrg_ln_bwd");
    if (rc) return rc;
    if (!dgamma && !dbeta && !dres_colsum) return MRG_OK;   // partials [nb][3][D] left to the caller (mrg_reduce_batch)
    hipLaunchKernelGGL(colsum_partials_kernel, dim3(cdiv(D, 64), 3), dim3(1024), 0, stream, partials, nb, 3, D, dgamma, dbeta,
                       dres_colsum);
    return mrg_check_launch("mrg_ln_bwd/colsum_partials");
}

extern "C" int mrg_colsum_nblocks(int M) {
    int nb = cdiv(M, 16);
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    return nb;
}

extern "C" int mrg_colsum(const void* x, int x_dtype, int M, int N, int ld, const int* row_map, const int* m_dev, float* partials,
                          float* out, hipStream_t stream) {
    MRG_REQUIRE(x && partials, MRG_ERR_ARG, "mrg_colsum: null pointer");
    MRG_REQUIRE(M > 0 && N > 0, MRG_ERR_SHAPE, "mrg_colsum: bad shape");
    const int nb = mrg_colsum_nblocks(M);
    const bool is16 = x_dtype == MRG_DT_BF16;
    const bool wide = (N % 8 == 0) && (ld % 8 == 0) && (((uintptr_t)x & 15) == 0);
    if (wide) {
        const int ncg = N / 8;
        const size_t sh = (size_t)256 * 8 * 4;
        (void)ncg;
        if (is16) hipLaunchKernelGGL((colsum_wide_kernel<true>), dim3(nb), dim3(256), sh, stream, x, M, N, ld, row_map, m_dev, partials);
        else hipLaunchKernelGGL((colsum_wide_kernel<false>), dim3(nb), dim3(256), sh, stream, x, M, N, ld, row_map, m_dev, partials);
    } else {
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64), nb), dim3(256), 0, stream, is16 ? (const bf16_t*)x : nullptr,
                           !is16 ? (const float*)x : nullptr, M, N, ld, row_map, m_dev, partials);
    }
    int rc = mrg_check_launch("mrg_colsum");
    if (rc) return rc;
    if (!out) return MRG_OK;                                 // partials [nb][N] left to the caller (mrg_reduce_batch)
    hipLaunchKernelGGL(colsum_partials_kernel, dim3(cdiv(N, 64), 1), dim3(1024), 0, stream, partials, nb, 1, N, out, (float*)nullptr,
                       (float*)nullptr);
    return mrg_check_launch("mrg_colsum/colsum_partials");
}

extern "C" int mrg_reduce_batch(const mrg_reduce_task* tasks, int ntasks, hipStream_t stream) {
    MRG_REQUIRE(tasks && ntasks > 0, MRG_ERR_ARG, "mrg_reduce_batch: no tasks");
    for (int pass = 0; pass < 2; ++pass) {                 // pass 0: wide (few slices), pass 1: tall
        RbArgs a;
        a.ntasks = 0;
        long nblocks = 0;
        auto flush = [&]() -> int {
            if (a.ntasks == 0) return MRG_OK;
            if (pass == 0) hipLaunchKernelGGL(reduce_batch_wide_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, a);
            else hipLaunchKernelGGL(reduce_batch_tall_kernel, dim3((unsigned)nblocks), dim3(1024), 0, stream, a);
            a.ntasks = 0;
            nblocks = 0;
            return mrg_check_launch("mrg_reduce_batch");
        };
        for (int i = 0; i < ntasks; ++i) {
            const mrg_reduce_task& t = tasks[i];
            MRG_REQUIRE(t.src && t.dst && t.n > 0 && t.nslices > 0, MRG_ERR_ARG, "mrg_reduce_batch: bad task %d", i);
            MRG_REQUIRE(t.n % 4 == 0 && t.stride % 4 == 0 && ((((uintptr_t)t.src) | ((uintptr_t)t.dst)) & 15) == 0, MRG_ERR_SHAPE,
                        "mrg_reduce_batch: task %d is not 16-byte aligned / a multiple of 4 floats", i);
            const bool tall = t.nslices >= 32;
            if (tall != (pass == 1)) continue;
            const long n4 = t.n / 4;
            const long nb = tall ? (n4 + 63) / 64 : (n4 + 255) / 256;
            if (a.ntasks == RB_TASKS || nblocks + nb > 0x7fffffffl) { int rc = flush(); if (rc) return rc; }
            a.t[a.ntasks] = RbTask{t.src, t.dst, n4, t.stride / 4, t.nslices, (int)nblocks};
            a.ntasks += 1;
            nblocks += nb;
        }
        int rc = flush();
        if (rc) return rc;
    }
    return MRG_OK;
}
