// bf16 MFMA GEMM for gfx950 with fused epilogues.
//
//   C[M,N] = A(M x K) * B(N x K)^T   (fp32 accumulate on v_mfma_f32_32x32x16_bf16)
//
// Operand modes (per operand):
//   KCONTIG  : element (row, k) at P[row*ld + k]   -- activations [M,K], weights [out,in]
//   KSTRIDED : element (row, k) at P[k*ld + row]   -- the "transposed" view needed by dgrad (W as [N',K'])
//              and wgrad (dY and X both contracted over the token axis).  Fragments come out of LDS with
//              ds_read_b64_tr_b16, so no transposed copies of weights or activations ever exist in HBM.
// This replaces the reference's aten addmm / mm dispatches (mirage/utils.py:155-157,177,186,209-221 and
// autograd mm x186).
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles, register-staged
// double-buffered LDS (one barrier per K tile), XOR-swizzled LDS images (conflict-free ds_read_b128 /
// tr reads), XCD-aware tile order.
#include "gemm_common.h"
#include <math.h>

namespace {

template <int AMODE, int BMODE, bool ALIGNED, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous band of logical tiles
    const int nwg = g.tiles_m * g.tiles_n;
    int bid, ksplit;
    wg_remap(nwg, g.xcd_split, bid, ksplit);
    const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Meff = g.m_dev ? min(g.M, *g.m_dev) : g.M;
    const int Keff = g.k_dev ? min(g.K, *g.k_dev) : g.K;
    if (m0 >= Meff) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int ntiles = (Keff + BK - 1) / BK;
    int t0 = 0, t1 = ntiles;
    if (g.splitk > 1) {
        const int per = (ntiles + g.splitk - 1) / g.splitk;
        t0 = ksplit * per;
        t1 = min(ntiles, t0 + per);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const uint32_t sbase = lds_addr(smem);

    uint4 ra[4], rb[4];
    if (t0 < t1) {
        load_tile<AMODE, ALIGNED>(ra, g.A, g.lda, m0, Meff, t0 * BK, Keff, g.a_map, tid);
        load_tile<BMODE, ALIGNED>(rb, g.B, g.ldb, n0, g.N, t0 * BK, Keff, g.b_map, tid);
        store_tile<AMODE>(ra, smem, tid);
        store_tile<BMODE>(rb, smem + TILE_BYTES, tid);
    }
    __syncthreads();
    int cur = 0;
    for (int t = t0; t < t1; ++t) {
        const bool more = (t + 1 < t1);
        if (more) {
            load_tile<AMODE, ALIGNED>(ra, g.A, g.lda, m0, Meff, (t + 1) * BK, Keff, g.a_map, tid);
            load_tile<BMODE, ALIGNED>(rb, g.B, g.ldb, n0, g.N, (t + 1) * BK, Keff, g.b_map, tid);
        }
        const char* la = smem + cur * (2 * TILE_BYTES);
        const char* lb = la + TILE_BYTES;
        const uint32_t la_u = sbase + cur * (2 * TILE_BYTES), lb_u = la_u + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = read_frag<AMODE>(la, la_u, wm * 64 + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = read_frag<BMODE>(lb, lb_u, wn * 64 + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            store_tile<AMODE>(ra, smem + (cur ^ 1) * (2 * TILE_BYTES), tid);
            store_tile<BMODE>(rb, smem + (cur ^ 1) * (2 * TILE_BYTES) + TILE_BYTES, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue through LDS: accumulators (lane = column lane&31, rows (r&3)+8*(r>>2)+4*(lane>>5)) are
    // parked as a 128x128 fp32 tile, then every thread streams 4 consecutive columns of a row so that bias /
    // residual / output traffic is 16-byte wide and fully coalesced.
    float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cl = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                ct[rl * BN + cl] = acc[i][j][r];
            }
        }
    __syncthreads();
    const int cl = (tid & 31) << 2;
    const int col = n0 + cl;
    float* ws = g.splitk > 1 ? g.ws + (long)ksplit * g.M * g.N : nullptr;
    if (col < g.N) {
        const int nv = min(4, g.N - col);
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias && !ws) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nv) bv[e] = g.bias[col + e];
        }
        MRG_RETIRE4(bv);
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            const int rl = (tid >> 5) + (it << 3);
            const int row = m0 + rl;
            if (row >= Meff) break;
            const float4 a4 = *reinterpret_cast<const float4*>(ct + rl * BN + cl);
            epi_any<VEC>(g, a4, bv, ws, row, col, nv);
        }
    }
    if (ws && g.tile_cnt) splitk_finish<VEC>(g, bid, m0, n0, BM, BN, Meff, smem, tid, 256);
}

// =====================================================================================================
// Fast path: aligned operands, no gather maps.  Operand tiles go HBM/L2 -> LDS directly with
// global_load_lds (16 B per lane, LDS destination = wave-uniform base + lane*16), so the XOR swizzle is applied to
// the per-lane SOURCE address while the LDS image stays linear; out-of-range rows / k read a zero page instead of
// branching.  One 32 KiB LDS stage per workgroup (A + B tile), ~130 VGPRs -> 3 workgroups per CU hide the
// load -> barrier -> MFMA latency chain; the epilogue reuses the stage in two 64-row passes.
__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// per-thread source pointer + per-tile increment for LDS block b (1 KiB) of an operand tile
template <int MODE>
__device__ __forceinline__ void glds_setup(const bf16_t* P, int ld, int row0, int nrows, int k_first, int b, int lane,
                                           const bf16_t*& ptr, long& inc, int& koff, const int* __restrict__ map = nullptr) {
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    if (MODE == MRG_OP_KCONTIG) {
        const int r = b * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int row = row0 + r;
        koff = c << 3;
        if (row < nrows) { ptr = P + (long)(map ? map[row] : row) * ld + k_first + koff; inc = BK; }   // map: gathered rows (embedding dgrad)
        else { ptr = zero; inc = 0; koff = -(1 << 30); }
    } else {
        const int kr = b * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((kr & 3) << 2);
        const int row = row0 + (c << 3);
        koff = kr;
        if (row < nrows) { ptr = P + (long)(k_first + kr) * ld + row; inc = (long)BK * ld; }
        else { ptr = zero; inc = 0; koff = -(1 << 30); }
    }
}

// DIRECT: the MFMA operands are swapped (the A fragment feeds the matrix core's B port), so a lane ends up with output ROW lane & 31
// and, per accumulator quad, four CONSECUTIVE columns: the fused epilogue runs straight on the registers and stores 8 B (bf16 x 4) /
// 16 B (fp32 x 4) per quad -- no fp32 round trip of the tile through LDS (64 ds_write_b32 + 16 ds_read_b128 per lane, at the LDS's
// ~64-80 B/clk store rate about 1000 cycles per tile) and none of its four barriers.
template <int AMODE, int BMODE, bool VEC, int OCC, bool CSUM = false, bool KF = false, bool DIRECT = false, bool DROP = false>
__device__ __forceinline__ void gemm_glds_body(const GemmArgs& g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = g.tiles_m * g.tiles_n;
    int bid, ksplit;
    wg_remap(nwg, g.xcd_split, bid, ksplit);
    // grouped rasterisation inside the XCD's band: GROUP_M row panels x all column tiles, column-major within the group,
    // so a B (weight) tile is reused by GROUP_M consecutive workgroups while their A panels stay L2-resident
    constexpr int GROUP_M = 8;      // round 6: 4 / 8 / 16 tie in the step (C2 8.98 / 8.98 / 8.98 ms, C4 19.49 / 19.49 / 19.53), 64 -- M-fastest over the whole band -- is slower (9.09 / 19.73)
    int tm, tn;
    {
        const int per_group = GROUP_M * g.tiles_n;
        const int grp = fast_udiv(bid, per_group), in_grp = bid - grp * per_group;
        const int first = grp * GROUP_M;
        const int rows = min(GROUP_M, g.tiles_m - first);
        tn = fast_udiv(in_grp, rows);
        tm = first + in_grp - tn * rows;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int Meff = g.m_dev ? min(g.M, *g.m_dev) : g.M;
    const int Keff = g.k_dev ? min(g.K, *g.k_dev) : g.K;
    if (m0 >= Meff) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = (Keff + BK - 1) / BK;
    int t0 = 0, t1 = ntiles;
    if (g.splitk > 1) {
        const int per = (ntiles + g.splitk - 1) / g.splitk;
        t0 = ksplit * per;
        t1 = min(ntiles, t0 + per);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    char* ldsA = smem;
    char* ldsB = smem + TILE_BYTES;
    const uint32_t la_u = lds_addr(ldsA), lb_u = lds_addr(ldsB);
    const bf16_t* pa[4]; const bf16_t* pb[4];
    long ia[4], ib[4];
    int ka[4], kb[4];
    uint32_t oa[4], ob[4];
    const char* sa = nullptr; const char* sb = nullptr;
    const uint32_t la_s = __builtin_amdgcn_readfirstlane(la_u), lb_s = __builtin_amdgcn_readfirstlane(lb_u);
    if constexpr (KF) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            oa[j] = glds_lane_off<AMODE>(g.lda, m0, Meff, wave * 4 + j, lane);
            ob[j] = glds_lane_off<BMODE>(g.ldb, n0, g.N, wave * 4 + j, lane);
        }
        sa = uniform_ptr(reinterpret_cast<const char*>(g.A) + (AMODE == MRG_OP_KCONTIG ? ((long)m0 * g.lda + (long)t0 * BK) * 2 : ((long)t0 * BK * g.lda + m0) * 2));
        sb = uniform_ptr(reinterpret_cast<const char*>(g.B) + (BMODE == MRG_OP_KCONTIG ? ((long)n0 * g.ldb + (long)t0 * BK) * 2 : ((long)t0 * BK * g.ldb + n0) * 2));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds_setup<AMODE>(g.A, g.lda, m0, Meff, t0 * BK, wave * 4 + j, lane, pa[j], ia[j], ka[j], g.a_map);
            glds_setup<BMODE>(g.B, g.ldb, n0, g.N, t0 * BK, wave * 4 + j, lane, pb[j], ib[j], kb[j]);
        }
    }
    const long step_a = AMODE == MRG_OP_KCONTIG ? BK * 2 : (long)BK * g.lda * 2, step_b = BMODE == MRG_OP_KCONTIG ? BK * 2 : (long)BK * g.ldb * 2;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    for (int t = t0; t < t1; ++t) {
        const int krem = Keff - t * BK;           // valid k in this tile (>= BK except for the last, partial tile)
        if (t > t0) __syncthreads();              // everyone finished reading the previous tile
        if constexpr (KF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                glds16_s(sa, oa[j], la_s + (wave * 4 + j) * 1024);
                glds16_s(sb, ob[j], lb_s + (wave * 4 + j) * 1024);
            }
            sa = uniform_ptr(sa + step_a); sb = uniform_ptr(sb + step_b);
        } else {
        if (krem >= BK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                glds16(pa[j], ldsA + (wave * 4 + j) * 1024);
                glds16(pb[j], ldsB + (wave * 4 + j) * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                glds16(ka[j] < krem ? pa[j] : zero, ldsA + (wave * 4 + j) * 1024);
                glds16(kb[j] < krem ? pb[j] : zero, ldsB + (wave * 4 + j) * 1024);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa[j] += ia[j]; pb[j] += ib[j]; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = read_frag<AMODE>(ldsA, la_u, wm * 64 + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = read_frag<BMODE>(ldsB, lb_u, wn * 64 + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    if constexpr (DIRECT) {
        // ---- epilogue on the accumulator registers: lane = row (lane & 31) of each 32-row tile, quad q of accumulator tile (i, j) =
        // columns j*32 + 8q + 4*(lane >> 5) .. +3
        float* ws = g.splitk > 1 ? g.ws + (long)ksplit * g.M * g.N : nullptr;
        const int hc = (lane >> 5) << 2;
        const bool bias_vec = VEC && ((reinterpret_cast<uintptr_t>(g.bias) & 15) == 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float bq[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + wn * 64 + j * 32 + 8 * q + hc;
#pragma unroll
                for (int e = 0; e < 4; ++e) bq[q][e] = 0.f;
                if (g.bias && !ws && col < g.N) {
                    if (bias_vec) { const float4 b4 = *reinterpret_cast<const float4*>(g.bias + col); bq[q][0] = b4.x; bq[q][1] = b4.y; bq[q][2] = b4.z; bq[q][3] = b4.w; }
                    else { for (int e = 0; e < 4; ++e) if (col + e < g.N) bq[q][e] = g.bias[col + e]; }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = m0 + wm * 64 + i * 32 + (lane & 31);
                if (row < Meff) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = n0 + wn * 64 + j * 32 + 8 * q + hc;
                        if (col < g.N) {
                            const float4 a4 = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                            epi_any<VEC>(g, a4, bq[q], ws, row, col, min(4, g.N - col));
                        }
                    }
                }
            }
        }
        if (ws && g.tile_cnt) splitk_finish<VEC>(g, bid, m0, n0, BM, BN, Meff, smem, tid, 256);
        return;
    }
    // ---- epilogue in two passes of 64 rows x 128 cols fp32 (32 KiB)
    float* ct = reinterpret_cast<float*>(smem);
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const int cl = (tid & 31) << 2;
    const int col = n0 + cl;
    const int nv = min(4, g.N - col);
    float* ws = g.splitk > 1 ? g.ws + (long)ksplit * g.M * g.N : nullptr;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && !ws && col < g.N) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) bv[e] = g.bias[col + e];
    }
    MRG_RETIRE4(bv);
#pragma unroll
    for (int pss = 0; pss < 2; ++pss) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c2 = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                ct[rl * BN + c2] = acc[pss][j][r];
            }
        }
        __syncthreads();
        if (col < g.N) {
            auto row_of = [&](int it, int& rl, int& row) { rl = (tid >> 5) + (it << 3); row = m0 + (rl >> 5) * 64 + pss * 32 + (rl & 31); };
            if constexpr (DROP) {           // bias -> Philox dropout -> row factor -> residual (the only instantiation that reads g.drop)
#pragma unroll 1
                for (int it = 0; it < 8; ++it) {
                    int rl, row;
                    row_of(it, rl, row);
                    if (row < Meff) epi_row_drop<VEC>(g, *reinterpret_cast<const float4*>(ct + rl * BN + cl), bv, row, col, nv);
                }
            } else
            if (VEC && !CSUM && (g.side_drain & 2) && ws) MRG_DRAIN8(7, false, ws);
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 4) MRG_DRAIN8(4, CSUM, cs);
            else if (VEC && !CSUM && (g.side_drain & 1) && g.epi_kind == 2) MRG_DRAIN8(2, false, cs);
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 1) MRG_DRAIN8(1, CSUM, cs);
            else if (VEC && !CSUM && (g.side_drain & 1) && g.epi_kind == 3) MRG_DRAIN8(3, false, cs);
            else if (VEC && !CSUM && (g.side_drain & 1) && g.epi_kind == 5) MRG_DRAIN8(5, false, cs);
            else {
#pragma unroll 1
            for (int it = 0; it < 8; ++it) {
                const int rl = (tid >> 5) + (it << 3);                       // 0..63
                const int row = m0 + (rl >> 5) * 64 + pss * 32 + (rl & 31);
                if (row < Meff) {
                    const float4 a4 = *reinterpret_cast<const float4*>(ct + rl * BN + cl);
                    if (CSUM) epi_any_cs<VEC>(g, a4, bv, ws, row, col, nv, cs);
                    else epi_any<VEC>(g, a4, bv, ws, row, col, nv);
                }
            }
            }
        }
    }
    if (!CSUM && ws && g.tile_cnt) splitk_finish<VEC>(g, bid, m0, n0, BM, BN, Meff, smem, tid, 256);
    if (CSUM) {   // column sums of this tile's final values (e.g. the bias gradient of the layer whose output gradient C is)
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) ct[(tid >> 5) * BN + cl + e] = cs[e];
        __syncthreads();
        if (tid < BN && n0 + tid < g.N) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += ct[k * BN + tid];
            g.colsum[(long)tm * g.N + n0 + tid] = t;
        }
    }
}

template <int AMODE, int BMODE, bool VEC, int OCC, bool CSUM = false, bool KF = false, bool DIRECT = false, bool DROP = false>
__global__ __launch_bounds__(256, OCC) void gemm_glds_kernel(const GemmArgs g) {
    gemm_glds_body<AMODE, BMODE, VEC, OCC, CSUM, KF, DIRECT, DROP>(g);
}

// Several independent problems in ONE launch (mrg_gemm_multi: the patch projections of a step's modalities, mirage/input_adapters.py:101,229):
// blockIdx.z selects the problem, the grid's x extent is the largest tile count rounded up to a multiple of 8 (so that a block's XCD is
// blockIdx.x & 7 for every z, as wg_remap assumes), surplus blocks leave.  Each problem alone is a latency chain over its k-tiles on
// fewer tiles than CUs; side by side their chains overlap (2 x 31 us -> one launch).
struct GemmMultiArgs { GemmArgs g[MRG_MAX_DOMAINS]; };
template <int AMODE, int BMODE, bool VEC, int OCC, bool KF>
__global__ __launch_bounds__(256, OCC) void gemm_glds_multi_kernel(const GemmMultiArgs m) {
    const GemmArgs& g = m.g[blockIdx.z];
    if ((int)blockIdx.x >= g.tiles_m * g.tiles_n) return;
    gemm_glds_body<AMODE, BMODE, VEC, OCC, false, KF, false, false>(g);
}

// LDS-DMA issued from inline asm: invisible to hipcc's s_waitcnt bookkeeping, so the compiler does not drain the
// in-flight tile (vmcnt(0)) in front of the ds_reads of the tile being multiplied; completion is waited for by hand.
__device__ __forceinline__ void glds16_asm(const bf16_t* src, uint32_t lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_wave_base) : "memory");
}

// Two-stage ring, 64 KiB LDS, 2 workgroups per CU: the DMA of tile t+1 flies while tile t is on the MFMA pipe;
// one raw s_barrier per k-tile.
template <int AMODE, int BMODE, bool VEC, bool KF = false>
__global__ __launch_bounds__(256, 2) void gemm_pipe_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = g.tiles_m * g.tiles_n;
    int bid, ksplit;
    wg_remap(nwg, g.xcd_split, bid, ksplit);
    constexpr int GROUP_M = 8;
    int tm, tn;
    {
        const int per_group = GROUP_M * g.tiles_n;
        const int grp = fast_udiv(bid, per_group), in_grp = bid - grp * per_group;
        const int first = grp * GROUP_M;
        const int rows = min(GROUP_M, g.tiles_m - first);
        tn = fast_udiv(in_grp, rows);
        tm = first + in_grp - tn * rows;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int Meff = g.m_dev ? min(g.M, *g.m_dev) : g.M;
    const int Keff = g.k_dev ? min(g.K, *g.k_dev) : g.K;
    if (m0 >= Meff) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = (Keff + BK - 1) / BK;
    int t0 = 0, t1 = ntiles;
    if (g.splitk > 1) {
        const int per = (ntiles + g.splitk - 1) / g.splitk;
        t0 = ksplit * per;
        t1 = min(ntiles, t0 + per);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const uint32_t sbase = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const bf16_t* pa[4]; const bf16_t* pb[4];
    long ia[4], ib[4];
    int ka[4], kb[4];
    uint32_t oa[4], ob[4];
    const char* sa = nullptr; const char* sb = nullptr;
    if constexpr (KF) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            oa[j] = glds_lane_off<AMODE>(g.lda, m0, Meff, wave * 4 + j, lane);
            ob[j] = glds_lane_off<BMODE>(g.ldb, n0, g.N, wave * 4 + j, lane);
        }
        sa = uniform_ptr(reinterpret_cast<const char*>(g.A) + (AMODE == MRG_OP_KCONTIG ? ((long)m0 * g.lda + (long)t0 * BK) * 2 : ((long)t0 * BK * g.lda + m0) * 2));
        sb = uniform_ptr(reinterpret_cast<const char*>(g.B) + (BMODE == MRG_OP_KCONTIG ? ((long)n0 * g.ldb + (long)t0 * BK) * 2 : ((long)t0 * BK * g.ldb + n0) * 2));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds_setup<AMODE>(g.A, g.lda, m0, Meff, t0 * BK, wave * 4 + j, lane, pa[j], ia[j], ka[j], g.a_map);
            glds_setup<BMODE>(g.B, g.ldb, n0, g.N, t0 * BK, wave * 4 + j, lane, pb[j], ib[j], kb[j]);
        }
    }
    const long step_a = AMODE == MRG_OP_KCONTIG ? BK * 2 : (long)BK * g.lda * 2, step_b = BMODE == MRG_OP_KCONTIG ? BK * 2 : (long)BK * g.ldb * 2;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    auto issue = [&](int t, int buf) {
        const uint32_t la = sbase + buf * (2 * TILE_BYTES) + wave * 4096;
        const uint32_t lb = la + TILE_BYTES;
        if constexpr (KF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                glds16_s(sa, oa[j], la + j * 1024);
                glds16_s(sb, ob[j], lb + j * 1024);
            }
            sa = uniform_ptr(sa + step_a); sb = uniform_ptr(sb + step_b);
        } else {
        const int krem = Keff - t * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16_asm((krem >= BK || ka[j] < krem) ? pa[j] : zero, la + j * 1024);
            glds16_asm((krem >= BK || kb[j] < krem) ? pb[j] : zero, lb + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa[j] += ia[j]; pb[j] += ib[j]; }
        }
    };
    if (t0 < t1) issue(t0, 0);
    for (int t = t0; t < t1; ++t) {
        const int cur = (t - t0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < t1) issue(t + 1, cur ^ 1);
        const char* ldsA = smem + cur * (2 * TILE_BYTES);
        const char* ldsB = ldsA + TILE_BYTES;
        const uint32_t la_u = sbase + cur * (2 * TILE_BYTES), lb_u = la_u + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = read_frag<AMODE>(ldsA, la_u, wm * 64 + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = read_frag<BMODE>(ldsB, lb_u, wn * 64 + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    // ---- epilogue: one pass through a 128x128 fp32 tile (64 KiB ring is free once every wave passed the barrier)
    __syncthreads();
    float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c2 = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                ct[rl * BN + c2] = acc[i][j][r];
            }
        }
    __syncthreads();
    const int cl = (tid & 31) << 2;
    const int col = n0 + cl;
    float* ws = g.splitk > 1 ? g.ws + (long)ksplit * g.M * g.N : nullptr;
    if (col < g.N) {
        const int nv = min(4, g.N - col);
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias && !ws) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nv) bv[e] = g.bias[col + e];
        }
        MRG_RETIRE4(bv);
        if (VEC && ((ws && (g.side_drain & 2)) || (!ws && (g.side_drain & 1) && g.epi_kind >= 1 && g.epi_kind <= 5))) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                auto row_of = [&](int it, int& rl, int& row) { rl = (tid >> 5) + ((it + 8 * hh) << 3); row = m0 + rl; };
                if (ws) MRG_DRAIN8(7, false, ws);
                else if (g.epi_kind == 2) MRG_DRAIN8(2, false, nullptr);
                else if (g.epi_kind == 4) MRG_DRAIN8(4, false, nullptr);
                else if (g.epi_kind == 1) MRG_DRAIN8(1, false, nullptr);
                else if (g.epi_kind == 3) MRG_DRAIN8(3, false, nullptr);
                else MRG_DRAIN8(5, false, nullptr);
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < 16; ++it) {
                const int rl = (tid >> 5) + (it << 3);
                const int row = m0 + rl;
                if (row >= Meff) break;
                const float4 a4 = *reinterpret_cast<const float4*>(ct + rl * BN + cl);
                epi_any<VEC>(g, a4, bv, ws, row, col, nv);
            }
        }
    }
    if (ws && g.tile_cnt) splitk_finish<VEC>(g, bid, m0, n0, BM, BN, Meff, smem, tid, 256);
}

// Warp-specialised 256 x 128 kernel: 8 compute waves (4 x 2, 64 x 64 each) + 4 loader waves.  Issuing a global_load_lds
// costs the issuing wave ~60-180 cycles during which it cannot issue MFMAs (in-order issue); the loaders take all 48 of them
// per k-tile off the compute waves and keep a 3-stage ring (144 KiB) two tiles ahead.  One s_barrier per k-tile for all 12 waves.
template <int AMODE, int BMODE, bool VEC, bool KF = false>
__global__ __launch_bounds__(768) void gemm_ws256_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 3 * TILE_BYTES, STAGES = 3;
    const int tiles_m = (g.M + 255) / 256;
    const int nwg = tiles_m * g.tiles_n;
    int bid, ksplit;
    wg_remap(nwg, g.xcd_split, bid, ksplit);
    constexpr int GROUP_M = 4;
    int tm, tn;
    {
        const int per_group = GROUP_M * g.tiles_n;
        const int grp = fast_udiv(bid, per_group), in_grp = bid - grp * per_group;
        const int first = grp * GROUP_M;
        const int rows = min(GROUP_M, tiles_m - first);
        tn = fast_udiv(in_grp, rows);
        tm = first + in_grp - tn * rows;
    }
    const int m0 = tm * 256, n0 = tn * BN;
    const int Meff = g.m_dev ? min(g.M, *g.m_dev) : g.M;
    const int Keff = g.k_dev ? min(g.K, *g.k_dev) : g.K;
    if (m0 >= Meff) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (Keff + BK - 1) / BK;
    int t0 = 0, t1 = ntiles;
    if (g.splitk > 1) {
        const int per = (ntiles + g.splitk - 1) / g.splitk;
        t0 = ksplit * per;
        t1 = min(ntiles, t0 + per);
    }
    const uint32_t sbase = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    if (wave >= 8) {
        // ------------------------------------------------------------------ loader waves
        const int L = wave - 8;
        const bf16_t* pa[8]; const bf16_t* pb[4];
        long ia[8], ib[4];
        int ka[8], kb[4];
        uint32_t oa[8], ob[4];
        const char* sa = nullptr; const char* sb = nullptr;
        if constexpr (KF) {
            const int ra = min(m0 + (L >> 1) * 128, (AMODE == MRG_OP_KCONTIG ? Meff - 1 : max(0, ((Meff - 1) & ~7))));
#pragma unroll
            for (int j = 0; j < 8; ++j) oa[j] = glds_lane_off<AMODE>(g.lda, ra, Meff, (L & 1) * 8 + j, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) ob[j] = glds_lane_off<BMODE>(g.ldb, n0, g.N, L * 4 + j, lane);
            sa = uniform_ptr(reinterpret_cast<const char*>(g.A) + (AMODE == MRG_OP_KCONTIG ? ((long)ra * g.lda + (long)t0 * BK) * 2 : ((long)t0 * BK * g.lda + ra) * 2));
            sb = uniform_ptr(reinterpret_cast<const char*>(g.B) + (BMODE == MRG_OP_KCONTIG ? ((long)n0 * g.ldb + (long)t0 * BK) * 2 : ((long)t0 * BK * g.ldb + n0) * 2));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) glds_setup<AMODE>(g.A, g.lda, m0 + (L >> 1) * 128, Meff, t0 * BK, (L & 1) * 8 + j, lane, pa[j], ia[j], ka[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) glds_setup<BMODE>(g.B, g.ldb, n0, g.N, t0 * BK, L * 4 + j, lane, pb[j], ib[j], kb[j]);
        }
        const long step_a = AMODE == MRG_OP_KCONTIG ? BK * 2 : (long)BK * g.lda * 2, step_b = BMODE == MRG_OP_KCONTIG ? BK * 2 : (long)BK * g.ldb * 2;
        const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
        auto issue = [&](int t, int buf) {
            const uint32_t la = sbase + buf * STAGE + (L >> 1) * TILE_BYTES + (L & 1) * 8192;
            const uint32_t lb = sbase + buf * STAGE + 2 * TILE_BYTES + L * 4096;
            if constexpr (KF) {
#pragma unroll
                for (int j = 0; j < 8; ++j) glds16_s(sa, oa[j], la + j * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) glds16_s(sb, ob[j], lb + j * 1024);
                sa = uniform_ptr(sa + step_a); sb = uniform_ptr(sb + step_b);
            } else {
            const int krem = Keff - t * BK;
#pragma unroll
            for (int j = 0; j < 8; ++j) glds16_asm((krem >= BK || ka[j] < krem) ? pa[j] : zero, la + j * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16_asm((krem >= BK || kb[j] < krem) ? pb[j] : zero, lb + j * 1024);
#pragma unroll
            for (int j = 0; j < 8; ++j) pa[j] += ia[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) pb[j] += ib[j];
            }
        };
        if (t0 < t1) issue(t0, 0);
        if (t0 + 1 < t1) issue(t0 + 1, 1);
        int cur = 0;
        for (int t = t0; t < t1; ++t) {
            if (t + 1 < t1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // tile t landed, tile t+1 may still fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 2 < t1) issue(t + 2, (cur + 2) % STAGES);    // stage of tile t-1: every compute wave is past it
            cur = (cur + 1 == STAGES) ? 0 : cur + 1;
        }
        // take part in the epilogue barriers
        __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();
        if (g.splitk > 1 && g.tile_cnt) splitk_finish<VEC>(g, bid, m0, n0, 256, BN, Meff, smem, tid, 768);   // same barriers as the compute waves
        return;
    }
    // ---------------------------------------------------------------------- compute waves
    const int half = wave >> 2, wl = wave & 3;
    const int wm = wl >> 1, wn = wl & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int cur = 0;
    for (int t = t0; t < t1; ++t) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* ldsA = smem + cur * STAGE + half * TILE_BYTES;
        const char* ldsB = smem + cur * STAGE + 2 * TILE_BYTES;
        const uint32_t la_u = sbase + cur * STAGE + half * TILE_BYTES, lb_u = sbase + cur * STAGE + 2 * TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = read_frag<AMODE>(ldsA, la_u, wm * 64 + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = read_frag<BMODE>(ldsB, lb_u, wn * 64 + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        cur = (cur + 1 == STAGES) ? 0 : cur + 1;
    }
    float* ct = reinterpret_cast<float*>(smem);
    const int cl = (tid & 31) << 2;
    const int col = n0 + cl;
    const int nv = min(4, g.N - col);
    float* ws = g.splitk > 1 ? g.ws + (long)ksplit * g.M * g.N : nullptr;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && !ws && col < g.N) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) bv[e] = g.bias[col + e];
    }
    MRG_RETIRE4(bv);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        if (half == h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c2 = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        ct[rl * BN + c2] = acc[i][j][r];
                    }
                }
        }
        __syncthreads();
        if (col < g.N) {
            auto row_of = [&](int it, int& rl, int& row) { rl = (tid >> 5) + (it << 4); row = m0 + h * 128 + rl; };
            if (VEC && (g.side_drain & 2) && ws) MRG_DRAIN8(7, false, ws);
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 2) MRG_DRAIN8(2, false, nullptr);
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 4) MRG_DRAIN8(4, false, nullptr);
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 1) MRG_DRAIN8(1, false, nullptr);
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 3) MRG_DRAIN8(3, false, nullptr);
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 5) MRG_DRAIN8(5, false, nullptr);
            else {
#pragma unroll 1
            for (int it = 0; it < 8; ++it) {
                const int rl = (tid >> 5) + (it << 4);
                const int row = m0 + h * 128 + rl;
                if (row < Meff) {
                    const float4 a4 = *reinterpret_cast<const float4*>(ct + rl * BN + cl);
                    epi_any<VEC>(g, a4, bv, ws, row, col, nv);
                }
            }
            }
        }
    }
    if (ws && g.tile_cnt) splitk_finish<VEC>(g, bid, m0, n0, 256, BN, Meff, smem, tid, 768);
}

// split-K finish: C (+)= sum_z ws[z]  (+ bias) ; fp32 out, optional bf16 copy
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splitk, long mn, int N, float* __restrict__ C, int ldc,
                                     int accumulate, const float* __restrict__ bias, bf16_t* __restrict__ C2, int ldc2,
                                     int c_f32, void* Cany) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splitk; ++z) s += ws[(long)z * mn + i];
        const long row = i / N;
        const int col = (int)(i - row * N);
        if (bias) s += bias[col];
        if (c_f32) {
            float* c = C + row * ldc + col;
            if (accumulate) s += *c;
            *c = s;
        } else {
            reinterpret_cast<bf16_t*>(Cany)[row * ldc + col] = f2bf(s);
        }
        if (C2) C2[row * ldc2 + col] = f2bf(s);
    }
}

}  // namespace

extern "C" int mrg_gemm(const mrg_gemm_desc* d, hipStream_t stream) {
    MRG_REQUIRE(d != nullptr, MRG_ERR_ARG, "mrg_gemm: null descriptor");
    MRG_REQUIRE(d->A && d->B && d->C, MRG_ERR_ARG, "mrg_gemm: null operand pointer");
    MRG_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, MRG_ERR_SHAPE, "mrg_gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    MRG_REQUIRE(d->a_mode == MRG_OP_KCONTIG || d->a_mode == MRG_OP_KSTRIDED, MRG_ERR_ARG, "mrg_gemm: bad a_mode");
    MRG_REQUIRE(d->b_mode == MRG_OP_KCONTIG || d->b_mode == MRG_OP_KSTRIDED, MRG_ERR_ARG, "mrg_gemm: bad b_mode");
    MRG_REQUIRE(!(d->a_mode == MRG_OP_KSTRIDED && d->b_mode == MRG_OP_KCONTIG), MRG_ERR_UNSUPPORTED,
                "mrg_gemm: (A k-strided, B k-contiguous) is not instantiated");
    MRG_REQUIRE(d->act >= MRG_ACT_NONE && d->act <= MRG_ACT_MUL_AUX_F16, MRG_ERR_ARG, "mrg_gemm: bad act");
    MRG_REQUIRE((d->act != MRG_ACT_GELU_BWD && d->act != MRG_ACT_MUL_AUX_F16) || d->aux_in, MRG_ERR_ARG, "mrg_gemm: this act needs aux_in");
    MRG_REQUIRE(!d->add_tab || d->add_row, MRG_ERR_ARG, "mrg_gemm: add_tab needs add_row");
    MRG_REQUIRE(!d->accumulate || d->c_dtype == MRG_DT_F32, MRG_ERR_ARG, "mrg_gemm: accumulate needs fp32 C");
    const int splitk = d->splitk > 1 ? d->splitk : 1;
    if (splitk > 1) {
        MRG_REQUIRE(d->ws, MRG_ERR_ARG, "mrg_gemm: split-K needs a workspace");
        MRG_REQUIRE(d->act == MRG_ACT_NONE && !d->resid && !d->add_tab && !d->row_map && !d->m_dev && !d->row_scale, MRG_ERR_UNSUPPORTED,
                    "mrg_gemm: split-K supports only bias/accumulate epilogues");
    }
    GemmArgs g;
    g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B;
    g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb;
    g.a_map = d->a_map; g.b_map = d->b_map; g.m_dev = d->m_dev; g.k_dev = d->k_dev;
    g.a_aligned = (((uintptr_t)d->A & 15) == 0) && (d->lda % 8 == 0);
    g.b_aligned = (((uintptr_t)d->B & 15) == 0) && (d->ldb % 8 == 0);
    g.C = d->C; g.ldc = d->ldc; g.c_f32 = (d->c_dtype == MRG_DT_F32);
    g.C2 = (bf16_t*)d->C2; g.ldc2 = d->ldc2;
    g.bias = d->bias; g.resid = d->resid; g.ldr = d->ldr; g.accumulate = d->accumulate; g.act = d->act;
    g.aux_out = (bf16_t*)d->aux_out; g.aux_in = (const bf16_t*)d->aux_in; g.ldaux = d->ldaux;
    g.row_map = d->row_map; g.add_tab = d->add_tab; g.add_row = d->add_row; g.ldt = d->ldt; g.row_scale = d->row_scale; g.colsum = d->colsum_out;
    g.splitk = splitk; g.ws = d->ws; g.tile_cnt = nullptr;
    g.drop.rng = nullptr; g.drop.thresh = 0; g.drop.scale = 1.f; g.drop.stream = 0;
    if (d->drop_rng && d->drop_p > 0.f) {
        MRG_REQUIRE(d->drop_p < 1.f, MRG_ERR_ARG, "mrg_gemm: drop_p = %g must be in [0, 1)", (double)d->drop_p);
        MRG_REQUIRE(splitk == 1 && d->N % 4 == 0 && !d->colsum_out && !d->accumulate, MRG_ERR_UNSUPPORTED,
                    "mrg_gemm: dropout needs N %% 4 == 0 and no split-K / column sums / accumulate");
        MRG_REQUIRE(d->act == MRG_ACT_NONE && !d->add_tab && !d->row_map && !d->a_map && !d->b_map && !d->m_dev && !d->k_dev &&
                    d->a_mode == MRG_OP_KCONTIG && d->b_mode == MRG_OP_KCONTIG, MRG_ERR_UNSUPPORTED,
                    "mrg_gemm: dropout is fused into the forward (+bias)(*row_scale)(+residual) epilogue only");
        const double t = floor((double)d->drop_p * 4294967296.0);
        g.drop.rng = (const unsigned long long*)d->drop_rng;
        g.drop.thresh = (uint32_t)(t > 4294967295.0 ? 4294967295.0 : t);
        g.drop.scale = 1.f / (1.f - d->drop_p);
        g.drop.stream = d->drop_stream;
    }
    MRG_REQUIRE(d->variant == MRG_GEMM_AUTO || d->variant == MRG_GEMM_GLDS3 || d->variant == MRG_GEMM_GLDS4 || d->variant == MRG_GEMM_PIPE128 ||
                d->variant == MRG_GEMM_WS256 || d->variant == MRG_GEMM_P8, MRG_ERR_ARG, "mrg_gemm: bad variant %d", d->variant);
    g.tiles_m = cdiv(d->M, BM); g.tiles_n = cdiv(d->N, BN);
    { static const int abl = [] { const char* e = getenv("MRG_GEMM_ABLATE"); return e ? atoi(e) : 0; }(); g.ablate = abl; }
    { static const int nt = [] { const char* e = getenv("MRG_GEMM_NT"); return e ? atoi(e) : 2; }(); g.nt = nt; }
    { static const int xs = [] { const char* e = getenv("MRG_GEMM_XCD_SPLIT"); return e ? atoi(e) : 1; }(); g.xcd_split = xs; }
    { static const int sd = [] { const char* e = getenv("MRG_GEMM_SIDE_DRAIN"); return e ? atoi(e) : 3; }(); g.side_drain = sd; }
    static const int kfast = [] { const char* e = getenv("MRG_GEMM_KFAST"); return e ? atoi(e) : 1; }();
    static const int epi_direct_env = [] { const char* e = getenv("MRG_EPI_DIRECT"); return e ? atoi(e) : 0; }();   // register-direct epilogue of the single-stage 128 x 128 kernel: measured SLOWER (12.95 vs 11.02 ms/step at C2)
    const int epi_direct = epi_direct_env;
    const bool kf = kfast && (d->K % BK == 0) && !d->k_dev && !d->a_map;      // scalar-base loaders
    {   // lean epilogue selection (uniform over the launch)
        static const int lean = [] { const char* e = getenv("MRG_GEMM_LEAN"); return e ? atoi(e) : 1; }();
        const bool base = splitk == 1 && !d->add_tab && !d->row_map && !d->accumulate && !d->resid && !d->row_scale && !d->C2 && !g.c_f32 && !g.drop.rng;
        g.epi_kind = 0;
        if (lean) {
            if (base && d->act == MRG_ACT_NONE) g.epi_kind = 1;
            else if (base && d->act == MRG_ACT_GELU_SAVE_GRAD && d->aux_out && !d->colsum_out) g.epi_kind = 3;
            else if (base && d->act == MRG_ACT_MUL_AUX_F16 && d->aux_in) g.epi_kind = 4;
            else if (base && d->act == MRG_ACT_GELU && !d->aux_out && !d->colsum_out) g.epi_kind = 5;
            else if (splitk == 1 && d->act == MRG_ACT_NONE && d->add_tab && d->add_row && d->row_map && !d->accumulate && !d->resid && !d->row_scale &&
                     !d->C2 && g.c_f32 && !d->colsum_out && !g.drop.rng) g.epi_kind = 6;
            else if (splitk == 1 && d->act == MRG_ACT_NONE && !d->add_tab && !d->row_map && !d->accumulate && !d->colsum_out && d->resid && g.c_f32) g.epi_kind = 2;
            else if (splitk == 1 && d->act == MRG_ACT_NONE && !d->add_tab && !d->row_map && !d->accumulate && !d->colsum_out && !d->resid && !d->row_scale &&
                     !d->C2 && !d->bias && g.c_f32 && !g.drop.rng && d->variant == MRG_GEMM_P8) g.epi_kind = 7;
        }
    }
    dim3 grid(g.tiles_m * g.tiles_n, splitk), block(256);
    const size_t shmem = 4 * TILE_BYTES;
    // vector (16 B) loads need 16-B aligned bases, ld % 8 == 0 and whole chunks inside the matrix
    bool al = g.a_aligned && g.b_aligned && !d->k_dev;
    if (d->a_mode == MRG_OP_KCONTIG) al = al && (d->K % 8 == 0); else al = al && (d->M % 8 == 0);
    if (d->b_mode == MRG_OP_KCONTIG) al = al && (d->K % 8 == 0); else al = al && (d->N % 8 == 0);
    if (d->k_dev && d->a_mode == MRG_OP_KSTRIDED && d->b_mode == MRG_OP_KSTRIDED)   // k rows are whole rows: count may be anything
        al = g.a_aligned && g.b_aligned && (d->M % 8 == 0) && (d->N % 8 == 0);
    // 16-byte epilogue accesses need N % 4 == 0 and every touched matrix 16-B (fp32) / 8-B (bf16) aligned per row
    auto al4 = [](const void* p, int ld, int bytes) { return p == nullptr || ((((uintptr_t)p) % (4 * bytes) == 0) && (ld % 4 == 0)); };
    const bool vec = (d->N % 4 == 0) && al4(d->C, d->ldc, g.c_f32 ? 4 : 2) && al4(d->C2, d->ldc2, 2) && al4(d->resid, d->ldr, 4) &&
                     al4(d->add_tab, d->ldt, 4) && al4(d->aux_out, d->ldaux, 2) && al4(d->aux_in, d->ldaux, 2) && al4(d->ws, d->N, 4);
    const bool fast = al && !d->b_map && (!d->a_map || d->a_mode == MRG_OP_KCONTIG);
    const bool mapped = d->a_map != nullptr;                       // gathered A rows: single-/two-stage 128 x 128 kernels, general loader
    if (d->ev_start) (void)hipEventRecord((hipEvent_t)d->ev_start, stream);
    if (g.drop.rng) {      // nn.Dropout after proj / fc2: the dedicated instantiation of the single-stage 128 x 128 kernel
        MRG_REQUIRE(fast, MRG_ERR_UNSUPPORTED, "mrg_gemm: dropout needs 16-byte aligned operands with K %% 8 == 0");
        g.epi_kind = 0;
        if (vec) hipLaunchKernelGGL((gemm_glds_kernel<0, 0, true, 3, false, false, false, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g);
        else hipLaunchKernelGGL((gemm_glds_kernel<0, 0, false, 3, false, false, false, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g);
        if (d->ev_stop) (void)hipEventRecord((hipEvent_t)d->ev_stop, stream);
        return mrg_check_launch("mrg_gemm");
    }
    // 4 workgroups per CU (<= 128 VGPRs) pay off only when there are enough tiles to fill them (measured: +13..18 % at
    // >= ~1500 tiles, -3..-15 % below and for the wgrad layout)
    // kernel choice (measured, tools/bench_gemm.py): long K loops (>= 24 k-tiles per workgroup) run the pipelined two-stage
    // kernel (+5..40 %), short ones the single-stage kernel at 3-4 workgroups/CU (+10..20 % there).  MRG_GEMM_PIPE=0/1 forces.
    // kernel choice: an explicit variant (host autotuner) wins; otherwise measured heuristics (tools/bench_gemm.py):
    // long K loops run the pipelined two-stage kernels, short K with many tiles the single-stage kernel at 4 WG/CU.
    const int ktiles_per_wg = cdiv(cdiv(d->K, BK), splitk);
    bool use_pipe = ktiles_per_wg >= 24;
    bool occ4 = fast && d->a_mode == MRG_OP_KCONTIG && (g.tiles_m * g.tiles_n >= 1400) && splitk == 1;
    if (d->variant == MRG_GEMM_GLDS3) { use_pipe = false; occ4 = false; }
    else if (d->variant == MRG_GEMM_GLDS4) { use_pipe = false; occ4 = true; }
    else if (d->variant == MRG_GEMM_PIPE128) { use_pipe = true; }
    if (d->colsum_out) {      // only the single-stage 128 x 128 kernel (dgrad layout) is instantiated with the column-sum epilogue
        MRG_REQUIRE(fast && splitk == 1 && !d->row_map && d->a_mode == MRG_OP_KCONTIG && d->b_mode == MRG_OP_KSTRIDED, MRG_ERR_UNSUPPORTED,
                    "mrg_gemm: colsum_out needs the dgrad layout (A k-contiguous, B k-strided), aligned unmapped operands and no split-K");
        if (vec && kf) { if (occ4) hipLaunchKernelGGL((gemm_glds_kernel<0, 1, true, 4, true, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g);
                         else hipLaunchKernelGGL((gemm_glds_kernel<0, 1, true, 3, true, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g); }
        else if (vec) { if (occ4) hipLaunchKernelGGL((gemm_glds_kernel<0, 1, true, 4, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g);
                   else hipLaunchKernelGGL((gemm_glds_kernel<0, 1, true, 3, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g); }
        else { if (occ4) hipLaunchKernelGGL((gemm_glds_kernel<0, 1, false, 4, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g);
               else hipLaunchKernelGGL((gemm_glds_kernel<0, 1, false, 3, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g); }
        if (d->ev_stop) (void)hipEventRecord((hipEvent_t)d->ev_stop, stream);
        return mrg_check_launch("mrg_gemm");
    }
    // split-K finished inside the launch (splitk_finish) when the caller supplies arrival counters and the selected kernel has the
    // protocol (generic, single-stage / two-stage 128 x 128, warp-specialised 256 x 128, 8-phase 256 x 256)
    bool inkernel = false;
    auto finish_splitk = [&]() -> int {
        if (inkernel) return 0;
        if (splitk > 1 && d->defer_reduce) {
            // the caller sums the split-K slabs ws[splitk][M*N] itself (mrg_reduce_batch): plain fp32 overwrite only
            MRG_REQUIRE(g.c_f32 && !d->accumulate && !d->bias && !d->C2 && d->ldc == d->N, MRG_ERR_UNSUPPORTED,
                        "mrg_gemm: defer_reduce needs a contiguous fp32 C without bias / accumulate / C2");
        } else if (splitk > 1) {
            const long mn = (long)d->M * d->N;
            const int nb = (int)((mn + 255) / 256 < 2048 ? (mn + 255) / 256 : 2048);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, stream, d->ws, splitk, mn, d->N, (float*)d->C, d->ldc,
                               d->accumulate, d->bias, (bf16_t*)d->C2, d->ldc2, g.c_f32, d->C);
            return mrg_check_launch("mrg_gemm/splitk_reduce");
        }
        return 0;
    };
    inkernel = splitk > 1 && d->splitk_counters && !d->defer_reduce && !d->colsum_out;
    g.tile_cnt = inkernel ? d->splitk_counters : nullptr;
    if (fast && !mapped && d->variant == MRG_GEMM_P8 && !d->colsum_out && d->M > 128 && d->N > 128 &&
        mrg_gemm8p_launch(g, d->a_mode, d->b_mode, vec, splitk, stream)) {
        if (d->ev_stop) (void)hipEventRecord((hipEvent_t)d->ev_stop, stream);
        int rc8 = mrg_check_launch("mrg_gemm");
        return rc8 ? rc8 : finish_splitk();
    }
    const bool use_ws = !d->colsum_out && fast && !mapped && d->variant == MRG_GEMM_WS256 && d->M > 128;
#define MRG_WS(AM, BM_, V) do { if (kf && V) { const void* fn = (const void*)gemm_ws256_kernel<AM, BM_, true, true>; \
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * TILE_BYTES); \
        hipLaunchKernelGGL((gemm_ws256_kernel<AM, BM_, true, true>), dim3(cdiv(d->M, 256) * g.tiles_n, splitk), dim3(768), (size_t)(9 * TILE_BYTES), stream, g); } else { \
        const void* fn = (const void*)gemm_ws256_kernel<AM, BM_, V>; \
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * TILE_BYTES); \
        hipLaunchKernelGGL((gemm_ws256_kernel<AM, BM_, V>), dim3(cdiv(d->M, 256) * g.tiles_n, splitk), dim3(768), (size_t)(9 * TILE_BYTES), stream, g); } } while (0)
#define MRG_FAST(AM, BM_, V) do { if (use_ws) MRG_WS(AM, BM_, V); else if (use_pipe && kf && V) hipLaunchKernelGGL((gemm_pipe_kernel<AM, BM_, true, true>), grid, block, (size_t)(4 * TILE_BYTES), stream, g); \
                                  else if (use_pipe) hipLaunchKernelGGL((gemm_pipe_kernel<AM, BM_, V>), grid, block, (size_t)(4 * TILE_BYTES), stream, g); \
                                  else if (occ4 && kf && V && epi_direct) hipLaunchKernelGGL((gemm_glds_kernel<AM, BM_, true, 4, false, true, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g); \
                                  else if (kf && V && epi_direct) hipLaunchKernelGGL((gemm_glds_kernel<AM, BM_, true, 3, false, true, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g); \
                                  else if (occ4 && kf && V) hipLaunchKernelGGL((gemm_glds_kernel<AM, BM_, true, 4, false, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g); \
                                  else if (kf && V) hipLaunchKernelGGL((gemm_glds_kernel<AM, BM_, true, 3, false, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, g); \
                                  else if (occ4) hipLaunchKernelGGL((gemm_glds_kernel<AM, BM_, V, 4>), grid, block, (size_t)(2 * TILE_BYTES), stream, g); \
                                  else hipLaunchKernelGGL((gemm_glds_kernel<AM, BM_, V, 3>), grid, block, (size_t)(2 * TILE_BYTES), stream, g); } while (0)
#define MRG_LAUNCH2(AM, BM_, AL) do { \
        if (fast) { if (vec) MRG_FAST(AM, BM_, true); else MRG_FAST(AM, BM_, false); } \
        else if (vec) hipLaunchKernelGGL((gemm_kernel<AM, BM_, AL, true>), grid, block, shmem, stream, g); \
        else hipLaunchKernelGGL((gemm_kernel<AM, BM_, AL, false>), grid, block, shmem, stream, g); } while (0)
#define MRG_LAUNCH(AM, BM_, AL) MRG_LAUNCH2(AM, BM_, AL)
    if (d->a_mode == MRG_OP_KCONTIG && d->b_mode == MRG_OP_KCONTIG) { if (al) MRG_LAUNCH(0, 0, true); else MRG_LAUNCH(0, 0, false); }
    else if (d->a_mode == MRG_OP_KCONTIG && d->b_mode == MRG_OP_KSTRIDED) { if (al) MRG_LAUNCH(0, 1, true); else MRG_LAUNCH(0, 1, false); }
    else { if (al) MRG_LAUNCH(1, 1, true); else MRG_LAUNCH(1, 1, false); }
#undef MRG_LAUNCH
#undef MRG_LAUNCH2
#undef MRG_FAST
#undef MRG_WS
    if (d->ev_stop) (void)hipEventRecord((hipEvent_t)d->ev_stop, stream);
    int rc = mrg_check_launch("mrg_gemm");
    if (rc) return rc;
    return finish_splitk();
}

// n independent forward-layout products in one launch of the single-stage 128 x 128 kernel (gemm_glds_multi_kernel).  The narrow case the
// step needs: both operands k-contiguous and 16-B aligned, K a multiple of 64, no split-K / maps on the operands / device-side K /
// column sums / dropout / accumulate; every fused epilogue of that layout (bias, table row at a mapped output row, fp32 or bf16 C, m_dev).
extern "C" int mrg_gemm_multi(const mrg_gemm_desc* ds, int n, hipStream_t stream) {
    MRG_REQUIRE(ds != nullptr && n >= 1 && n <= MRG_MAX_DOMAINS, MRG_ERR_ARG, "mrg_gemm_multi: %d problems (1 .. %d)", n, MRG_MAX_DOMAINS);
    GemmMultiArgs m;
    int gx = 0;
    for (int p = 0; p < n; ++p) {
        const mrg_gemm_desc* d = ds + p;
        GemmArgs& g = m.g[p];
        MRG_REQUIRE(d->A && d->B && d->C && d->M > 0 && d->N > 0 && d->K > 0, MRG_ERR_ARG, "mrg_gemm_multi: problem %d: null operand or empty shape", p);
        MRG_REQUIRE(d->a_mode == MRG_OP_KCONTIG && d->b_mode == MRG_OP_KCONTIG && d->K % BK == 0 && d->splitk <= 1 && !d->a_map && !d->b_map && !d->k_dev &&
                    !d->colsum_out && !d->accumulate && !(d->drop_rng && d->drop_p > 0.f) && !d->resid && !d->row_scale && !d->C2 && d->act == MRG_ACT_NONE,
                    MRG_ERR_UNSUPPORTED, "mrg_gemm_multi: problem %d is outside the forward-layout case this entry point launches", p);
        MRG_REQUIRE(!d->add_tab || d->add_row, MRG_ERR_ARG, "mrg_gemm_multi: add_tab needs add_row");
        g = GemmArgs{};
        g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B;
        g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb;
        g.m_dev = d->m_dev;
        g.a_aligned = (((uintptr_t)d->A & 15) == 0) && (d->lda % 8 == 0);
        g.b_aligned = (((uintptr_t)d->B & 15) == 0) && (d->ldb % 8 == 0);
        g.C = d->C; g.ldc = d->ldc; g.c_f32 = (d->c_dtype == MRG_DT_F32);
        g.bias = d->bias; g.act = d->act;
        g.row_map = d->row_map; g.add_tab = d->add_tab; g.add_row = d->add_row; g.ldt = d->ldt;
        g.splitk = 1; g.drop.scale = 1.f;
        g.tiles_m = cdiv(d->M, BM); g.tiles_n = cdiv(d->N, BN);
        g.nt = 2; g.xcd_split = 1; g.side_drain = 3;
        auto al4 = [](const void* q, int ld, int bytes) { return q == nullptr || ((((uintptr_t)q) % (4 * bytes) == 0) && (ld % 4 == 0)); };
        const bool vec = (d->N % 4 == 0) && al4(d->C, d->ldc, g.c_f32 ? 4 : 2) && al4(d->add_tab, d->ldt, 4);
        MRG_REQUIRE(g.a_aligned && g.b_aligned && vec, MRG_ERR_UNSUPPORTED, "mrg_gemm_multi: problem %d: operands / outputs must be 16-B aligned with N %% 4 == 0", p);
        // lean epilogues of this layout (mrg_gemm's selection): bias -> bf16 (1), bias + table row -> fp32 C at a mapped row (6); else generic
        g.epi_kind = 0;
        if (!d->add_tab && !d->row_map && !g.c_f32) g.epi_kind = 1;
        else if (d->add_tab && d->add_row && d->row_map && g.c_f32) g.epi_kind = 6;
        const int nt = g.tiles_m * g.tiles_n;
        gx = nt > gx ? nt : gx;
    }
    for (int p = n; p < MRG_MAX_DOMAINS; ++p) m.g[p] = m.g[0];
    const dim3 grid((unsigned)((gx + 7) & ~7), 1, (unsigned)n), block(256);
    hipLaunchKernelGGL((gemm_glds_multi_kernel<0, 0, true, 3, true>), grid, block, (size_t)(2 * TILE_BYTES), stream, m);
    return mrg_check_launch("mrg_gemm_multi");
}

// Grouped weight gradients (gemm8p.hip gemm_p8_group_kernel): C_p[M_p, N_p] (fp32, overwritten) = A_p^T B_p, A_p = dY [K, M_p],
// B_p = X [K, N_p] -- the mm of every nn.Linear backward of a group of encoder blocks (mirage/utils.py:155-157,177,186) in one launch.
extern "C" int mrg_wgrad_group(const mrg_wgrad_group_desc* d, hipStream_t stream) {
    MRG_REQUIRE(d != nullptr, MRG_ERR_ARG, "mrg_wgrad_group: null descriptor");
    MRG_REQUIRE(d->count >= 1 && d->count <= MRG_WGRAD_GROUP_MAX, MRG_ERR_SHAPE, "mrg_wgrad_group: count %d not in [1, %d]", d->count, MRG_WGRAD_GROUP_MAX);
    MRG_REQUIRE(d->K > 0 && d->K % BK == 0, MRG_ERR_SHAPE, "mrg_wgrad_group: K = %d must be a positive multiple of %d", d->K, BK);
    static_assert(MRG_WGRAD_GROUP_MAX == P8_GROUP_MAX, "group size of the C ABI and of the kernel arguments");
    P8GroupArgs ga;
    GemmArgs& g = ga.base;
    g = GemmArgs();
    g.K = d->K;
    g.a_aligned = 1; g.b_aligned = 1;
    const int splitk = d->splitk > 1 ? d->splitk : 1;
    MRG_REQUIRE(splitk == 1 || d->ws, MRG_ERR_ARG, "mrg_wgrad_group: split-K needs a workspace");
    MRG_REQUIRE(splitk <= (d->K / BK), MRG_ERR_SHAPE, "mrg_wgrad_group: split-K factor %d exceeds the %d k-tiles", splitk, d->K / BK);
    g.c_f32 = 1; g.splitk = splitk; g.ws = d->ws; g.epi_kind = splitk == 1 ? 7 : 0; g.act = MRG_ACT_NONE; g.xcd_split = 1;
    { static const int nt = [] { const char* e = getenv("MRG_GEMM_NT"); return e ? atoi(e) : 2; }(); g.nt = nt; }
    { static const int sd = [] { const char* e = getenv("MRG_GEMM_SIDE_DRAIN"); return e ? atoi(e) : 3; }(); g.side_drain = sd; }
    ga.nprob = d->count;
    bool vec = true;
    int tiles = 0;
    long mn_before = 0;
    for (int p = 0; p < d->count; ++p) {
        MRG_REQUIRE(d->A[p] && d->B[p] && d->C[p], MRG_ERR_ARG, "mrg_wgrad_group: null operand pointer (problem %d)", p);
        MRG_REQUIRE(d->M[p] > 0 && d->N[p] > 0, MRG_ERR_SHAPE, "mrg_wgrad_group: bad shape (problem %d)", p);
        MRG_REQUIRE(d->M[p] % 8 == 0 && d->N[p] % 8 == 0 && d->lda[p] % 8 == 0 && d->ldb[p] % 8 == 0 &&
                    (((uintptr_t)d->A[p] | (uintptr_t)d->B[p]) & 15) == 0, MRG_ERR_UNSUPPORTED,
                    "mrg_wgrad_group: operands must be 16-B aligned with M, N, lda, ldb multiples of 8 (problem %d)", p);
        ga.tile_start[p] = tiles;
        ga.ws_off[p] = (long)splitk * mn_before;
        mn_before += (long)d->M[p] * d->N[p];
        MRG_REQUIRE(splitk == 1 || (d->ldc[p] == d->N[p]), MRG_ERR_UNSUPPORTED, "mrg_wgrad_group: split-K slabs need contiguous C (problem %d)", p);
        tiles += cdiv(d->M[p], 256) * cdiv(d->N[p], 256);
        ga.A[p] = (const bf16_t*)d->A[p]; ga.B[p] = (const bf16_t*)d->B[p]; ga.C[p] = d->C[p];
        ga.M[p] = d->M[p]; ga.N[p] = d->N[p]; ga.lda[p] = d->lda[p]; ga.ldb[p] = d->ldb[p]; ga.ldc[p] = d->ldc[p];
        vec = vec && (d->N[p] % 4 == 0) && (d->ldc[p] % 4 == 0) && (((uintptr_t)d->C[p]) % 16 == 0);
    }
    for (int p = 0; p < P8_GROUP_MAX; ++p) ga.ss[p] = nullptr;
    ga.af = AdamFuseArgs();
    if (d->fuse) {
        const mrg_adamw_fuse* f = d->fuse;
        MRG_REQUIRE(splitk == 1, MRG_ERR_UNSUPPORTED, "mrg_wgrad_group: the fused AdamW update needs splitk <= 1");
        MRG_REQUIRE(f->g0 && f->p && f->m && f->v && f->hp && f->n_decay >= 0, MRG_ERR_ARG, "mrg_wgrad_group: fuse: null arena / hp pointer");
        MRG_REQUIRE((((uintptr_t)f->g0 | (uintptr_t)f->p | (uintptr_t)f->m | (uintptr_t)f->v) & 15) == 0 && (((uintptr_t)f->shadow_bf16) & 7) == 0,
                    MRG_ERR_ARG, "mrg_wgrad_group: fuse: arenas must be 16-B aligned");
        for (int p = 0; p < d->count; ++p) {
            MRG_REQUIRE(d->ldc[p] == d->N[p] && d->N[p] % 4 == 0 && (const float*)d->C[p] >= f->g0 && ((const float*)d->C[p] - f->g0) % 4 == 0 && f->ss[p],
                        MRG_ERR_ARG, "mrg_wgrad_group: fuse: C must be a contiguous 16-B aligned view into the gradient arena, ss given (problem %d)", p);
            ga.ss[p] = f->ss[p];
        }
        g.epi_kind = 9;
        ga.af.g0 = f->g0; ga.af.p = f->p; ga.af.m = f->m; ga.af.v = f->v; ga.af.sh = (bf16_t*)f->shadow_bf16; ga.af.hp = f->hp; ga.af.ndecay = f->n_decay;
        ga.af.b1 = f->beta1; ga.af.b2 = f->beta2; ga.af.eps = f->eps;
    }
    for (int p = d->count; p <= P8_GROUP_MAX; ++p) ga.tile_start[p] = tiles;
    for (int p = d->count; p < P8_GROUP_MAX; ++p) { ga.A[p] = nullptr; ga.B[p] = nullptr; ga.C[p] = nullptr; ga.M[p] = ga.N[p] = ga.lda[p] = ga.ldb[p] = ga.ldc[p] = 0; ga.ws_off[p] = 0; }
    if (splitk > 1) vec = vec && (((uintptr_t)d->ws) % 16 == 0);
    if (d->ev_start) (void)hipEventRecord((hipEvent_t)d->ev_start, stream);
    MRG_REQUIRE(mrg_gemm8p_group_launch(ga, vec, stream), MRG_ERR_SHAPE, "mrg_wgrad_group: nothing to launch");
    if (d->ev_stop) (void)hipEventRecord((hipEvent_t)d->ev_stop, stream);
    return mrg_check_launch("mrg_wgrad_group");
}
