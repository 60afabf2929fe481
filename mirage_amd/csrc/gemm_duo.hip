// 128 x 256 x 64 bf16 MFMA GEMM for gfx950 with TWO co-resident workgroups per CU ("duo"): the short-K forward / dgrad products of the
// encoder (mirage/utils.py:155-157,177,186: qkv, proj, fc1, fc2 and their input gradients), where the fused epilogue of a tile is as
// long as its 12-16 k-tiles.
//
//   C[M,N] = A(M x K, k-contiguous) * B(N x K)^T, B k-contiguous (forward) or k-strided (dgrad: W as stored); same fused epilogues
//   as gemm.hip / gemm8p.hip.
//
// Why this shape.  The 256 x 256 8-phase kernel (gemm8p.hip) owns a CU alone: its main loop runs at the matrix unit's issue limit, but
// nothing covers its epilogue (4-17 us per tile against 14-19 us of main loop at K = 768, profiles/r04_streamk.md) nor its pipeline
// fill.  The 128 x 128 kernels (gemm.hip) hide both under 3-4 co-resident workgroups but move twice the operand bytes per flop and
// read one LDS fragment per MFMA.  Here a workgroup is HALF of the 8-phase kernel -- 4 waves (one per SIMD) x (64 x 128) outputs, 128
// accumulator registers, 1.33 MFMAs per fragment read, 85 flop per operand byte -- with its own 80 KiB operand ring, so that exactly
// two workgroups fit a CU (2 x 80 KiB LDS, 2 x 256 VGPRs per SIMD).  The hardware interleaves them: every SIMD holds one wave of each
// workgroup, one wave's MFMA cluster runs under its partner's LDS reads / DMA issue / barrier waits, and -- the point -- one
// workgroup's EPILOGUE (VALU + stores) and the next tile's prologue run under the other's main loop.  No cross-workgroup
// synchronisation: the two tiles drift to complementary phases by themselves.
//
// Ring.  A k-tile is three 16 KiB chunks in the images of gemm8p.hip: A = [128 rows][64 k]; B0 / B1 = the tile columns c with
// ((c >> 6) & 1) == 0 / 1 (so each wave's 128 columns are 64 of B0 and 64 of B1).  Chunks are staged in the order of use
// A(t), B0(t), B1(t), A(t+1), ... into 5 slots (chunk c -> slot c mod 5).  Two phases per k-tile and wave:
//   phase a:  wait for A(t), B0(t) (own DMA: counted vmcnt, then the barrier for the others');  stage B0(t+1) into the slot B1(t-1)
//             left;  16 ds_read_b128 (A rows of the wave: kept for the whole k-tile; B0 columns);  16 MFMA 32x32x16
//   phase b:  wait for B1(t);  stage B1(t+1), A(t+2) into the slots of A(t), B0(t) (all waves passed their phase-a reads);
//             8 ds_read_b128 (B1 columns);  16 MFMA
// i.e. two barriers and two counted waits per k-tile, 3-4 chunks (48-64 KiB) in flight per workgroup.
//
// Epilogue.  The MFMA operands are swapped (D = B_frag x A_frag^T), so a lane owns one ROW of a 32 x 32 block and four consecutive
// columns per register quad.  Each wave transposes its 64 x 128 outputs through a PRIVATE 16 KiB staging area (two passes of 32 rows,
// 16-byte LDS stores, XOR-swizzled) -- no workgroup barrier inside the epilogue -- and drains full 512-byte row segments through the
// shared fused-epilogue code (gemm_common.h drain_side / epi_any).
#include "gemm_common.h"

namespace {

constexpr int D2_CHUNK = 16384;
constexpr int D2_NSLOT = 5;
constexpr int D2_LDS = D2_NSLOT * D2_CHUNK;            // 80 KiB
constexpr int D2_BM = 128, D2_BN = 256;
constexpr int D2_GROUP_M = 8;                          // row tiles that walk a B panel together

typedef __attribute__((address_space(3))) const bf16x8 d2_lds_bf16x8_t;
__device__ __forceinline__ bf16x8 d2_lds_b128(uint32_t byte_addr, int byte_off) {
    return *((d2_lds_bf16x8_t*)(uintptr_t)byte_addr + (byte_off >> 4));
}

// tile column of local row lr of B half-chunk `half`
__device__ __forceinline__ int d2_b_col(int half, int lr) { return ((lr >> 6) << 7) + (half << 6) + (lr & 63); }

// Per-lane byte offset (relative to the operand's tile origin at k-tile 0) of LDS block `b` (1 KiB) of a chunk; rows past the operand's
// edge are clamped to a valid row (their products land in output rows / columns that are never stored).
template <int MODE, bool IS_B>
__device__ __forceinline__ uint32_t d2_lane_off(int ld, int row0, int nrows, int half, int b, int lane) {
    if (MODE == MRG_OP_KCONTIG) {
        const int lr = b * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((lr >> 1) & 7);
        const int tr = IS_B ? d2_b_col(half, lr) : lr;
        const int rr = min(row0 + tr, nrows - 1) - row0;
        return (uint32_t)(rr * ld * 2 + (c << 4));
    } else {
        const int kr = b * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((kr & 3) << 2);
        const int tr = IS_B ? d2_b_col(half, c << 3) : (c << 3);        // 8 consecutive local rows are consecutive rows
        const int col = (row0 + tr + 8 <= nrows) ? tr : 0;
        return (uint32_t)(kr * ld * 2 + col * 2);
    }
}

// LDS fragment addresses of one operand of a wave inside a chunk (two 32-row tiles from local row base_lr); at(soff) adds the chunk's
// slot offset.  KCONTIG: one address per 16-deep k-step (row tiles by immediate offset); KSTRIDED: one per row tile (k-steps by immediate).
template <int MODE>
struct D2Frag {
    uint32_t a[MODE == MRG_OP_KCONTIG ? 4 : 2];
    __device__ __forceinline__ void init(uint32_t base, int base_lr, int lane) {
        if (MODE == MRG_OP_KCONTIG) {
            const int r = lane & 31;
            const uint32_t a0 = base + (uint32_t)((base_lr + r) * 128 + (((lane >> 5) ^ ((r >> 1) & 7)) << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a[kk] = a0 ^ (uint32_t)(kk << 5);
        } else {
            const int i = lane & 15;
            const int krl = ((lane >> 5) << 3) + (i >> 2);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int col = base_lr + rt * 32 + (lane & 16) + ((i & 3) << 2);
                a[rt] = base + (uint32_t)(krl * 256 + ((((col >> 3) ^ ((krl & 3) << 2))) << 4) + ((col & 4) << 1));
            }
        }
    }
    __device__ __forceinline__ D2Frag at(uint32_t soff) const {
        D2Frag f;
#pragma unroll
        for (int x = 0; x < (MODE == MRG_OP_KCONTIG ? 4 : 2); ++x) f.a[x] = a[x] + soff;
        return f;
    }
    __device__ __forceinline__ bf16x8 read(int rt, int kk) const {
        if (MODE == MRG_OP_KCONTIG) {
            return d2_lds_b128(a[kk], rt * 4096);
        } else {
            const s16x4 lo = lds_read_tr16_off(a[rt], kk * 4096);
            const s16x4 hi = lds_read_tr16_off(a[rt], kk * 4096 + 1024);
            bf16x8 f;
            f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
            f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
            return f;
        }
    }
};

__device__ __forceinline__ void d2_tile_origin(const GemmArgs& g, const int bid, int& m0, int& n0) {
    const int tiles_m = (g.M + D2_BM - 1) / D2_BM, tiles_n = (g.N + D2_BN - 1) / D2_BN;
    const int per_group = D2_GROUP_M * tiles_n;
    const int grp = fast_udiv(bid, per_group), in_grp = bid - grp * per_group;
    const int first = grp * D2_GROUP_M;
    const int rows = min(D2_GROUP_M, tiles_m - first);
    const int tn = fast_udiv(in_grp, rows);
    const int tm = first + in_grp - tn * rows;
    m0 = tm * D2_BM; n0 = tn * D2_BN;
}

// next ring slot offset (bytes)
__device__ __forceinline__ uint32_t d2_next(uint32_t soff) { return soff == (uint32_t)((D2_NSLOT - 1) * D2_CHUNK) ? 0u : soff + (uint32_t)D2_CHUNK; }

// main loop of one 128 x 256 tile at (m0, n0) over the k-tiles [t0, t0 + nt).  acc[i][j]: rows wm*64 + i*32.., columns wn*128 + (j>>1)*64
// + (j&1)*32.. in the TRANSPOSED register layout (lane & 31 = row, register 4q + e = column 8q + 4 (lane >> 5) + e of the block).
// On return no LDS-DMA of this call is outstanding; the caller must pass a barrier before it reuses the ring.
// FILL (experiment builds only, -DMRG_DUO_EXPERIMENT): FILL dummy v_fma_f32 behind every MFMA of a cluster, pinned there by
// sched_group_barrier -- does VALU work issued from INSIDE a wave's own MFMA cluster cost time (profiles/r05_gemm_epilogue.md)?
template <int BMODE, int FILL = 0>
__device__ __forceinline__ void d2_mainloop(const GemmArgs& g, const int m0, const int n0, const int t0, const int nt,
                                            f32x16 (&acc)[2][4], char* smem, const int tid, float* fill_out = nullptr) {
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (nt <= 0) return;

    const uint32_t sbase = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    // ---- LDS-DMA setup: every chunk is 16 blocks of 1 KiB, four per wave
    uint32_t oa[4], ob[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        oa[j] = d2_lane_off<MRG_OP_KCONTIG, false>(g.lda, m0, g.M, 0, wave * 4 + j, lane);
        ob[0][j] = d2_lane_off<BMODE, true>(g.ldb, n0, g.N, 0, wave * 4 + j, lane);
        ob[1][j] = d2_lane_off<BMODE, true>(g.ldb, n0, g.N, 1, wave * 4 + j, lane);
    }
    const char* a_org = reinterpret_cast<const char*>(g.A) + ((long)m0 * g.lda + (long)t0 * BK) * 2;
    const char* b_org = reinterpret_cast<const char*>(g.B) + (BMODE == MRG_OP_KCONTIG ? ((long)n0 * g.ldb + (long)t0 * BK) * 2 : ((long)t0 * BK * g.ldb + n0) * 2);
    const long step_a = BK * 2, step_b = BMODE == MRG_OP_KCONTIG ? BK * 2 : (long)BK * g.ldb * 2;
    const uint32_t dst_w = sbase + (uint32_t)(wave * 4096);            // this wave's four 1 KiB blocks inside any chunk
    uint32_t wslot = 0;                                                 // slot offset of the next chunk to stage
    auto stage_a = [&](int T) {
        const char* s = uniform_ptr(a_org + (long)T * step_a);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_s(s, oa[j], dst_w + wslot + (uint32_t)(j * 1024));
        wslot = d2_next(wslot);
    };
    auto stage_b = [&](int T, int h) {
        const char* s = uniform_ptr(b_org + (long)T * step_b);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_s(s, ob[h][j], dst_w + wslot + (uint32_t)(j * 1024));
        wslot = d2_next(wslot);
    };
    // ---- fragment addresses (slot 0): A chunk local rows wm*64.., B half-chunks local rows wn*64..
    D2Frag<MRG_OP_KCONTIG> fra0;
    D2Frag<BMODE> frb0;
    fra0.init(sbase, wm * 64, lane);
    frb0.init(sbase, wn * 64, lane);

    // prologue: A(0), B0(0), B1(0), A(1)
    stage_a(0); stage_b(0, 0); stage_b(0, 1);
    if (nt > 1) stage_a(1);
    uint32_t rslot = 0;                                                 // slot offset of the next chunk to read
    bf16x8 fa[2][4], fb[2][4];
    float fd[8];
    if constexpr (FILL > 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) fd[q] = (float)(lane + q);
    }
    for (int t = 0; t < nt; ++t) {
        const bool more = t + 1 < nt;
        // ---- phase a: A(t) x B0(t)
        if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more) stage_b(t + 1, 0);
        {
            const D2Frag<MRG_OP_KCONTIG> fra = fra0.at(rslot);
            rslot = d2_next(rslot);
            const D2Frag<BMODE> frb = frb0.at(rslot);
            rslot = d2_next(rslot);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { fb[0][kk] = frb.read(0, kk); fb[1][kk] = frb.read(1, kk); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { fa[0][kk] = fra.read(0, kk); fa[1][kk] = fra.read(1, kk); }
        }
#define D2_MFMA16(J0) do { \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_setprio(1); \
        asm volatile("" : "+v"(acc[0][J0]), "+v"(acc[1][J0]), "+v"(acc[0][J0 + 1]), "+v"(acc[1][J0 + 1]) :: "memory"); \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { \
            acc[0][J0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][kk], fa[0][kk], acc[0][J0], 0, 0, 0); \
            acc[1][J0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][kk], fa[1][kk], acc[1][J0], 0, 0, 0); \
            acc[0][J0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][kk], fa[0][kk], acc[0][J0 + 1], 0, 0, 0); \
            acc[1][J0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][kk], fa[1][kk], acc[1][J0 + 1], 0, 0, 0); \
            if constexpr (FILL > 0) { \
                _Pragma("unroll") for (int q = 0; q < 4 * FILL; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fd[q & 7]) : "v"(fd[(q + 1) & 7] )); \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0); } \
            } \
        } \
        asm volatile("" : "+v"(acc[0][J0]), "+v"(acc[1][J0]), "+v"(acc[0][J0 + 1]), "+v"(acc[1][J0 + 1]) :: "memory"); \
        __builtin_amdgcn_s_setprio(0); } while (0)
        D2_MFMA16(0);
        // ---- phase b: A(t) (still in registers) x B1(t)
        if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more) stage_b(t + 1, 1);
        if (t + 2 < nt) stage_a(t + 2);
        {
            const D2Frag<BMODE> frb = frb0.at(rslot);
            rslot = d2_next(rslot);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { fb[0][kk] = frb.read(0, kk); fb[1][kk] = frb.read(1, kk); }
        }
        D2_MFMA16(2);
#undef D2_MFMA16
    }
    if constexpr (FILL > 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += fd[q];
        if (fill_out && t == 12345.678f) fill_out[tid] = t;      // keeps the dummy chains alive; never true in practice
    }
}

// Epilogue of one tile: per wave, two passes of 32 rows x 128 columns through a private fp32 staging area (16 KiB at smem + wave * 20 KiB).
// The caller has passed a barrier after the main loop (the staging areas overlay the operand ring).
template <bool VEC, int ps>
__device__ __forceinline__ void d2_epilogue_pass(const GemmArgs& g, const int m0, const int n0, f32x16 (&acc)[2][4], char* smem, float* ws,
                                                 const float (&bv)[4], const int tid) {
    const int Meff = g.M;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    float* ct = reinterpret_cast<float*>(smem + wave * 20480);
    {
        const int m = lane & 31, h = lane >> 5;
        float* rowp = ct + m * 128;
        const int sw = (m & 7) << 2;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = ((j >> 1) * 64 + (j & 1) * 32 + 8 * q + 4 * h) ^ sw;
                *reinterpret_cast<float4*>(rowp + c) = make_float4(acc[ps][j][4 * q], acc[ps][j][4 * q + 1], acc[ps][j][4 * q + 2], acc[ps][j][4 * q + 3]);
            }
    }
    const int cl = (lane & 31) << 2;
    const int col = n0 + wn * 128 + cl;
    const int nv = min(4, g.N - col);
    if (col < g.N) {
        const int rbase = m0 + wm * 64 + ps * 32;
        auto row_of = [&](int it, int& rl, int& row) { rl = (it << 1) + (lane >> 5); row = rbase + rl; };
#define D2_DRAIN16(KIND_, CSP_) do { \
        drain_side<KIND_, false, 8, 128, true>(g, ct, cl, col, bv, Meff, [&](int it, int& rl, int& row) { row_of(it, rl, row); }, CSP_); \
        drain_side<KIND_, false, 8, 128, true>(g, ct, cl, col, bv, Meff, [&](int it, int& rl, int& row) { row_of(it + 8, rl, row); }, CSP_); } while (0)
        if (VEC && ws) { D2_DRAIN16(7, ws); }
        else if (VEC && g.epi_kind == 2) { D2_DRAIN16(2, nullptr); }
        else if (VEC && g.epi_kind == 4) { D2_DRAIN16(4, nullptr); }
        else if (VEC && g.epi_kind == 1) { D2_DRAIN16(1, nullptr); }
        else if (VEC && g.epi_kind == 3) { D2_DRAIN16(3, nullptr); }
        else if (VEC && g.epi_kind == 5) { D2_DRAIN16(5, nullptr); }
        else if (VEC && g.epi_kind == 7) { D2_DRAIN16(8, nullptr); }
        else {
#pragma unroll 1
            for (int it = 0; it < 16; ++it) {
                int rl, row;
                row_of(it, rl, row);
                if (row < Meff) {
                    const float4 a4 = *reinterpret_cast<const float4*>(ct + rl * 128 + (cl ^ ((rl & 7) << 2)));
                    epi_any<VEC>(g, a4, bv, ws, row, col, nv);
                }
            }
        }
#undef D2_DRAIN16
    }
}

template <int BMODE, bool VEC, int FILL = 0>
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = ((g.M + D2_BM - 1) / D2_BM) * ((g.N + D2_BN - 1) / D2_BN);
    int bid, ksplit;
    wg_remap(nwg, g.xcd_split, bid, ksplit);
    int m0, n0;
    d2_tile_origin(g, bid, m0, n0);
    const int ntiles = g.K / BK;
    int t0 = 0, t1 = ntiles;
    if (g.splitk > 1) {
        const int per = (ntiles + g.splitk - 1) / g.splitk;
        t0 = ksplit * per;
        t1 = min(ntiles, t0 + per);
    }
    const int tid = threadIdx.x;
    float bv[4];
    {
        const int col = n0 + ((tid >> 6) & 1) * 128 + ((tid & 31) << 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (g.splitk <= 1 && g.bias && col + e < g.N) ? g.bias[col + e] : 0.f;
    }
    f32x16 acc[2][4];
    d2_mainloop<BMODE, FILL>(g, m0, n0, t0, t1 - t0, acc, smem, tid, FILL > 0 ? reinterpret_cast<float*>(g.C) : nullptr);
    __syncthreads();                                   // every wave has read its last fragments: the ring becomes the staging areas
    // The bias registers were loaded before the main loop and are first used inside the epilogue's `if (row < M)` blocks.  hipcc's wait
    // insertion then keeps them "pending" at every join (the block that waited may have been skipped) and opens EVERY row segment with
    // s_waitcnt vmcnt(0) -- which also drains the stores of the segment before.  One unconditional use here retires them for good.
    MRG_RETIRE4(bv);
    float* ws = g.splitk > 1 ? g.ws + (long)ksplit * g.M * g.N : nullptr;
    d2_epilogue_pass<VEC, 0>(g, m0, n0, acc, smem, ws, bv, tid);
    d2_epilogue_pass<VEC, 1>(g, m0, n0, acc, smem, ws, bv, tid);
    if (ws && g.tile_cnt) splitk_finish<VEC>(g, bid, m0, n0, D2_BM, D2_BN, g.M, smem, tid, 256);
}

}  // namespace

bool mrg_gemm_duo_launch(const mrg_detail::GemmArgs& g, int a_mode, int b_mode, bool vec, int splitk, hipStream_t stream) {
    if (g.K % BK != 0 || g.a_map || g.b_map || g.m_dev || g.k_dev || g.colsum) return false;
    if (!g.a_aligned || !g.b_aligned) return false;
    if (a_mode != MRG_OP_KCONTIG) return false;
    if (b_mode == MRG_OP_KSTRIDED && (g.N % 8 != 0)) return false;
    const dim3 grid(cdiv(g.M, D2_BM) * cdiv(g.N, D2_BN), splitk), block(256);
#define MRG_DUO(BM_, V) do { const void* fn = (const void*)gemm_duo_kernel<BM_, V>; \
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, D2_LDS); \
        hipLaunchKernelGGL((gemm_duo_kernel<BM_, V>), grid, block, (size_t)D2_LDS, stream, g); } while (0)
#ifdef MRG_DUO_EXPERIMENT
    static const int fill = [] { const char* e = getenv("MRG_DUO_FILL"); return e ? atoi(e) : 0; }();
#define MRG_DUOF(F) do { const void* fn = (const void*)gemm_duo_kernel<0, true, F>; \
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, D2_LDS); \
        hipLaunchKernelGGL((gemm_duo_kernel<0, true, F>), grid, block, (size_t)D2_LDS, stream, g); } while (0)
    if (fill > 0 && b_mode == MRG_OP_KCONTIG && vec) {
        if (fill == 1) MRG_DUOF(1); else if (fill == 2) MRG_DUOF(2); else if (fill == 3) MRG_DUOF(3); else if (fill == 4) MRG_DUOF(4);
        else if (fill == 6) MRG_DUOF(6); else MRG_DUOF(8);
        return true;
    }
#undef MRG_DUOF
#endif
    if (b_mode == MRG_OP_KCONTIG) { if (vec) MRG_DUO(0, true); else MRG_DUO(0, false); }
    else { if (vec) MRG_DUO(1, true); else MRG_DUO(1, false); }
#undef MRG_DUO
    return true;
}
