// 256 x 256 x 64 bf16 MFMA GEMM with an 8-phase (two k-tiles x four phases) ping-pong schedule for gfx950.
//
//   C[M,N] = A(M x K) * B(N x K)^T, same operand modes and fused epilogues as gemm.hip (replaces the same aten addmm / mm
//   dispatches: mirage/utils.py:155-157,177,186,209-221), for problems with enough 256 x 256 tiles or a long K.
//
// Geometry.  8 waves = 2 (M) x 4 (N), each wave owns 128 x 64 of the output = 4 x 2 MFMA 32x32 tiles (128 accumulator
// registers).  A k-tile (64 deep) is four 16 KiB LDS "half-tiles": A0/A1 hold the tile rows r with ((r >> 6) & 1) == 0 / 1,
// B0/B1 the tile columns c with ((c >> 5) & 1) == 0 / 1, each as a [128][64] (k-contiguous) or [64][128] (k-strided) image
// with the XOR swizzles of gemm.hip.  So every wave's output quadrant (64 rows x 32 columns) needs exactly one A half and
// one B half.  Two buffers of four halves = 128 KiB, one workgroup per CU.
//
// Schedule (cdna_hip_programming.md "The 256^2 8-phase template").  A k-tile is four phases, one output quadrant each:
//   (A0,B0) (A0,B1) (A1,B1) (A1,B0); a phase = { LDS fragment reads of the halves it newly needs (12 / 4 / 8 / 0 ds_read_b128),
//   LDS-DMA of ONE half-tile of a later k-tile (2 x global_load_lds_dwordx4 per lane), s_barrier, 8 MFMA 32x32x16 under
//   s_setprio 1, s_barrier }.  The two M-halves of the workgroup (waves 0-3 / 4-7, which share the four SIMDs pairwise) run
//   one barrier apart, so that on every SIMD one wave is in its MFMA cluster while its partner reads LDS and issues DMA.
//   Half-tiles are staged 7 slots ahead in the order of first use / last read (A0, B0, B1, A1); the only vector-memory wait in
//   the loop is a counted s_waitcnt vmcnt(6) in the last phase of a k-tile (the three youngest half-tiles stay in flight across
//   the barriers).  Hazards: a half-tile is re-staged >= 1 phase after its last ds_read (which an lgkmcnt(0) retired before that
//   phase's MFMAs); staged data is read only after the issuing waves' vmcnt AND a barrier both wave groups have passed.
#include "gemm_common.h"

namespace {

constexpr int P8_REGION = 16384;                       // one half-tile
constexpr int P8_LDS = 8 * P8_REGION;                  // 128 KiB
// LDS byte address of half-tile (op: 0 = A, 1 = B; half; buf): buf toggles bit 14, half bit 15, op bit 16
__device__ __forceinline__ constexpr uint32_t p8_region(int op, int half, int buf) { return (uint32_t)(((op * 2 + half) * 2 + buf) * P8_REGION); }

typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8_t;
__device__ __forceinline__ bf16x8 lds_read_b128_off(uint32_t byte_addr, int byte_off) {
    return *((lds_bf16x8_t*)(uintptr_t)byte_addr + (byte_off >> 4));
}

// tile row (A) / tile column (B) of local row lr of half-tile `half`
__device__ __forceinline__ int p8_a_row(int half, int lr) { return ((lr >> 6) << 7) + (half << 6) + (lr & 63); }
__device__ __forceinline__ int p8_b_row(int half, int lr) { return ((lr >> 5) << 6) + (half << 5) + (lr & 31); }

// Per-lane byte offset (relative to the operand's tile origin at k-tile 0) of LDS block `b` (1 KiB) of a half-tile; rows past the
// operand's edge are clamped to its last row (their products land in output rows / columns that are never stored).
template <int MODE, bool IS_A>
__device__ __forceinline__ uint32_t p8_lane_off(int ld, int row0, int nrows, int half, int b, int lane) {
    if (MODE == MRG_OP_KCONTIG) {
        const int lr = b * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((lr >> 1) & 7);
        const int tr = IS_A ? p8_a_row(half, lr) : p8_b_row(half, lr);
        const int rr = min(row0 + tr, nrows - 1) - row0;
        return (uint32_t)(rr * ld * 2 + (c << 4));
    } else {
        const int kr = b * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((kr & 3) << 2);
        const int tr = IS_A ? p8_a_row(half, c << 3) : p8_b_row(half, c << 3);     // 8 consecutive local rows are consecutive rows
        const int col = (row0 + tr + 8 <= nrows) ? tr : 0;
        return (uint32_t)(kr * ld * 2 + col * 2);
    }
}

// LDS fragment addresses of one operand for a wave.  KCONTIG: one address per 16-deep k-step (row tiles by immediate offset);
// KSTRIDED: one per 32-row tile (k-steps by immediate).  `base_lr` = first local row of the wave inside the half-tiles.
template <int MODE, int NRT>
struct P8Frag {
    uint32_t a[MODE == MRG_OP_KCONTIG ? 4 : NRT];
    __device__ __forceinline__ void init(uint32_t region0, int base_lr, int lane) {
        if (MODE == MRG_OP_KCONTIG) {
            const int r = lane & 31;                                   // base_lr is a multiple of 32: (row >> 1) & 7 = (r >> 1) & 7
            const uint32_t a0 = region0 + (uint32_t)((base_lr + r) * 128 + (((lane >> 5) ^ ((r >> 1) & 7)) << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a[kk] = a0 ^ (uint32_t)(kk << 5);
        } else {
            const int i = lane & 15;
            const int krl = ((lane >> 5) << 3) + (i >> 2);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const int col = base_lr + rt * 32 + (lane & 16) + ((i & 3) << 2);
                a[rt] = region0 + (uint32_t)(krl * 256 + ((((col >> 3) ^ ((krl & 3) << 2))) << 4) + ((col & 4) << 1));
            }
        }
    }
    __device__ __forceinline__ void toggle() {
#pragma unroll
        for (int x = 0; x < (MODE == MRG_OP_KCONTIG ? 4 : NRT); ++x) a[x] ^= (uint32_t)P8_REGION;
    }
    // fragment of row tile rt, k-step kk, from half-tile `half` (compile-time)
    __device__ __forceinline__ bf16x8 read(int half, int rt, int kk) const {
        if (MODE == MRG_OP_KCONTIG) {
            return lds_read_b128_off(a[kk], half * 2 * P8_REGION + rt * 4096);
        } else {
            const s16x4 lo = lds_read_tr16_off(a[rt], half * 2 * P8_REGION + kk * 4096);
            const s16x4 hi = lds_read_tr16_off(a[rt], half * 2 * P8_REGION + kk * 4096 + 1024);
            bf16x8 f;
            f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
            f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
            return f;
        }
    }
};

// (row, column) origin of tile `bid` of the problem in grouped row-major order (GROUP_M row tiles share a B panel walk)
__device__ __forceinline__ void p8_tile_origin(const GemmArgs& g, const int bid, int& m0, int& n0) {
    const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + 255) / 256;
    constexpr int GROUP_M = 4;
    const int per_group = GROUP_M * tiles_n;
    const int grp = fast_udiv(bid, per_group), in_grp = bid - grp * per_group;
    const int first = grp * GROUP_M;
    const int rows = min(GROUP_M, tiles_m - first);
    const int tn = fast_udiv(in_grp, rows);
    const int tm = first + in_grp - tn * rows;
    m0 = tm * 256; n0 = tn * 256;
}

// main loop of one 256 x 256 tile at (m0, n0) over the k-tiles [t0, t0 + nt): acc = A[m0.., k] B[n0.., k]^T (accumulators zeroed here).
// On return both wave groups are aligned again and no LDS read or LDS-DMA of this call is outstanding.
template <int AMODE, int BMODE, int PHASES>
__device__ __forceinline__ void p8_mainloop(const GemmArgs& g, const int m0, const int n0, const int t0, const int nt,
                                            f32x16 (&acc)[4][2], char* smem, const int tid) {
    const int Meff = g.M;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;             // rows wm*128.., cols wn*64..
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const uint32_t sbase = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    // ---- LDS-DMA setup: per (operand, half) two blocks per wave
    uint32_t oa[2][2], ob[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            oa[h][j] = p8_lane_off<AMODE, true>(g.lda, m0, Meff, h, wave * 2 + j, lane);
            ob[h][j] = p8_lane_off<BMODE, false>(g.ldb, n0, g.N, h, wave * 2 + j, lane);
        }
    const char* a_org = reinterpret_cast<const char*>(g.A) + (AMODE == MRG_OP_KCONTIG ? ((long)m0 * g.lda + (long)t0 * BK) * 2 : ((long)t0 * BK * g.lda + m0) * 2);
    const char* b_org = reinterpret_cast<const char*>(g.B) + (BMODE == MRG_OP_KCONTIG ? ((long)n0 * g.ldb + (long)t0 * BK) * 2 : ((long)t0 * BK * g.ldb + n0) * 2);
    const long step_a = AMODE == MRG_OP_KCONTIG ? BK * 2 : (long)BK * g.lda * 2, step_b = BMODE == MRG_OP_KCONTIG ? BK * 2 : (long)BK * g.ldb * 2;
    const uint32_t dst_w = sbase + (uint32_t)(wave * 2048);            // this wave's two 1 KiB blocks inside any half-tile
    // slot s of k-tile T (relative to t0): 0 = A half 0, 1 = B half 0, 2 = B half 1, 3 = A half 1 (order of first use = last read)
    // MRG_GEMM_ABLATE bits (roofline analysis, builds with -DMRG_P8_ABLATE only: the run-time checks alone cost ~20 %):
    // 1 no DMA in the loop, 2 no MFMA, 4 no LDS fragment reads, 8 no setprio, 16 no stagger
#ifdef MRG_P8_ABLATE
    const int abl = g.ablate;
#else
    constexpr int abl = 0;
#endif
    bool in_loop = false;
    auto stage = [&](int T, int slot) {
        if (T >= nt || (in_loop && (abl & 1))) return;
        const int buf = T & 1;
        if (slot == 0 || slot == 3) {
            const int h = slot == 3;
            const char* sa = uniform_ptr(a_org + (long)T * step_a);
            glds16_s(sa, oa[h][0], dst_w + p8_region(0, h, 0) + (uint32_t)(buf * P8_REGION));
            glds16_s(sa, oa[h][1], dst_w + p8_region(0, h, 0) + (uint32_t)(buf * P8_REGION) + 1024);
        } else {
            const int h = slot == 2;
            const char* sb = uniform_ptr(b_org + (long)T * step_b);
            glds16_s(sb, ob[h][0], dst_w + p8_region(1, h, 0) + (uint32_t)(buf * P8_REGION));
            glds16_s(sb, ob[h][1], dst_w + p8_region(1, h, 0) + (uint32_t)(buf * P8_REGION) + 1024);
        }
    };
    // ---- fragment addresses: A half-tiles local rows wm*64 .. +63 (two 32-row tiles), B half-tiles local rows wn*32 .. +31
    P8Frag<AMODE, 2> fra;
    P8Frag<BMODE, 1> frb;
    fra.init(sbase + p8_region(0, 0, 0), wm * 64, lane);
    frb.init(sbase + p8_region(1, 0, 0), wn * 32, lane);

    if (nt > 0) {
        // prologue: k-tile 0 completely, three half-tiles of k-tile 1
        stage(0, 0); stage(0, 1); stage(0, 2); stage(0, 3);
        stage(1, 0); stage(1, 1); stage(1, 2);
        if (nt >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wm == 1 && !(abl & 16)) __builtin_amdgcn_s_barrier();               // the second M-half runs one barrier behind the first
        __builtin_amdgcn_sched_barrier(0);
        in_loop = true;

        bf16x8 fa[2][4] = {}, fb0[4] = {}, fb1[4] = {};
        // The MFMAs are pure register instructions: nothing but data dependences keeps the compiler from floating them across
        // the barriers / priority changes (sched_barrier only pins what instruction selection already ordered).  The two empty
        // asm statements make the cluster's accumulators "produced" after the first barrier and "consumed" before the second.
#define P8_MFMA(I0, J, FB) do { \
        __builtin_amdgcn_s_barrier(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if (!(abl & 8)) __builtin_amdgcn_s_setprio(1); \
        asm volatile("" : "+v"(acc[I0][J]), "+v"(acc[I0 + 1][J]) :: "memory"); \
        if (!(abl & 2)) { \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { \
            acc[I0][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][kk], FB[kk], acc[I0][J], 0, 0, 0); \
            acc[I0 + 1][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][kk], FB[kk], acc[I0 + 1][J], 0, 0, 0); \
        } } else { asm volatile("" :: "v"(fa[0][0]), "v"(fa[1][3]), "v"(FB[0]), "v"(FB[3])); } \
        asm volatile("" : "+v"(acc[I0][J]), "+v"(acc[I0 + 1][J]) :: "memory"); \
        __builtin_amdgcn_s_setprio(0); \
        __builtin_amdgcn_s_barrier(); \
        asm volatile("" ::: "memory"); } while (0)

        if constexpr (PHASES == 2) {
            // Two phases per k-tile (half the barriers): phase 0 = A0 x (B0, B1), phase 1 = A1 x (B0, B1), 16 MFMAs each.  A1 of the
            // next k-tile is staged in phase 0, A0 / B0 / B1 of the one after in phase 1 (their previous contents were last read
            // in phase 0); the wait before the second phase's barrier leaves those three half-tiles in flight.
#define P8_MFMA16(I0) do { \
        __builtin_amdgcn_s_barrier(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if (!(abl & 8)) __builtin_amdgcn_s_setprio(1); \
        asm volatile("" : "+v"(acc[I0][0]), "+v"(acc[I0 + 1][0]), "+v"(acc[I0][1]), "+v"(acc[I0 + 1][1]) :: "memory"); \
        if (!(abl & 2)) { \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { \
            acc[I0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][kk], fb0[kk], acc[I0][0], 0, 0, 0); \
            acc[I0 + 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][kk], fb0[kk], acc[I0 + 1][0], 0, 0, 0); \
            acc[I0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][kk], fb1[kk], acc[I0][1], 0, 0, 0); \
            acc[I0 + 1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][kk], fb1[kk], acc[I0 + 1][1], 0, 0, 0); \
        } } else { asm volatile("" :: "v"(fa[0][0]), "v"(fa[1][3]), "v"(fb0[0]), "v"(fb1[3])); } \
        asm volatile("" : "+v"(acc[I0][0]), "+v"(acc[I0 + 1][0]), "+v"(acc[I0][1]), "+v"(acc[I0 + 1][1]) :: "memory"); \
        __builtin_amdgcn_s_setprio(0); \
        __builtin_amdgcn_s_barrier(); \
        asm volatile("" ::: "memory"); } while (0)
            for (int t = 0; t < nt; ++t) {
                if (!(abl & 4)) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) { fb0[kk] = frb.read(0, 0, kk); fb1[kk] = frb.read(1, 0, kk); }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) { fa[0][kk] = fra.read(0, 0, kk); fa[1][kk] = fra.read(0, 1, kk); }
                }
                stage(t + 1, 3);
                P8_MFMA16(0);
                if (!(abl & 4)) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) { fa[0][kk] = fra.read(1, 0, kk); fa[1][kk] = fra.read(1, 1, kk); }
                }
                stage(t + 2, 0); stage(t + 2, 1); stage(t + 2, 2);
                if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                P8_MFMA16(2);
                fra.toggle(); frb.toggle();
            }
#undef P8_MFMA16
        } else {
        for (int t = 0; t < nt; ++t) {
            // ---- phase 0: quadrant (A0, B0)
            if (!(abl & 4)) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fb0[kk] = frb.read(0, 0, kk);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { fa[0][kk] = fra.read(0, 0, kk); fa[1][kk] = fra.read(0, 1, kk); }
            }
            stage(t + 1, 3);
            P8_MFMA(0, 0, fb0);
            // ---- phase 1: quadrant (A0, B1)
            if (!(abl & 4)) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fb1[kk] = frb.read(1, 0, kk);
            }
            stage(t + 2, 0);
            P8_MFMA(0, 1, fb1);
            // ---- phase 2: quadrant (A1, B1)
            if (!(abl & 4)) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { fa[0][kk] = fra.read(1, 0, kk); fa[1][kk] = fra.read(1, 1, kk); }
            }
            stage(t + 2, 1);
            P8_MFMA(2, 1, fb1);
            // ---- phase 3: quadrant (A1, B0); k-tile t+1 must have landed before the next phase reads it
            stage(t + 2, 2);
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            P8_MFMA(2, 0, fb0);
            fra.toggle(); frb.toggle();
        }
        }
#undef P8_MFMA
        if (wm == 0 && !(abl & 16)) __builtin_amdgcn_s_barrier();               // re-align the two halves
    }
}

// bias values of this thread's four output columns (requested before the main loop: the load's latency would otherwise open the epilogue)
__device__ __forceinline__ void p8_bias(const GemmArgs& g, const int n0, bool want, float (&bv)[4], const int tid) {
    const int col = n0 + ((tid & 63) << 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (want && g.bias && col + e < g.N) ? g.bias[col + e] : 0.f;
}

// Epilogue of one tile: two passes of 128 rows (64 of each M-half, so that all eight waves stage at once) through a 128 KiB fp32 staging
// tile -- the whole operand ring, idle by now -- drained as 16-B row segments in batches of eight (side operands of a batch are
// requested before the first is consumed).  One workgroup per CU means nothing hides this phase: at K = 768 it is as long as the main
// loop (profiles/r04_streamk.md), hence two barriers per pass instead of four half-used ones and no dependent global round trip per row.
// ws != nullptr: split-K slab stores instead.
template <bool VEC, int ps>
__device__ __forceinline__ void p8_epilogue_pass(const GemmArgs& g, const int m0, const int n0, f32x16 (&acc)[4][2], char* smem, float* ws,
                                                 const float (&bv)[4], unsigned long long* etr, const int tid, float* ssacc = nullptr,
                                                 const AdamFuseArgs* af = nullptr) {
    const int Meff = g.M;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    float* ct = reinterpret_cast<float*>(smem);
    const int cl = (tid & 63) << 2;
    const int col = n0 + cl;
    const int nv = min(4, g.N - col);
    {
        __syncthreads();
        if (etr && tid == 0) etr[3 * ps] = wall_clock64();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c2 = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * 64 + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    ct[rl * 256 + c2] = acc[2 * ps + ii][j][r];
                }
            }
        __syncthreads();
        if (etr && tid == 0) etr[3 * ps + 1] = wall_clock64();
        if (col < g.N) {
            // staging row rl (0..127) = M-half (rl >> 6), row (rl & 63) of this pass's 64-row band of that half
            auto row_of = [&](int it, int& rl, int& row) { rl = (tid >> 6) + (it << 3); row = m0 + ((rl >> 6) << 7) + ps * 64 + (rl & 63); };
#define P8_DRAIN16(KIND_, CSP_) do { \
            drain_side<KIND_, false, 8, 256>(g, ct, cl, col, bv, Meff, [&](int it, int& rl, int& row) { row_of(it, rl, row); }, CSP_); \
            drain_side<KIND_, false, 8, 256>(g, ct, cl, col, bv, Meff, [&](int it, int& rl, int& row) { row_of(it + 8, rl, row); }, CSP_); } while (0)
            if (VEC && (g.side_drain & 2) && ws) { P8_DRAIN16(7, ws); }
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 2) { P8_DRAIN16(2, nullptr); }
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 4) { P8_DRAIN16(4, nullptr); }
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 1) { P8_DRAIN16(1, nullptr); }
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 3) { P8_DRAIN16(3, nullptr); }
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 5) { P8_DRAIN16(5, nullptr); }
            else if (VEC && (g.side_drain & 1) && g.epi_kind == 7) { P8_DRAIN16(8, nullptr); }
            else if (VEC && g.epi_kind == 9) {   // fused AdamW: three more 16-B operands per row segment -> batches of four
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    drain_side<9, false, 4, 256>(g, ct, cl, col, bv, Meff, [&](int it, int& rl, int& row) { row_of(it + 4 * q, rl, row); }, ssacc, af);
            }
            else {
#pragma unroll 1
                for (int it = 0; it < 16; ++it) {
                    int rl, row;
                    row_of(it, rl, row);
                    if (row < Meff) {
                        const float4 a4 = *reinterpret_cast<const float4*>(ct + rl * 256 + cl);
                        epi_any<VEC>(g, a4, bv, ws, row, col, nv);
                    }
                }
            }
#undef P8_DRAIN16
        }
        if (etr && tid == 0) etr[3 * ps + 2] = wall_clock64();
    }
}

template <bool VEC>
__device__ __forceinline__ void p8_epilogue(const GemmArgs& g, const int m0, const int n0, f32x16 (&acc)[4][2], char* smem, float* ws,
                                            const float (&bv)[4], const int tid, unsigned long long* etr = nullptr, float* ssacc = nullptr,
                                            const AdamFuseArgs* af = nullptr) {
    // two calls with a compile-time pass index: inside a loop the accumulator array is indexed at run time and moves to scratch memory
    p8_epilogue_pass<VEC, 0>(g, m0, n0, acc, smem, ws, bv, etr, tid, ssacc, af);
    p8_epilogue_pass<VEC, 1>(g, m0, n0, acc, smem, ws, bv, etr, tid, ssacc, af);
}

// one 256 x 256 output tile (tile `bid` of the problem in grouped row-major order, k-range `ksplit` of g.splitk)
template <int AMODE, int BMODE, bool VEC, int PHASES>
__device__ __forceinline__ void p8_tile(const GemmArgs& g, const int bid, const int ksplit, char* smem, float* ssacc = nullptr,
                                        const AdamFuseArgs* af = nullptr) {
    int m0, n0;
    p8_tile_origin(g, bid, m0, n0);
    const int ntiles = g.K / BK;
    int t0 = 0, t1 = ntiles;
    if (g.splitk > 1) {
        const int per = (ntiles + g.splitk - 1) / g.splitk;
        t0 = ksplit * per;
        t1 = min(ntiles, t0 + per);
    }
    const int tid = threadIdx.x;
    float bv[4];
    p8_bias(g, n0, g.splitk <= 1, bv, tid);
    f32x16 acc[4][2];
    p8_mainloop<AMODE, BMODE, PHASES>(g, m0, n0, t0, t1 - t0, acc, smem, tid);
    MRG_RETIRE4(bv);
    float* ws = g.splitk > 1 ? g.ws + (long)ksplit * g.M * g.N : nullptr;
    p8_epilogue<VEC>(g, m0, n0, acc, smem, ws, bv, tid, nullptr, ssacc, af);
    if (ws && g.tile_cnt) splitk_finish<VEC>(g, bid, m0, n0, 256, 256, g.M, smem, threadIdx.x, 512);
}

template <int AMODE, int BMODE, bool VEC, int PHASES>
__global__ __launch_bounds__(512) void gemm_p8_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    int bid, ksplit;
    wg_remap(nwg, g.xcd_split, bid, ksplit);
    p8_tile<AMODE, BMODE, VEC, PHASES>(g, bid, ksplit, smem);
}

// Grouped weight gradients: up to P8_GROUP_MAX independent problems C_p (fp32, overwritten) = A_p^T B_p with BOTH operands contracted
// over their leading (token) axis, in ONE launch and WITHOUT split-K.  A single weight gradient of the encoder has 9-36 output
// tiles of 256 x 256 -- too few for 256 CUs, hence split-K, fp32 slabs (66 MB per fc1 gradient at C2) and a reduction pass.  The
// weight gradients of several Linears / blocks are independent and become available together (the side stream is forked once per
// group of blocks), so their tiles fill the chip side by side and every tile contracts over ALL tokens (129 / 99 k-tiles: the
// 8-phase loop at its long-K rate).  XCD x gets one contiguous band of the concatenated tile list (tiles of a problem that share an
// A panel are neighbours).  Measured (tools/debug/exp_group_wgrad.py): 100-106 us per Base block at C2 against 197 us for the four
// tuned split-K GEMMs + reductions, 130 against 215 us per Large block at C4.
template <bool VEC, int PHASES>
__global__ __launch_bounds__(512) void gemm_p8_group_kernel(const P8GroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int total = ga.tile_start[ga.nprob];
    int bid, ksplit;
    wg_remap(total, 1, bid, ksplit);             // gridDim.y = split-K factor (1 for groups that fill the chip by themselves)
    int p = 0;
    while (p + 1 < ga.nprob && bid >= ga.tile_start[p + 1]) ++p;
    GemmArgs g = ga.base;
    g.A = ga.A[p]; g.B = ga.B[p]; g.C = ga.C[p];
    g.M = ga.M[p]; g.N = ga.N[p]; g.lda = ga.lda[p]; g.ldb = ga.ldb[p]; g.ldc = ga.ldc[p];
    if (g.splitk > 1) g.ws = ga.base.ws + ga.ws_off[p];
    // fused AdamW (epi_kind 9): the epilogue updates the parameter elements of its tile and every wave leaves its partial sum of g^2
    float ss = 0.f;
    p8_tile<MRG_OP_KSTRIDED, MRG_OP_KSTRIDED, VEC, PHASES>(g, bid - ga.tile_start[p], ksplit, smem, &ss, &ga.af);
    if (g.epi_kind == 9) {
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) ga.ss[p][(long)(bid - ga.tile_start[p]) * 8 + (threadIdx.x >> 6)] = ss;
    }
}

}  // namespace

bool mrg_gemm8p_group_launch(const mrg_detail::P8GroupArgs& ga, bool vec, hipStream_t stream) {
    const int total = ga.tile_start[ga.nprob];
    if (total <= 0) return false;
    static const int phases = [] { const char* e = getenv("MRG_P8_PHASES"); return e ? atoi(e) : 2; }();
#define MRG_P8G(V, PH) do { const void* fn = (const void*)gemm_p8_group_kernel<V, PH>; \
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS); \
        hipLaunchKernelGGL((gemm_p8_group_kernel<V, PH>), dim3(total, ga.base.splitk), dim3(512), (size_t)P8_LDS, stream, ga); } while (0)
    if (vec) { if (phases == 4) MRG_P8G(true, 4); else MRG_P8G(true, 2); }
    else { if (phases == 4) MRG_P8G(false, 4); else MRG_P8G(false, 2); }
#undef MRG_P8G
    return true;
}

bool mrg_gemm8p_launch(const mrg_detail::GemmArgs& g, int a_mode, int b_mode, bool vec, int splitk, hipStream_t stream) {
    if (g.K % BK != 0 || g.a_map || g.b_map || g.m_dev || g.k_dev || g.colsum) return false;
    if (!g.a_aligned || !g.b_aligned) return false;
    if (a_mode == MRG_OP_KSTRIDED && (g.M % 8 != 0)) return false;
    if (b_mode == MRG_OP_KSTRIDED && (g.N % 8 != 0)) return false;
    const dim3 grid(cdiv(g.M, 256) * cdiv(g.N, 256), splitk), block(512);
    static const int phases = [] { const char* e = getenv("MRG_P8_PHASES"); return e ? atoi(e) : 2; }();   // phases per k-tile: 2 (default) or 4
#define MRG_P8_(AM, BM_, V, PH) do { const void* fn = (const void*)gemm_p8_kernel<AM, BM_, V, PH>; \
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS); \
        hipLaunchKernelGGL((gemm_p8_kernel<AM, BM_, V, PH>), grid, block, (size_t)P8_LDS, stream, g); } while (0)
#define MRG_P8(AM, BM_, V) do { if (phases == 4) MRG_P8_(AM, BM_, V, 4); else MRG_P8_(AM, BM_, V, 2); } while (0)
    if (a_mode == MRG_OP_KCONTIG && b_mode == MRG_OP_KCONTIG) { if (vec) MRG_P8(0, 0, true); else MRG_P8(0, 0, false); }
    else if (a_mode == MRG_OP_KCONTIG && b_mode == MRG_OP_KSTRIDED) { if (vec) MRG_P8(0, 1, true); else MRG_P8(0, 1, false); }
    else if (a_mode == MRG_OP_KSTRIDED && b_mode == MRG_OP_KSTRIDED) { if (vec) MRG_P8(1, 1, true); else MRG_P8(1, 1, false); }
    else return false;
#undef MRG_P8
#undef MRG_P8_
    return true;
}

int mrg_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
        return cus;
    }();
    return n;
}
