// Fused softmax attention, forward and backward, for the short sequences of the MIRAGE path
// (encoder N = K+1 in {99,129,257}, hd 64; decoder self N = 256 and cross 256 x N, hd 32).
// Replaces F.scaled_dot_product_attention (mirage/utils.py:181,216; scale = hd^-0.5, no mask, p = 0) and
// aten::_scaled_dot_product_*_backward.
//
// One workgroup (4 waves) per (batch, head).  The whole K/V of a head fits in LDS (<= 288 keys: 36 KiB each),
// so forward is "single KV tile": each wave owns 32-query tiles, computes S^T = K Q^T on MFMA 32x32x16 so that a
// lane owns one query column (row-reductions are in-lane + one cross-half exchange), keeps every key block's
// scores in registers, and feeds P straight back into the PV MFMAs as the A operand (the accumulator layout of
// S^T is exactly a k-permuted A-operand layout; V fragments come from LDS with ds_read_b64_tr_b16 using the same
// k permutation).  Longer sequences loop over KV chunks with an online-softmax rescale.
// Backward recomputes P from the saved log-sum-exp: a query-owner kernel produces dQ and a key-owner kernel
// produces dK, dV (no atomics, deterministic).
//
// All tensors are bf16 with explicit (batch, row, head) strides in elements so the kernels read q/k/v directly
// out of the fused qkv / kv projection buffers ([B, N, 3, H, hd] / [B, N, 2, H, hd]) and write the merged-head
// [B, N, H*hd] layout the proj GEMM consumes: the reference's reshape/permute/transpose copies do not exist.
#include "common.h"
#include "mirage_hip.h"
#include <stdlib.h>

namespace {

constexpr float LOG2E = 1.4426950408889634f;

enum { SWZ_ROW = 0, SWZ_TR = 1 };

// byte offset of 16-B chunk c of row r in a [rows][HD] bf16 LDS tile
template <int HD, int SWZ>
__device__ __forceinline__ int tile_off(int r, int c) {
    if (HD == 64) {
        if (SWZ == SWZ_ROW) return r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
        return r * 128 + ((c ^ ((r & 2) << 1)) << 4);
    } else {
        if (SWZ == SWZ_ROW) return r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
        return r * 64 + (c << 4);
    }
}

// cooperative global -> LDS copy of `rows` rows of two tensors (zero-filled beyond nvalid); src row strides in elements.
// Loads are issued in branch-free batches of 4 + 4 before the first LDS store so that the whole tile costs one or two memory
// round trips (a plain load -> store loop is compiled into one serialized round trip per 16 B).
template <int HD, int SWZ_A, int SWZ_B>
__device__ __forceinline__ void load_rows2(char* ldsA, const bf16_t* __restrict__ srcA, long strideA,
                                           char* ldsB, const bf16_t* __restrict__ srcB, long strideB,
                                           int rows, int nvalid, int tid, int nthreads) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int CPR = HD / 8;
    const int total = rows * CPR;
    for (int q0 = tid; q0 < total; q0 += 4 * nthreads) {
        const int qa = q0, qb = q0 + nthreads, qc = q0 + 2 * nthreads, qd = q0 + 3 * nthreads;
#define MRG_LD(Q, VA, VB) const int r##Q = Q / CPR, c##Q = Q % CPR; const bool ok##Q = Q < total && r##Q < nvalid; \
        const long o##Q = ok##Q ? (long)r##Q : 0l; \
        u32x4 VA = *reinterpret_cast<const u32x4*>(srcA + o##Q * strideA + c##Q * 8); \
        u32x4 VB = *reinterpret_cast<const u32x4*>(srcB + o##Q * strideB + c##Q * 8);
        MRG_LD(qa, a0, b0) MRG_LD(qb, a1, b1) MRG_LD(qc, a2, b2) MRG_LD(qd, a3, b3)
#undef MRG_LD
        const u32x4 z = {0u, 0u, 0u, 0u};
#define MRG_ST(Q, VA, VB) if (Q < total) { \
            *reinterpret_cast<u32x4*>(ldsA + tile_off<HD, SWZ_A>(r##Q, c##Q)) = ok##Q ? VA : z; \
            *reinterpret_cast<u32x4*>(ldsB + tile_off<HD, SWZ_B>(r##Q, c##Q)) = ok##Q ? VB : z; }
        MRG_ST(qa, a0, b0) MRG_ST(qb, a1, b1) MRG_ST(qc, a2, b2) MRG_ST(qd, a3, b3)
#undef MRG_ST
    }
}

// A/B operand fragment with the contraction along the row (hd) direction: lane -> row (l&31), chunk 2*kk + (l>>5)
template <int HD, int SWZ>
__device__ __forceinline__ bf16x8 frag_rows(const char* lds, int row0, int kk, int lane) {
    const int r = row0 + (lane & 31);
    return *reinterpret_cast<const bf16x8*>(lds + tile_off<HD, SWZ>(r, kk * 2 + (lane >> 5)));
}

// B operand fragment with the contraction along the tile's ROW direction (keys / queries), output column = hd index.
// k-slot order matches an accumulator-layout A operand: slots 0-3 = rows base+0..3, slots 4-7 = rows base+8..11 with
// base = row0 + 16*mf + 4*(lane>>5).
template <int HD, int SWZ>
__device__ __forceinline__ bf16x8 frag_cols_tr(uint32_t lds_base, int row0, int mf, int dj, int lane) {
    const int i = lane & 15;
    const int d = dj * 32 + (lane & 16) + ((i & 3) << 2);
    const int r0 = row0 + 16 * mf + 4 * (lane >> 5) + (i >> 2);
    const int r1 = r0 + 8;
    const s16x4 lo = lds_read_tr16(lds_base + tile_off<HD, SWZ>(r0, d >> 3) + ((d & 4) << 1));
    const s16x4 hi = lds_read_tr16(lds_base + tile_off<HD, SWZ>(r1, d >> 3) + ((d & 4) << 1));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// the same two with the row index clamped to `rmax` (the zero row of a compactly stored tile: rows N .. 32 * ceil(N / 32) - 1 of a
// padded tile do not exist in LDS, they all alias one zero row -- 132 instead of 160 rows at N = 129)
template <int HD, int SWZ>
__device__ __forceinline__ bf16x8 frag_rows_cl(const char* lds, int row0, int kk, int lane, int rmax) {
    const int r = min(row0 + (lane & 31), rmax);
    return *reinterpret_cast<const bf16x8*>(lds + tile_off<HD, SWZ>(r, kk * 2 + (lane >> 5)));
}
template <int HD, int SWZ>
__device__ __forceinline__ bf16x8 frag_cols_tr_cl(uint32_t lds_base, int row0, int mf, int dj, int lane, int rmax) {
    const int i = lane & 15;
    const int d = dj * 32 + (lane & 16) + ((i & 3) << 2);
    const int r0u = row0 + 16 * mf + 4 * (lane >> 5) + (i >> 2);
    const int r0 = min(r0u, rmax), r1 = min(r0u + 8, rmax);
    const s16x4 lo = lds_read_tr16(lds_base + tile_off<HD, SWZ>(r0, d >> 3) + ((d & 4) << 1));
    const s16x4 hi = lds_read_tr16(lds_base + tile_off<HD, SWZ>(r1, d >> 3) + ((d & 4) << 1));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int mf) {
    bf16x8 f;
    const int b = mf * 8;
    uint32_t w0 = pack_bf2(s[b + 0], s[b + 1]), w1 = pack_bf2(s[b + 2], s[b + 3]);
    uint32_t w2 = pack_bf2(s[b + 4], s[b + 5]), w3 = pack_bf2(s[b + 6], s[b + 7]);
    f[0] = (short)(w0 & 0xffff); f[1] = (short)(w0 >> 16); f[2] = (short)(w1 & 0xffff); f[3] = (short)(w1 >> 16);
    f[4] = (short)(w2 & 0xffff); f[5] = (short)(w2 >> 16); f[6] = (short)(w3 & 0xffff); f[7] = (short)(w3 >> 16);
    return f;
}

// row index (0..31) of accumulator register r for this lane
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v;
    bf16_t* o; float* lse;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;  // batch / row strides (elements); head stride = HD
    int B, H, Nq, Nk;
    float scale;
    // backward
    const bf16_t* d_o; long do_bs, do_rs;
    float* dsum;
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    long dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    int chunk_rows;
    int rmax;         // forward, resident K / V stored compactly: index of the zero row every out-of-range key row is clamped to (else INT_MAX)
    int stage_tail;   // forward: rows of the LAST wave's staging tile when it was allocated short (0: every wave has 32 rows)
    float* cs_q; float* cs_k; float* cs_v; long cs_bs;   // optional per-(batch, head) column sums of dQ / dK / dV (bias gradients)
    int tail_off;     // forward TAIL: byte offset (in the dynamic LDS) of the waves' exchange area for the last query row, [waves][66] floats
    int tail;         // forward, Nq == Nk == 32 m + 1 (the global token is appended LAST, mirage/model.py:391): the tiled kernel covers the 32 m
                      // leading tokens, folds the last KEY into every row's softmax state by hand and does the last QUERY row with plain FMAs at the end
};

// ---- transposed accumulators -> full-line stores (forward output, fused backward outputs) ----------------------------------
// An output accumulated with the MFMA operands swapped has lane = token row of the 32-row tile and register r = column
// 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of the 32-column block dj: per-row factors (softmax rescale, 1 / l) are lane-local, four
// consecutive columns pack into one 8-byte LDS store, and the wave's staging tile is read back as 16-B chunks so that a store
// instruction covers whole rows (8 lanes per 128-B row at hd 64).  Straight from the usual layout (lane = column) a lane could only
// store 2 bytes at a time.
constexpr int FUSED_SP = 144;                     // staging pitch (bytes) of a 64-column bf16 row: 16-B aligned, 36 banks
template <int DJ>
__device__ __forceinline__ void fused_stage(char* stg, const f32x16 (&acc)[DJ], int lane, int nrows = 32) {
    const int rr = lane & 31, hf = lane >> 5;
    if (rr >= nrows) return;                      // a short staging tile (the row tail of the last query tile) holds only nrows rows
#pragma unroll
    for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack_bf2(acc[dj][4 * g + 0], acc[dj][4 * g + 1]);
            w.y = pack_bf2(acc[dj][4 * g + 2], acc[dj][4 * g + 3]);
            *reinterpret_cast<uint2*>(stg + rr * FUSED_SP + (dj * 32 + 8 * g + 4 * hf) * 2) = w;
        }
}
// staging tile -> global rows row0 .. row0 + 31 (valid below N) as 16-B stores; CS: also this lane's partial column sums (columns
// 8 (lane & 7) .. + 7 over the rows it stored, reduced over the lanes that share the chunk) in cs[8] (hd 64 only)
template <int DJ, bool CS>
__device__ __forceinline__ void fused_flush(const char* stg, bf16_t* __restrict__ dst, long rs, int row0, int N, int lane, float (&cs)[8]) {
    constexpr int CPR = DJ * 4;                    // 16-B chunks per row: 8 (hd 64) or 4 (hd 32)
    constexpr int RPI = 64 / CPR;                  // rows per store instruction
    if (CS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 32 / RPI; ++j) {
        const int rr = lane / CPR + RPI * j, c = lane % CPR;
        if (row0 + rr < N) {
            const uint4 v = *reinterpret_cast<const uint4*>(stg + rr * FUSED_SP + c * 16);
            *reinterpret_cast<uint4*>(dst + (long)(row0 + rr) * rs + c * 8) = v;
            if (CS) {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[2 * e] += __uint_as_float(w[e] << 16);
                    cs[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                }
            }
        }
    }
    if (CS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cs[e] += __shfl_xor(cs[e], 8, 64);
            cs[e] += __shfl_xor(cs[e], 16, 64);
            cs[e] += __shfl_xor(cs[e], 32, 64);
        }
    }
}

// ================================================ forward ===============================================
// STAGE: the output tile goes through a per-wave LDS staging tile (behind K / V) and leaves as full rows; otherwise (K / V + staging
// would cost the second workgroup per CU) as 8-byte stores from the registers.
// TAIL (a.tail): sequence = 32 m tokens + one more (N = 129 / 257: the patches and the global token).  As a ninth tile that one token costs a
// whole second round of tiles and a padded key block per row (N = 257: 252 us against 170 us at N = 256).  Here the tiles cover the 32 m
// leading queries and keys; the last key enters every row's online-softmax state as its INITIAL value (score by a 64-term dot product per
// lane pair, weight exp(0) = 1, output = its value row); the last query's row is computed at the end by all waves together from the K / V
// rows already in LDS (wave w: keys of block w with lane = key for the scores and lane = head-dim column for P V; partial softmax states
// are merged by wave 0 through a small LDS area).
template <int HD, int CB, bool STAGE, bool TAIL = false>
__global__ __launch_bounds__(512, 4) void attn_fwd_kernel(const AttnArgs a0) {     // (HIP: second value = minimum waves per SIMD -> at most 128 VGPRs)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 16, DJ = HD / 32, ROWS = CB * 32;
    AttnArgs a = a0;
    if (TAIL) { a.Nq -= 1; a.Nk -= 1; }                   // the tiled part; strides, lse and output rows keep the full length below
    // a.chunk_rows = rows of K and of V resident in LDS: all keys (padded to 32) when they fit, so that longer rows stage K/V
    // once and the chunk loop below only walks LDS; otherwise ROWS and every chunk is re-staged
    const int lds_rows = a.chunk_rows;
    const bool resident = lds_rows >= a.Nk;
    char* ldsK = smem;
    char* ldsV = smem + lds_rows * HD * 2;
    const uint32_t ldsV_u = lds_addr(ldsV);
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nwav = nthr >> 6;
    const bf16_t* qp = a.q + (long)b * a.q_bs + h * HD;
    const bf16_t* kp = a.k + (long)b * a.k_bs + h * HD;
    const bf16_t* vp = a.v + (long)b * a.v_bs + h * HD;
    const int nqt = (a.Nq + 31) >> 5;
    const int nchunks = (a.Nk + ROWS - 1) / ROWS;
    const float sc = a.scale * LOG2E;
#ifdef MRG_ATTN_TS      // debug build: phase timestamps of this workgroup's wave 0 (100 MHz wall clock), written over its lse rows at the end
    unsigned long long fts0 = wall_clock64(), fts1 = 0, fts2 = 0, fts3 = 0;
#endif

    // TAIL: the last token's q, k and v rows (HD bf16 each, plain layout) in front of the exchange area -- read several times below, and a
    // global round trip in front of every tile and of the last query's row is what made the path lose at N = 129
    const bf16_t* tl = reinterpret_cast<const bf16_t*>(smem + (TAIL ? a.tail_off : 0));
    if (TAIL && tid < 3 * (HD / 8)) {
        const int rw = tid / (HD / 8), c = tid % (HD / 8);
        const bf16_t* src = rw == 0 ? qp + (long)a.Nq * a.q_rs : rw == 1 ? kp + (long)a.Nk * a.k_rs : vp + (long)a.Nk * a.v_rs;
        *reinterpret_cast<uint4*>(smem + a.tail_off + rw * HD * 2 + c * 16) = *reinterpret_cast<const uint4*>(src + c * 8);
    }
    if (resident) {
        load_rows2<HD, SWZ_ROW, SWZ_TR>(ldsK, kp, a.k_rs, ldsV, vp, a.v_rs, lds_rows, a.Nk, tid, nthr);
        __syncthreads();
    }
#ifdef MRG_ATTN_TS
    fts1 = wall_clock64();
#endif
    // uniform rounds so that every wave takes part in the staging barriers of the multi-chunk path
    const int rounds = (nqt + nwav - 1) / nwav;
    for (int rd = 0; rd < rounds; ++rd) {
        const int qt = rd * nwav + wave;
        const bool active = qt < nqt;
        const int q0 = qt * 32;
        bf16x8 qf[KS];
        {
            const int qr = q0 + (lane & 31);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (active && qr < a.Nq) v = *reinterpret_cast<const uint4*>(qp + (long)qr * a.q_rs + kk * 16 + 8 * (lane >> 5));
                qf[kk] = *reinterpret_cast<bf16x8*>(&v);
            }
        }
        f32x16 o[DJ];
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dj][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f, s_tail = 0.f;
        if (TAIL) {
            // the last key (row a.Nk of k / v): this lane pair's query . k_tail, each lane over the half of the head dimension it holds
            const bf16_t* kt = tl + HD;
            float part = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const uint4 kv4 = *reinterpret_cast<const uint4*>(kt + kk * 16 + 8 * (lane >> 5));
                const uint32_t kw[4] = {kv4.x, kv4.y, kv4.z, kv4.w};
                const uint32_t* qw = reinterpret_cast<const uint32_t*>(&qf[kk]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    part = fmaf(__uint_as_float(qw[e] << 16), __uint_as_float(kw[e] << 16), part);
                    part = fmaf(__uint_as_float(qw[e] & 0xffff0000u), __uint_as_float(kw[e] & 0xffff0000u), part);
                }
            }
            s_tail = part + __shfl_xor(part, 32, 64);
            m_run = s_tail;                                   // the running maximum knows the last key from the start; its weight and its value
        }                                                     // row are added after the tiles (fewer registers live across the chunk loop)

        for (int ch = 0; ch < nchunks; ++ch) {
            const int kbase = ch * ROWS;
            const int kvalid = min(ROWS, a.Nk - kbase);
            if (!resident) {
                __syncthreads();
                load_rows2<HD, SWZ_ROW, SWZ_TR>(ldsK, kp + (long)kbase * a.k_rs, a.k_rs, ldsV, vp + (long)kbase * a.v_rs, a.v_rs, ROWS, kvalid, tid, nthr);
                __syncthreads();
            }
            if (!active) continue;
            const int lbase = resident ? kbase : 0;      // first LDS row of this chunk
            f32x16 s[CB];
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < CB; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
                if (kb * 32 < kvalid) {                // rows past the last key alias the zero row of a compactly stored tile (a.rmax)
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk)
                        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_cl<HD, SWZ_ROW>(ldsK, lbase + kb * 32, kk, lane, a.rmax), qf[kk], s[kb], 0, 0, 0);
                }
                if (!TAIL && kb * 32 + 32 > kvalid) {   // only the last, partial key block needs the key-range test (TAIL: the tiles cover 32 m keys)
                    asm volatile("" ::: "memory");        // keeps this a BRANCH: if-converted it cost every block 58 compare / select instructions
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (kb * 32 + acc_row(r, lane) >= kvalid) s[kb][r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // Thresholded rescale: while no query of the wave sees its running maximum grow by more than 2^8 (in the exponent's units),
            // the stale maximum stays the reference point -- p <= 256 instead of <= 1, the normalisation by l at the end is the same --
            // and the block skips the rescale of the 32 output registers and its exponential: ~70 of the ~550 issue cycles of a block
            // step on a SIMD-issue-bound kernel (profiles/r05_attn_fwd.md).  The first block of a row always rescales (m_run = -inf).
            const bool grow = (fmaxf(m_run, mx) - m_run) * sc > 8.0f;         // -inf start: inf > 8
            const bool rescale = __builtin_amdgcn_ballot_w64(grow) != 0;
            const float m_new = rescale ? fmaxf(m_run, mx) : m_run;
            const float alpha = rescale ? fast_exp2((m_run - m_new) * sc) : 1.0f;   // first chunk: exp2(-inf) = 0
            float ls = 0.f;
            const float nms = -m_new * sc;               // p = 2^(s sc - m sc): one FMA per score instead of a subtract and a multiply
#pragma unroll
            for (int kb = 0; kb < CB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = fast_exp2(fmaf(s[kb][r], sc, nms));
                    s[kb][r] = p;
                    ls += p;
                }
            ls += __shfl_xor(ls, 32, 64);
            l_run = l_run * alpha + ls;
            m_run = m_new;
            if ((ch > 0 || TAIL) && rescale) {       // o is accumulated transposed (lane = query): the row factor is lane-local
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int dj = 0; dj < DJ; ++dj) o[dj][r] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < CB; ++kb) {
                __builtin_amdgcn_sched_barrier(0);   // keep the V fragments of later key blocks from being hoisted (VGPR budget)
                if (kb * 32 < kvalid) {
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) {
                        const bf16x8 pa = pack8(s[kb], mf);
#pragma unroll
                        for (int dj = 0; dj < DJ; ++dj)
                            o[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr_cl<HD, SWZ_TR>(ldsV_u, lbase + kb * 32, mf, dj, lane, a.rmax), pa, o[dj], 0, 0, 0);
                    }
                }
            }
        }
        if (!active) continue;
        if (TAIL) {                                           // m_run >= s_tail by construction
            const float w = fast_exp2((s_tail - m_run) * sc);
            l_run += w;
            const bf16_t* vt = tl + 2 * HD;
#pragma unroll
            for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {                 // registers 4 g .. 4 g + 3 = columns dj 32 + 8 g + 4 (lane >> 5) + 0 .. 3
                    const uint2 v2 = *reinterpret_cast<const uint2*>(vt + dj * 32 + 8 * g + 4 * (lane >> 5));
                    o[dj][4 * g + 0] = fmaf(w, __uint_as_float(v2.x << 16), o[dj][4 * g + 0]);
                    o[dj][4 * g + 1] = fmaf(w, __uint_as_float(v2.x & 0xffff0000u), o[dj][4 * g + 1]);
                    o[dj][4 * g + 2] = fmaf(w, __uint_as_float(v2.y << 16), o[dj][4 * g + 2]);
                    o[dj][4 * g + 3] = fmaf(w, __uint_as_float(v2.y & 0xffff0000u), o[dj][4 * g + 3]);
                }
        }
        const float linv = 1.f / l_run;
        bf16_t* op = a.o + (long)b * a.o_bs + h * HD;
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dj][r] *= linv;
        if (STAGE) {
            char* stg = smem + 2 * lds_rows * HD * 2 + wave * (32 * FUSED_SP);
            float unused[8];
            fused_stage<DJ>(stg, o, lane, (a.stage_tail && qt == nqt - 1) ? a.stage_tail : 32);
            fused_flush<DJ, false>(stg, op, a.o_rs, q0, a.Nq, lane, unused);
        } else {
            const int qr = q0 + (lane & 31);
            if (qr < a.Nq) {
#pragma unroll
                for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 w;
                        w.x = pack_bf2(o[dj][4 * g + 0], o[dj][4 * g + 1]);
                        w.y = pack_bf2(o[dj][4 * g + 2], o[dj][4 * g + 3]);
                        *reinterpret_cast<uint2*>(op + (long)qr * a.o_rs + dj * 32 + 8 * g + 4 * (lane >> 5)) = w;
                    }
            }
        }
        if (a.lse && lane < 32 && q0 + lane < a.Nq)
            a.lse[((long)b * a.H + h) * a0.Nq + q0 + lane] = m_run * a.scale + __logf(l_run);
    }
#ifdef MRG_ATTN_TS
    fts2 = wall_clock64();
#endif
    if (TAIL) {
        // ---- the last query row (index a.Nq), as a query tile with ONE valid row on the matrix units: every wave takes the key blocks kb = wave,
        // wave + nwav, ... with the tiles' own block step (S^T = K q^T -> online softmax -> P V, transposed accumulators: lanes 0 and 32 own the
        // row), wave 0 adds the last key, and the waves' partial softmax states are merged through LDS.  (Round 4 did this row with scalar
        // reads -- 32 readlane + LDS + FMA steps per key block -- which cost 4.3 us of a workgroup's 20 at N = 257, profiles/r05_attn_fwd.md.)
        float* xch = reinterpret_cast<float*>(smem + a.tail_off + 3 * HD * 2);
        const bf16_t* qt = tl;
        bf16x8 qf2[KS];
        int hf = lane >> 5;
        asm volatile("" : "+v"(hf));          // opaque: the loads below depend on it and cannot be hoisted above the tiles (the register budget is theirs)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((lane & 31) == 0) v = *reinterpret_cast<const uint4*>(qt + kk * 16 + 8 * hf);
            qf2[kk] = *reinterpret_cast<bf16x8*>(&v);
        }
        f32x16 ot[DJ];
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[dj][r] = 0.f;
        float mrun = -INFINITY, lrun = 0.f;
        const int nkb = a.Nk >> 5;
        for (int kb = wave; kb < nkb; kb += nwav) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_cl<HD, SWZ_ROW>(ldsK, kb * 32, kk, lane, a.rmax), qf2[kk], st, 0, 0, 0);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(mrun, mx);
            const float alpha = fast_exp2((mrun - mn) * sc);
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pp = fast_exp2((st[r] - mn) * sc);
                st[r] = pp;
                ls += pp;
            }
            ls += __shfl_xor(ls, 32, 64);
            lrun = lrun * alpha + ls;
            mrun = mn;
#pragma unroll
            for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[dj][r] *= alpha;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                const bf16x8 pa = pack8(st, mf);
#pragma unroll
                for (int dj = 0; dj < DJ; ++dj)
                    ot[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr_cl<HD, SWZ_TR>(ldsV_u, kb * 32, mf, dj, lane, a.rmax), pa, ot[dj], 0, 0, 0);
            }
        }
        if (wave == 0) {                                      // the last key, for the last query: one more single-key block (lanes 0 and 32 hold the row)
            const bf16_t* kt = tl + HD;
            float part = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const uint4 kv4 = *reinterpret_cast<const uint4*>(kt + kk * 16 + 8 * (lane >> 5));
                const uint32_t kw[4] = {kv4.x, kv4.y, kv4.z, kv4.w};
                const uint32_t* qw = reinterpret_cast<const uint32_t*>(&qf2[kk]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    part = fmaf(__uint_as_float(qw[e] << 16), __uint_as_float(kw[e] << 16), part);
                    part = fmaf(__uint_as_float(qw[e] & 0xffff0000u), __uint_as_float(kw[e] & 0xffff0000u), part);
                }
            }
            const float sj = part + __shfl_xor(part, 32, 64);
            const float mn = fmaxf(mrun, sj);
            const float alpha = fast_exp2((mrun - mn) * sc);
            const float pp = fast_exp2((sj - mn) * sc);
            lrun = lrun * alpha + pp;
            mrun = mn;
            const float pb = __uint_as_float(pack_bf2(pp, 0.f) << 16);      // P enters the P V product rounded to bf16, as in the tiles
            const bf16_t* vt = tl + 2 * HD;
#pragma unroll
            for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {                 // registers 4 g .. 4 g + 3 = columns dj 32 + 8 g + 4 (lane >> 5) + 0 .. 3
                    const uint2 v2 = *reinterpret_cast<const uint2*>(vt + dj * 32 + 8 * g + 4 * (lane >> 5));
                    ot[dj][4 * g + 0] = fmaf(pb, __uint_as_float(v2.x << 16), ot[dj][4 * g + 0] * alpha);
                    ot[dj][4 * g + 1] = fmaf(pb, __uint_as_float(v2.x & 0xffff0000u), ot[dj][4 * g + 1] * alpha);
                    ot[dj][4 * g + 2] = fmaf(pb, __uint_as_float(v2.y << 16), ot[dj][4 * g + 2] * alpha);
                    ot[dj][4 * g + 3] = fmaf(pb, __uint_as_float(v2.y & 0xffff0000u), ot[dj][4 * g + 3] * alpha);
                }
        }
        if ((lane & 31) == 0) {                               // lanes 0 and 32 hold the row: 32 of its 64 columns each
#pragma unroll
            for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xch[wave * 66 + dj * 32 + 8 * g + 4 * (lane >> 5) + e] = ot[dj][4 * g + e];
        }
        if (lane == 0) { xch[wave * 66 + 64] = mrun; xch[wave * 66 + 65] = lrun; }
        __syncthreads();
        if (wave == 0) {
            float M = -INFINITY;
            for (int w = 0; w < nwav; ++w) M = fmaxf(M, xch[w * 66 + 64]);
            float L = 0.f, O = 0.f;
            for (int w = 0; w < nwav; ++w) {
                const float f = fast_exp2((xch[w * 66 + 64] - M) * sc);      // a wave without keys has m = -inf: factor 0
                L = fmaf(xch[w * 66 + 65], f, L);
                O = fmaf(xch[w * 66 + lane], f, O);
            }
            a.o[(long)b * a.o_bs + (long)a.Nq * a.o_rs + h * HD + lane] = f2bf(O / L);
            if (a.lse && lane == 0) a.lse[((long)b * a.H + h) * a0.Nq + a.Nq] = M * a.scale + __logf(L);
        }
    }
#ifdef MRG_ATTN_TS
    __syncthreads();
    fts3 = wall_clock64();
    if (a.lse && tid == 0) {
        unsigned long long* tsp = reinterpret_cast<unsigned long long*>(a.lse + ((long)b * a.H + h) * a0.Nq);
        tsp[0] = fts0; tsp[1] = fts1; tsp[2] = fts2; tsp[3] = fts3;
    }
#endif
}

// ================================================ backward ==============================================
// query-owner pass: dQ (and dsum[b,h,q] = sum_d dO * O for the key-owner pass).  LDS: K chunk (row + tr reads), V chunk (row reads).
template <int HD>
__global__ __launch_bounds__(512, 3) void attn_bwd_dq_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 16, DJ = HD / 32;
    const int ROWS = a.chunk_rows;
    char* ldsK = smem;
    char* ldsV = smem + ROWS * HD * 2;
    const uint32_t ldsK_u = lds_addr(ldsK);
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nwav = nthr >> 6;
    const bf16_t* qp = a.q + (long)b * a.q_bs + h * HD;
    const bf16_t* kp = a.k + (long)b * a.k_bs + h * HD;
    const bf16_t* vp = a.v + (long)b * a.v_bs + h * HD;
    const bf16_t* dop = a.d_o + (long)b * a.do_bs + h * HD;
    const int nqt = (a.Nq + 31) >> 5;
    const int nchunks = (a.Nk + ROWS - 1) / ROWS;
    const int rounds = (nqt + nwav - 1) / nwav;
    const float sc = a.scale * LOG2E;
    if (nchunks == 1) {
        load_rows2<HD, SWZ_ROW, SWZ_ROW>(ldsK, kp, a.k_rs, ldsV, vp, a.v_rs, ROWS, a.Nk, tid, nthr);
        __syncthreads();
    }
    float csq[DJ];
#pragma unroll
    for (int dj = 0; dj < DJ; ++dj) csq[dj] = 0.f;
    for (int rd = 0; rd < rounds; ++rd) {
        const int qt = rd * nwav + wave;
        const bool active = qt < nqt;
        const int q0 = qt * 32;
        const int qr = q0 + (lane & 31);
        const bool qok = active && qr < a.Nq;
        bf16x8 qf[KS], dof[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            uint4 v = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0);
            if (qok) {
                v = *reinterpret_cast<const uint4*>(qp + (long)qr * a.q_rs + kk * 16 + 8 * (lane >> 5));
                w = *reinterpret_cast<const uint4*>(dop + (long)qr * a.do_rs + kk * 16 + 8 * (lane >> 5));
            }
            qf[kk] = *reinterpret_cast<bf16x8*>(&v);
            dof[kk] = *reinterpret_cast<bf16x8*>(&w);
        }
        const long sidx = ((long)b * a.H + h) * a.Nq + qr;
        const float lse2 = qok ? a.lse[sidx] * LOG2E : INFINITY;   // invalid query -> p = exp2(-inf) = 0
        // dsum[q] = sum_d dO[q,d] * O[q,d] (the softmax-backward row term): this lane holds half of the row's dO already, so the
        // dot product is formed here (and published for the key-owner kernel) instead of in a separate pass over O and dO
        float dsm = 0.f;
        if (qok) {
            const bf16_t* op = a.o + (long)b * a.o_bs + (long)qr * a.o_rs + h * HD + 8 * (lane >> 5);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const uint4 ov = *reinterpret_cast<const uint4*>(op + kk * 16);
                const uint32_t os[4] = {ov.x, ov.y, ov.z, ov.w};
                const uint32_t* ds = reinterpret_cast<const uint32_t*>(&dof[kk]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dsm += __uint_as_float(os[e] << 16) * __uint_as_float(ds[e] << 16);
                    dsm += __uint_as_float(os[e] & 0xffff0000u) * __uint_as_float(ds[e] & 0xffff0000u);
                }
            }
        }
        dsm += __shfl_xor(dsm, 32, 64);
        if (qok && lane < 32) a.dsum[sidx] = dsm;
        f32x16 dq[DJ];
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dj][r] = 0.f;
        for (int ch = 0; ch < nchunks; ++ch) {
            const int kbase = ch * ROWS;
            const int kvalid = min(ROWS, a.Nk - kbase);
            if (nchunks > 1) {
                __syncthreads();
                load_rows2<HD, SWZ_ROW, SWZ_ROW>(ldsK, kp + (long)kbase * a.k_rs, a.k_rs, ldsV, vp + (long)kbase * a.v_rs, a.v_rs, ROWS, kvalid, tid, nthr);
                __syncthreads();
            }
            if (!active) continue;
            const int nkb = (kvalid + 31) >> 5;
            for (int kb = 0; kb < nkb; ++kb) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD, SWZ_ROW>(ldsK, kb * 32, kk, lane), qf[kk], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD, SWZ_ROW>(ldsV, kb * 32, kk, lane), dof[kk], dp, 0, 0, 0);
                }
                if (kb * 32 + 32 <= kvalid) {        // full key block: no per-element key-range test (a third of the loop's VALU)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r] * sc - lse2) * (dp[r] - dsm) * a.scale;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb * 32 + acc_row(r, lane);
                        const float p = key < kvalid ? fast_exp2(s[r] * sc - lse2) : 0.f;
                        s[r] = p * (dp[r] - dsm) * a.scale;
                    }
                }
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    const bf16x8 da = pack8(s, mf);
#pragma unroll
                    for (int dj = 0; dj < DJ; ++dj)
                        dq[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, frag_cols_tr<HD, SWZ_ROW>(ldsK_u, kb * 32, mf, dj, lane), dq[dj], 0, 0, 0);
                }
            }
        }
        if (!active) continue;
        bf16_t* dqp = a.dq + (long)b * a.dq_bs + h * HD;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = q0 + acc_row(r, lane);
            if (row < a.Nq) {
#pragma unroll
                for (int dj = 0; dj < DJ; ++dj) {
                    dqp[(long)row * a.dq_rs + dj * 32 + (lane & 31)] = f2bf(dq[dj][r]);
                    csq[dj] += dq[dj][r];
                }
            }
        }
    }
    if (a.cs_q) {   // column sums of this (batch, head)'s dQ: the q-bias gradient before the reduction over the batch
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj) {
            const float v = csq[dj] + __shfl_xor(csq[dj], 32, 64);
            if (lane < 32) red[wave * HD + dj * 32 + lane] = v;
        }
        __syncthreads();
        for (int c = tid; c < HD; c += nthr) {
            float t = 0.f;
            for (int w = 0; w < nwav; ++w) t += red[w * HD + c];
            a.cs_q[(long)b * a.cs_bs + h * HD + c] = t;
        }
    }
}

// key-owner pass: dK, dV.  LDS: Q chunk and dO chunk (row + tr reads), lse*log2e and dsum per query of the chunk.
template <int HD>
__global__ __launch_bounds__(512, HD == 32 ? 4 : 2) void attn_bwd_dkv_kernel(const AttnArgs a) {   // (HIP: minimum waves per SIMD; 4 = at most 128 VGPRs: two 8-wave workgroups per CU at head_dim 32)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 16, DJ = HD / 32;
    const int ROWS = a.chunk_rows;
    char* ldsQ = smem;
    char* ldsDO = smem + ROWS * HD * 2;
    float* ldsL = reinterpret_cast<float*>(smem + 2 * ROWS * HD * 2);
    float* ldsD = ldsL + ROWS;
    const uint32_t ldsQ_u = lds_addr(ldsQ), ldsDO_u = lds_addr(ldsDO);
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nwav = nthr >> 6;
    const bf16_t* qp = a.q + (long)b * a.q_bs + h * HD;
    const bf16_t* kp = a.k + (long)b * a.k_bs + h * HD;
    const bf16_t* vp = a.v + (long)b * a.v_bs + h * HD;
    const bf16_t* dop = a.d_o + (long)b * a.do_bs + h * HD;
    const int nkt = (a.Nk + 31) >> 5;
    const int nchunks = (a.Nq + ROWS - 1) / ROWS;
    const int rounds = (nkt + nwav - 1) / nwav;
    const float sc = a.scale * LOG2E;
    float csk[DJ], csv[DJ];
#pragma unroll
    for (int dj = 0; dj < DJ; ++dj) { csk[dj] = 0.f; csv[dj] = 0.f; }
    for (int rd = 0; rd < rounds; ++rd) {
        const int kt = rd * nwav + wave;
        const bool active = kt < nkt;
        const int k0 = kt * 32;
        const int kr = k0 + (lane & 31);
        const bool kok = active && kr < a.Nk;
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            uint4 v = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0);
            if (kok) {
                v = *reinterpret_cast<const uint4*>(kp + (long)kr * a.k_rs + kk * 16 + 8 * (lane >> 5));
                w = *reinterpret_cast<const uint4*>(vp + (long)kr * a.v_rs + kk * 16 + 8 * (lane >> 5));
            }
            kf[kk] = *reinterpret_cast<bf16x8*>(&v);
            vf[kk] = *reinterpret_cast<bf16x8*>(&w);
        }
        f32x16 dk[DJ], dv[DJ];
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dj][r] = 0.f; dv[dj][r] = 0.f; }
        for (int ch = 0; ch < nchunks; ++ch) {
            const int qbase = ch * ROWS;
            const int qvalid = min(ROWS, a.Nq - qbase);
            if (nchunks > 1 || rd == 0) {
                __syncthreads();
                load_rows2<HD, SWZ_ROW, SWZ_ROW>(ldsQ, qp + (long)qbase * a.q_rs, a.q_rs, ldsDO, dop + (long)qbase * a.do_rs, a.do_rs, ROWS, qvalid, tid, nthr);
                for (int i = tid; i < ROWS; i += nthr) {
                    const long sidx = ((long)b * a.H + h) * a.Nq + qbase + i;
                    ldsL[i] = i < qvalid ? a.lse[sidx] * LOG2E : INFINITY;
                    ldsD[i] = i < qvalid ? a.dsum[sidx] : 0.f;
                }
                __syncthreads();
            }
            if (!active) continue;
            const int nqb = (qvalid + 31) >> 5;
            for (int qb = 0; qb < nqb; ++qb) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD, SWZ_ROW>(ldsQ, qb * 32, kk, lane), kf[kk], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD, SWZ_ROW>(ldsDO, qb * 32, kk, lane), vf[kk], dp, 0, 0, 0);
                }
                f32x16 pr;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int qrow = qb * 32 + 8 * g + 4 * (lane >> 5);
                    const float4 l4 = *reinterpret_cast<const float4*>(ldsL + qrow);
                    const float4 d4 = *reinterpret_cast<const float4*>(ldsD + qrow);
                    const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, ds[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = g * 4 + e;
                        const float p = fast_exp2(s[r] * sc - ls[e]);
                        pr[r] = p;
                        s[r] = p * (dp[r] - ds[e]) * a.scale;
                    }
                }
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    const bf16x8 pa = pack8(pr, mf);
                    const bf16x8 da = pack8(s, mf);
#pragma unroll
                    for (int dj = 0; dj < DJ; ++dj) {
                        dv[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_cols_tr<HD, SWZ_ROW>(ldsDO_u, qb * 32, mf, dj, lane), dv[dj], 0, 0, 0);
                        dk[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, frag_cols_tr<HD, SWZ_ROW>(ldsQ_u, qb * 32, mf, dj, lane), dk[dj], 0, 0, 0);
                    }
                }
            }
        }
        if (!active) continue;
        bf16_t* dkp = a.dk + (long)b * a.dk_bs + h * HD;
        bf16_t* dvp = a.dv + (long)b * a.dv_bs + h * HD;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = k0 + acc_row(r, lane);
            if (row < a.Nk) {
#pragma unroll
                for (int dj = 0; dj < DJ; ++dj) {
                    dkp[(long)row * a.dk_rs + dj * 32 + (lane & 31)] = f2bf(dk[dj][r]);
                    dvp[(long)row * a.dv_rs + dj * 32 + (lane & 31)] = f2bf(dv[dj][r]);
                    csk[dj] += dk[dj][r];
                    csv[dj] += dv[dj][r];
                }
            }
        }
    }
    if (a.cs_k) {   // column sums of this (batch, head)'s dK and dV (valid key rows only): k / v bias gradients before the batch sum
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj) {
            const float vk = csk[dj] + __shfl_xor(csk[dj], 32, 64);
            const float vv = csv[dj] + __shfl_xor(csv[dj], 32, 64);
            if (lane < 32) { red[wave * 2 * HD + dj * 32 + lane] = vk; red[wave * 2 * HD + HD + dj * 32 + lane] = vv; }
        }
        __syncthreads();
        for (int c = tid; c < 2 * HD; c += nthr) {
            float t = 0.f;
            for (int w = 0; w < nwav; ++w) t += red[w * 2 * HD + c];
            float* dst = c < HD ? a.cs_k : a.cs_v;
            dst[(long)b * a.cs_bs + h * HD + (c < HD ? c : c - HD)] = t;
        }
    }
}

// ------------------------------------------------ fused backward for short self-attention rows ----------------------------
// Nq == Nk in (96, 160], hd 64 (the encoder rows of the MIRAGE path: 99 and 129 tokens): ONE launch produces dQ, dK and dV.
// One workgroup per (batch, head), one wave per 32-row tile.  The scores of a (query block, key tile) pair are computed once:
//   phase A (wave = key tile):  S, dP -> P, dS (accumulator layout: lane = key, registers = queries);  dV^T += dO^T P and
//                               dK^T += Q^T dS straight from the registers;  dS also goes to LDS as packed bf16,
//                               T[query block][key][32 queries] (8-byte stores, XOR-swizzled: stores and transposed reads are
//                               conflict free).  Two query blocks are in flight per wave (four independent MFMA chains; the
//                               exponentials of one block run under the matrix instructions of the other);
//   phase B (wave = query tile): dQ^T += K^T dS^T with both operands by ds_read_b64_tr_b16 (T gives lane = query, k-slots = keys
//                               in the same permutation frag_cols_tr uses for K).
// The three outputs are accumulated TRANSPOSED (lane = token row, registers = 4 consecutive head-dim columns), packed to bf16,
// staged per wave in LDS and written as full 128-B lines; the bias-gradient column sums are formed in that pass.
// LDS holds only NR = N + 1 (rounded to 4) rows per tile: row NR - 1 is a zero row that every out-of-range row index is clamped
// to (N = 99: 78 KiB, two workgroups per CU; N = 129: 111 KiB, one).
// Against the query-owner + key-owner pair of kernels: 5 instead of 7 tile products, Q / K / V / dO are read once, one launch.
template <int NT> struct FusedCfg { static constexpr int ROWS = NT * 32; };
static inline int fused_nr(int N, int rows) { const int nr = (N + 1 + 3) & ~3; return nr < rows ? nr : rows; }
static inline size_t fused_lds(int nr, int nt) { return (size_t)4 * nr * 128 + (size_t)nt * nr * 64 + (size_t)2 * nt * 32 * 4 + 32 * 64; }

// byte offset inside a T block of the 8-byte chunk `c` (4 queries, c in 0..7) of key row `r`.  The row pitch of 64 B is 16 banks
// and the chunk index is XORed with (r >> 2): the 32 lanes of an 8-byte store (32 keys, one chunk) and of a transposed read
// (4 keys x 8 chunks) each touch 64 distinct banks.
__device__ __forceinline__ int t_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 7)) << 3); }

// fragment reads with the row index clamped to the zero row (CL: only the last 32-row tile can reach beyond NR)
__device__ __forceinline__ bf16x8 fz_rows(const bool CL, const char* lds, int row0, int kk, int lane, int nr1) {
    int r = row0 + (lane & 31);
    if (CL) r = min(r, nr1);
    return *reinterpret_cast<const bf16x8*>(lds + tile_off<64, SWZ_ROW>(r, kk * 2 + (lane >> 5)));
}
__device__ __forceinline__ bf16x8 fz_cols_tr(const bool CL, uint32_t lds_base, int row0, int mf, int dj, int lane, int nr1) {
    const int i = lane & 15;
    const int d = dj * 32 + (lane & 16) + ((i & 3) << 2);
    int r0 = row0 + 16 * mf + 4 * (lane >> 5) + (i >> 2);
    int r1 = r0 + 8;
    if (CL) { r0 = min(r0, nr1); r1 = min(r1, nr1); }
    const s16x4 lo = lds_read_tr16(lds_base + tile_off<64, SWZ_ROW>(r0, d >> 3) + ((d & 4) << 1));
    const s16x4 hi = lds_read_tr16(lds_base + tile_off<64, SWZ_ROW>(r1, d >> 3) + ((d & 4) << 1));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

template <int NT>
__global__ __launch_bounds__(64 * NT, 2) void attn_bwd_fused_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HD = 64, KS = 4, DJ = 2, ROWS = NT * 32, NTHR = 64 * NT;
    const int NR = a.chunk_rows, nr1 = NR - 1;
    const int TB = NR * 64;                            // bytes of one T block: [NR keys][32 queries] bf16
    char* ldsQ = smem;
    char* ldsDO = smem + NR * 128;
    char* ldsK = smem + 2 * NR * 128;
    char* ldsV = smem + 3 * NR * 128;
    char* ldsT = smem + 4 * NR * 128;
    float* ldsL = reinterpret_cast<float*>(ldsT + NT * TB);
    float* ldsD = ldsL + ROWS;
    const uint32_t ldsQ_u = lds_addr(ldsQ), ldsDO_u = lds_addr(ldsDO), ldsK_u = lds_addr(ldsK), ldsT_u = lds_addr(ldsT);
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
    const int N = a.Nq;
    const float sc = a.scale * LOG2E;
    const int row = wave * 32 + (lane & 31);          // this lane's key row in phase A
#ifdef MRG_ATTN_TS      // debug build: phase timestamps of every workgroup (100 MHz wall clock) into the unused dsum workspace
    unsigned long long* ts = reinterpret_cast<unsigned long long*>(a.dsum) + ((long)b * a.H + h) * 8;
#define MRG_TS(i) do { if (tid == 0) ts[i] = wall_clock64(); } while (0)
#else
#define MRG_TS(i) do { } while (0)
#endif
    MRG_TS(0);

    // ---- staging: Q, dO, K, V rows [0, NR) (zero beyond N) with 16-B loads, 8 lanes per 128-B row; dsum[q] = sum_d dO * O on the way
    {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const bf16_t* qp = a.q + (long)b * a.q_bs + h * HD;
        const bf16_t* kp = a.k + (long)b * a.k_bs + h * HD;
        const bf16_t* vp = a.v + (long)b * a.v_bs + h * HD;
        const bf16_t* dop = a.d_o + (long)b * a.do_bs + h * HD;
        const bf16_t* op = a.o + (long)b * a.o_bs + h * HD;
        // log-sum-exp of this thread's row (ROWS <= NTHR): requested with the tile loads, consumed after them
        const float lse_v = (tid < N) ? a.lse[((long)b * a.H + h) * N + tid] : INFINITY;     // invalid query -> p = exp2(-inf) = 0
        const int total = NR * 8;
        constexpr int PER = (ROWS * 8 + NTHR - 1) / NTHR;      // 4
        u32x4 vq[PER], vdo[PER], vk[PER], vv[PER], vo[PER];
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int q = tid + j * NTHR, r = q >> 3, c = q & 7;
            const bool ok = q < total && r < N;
            const long ro = ok ? (long)r : 0l;
            vq[j] = *reinterpret_cast<const u32x4*>(qp + ro * a.q_rs + c * 8);
            vdo[j] = *reinterpret_cast<const u32x4*>(dop + ro * a.do_rs + c * 8);
            vk[j] = *reinterpret_cast<const u32x4*>(kp + ro * a.k_rs + c * 8);
            vv[j] = *reinterpret_cast<const u32x4*>(vp + ro * a.v_rs + c * 8);
            vo[j] = *reinterpret_cast<const u32x4*>(op + ro * a.o_rs + c * 8);
        }
        MRG_TS(1);
        if (tid < ROWS) {
            ldsL[tid] = lse_v * LOG2E;
            if (tid >= N) ldsD[tid] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int q = tid + j * NTHR, r = q >> 3, c = q & 7;
            const bool ok = q < total && r < N;
            float dsm = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dsm += __uint_as_float(vo[j][e] << 16) * __uint_as_float(vdo[j][e] << 16);
                dsm += __uint_as_float(vo[j][e] & 0xffff0000u) * __uint_as_float(vdo[j][e] & 0xffff0000u);
            }
            dsm += __shfl_xor(dsm, 1, 64);
            dsm += __shfl_xor(dsm, 2, 64);
            dsm += __shfl_xor(dsm, 4, 64);
            if (q < total) {
                const int off = tile_off<HD, SWZ_ROW>(r, c);
                *reinterpret_cast<u32x4*>(ldsQ + off) = ok ? vq[j] : z;
                *reinterpret_cast<u32x4*>(ldsDO + off) = ok ? vdo[j] : z;
                *reinterpret_cast<u32x4*>(ldsK + off) = ok ? vk[j] : z;
                *reinterpret_cast<u32x4*>(ldsV + off) = ok ? vv[j] : z;
                if (ok && c == 0) ldsD[r] = dsm;
            }
        }
    }
    __syncthreads();
    MRG_TS(2);

    // ---- phase A: this wave's key tile against every query block, two blocks at a time
    f32x16 dvT[DJ], dkT[DJ];
#pragma unroll
    for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dvT[dj][r] = 0.f; dkT[dj][r] = 0.f; }
    {
        constexpr bool WL = true;                      // this wave's own tile may be the last one: clamp (uniform per wave)
        // dS store of this lane's key row: rows beyond NR go to a dump row of their own (branch-free: the stores stay inside the
        // scheduling region of the matrix instructions)
        char* tw = row < NR ? ldsT + row * 64 : reinterpret_cast<char*>(ldsD + ROWS) + (lane & 31) * 64;
        const int twq = row < NR ? TB : 0;
        const int tws = (row >> 2) & 7;
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            kf[kk] = fz_rows(WL, ldsK, wave * 32, kk, lane, nr1);
            vf[kk] = fz_rows(WL, ldsV, wave * 32, kk, lane, nr1);
        }
#define MRG_SDP(QB, S, DP) { \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) { S[r] = 0.f; DP[r] = 0.f; } \
            _Pragma("unroll") for (int kk = 0; kk < KS; ++kk) { \
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_rows((QB) == NT - 1, ldsQ, (QB) * 32, kk, lane, nr1), kf[kk], S, 0, 0, 0); \
                DP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_rows((QB) == NT - 1, ldsDO, (QB) * 32, kk, lane, nr1), vf[kk], DP, 0, 0, 0); } }
#define MRG_TAIL(QB, S, DP) { \
            _Pragma("unroll") for (int mf = 0; mf < 2; ++mf) { \
                f32x16 pr; \
                _Pragma("unroll") for (int g = 2 * mf; g < 2 * mf + 2; ++g) { \
                    const int qrow = (QB) * 32 + 8 * g + 4 * hf; \
                    const float4 l4 = *reinterpret_cast<const float4*>(ldsL + qrow); \
                    const float4 d4 = *reinterpret_cast<const float4*>(ldsD + qrow); \
                    const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, ds[4] = {d4.x, d4.y, d4.z, d4.w}; \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) { \
                        const int r = g * 4 + e; \
                        const float p = fast_exp2(S[r] * sc - ls[e]); \
                        pr[r] = p; \
                        S[r] = p * (DP[r] - ds[e]) * a.scale; } } \
                const bf16x8 pa = pack8(pr, mf); \
                const bf16x8 da = pack8(S, mf); \
                { \
                    const uint2* w2 = reinterpret_cast<const uint2*>(&da); \
                    _Pragma("unroll") for (int x = 0; x < 2; ++x) \
                        *reinterpret_cast<uint2*>(tw + (QB) * twq + (((4 * mf + 2 * x + hf) ^ tws) << 3)) = w2[x]; } \
                _Pragma("unroll") for (int dj = 0; dj < DJ; ++dj) { \
                    dvT[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_cols_tr((QB) == NT - 1, ldsDO_u, (QB) * 32, mf, dj, lane, nr1), pa, dvT[dj], 0, 0, 0); \
                    dkT[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_cols_tr((QB) == NT - 1, ldsQ_u, (QB) * 32, mf, dj, lane, nr1), da, dkT[dj], 0, 0, 0); } } }
#pragma unroll
        for (int qp2 = 0; qp2 < NT / 2; ++qp2) {
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s0, dp0, s1, dp1;
            MRG_SDP(2 * qp2, s0, dp0)
            MRG_SDP(2 * qp2 + 1, s1, dp1)
            MRG_TAIL(2 * qp2, s0, dp0)
            MRG_TAIL(2 * qp2 + 1, s1, dp1)
        }
        if (NT & 1) {
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s0, dp0;
            MRG_SDP(NT - 1, s0, dp0)
            MRG_TAIL(NT - 1, s0, dp0)
        }
#undef MRG_SDP
#undef MRG_TAIL
    }
    MRG_TS(3);
    __syncthreads();                                   // T complete; Q, dO, V are dead: their area becomes the per-wave staging tiles
    MRG_TS(4);
    char* stg = smem + wave * (32 * FUSED_SP);
    float cs[3][8];
    fused_stage<2>(stg, dvT, lane);
    fused_flush<2, true>(stg, a.dv + (long)b * a.dv_bs + h * HD, a.dv_rs, wave * 32, N, lane, cs[2]);
    fused_stage<2>(stg, dkT, lane);
    fused_flush<2, true>(stg, a.dk + (long)b * a.dk_bs + h * HD, a.dk_rs, wave * 32, N, lane, cs[1]);
    MRG_TS(5);

    // ---- phase B: this wave's query tile (T block `wave`), contraction over all keys
    {
        f32x16 dqT[DJ];
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqT[dj][r] = 0.f;
        const int i16 = lane & 15;
        const int chunk = ((lane & 16) >> 2) + (i16 & 3);
        const uint32_t tb = ldsT_u + wave * TB;
#pragma unroll
        for (int kb = 0; kb < NT; ++kb)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                int r0 = kb * 32 + 16 * mf + 4 * hf + (i16 >> 2);
                int r1 = r0 + 8;
                if (kb == NT - 1) { r0 = min(r0, nr1); r1 = min(r1, nr1); }
                const s16x4 lo = lds_read_tr16(tb + t_off(r0, chunk));
                const s16x4 hi = lds_read_tr16(tb + t_off(r1, chunk));
                bf16x8 da;
                da[0] = lo[0]; da[1] = lo[1]; da[2] = lo[2]; da[3] = lo[3];
                da[4] = hi[0]; da[5] = hi[1]; da[6] = hi[2]; da[7] = hi[3];
#pragma unroll
                for (int dj = 0; dj < DJ; ++dj)
                    dqT[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_cols_tr(kb == NT - 1, ldsK_u, kb * 32, mf, dj, lane, nr1), da, dqT[dj], 0, 0, 0);
            }
        MRG_TS(6);
        fused_stage<2>(stg, dqT, lane);
        fused_flush<2, true>(stg, a.dq + (long)b * a.dq_bs + h * HD, a.dq_rs, wave * 32, N, lane, cs[0]);
    }
    MRG_TS(7);
    if (a.cs_q) {   // per-(batch, head) column sums of dQ / dK / dV: the bias gradients before the reduction over the batch
        __syncthreads();                               // every wave is done with K and T
        float* red = reinterpret_cast<float*>(ldsK);
        if (lane < 8) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) red[(wave * 3 + m) * HD + lane * 8 + e] = cs[m][e];
        }
        __syncthreads();
        for (int c = tid; c < 3 * HD; c += NTHR) {
            float t = 0.f;
            for (int w = 0; w < NT; ++w) t += red[w * 3 * HD + c];
            float* dst = c < HD ? a.cs_q : (c < 2 * HD ? a.cs_k : a.cs_v);
            dst[(long)b * a.cs_bs + h * HD + (c % HD)] = t;
        }
    }
}

template <int NT>
int launch_bwd_fused(const AttnArgs& a0, hipStream_t stream) {
    AttnArgs a = a0;
    a.chunk_rows = fused_nr(a.Nq, NT * 32);
    const size_t sh = fused_lds(a.chunk_rows, NT);
    static size_t attr_sh = 0;
    if (sh > 65536 && sh > attr_sh) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        attr_sh = sh;
    }
    hipLaunchKernelGGL(attn_bwd_fused_kernel<NT>, dim3(a.H, a.B), dim3(64 * NT), sh, stream, a);
    return mrg_check_launch("mrg_attn_bwd/fused");
}

// ------------------------------------------------ fused backward, N = 32 NT + 1: the patches and the global token ---------------
// The MIRAGE encoder rows are 32 m kept patches + ONE global token appended last (mirage/model.py:391): as a fifth 32-row tile that token
// costs the general kernel a fifth wave (whose tile holds one row), 9 of 25 padded tile products, 111 KiB of LDS and with them the second
// workgroup of the CU (five-wave workgroups do not pair up even at 80 KiB: the dispatcher wants waves 0 and 4 of both on the same SIMD,
// i.e. <= 128 registers -- measured, profiles/r05_attn_bwd.md).  Here the NT tiles cover the patches only and the global token's row and
// column are plain vector arithmetic in fp32:
//   column (every query against the global token's key): one dot-product pair per query, 2 lanes each -> p, ds per query (LDS); their
//       contribution to dQ is a rank-1 update of the finished accumulators, dK[g] / dV[g] are vector-matrix products with lane = column;
//   row (the global token's query against every key): each key lane forms its scores from the K / V fragments it holds for phase A,
//       adds p dO[g] / ds Q[g] to its dV / dK accumulators, and dQ[g] is a vector-matrix product against K once K is in LDS.
// K and V never meet Q and dO in LDS (the owner waves hold them in registers through phase A and write K out for phase B): 73 KiB at N = 129,
// four waves -> two workgroups per CU, whose memory and matrix phases overlap.
template <int NT>
__global__ __launch_bounds__(64 * NT, 2) void attn_bwd_fused_gt_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HD = 64, KS = 4, DJ = 2, NM = NT * 32, NTHR = 64 * NT, G = NM;
    constexpr int TB = NM * 64;                        // bytes of one T block: [NM keys][32 queries] bf16
    constexpr int REG_A = 2 * NM * 128, REG_B = NM * 128 + NT * 32 * FUSED_SP, REG = REG_A > REG_B ? REG_A : REG_B;
    char* ldsT = smem;
    char* reg = smem + NT * TB;                        // phase A: Q, dO; afterwards: K, the per-wave staging tiles
    char* ldsQ = reg;
    char* ldsDO = reg + NM * 128;
    char* ldsK = reg;
    float* ldsL = reinterpret_cast<float*>(reg + REG); // log-sum-exp (x log2 e) and dsum per query, [NM + 4] each
    float* ldsD = ldsL + (NM + 4);
    float* ldsPG = ldsD + (NM + 4);                    // p and ds of (query, global key)
    float* ldsDSG = ldsPG + (NM + 4);
    float* ldsDSQ = ldsDSG + (NM + 4);                 // ds of (global query, key), [NM]
    float* kg = ldsDSQ + NM;                           // the global token's K, V, Q, dO rows in fp32
    float* vg = kg + HD;
    float* qg = vg + HD;
    float* dog = qg + HD;
    float* redKV = dog + HD;                           // [NT][2][HD] partial dK[g] / dV[g]
    float* redQ = redKV + NT * 2 * HD;                 // [NT][HD] partial dQ[g]
    const uint32_t ldsQ_u = lds_addr(ldsQ), ldsDO_u = lds_addr(ldsDO), ldsK_u = lds_addr(ldsK), ldsT_u = lds_addr(ldsT);
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
    const float sc = a.scale * LOG2E;
    const int row = wave * 32 + (lane & 31);           // this lane's key row in phase A, its query row in phase B
#ifdef MRG_ATTN_TS      // debug build: phase timestamps of every workgroup (100 MHz wall clock) into the unused dsum workspace
    unsigned long long* ts = reinterpret_cast<unsigned long long*>(a.dsum) + ((long)b * a.H + h) * 8;
#define MRG_TS(i) do { if (tid == 0) ts[i] = wall_clock64(); } while (0)
#else
#define MRG_TS(i) do { } while (0)
#endif
    MRG_TS(0);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto lo = [](uint32_t w) { return __uint_as_float(w << 16); };
    auto hi = [](uint32_t w) { return __uint_as_float(w & 0xffff0000u); };
    // dot product of 8 packed bf16 with 8 floats
    auto dot8 = [&](const uint32_t* w, const float* f) {
        const float4 f0 = *reinterpret_cast<const float4*>(f), f1 = *reinterpret_cast<const float4*>(f + 4);
        return ((lo(w[0]) * f0.x + hi(w[0]) * f0.y) + (lo(w[1]) * f0.z + hi(w[1]) * f0.w)) + ((lo(w[2]) * f1.x + hi(w[2]) * f1.y) + (lo(w[3]) * f1.z + hi(w[3]) * f1.w));
    };

    // ---- this wave's key tile: K / V fragments (lane = key row, 8 consecutive columns per k-slot group) straight from global memory
    bf16x8 kf[KS], vf[KS];
    {
        const bf16_t* kp = a.k + (long)b * a.k_bs + h * HD + (long)row * a.k_rs;
        const bf16_t* vp = a.v + (long)b * a.v_bs + h * HD + (long)row * a.v_rs;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            kf[kk] = *reinterpret_cast<const bf16x8*>(kp + (kk * 2 + hf) * 8);
            vf[kk] = *reinterpret_cast<const bf16x8*>(vp + (kk * 2 + hf) * 8);
        }
    }
    // ---- staging: Q, dO rows [0, NM) with 16-B loads, 8 lanes per 128-B row; dsum[q] = sum_d dO * O on the way; the global token's rows -> fp32
    {
        const bf16_t* qp = a.q + (long)b * a.q_bs + h * HD;
        const bf16_t* dop = a.d_o + (long)b * a.do_bs + h * HD;
        const bf16_t* op = a.o + (long)b * a.o_bs + h * HD;
        const float lse_v = (tid <= NM) ? a.lse[((long)b * a.H + h) * (NM + 1) + tid] : 0.f;
        u32x4 vq[4], vdo[4], vo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = tid + j * NTHR, r = q >> 3, c = q & 7;          // 4 NTHR / 8 = NM rows exactly
            vq[j] = *reinterpret_cast<const u32x4*>(qp + (long)r * a.q_rs + c * 8);
            vdo[j] = *reinterpret_cast<const u32x4*>(dop + (long)r * a.do_rs + c * 8);
            vo[j] = *reinterpret_cast<const u32x4*>(op + (long)r * a.o_rs + c * 8);
        }
        u32x4 gq = {0u, 0u, 0u, 0u}, gdo = gq, go = gq, gk = gq, gv = gq;
        if (tid < 8) {
            gq = *reinterpret_cast<const u32x4*>(qp + (long)G * a.q_rs + tid * 8);
            gdo = *reinterpret_cast<const u32x4*>(dop + (long)G * a.do_rs + tid * 8);
            go = *reinterpret_cast<const u32x4*>(op + (long)G * a.o_rs + tid * 8);
            gk = *reinterpret_cast<const u32x4*>(a.k + (long)b * a.k_bs + h * HD + (long)G * a.k_rs + tid * 8);
            gv = *reinterpret_cast<const u32x4*>(a.v + (long)b * a.v_bs + h * HD + (long)G * a.v_rs + tid * 8);
        }
        MRG_TS(1);
        if (tid <= NM) ldsL[tid] = lse_v * LOG2E;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = tid + j * NTHR, r = q >> 3, c = q & 7;
            float dsm = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) dsm += lo(vo[j][e]) * lo(vdo[j][e]) + hi(vo[j][e]) * hi(vdo[j][e]);
            dsm += __shfl_xor(dsm, 1, 64);
            dsm += __shfl_xor(dsm, 2, 64);
            dsm += __shfl_xor(dsm, 4, 64);
            const int off = tile_off<HD, SWZ_ROW>(r, c);
            *reinterpret_cast<u32x4*>(ldsQ + off) = vq[j];
            *reinterpret_cast<u32x4*>(ldsDO + off) = vdo[j];
            if (c == 0) ldsD[r] = dsm;
        }
        {
            float dsm = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) dsm += lo(go[e]) * lo(gdo[e]) + hi(go[e]) * hi(gdo[e]);
            dsm += __shfl_xor(dsm, 1, 64);
            dsm += __shfl_xor(dsm, 2, 64);
            dsm += __shfl_xor(dsm, 4, 64);
            if (tid < 8) {
                if (tid == 0) ldsD[G] = dsm;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    qg[tid * 8 + 2 * e] = lo(gq[e]); qg[tid * 8 + 2 * e + 1] = hi(gq[e]);
                    dog[tid * 8 + 2 * e] = lo(gdo[e]); dog[tid * 8 + 2 * e + 1] = hi(gdo[e]);
                    kg[tid * 8 + 2 * e] = lo(gk[e]); kg[tid * 8 + 2 * e + 1] = hi(gk[e]);
                    vg[tid * 8 + 2 * e] = lo(gv[e]); vg[tid * 8 + 2 * e + 1] = hi(gv[e]);
                }
            }
        }
    }
    __syncthreads();
    MRG_TS(2);

    // ---- the global token's KEY against every query: two lanes per query
    {
        const int q = tid >> 1, half = tid & 1;        // NTHR / 2 = NM queries
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int c = half * 4 + c4;
            const u32x4 qv = *reinterpret_cast<const u32x4*>(ldsQ + tile_off<HD, SWZ_ROW>(q, c));
            const u32x4 dv = *reinterpret_cast<const u32x4*>(ldsDO + tile_off<HD, SWZ_ROW>(q, c));
            const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
            s += dot8(qw, kg + c * 8);
            dp += dot8(dw, vg + c * 8);
        }
        s += __shfl_xor(s, 1, 64);
        dp += __shfl_xor(dp, 1, 64);
        const float p = fast_exp2(s * sc - ldsL[q]);
        if (half == 0) { ldsPG[q] = p; ldsDSG[q] = p * (dp - ldsD[q]) * a.scale; }
        if (wave == 0) {                               // (global query, global key)
            const float sg = wave_sum(qg[lane] * kg[lane]), dpg = wave_sum(dog[lane] * vg[lane]);
            const float pg = fast_exp2(sg * sc - ldsL[G]);
            if (lane == 0) { ldsPG[G] = pg; ldsDSG[G] = pg * (dpg - ldsD[G]) * a.scale; }
        }
    }
    __syncthreads();
    // dK[g] = sum_q ds[q, g] Q[q], dV[g] = sum_q p[q, g] dO[q]: lane = column, this wave's 32 queries
    {
        float ak = 0.f, av = 0.f;
        const int cb = tile_off<HD, SWZ_ROW>(0, 0);
        (void)cb;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int q = wave * 32 + i;
            const int off = tile_off<HD, SWZ_ROW>(q, lane >> 3) + (lane & 7) * 2;
            ak += ldsDSG[q] * bf2f(*reinterpret_cast<const bf16_t*>(ldsQ + off));
            av += ldsPG[q] * bf2f(*reinterpret_cast<const bf16_t*>(ldsDO + off));
        }
        if (wave == 0) { ak += ldsDSG[G] * qg[lane]; av += ldsPG[G] * dog[lane]; }
        redKV[(wave * 2 + 0) * HD + lane] = ak;
        redKV[(wave * 2 + 1) * HD + lane] = av;
    }

    // ---- phase A: this wave's key tile against every query block, two blocks at a time (attn_bwd_fused_kernel's, without row clamps)
    f32x16 dvT[DJ], dkT[DJ];
#pragma unroll
    for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dvT[dj][r] = 0.f; dkT[dj][r] = 0.f; }
    {
        char* tw = ldsT + row * 64;
        const int tws = (row >> 2) & 7;
#define MRG_SDP(QB, S, DP) { \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) { S[r] = 0.f; DP[r] = 0.f; } \
            _Pragma("unroll") for (int kk = 0; kk < KS; ++kk) { \
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_rows(false, ldsQ, (QB) * 32, kk, lane, 0), kf[kk], S, 0, 0, 0); \
                DP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_rows(false, ldsDO, (QB) * 32, kk, lane, 0), vf[kk], DP, 0, 0, 0); } }
#define MRG_TAIL(QB, S, DP) { \
            _Pragma("unroll") for (int mf = 0; mf < 2; ++mf) { \
                f32x16 pr; \
                _Pragma("unroll") for (int g = 2 * mf; g < 2 * mf + 2; ++g) { \
                    const int qrow = (QB) * 32 + 8 * g + 4 * hf; \
                    const float4 l4 = *reinterpret_cast<const float4*>(ldsL + qrow); \
                    const float4 d4 = *reinterpret_cast<const float4*>(ldsD + qrow); \
                    const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, ds[4] = {d4.x, d4.y, d4.z, d4.w}; \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) { \
                        const int r = g * 4 + e; \
                        const float p = fast_exp2(S[r] * sc - ls[e]); \
                        pr[r] = p; \
                        S[r] = p * (DP[r] - ds[e]) * a.scale; } } \
                const bf16x8 pa = pack8(pr, mf); \
                const bf16x8 da = pack8(S, mf); \
                { \
                    const uint2* w2 = reinterpret_cast<const uint2*>(&da); \
                    _Pragma("unroll") for (int x = 0; x < 2; ++x) \
                        *reinterpret_cast<uint2*>(tw + (QB) * TB + (((4 * mf + 2 * x + hf) ^ tws) << 3)) = w2[x]; } \
                _Pragma("unroll") for (int dj = 0; dj < DJ; ++dj) { \
                    dvT[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_cols_tr(false, ldsDO_u, (QB) * 32, mf, dj, lane, 0), pa, dvT[dj], 0, 0, 0); \
                    dkT[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_cols_tr(false, ldsQ_u, (QB) * 32, mf, dj, lane, 0), da, dkT[dj], 0, 0, 0); } } }
#pragma unroll
        for (int qp2 = 0; qp2 < NT / 2; ++qp2) {
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s0, dp0, s1, dp1;
            MRG_SDP(2 * qp2, s0, dp0)
            MRG_SDP(2 * qp2 + 1, s1, dp1)
            MRG_TAIL(2 * qp2, s0, dp0)
            MRG_TAIL(2 * qp2 + 1, s1, dp1)
        }
        if (NT & 1) {
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s0, dp0;
            MRG_SDP(NT - 1, s0, dp0)
            MRG_TAIL(NT - 1, s0, dp0)
        }
#undef MRG_SDP
#undef MRG_TAIL
        // ---- the global token's QUERY against this lane's key: scores from the fragments (a lane pair holds the key's 64 columns)
        __builtin_amdgcn_sched_barrier(0);
        float sp = 0.f, dpp = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int d0 = kk * 16 + hf * 8;
            sp += dot8(reinterpret_cast<const uint32_t*>(&kf[kk]), qg + d0);
            dpp += dot8(reinterpret_cast<const uint32_t*>(&vf[kk]), dog + d0);
        }
        sp += __shfl_xor(sp, 32, 64);
        dpp += __shfl_xor(dpp, 32, 64);
        const float pk = fast_exp2(sp * sc - ldsL[G]);
        const float dsk = pk * (dpp - ldsD[G]) * a.scale;
        if (hf == 0) ldsDSQ[row] = dsk;
        // dV[k] += p dO[g], dK[k] += ds Q[g]: register r of block dj is column 32 dj + 8 (r >> 2) + 4 hf + (r & 3)
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c0 = 32 * dj + 8 * g4 + 4 * hf;
                const float4 o4 = *reinterpret_cast<const float4*>(dog + c0), q4 = *reinterpret_cast<const float4*>(qg + c0);
                dvT[dj][4 * g4 + 0] += pk * o4.x; dvT[dj][4 * g4 + 1] += pk * o4.y; dvT[dj][4 * g4 + 2] += pk * o4.z; dvT[dj][4 * g4 + 3] += pk * o4.w;
                dkT[dj][4 * g4 + 0] += dsk * q4.x; dkT[dj][4 * g4 + 1] += dsk * q4.y; dkT[dj][4 * g4 + 2] += dsk * q4.z; dkT[dj][4 * g4 + 3] += dsk * q4.w;
            }
    }
    MRG_TS(3);
    __syncthreads();                                   // T complete; Q and dO are dead: their area takes K and the per-wave staging tiles
    MRG_TS(4);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)                    // this wave's K rows, from the registers, in the layout the transposed reads of phase B expect
        *reinterpret_cast<bf16x8*>(ldsK + tile_off<HD, SWZ_ROW>(row, kk * 2 + hf)) = kf[kk];
    char* stg = reg + NM * 128 + wave * (32 * FUSED_SP);
    float cs[3][8];
    fused_stage<2>(stg, dvT, lane);
    fused_flush<2, true>(stg, a.dv + (long)b * a.dv_bs + h * HD, a.dv_rs, wave * 32, NM, lane, cs[2]);
    fused_stage<2>(stg, dkT, lane);
    fused_flush<2, true>(stg, a.dk + (long)b * a.dk_bs + h * HD, a.dk_rs, wave * 32, NM, lane, cs[1]);
    MRG_TS(5);
    __syncthreads();                                   // every wave's K rows are in LDS

    // ---- phase B: this wave's query tile (T block `wave`), contraction over the patch keys; then the global key as a rank-1 update
    {
        f32x16 dqT[DJ];
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqT[dj][r] = 0.f;
        const int i16 = lane & 15;
        const int chunk = ((lane & 16) >> 2) + (i16 & 3);
        const uint32_t tb = ldsT_u + wave * TB;
#pragma unroll
        for (int kb = 0; kb < NT; ++kb)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                const int r0 = kb * 32 + 16 * mf + 4 * hf + (i16 >> 2), r1 = r0 + 8;
                const s16x4 l4 = lds_read_tr16(tb + t_off(r0, chunk));
                const s16x4 h4 = lds_read_tr16(tb + t_off(r1, chunk));
                bf16x8 da;
                da[0] = l4[0]; da[1] = l4[1]; da[2] = l4[2]; da[3] = l4[3];
                da[4] = h4[0]; da[5] = h4[1]; da[6] = h4[2]; da[7] = h4[3];
#pragma unroll
                for (int dj = 0; dj < DJ; ++dj)
                    dqT[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_cols_tr(false, ldsK_u, kb * 32, mf, dj, lane, 0), da, dqT[dj], 0, 0, 0);
            }
        MRG_TS(6);
        const float dsg = ldsDSG[row];
#pragma unroll
        for (int dj = 0; dj < DJ; ++dj)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 k4 = *reinterpret_cast<const float4*>(kg + 32 * dj + 8 * g4 + 4 * hf);
                dqT[dj][4 * g4 + 0] += dsg * k4.x; dqT[dj][4 * g4 + 1] += dsg * k4.y; dqT[dj][4 * g4 + 2] += dsg * k4.z; dqT[dj][4 * g4 + 3] += dsg * k4.w;
            }
        fused_stage<2>(stg, dqT, lane);
        fused_flush<2, true>(stg, a.dq + (long)b * a.dq_bs + h * HD, a.dq_rs, wave * 32, NM, lane, cs[0]);
    }
    // dQ[g] = sum_k ds[g, k] K[k]: lane = column, this wave's 32 keys
    {
        float aq = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int k = wave * 32 + i;
            aq += ldsDSQ[k] * bf2f(*reinterpret_cast<const bf16_t*>(ldsK + tile_off<HD, SWZ_ROW>(k, lane >> 3) + (lane & 7) * 2));
        }
        if (wave == 0) aq += ldsDSG[G] * kg[lane];
        redQ[wave * HD + lane] = aq;
    }
    MRG_TS(7);
    __syncthreads();                                   // partial sums complete; every wave is done with K and T
    float* red = reinterpret_cast<float*>(ldsK);
    if (wave == 0) {                                   // the global token's three output rows
        float tq = 0.f, tk = 0.f, tv = 0.f;
#pragma unroll
        for (int w = 0; w < NT; ++w) { tq += redQ[w * HD + lane]; tk += redKV[(w * 2 + 0) * HD + lane]; tv += redKV[(w * 2 + 1) * HD + lane]; }
        const bf16_t q16 = f2bf(tq), k16 = f2bf(tk), v16 = f2bf(tv);
        a.dq[(long)b * a.dq_bs + (long)G * a.dq_rs + h * HD + lane] = q16;
        a.dk[(long)b * a.dk_bs + (long)G * a.dk_rs + h * HD + lane] = k16;
        a.dv[(long)b * a.dv_bs + (long)G * a.dv_rs + h * HD + lane] = v16;
        if (a.cs_q) { red[(NT * 3 + 0) * HD + lane] = bf2f(q16); red[(NT * 3 + 1) * HD + lane] = bf2f(k16); red[(NT * 3 + 2) * HD + lane] = bf2f(v16); }
    }
    if (a.cs_q) {   // per-(batch, head) column sums of dQ / dK / dV: the bias gradients before the reduction over the batch
        if (lane < 8) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) red[(wave * 3 + m) * HD + lane * 8 + e] = cs[m][e];
        }
        __syncthreads();
        for (int c = tid; c < 3 * HD; c += NTHR) {
            float t = 0.f;
            for (int w = 0; w <= NT; ++w) t += red[w * 3 * HD + c];
            float* dst = c < HD ? a.cs_q : (c < 2 * HD ? a.cs_k : a.cs_v);
            dst[(long)b * a.cs_bs + h * HD + (c % HD)] = t;
        }
    }
}
#undef MRG_TS

template <int NT>
int launch_bwd_fused_gt(const AttnArgs& a, hipStream_t stream) {
    constexpr int NM = NT * 32;
    constexpr size_t REG_A = 2 * NM * 128, REG_B = NM * 128 + NT * 32 * FUSED_SP;
    const size_t sh = (size_t)NT * NM * 64 + (REG_A > REG_B ? REG_A : REG_B) + sizeof(float) * (4 * (NM + 4) + NM + 4 * 64 + NT * 3 * 64);
    static bool attr_set = false;
    if (sh > 65536 && !attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_fused_gt_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_bwd_fused_gt_kernel<NT>, dim3(a.H, a.B), dim3(64 * NT), sh, stream, a);
    return mrg_check_launch("mrg_attn_bwd/fused_gt");
}

// MRG_ATTN_GT=0: N = 32 m + 1 goes through the general fused kernel (five tiles) like any other length (A/B knob)
bool fused_gt_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MRG_ATTN_GT"); on = (e && e[0] == '0') ? 0 : 1; }
    return on == 1;
}

// MRG_ATTN_FUSED=0 selects the query-owner + key-owner pair of kernels for every shape (A/B knob; the fused kernel is the default)
bool fused_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MRG_ATTN_FUSED"); on = (e && e[0] == '0') ? 0 : 1; }
    return on == 1;
}

// waves per workgroup so that every wave owns the same number of 32-row tiles: ceil(tiles / ceil(tiles / 8))
static int waves_for(int ntiles) {
    const int rounds = (ntiles + 7) / 8;
    int w = (ntiles + rounds - 1) / rounds;
    return w < 1 ? 1 : w;
}

template <int HD>
int launch_fwd(const AttnArgs& a0, hipStream_t stream) {
    AttnArgs a = a0;
    // one token more than a multiple of 32 (the global token behind 128 / 256 patches): see attn_fwd_kernel<.., TAIL>.  MRG_ATTN_TAIL=0: off
    static const int tail_on = [] { const char* e = getenv("MRG_ATTN_TAIL"); return e ? atoi(e) : 1; }();
    const bool tail = tail_on && HD == 64 && a.Nq == a.Nk && (a.Nk & 31) == 1 && a.Nk >= (tail_on == 1 ? 193 : 33) &&   /* N = 129: 21.2 against 20.2 us for the compact five-tile form; N = 257: 217 against 251 */
                      (size_t)2 * (a.Nk - 1) * HD * 2 <= 80 * 1024;
    const int Nq = tail ? a.Nq - 1 : a.Nq, Nk = tail ? a.Nk - 1 : a.Nk;       // what the tiles cover
    a.tail = tail ? 1 : 0;
    const int nkb = (Nk + 31) / 32;
    dim3 grid(a.H, a.B), block(64 * waves_for((Nq + 31) / 32));
    // K and V of the whole row stay in LDS when that leaves room for two workgroups per CU (<= 80 KiB)
    // Resident K / V are stored compactly: rows N .. (padded to 32) alias ONE zero row (index rmax), and the last wave's staging tile holds
    // only its valid rows -- at N = 129 that is 132 instead of 160 rows per tile and 52.8 instead of 63 KiB per workgroup: three
    // workgroups per CU instead of two, i.e. the 768 (batch, head) pairs of C2 in ONE round instead of two (19.4 -> ~14 us);
    // N = 99 (C4): 40 instead of 50 KiB, four per CU instead of three (1024 pairs in one round).
    static const int compact = [] { const char* e = getenv("MRG_ATTN_COMPACT"); return e ? atoi(e) : 1; }();
#define MRG_LAUNCH(CB_, ST_, TL_) do { \
        if (sh > 65536) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<HD, CB_, ST_, TL_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((attn_fwd_kernel<HD, CB_, ST_, TL_>), grid, block, sh, stream, a); } while (0)
#define MRG_FWD(CB_) do { const bool fits = (size_t)2 * nkb * 32 * HD * 2 <= 80 * 1024; \
        a.chunk_rows = fits ? nkb * 32 : CB_ * 32; \
        if (a.chunk_rows < CB_ * 32) a.chunk_rows = CB_ * 32; \
        a.rmax = 0x7fffffff; a.stage_tail = 0; \
        const int nwav = block.x / 64, nqt_ = (Nq + 31) / 32; \
        if (compact && fits && (Nk & 31) && CB_ == 1) { a.chunk_rows = ((Nk + 1 + 3) / 4) * 4; a.rmax = a.chunk_rows - 1; } \
        const size_t kv = (size_t)2 * a.chunk_rows * HD * 2; \
        size_t stg = (size_t)nwav * 32 * FUSED_SP; \
        if (compact && nqt_ == nwav && (Nq & 31) && (Nq & 31) <= 16) { a.stage_tail = (((Nq & 31) + 3) / 4) * 4; \
                                                                      stg = (size_t)(nwav - 1) * 32 * FUSED_SP + (size_t)a.stage_tail * FUSED_SP; } \
        const size_t xch = tail ? (size_t)3 * HD * 2 + (size_t)nwav * 66 * 4 + 16 : 0;     /* TAIL: the last token's q / k / v rows + the waves' exchange area */ \
        const bool stage = kv + stg + xch <= 80 * 1024;        /* staging must not cost the second workgroup per CU */ \
        if (!stage) a.stage_tail = 0; \
        a.tail_off = (int)(stage ? kv + stg : kv); \
        const size_t sh = (stage ? kv + stg : kv) + xch; \
        if (tail && HD == 64) { if (stage) MRG_LAUNCH(1, true, true); else MRG_LAUNCH(1, false, true); } \
        else if (stage) MRG_LAUNCH(CB_, true, false); \
        else MRG_LAUNCH(CB_, false, false); } while (0)
    // Scores are kept in registers one chunk of CB key blocks at a time with an online-softmax rescale between chunks.  Small
    // chunks win on these short rows: CB = 1 needs 128 VGPRs (4 waves per SIMD) against 256 for CB = 5, and the kernel is bound
    // by latency chains, not MFMA throughput (C2 30.7 -> 23.2 us, C5 400 -> 297 us; tools/bench_attn_fwd.py)
    if (HD == 64) MRG_FWD(1);
    else MRG_FWD(2);
#undef MRG_FWD
#undef MRG_LAUNCH
    return mrg_check_launch("mrg_attn_fwd");
}

int check_common(const mrg_attn_desc* d, const char* who) {
    MRG_REQUIRE(d != nullptr, MRG_ERR_ARG, "%s: null descriptor", who);
    MRG_REQUIRE(d->q && d->k && d->v && d->o, MRG_ERR_ARG, "%s: null tensor", who);
    MRG_REQUIRE(d->head_dim == 32 || d->head_dim == 64, MRG_ERR_UNSUPPORTED, "%s: head_dim %d not in {32, 64}", who, d->head_dim);
    MRG_REQUIRE(d->B > 0 && d->H > 0 && d->Nq > 0 && d->Nk > 0, MRG_ERR_SHAPE, "%s: bad shape", who);
    const long strides[] = {d->q_bs, d->q_rs, d->k_bs, d->k_rs, d->v_bs, d->v_rs, d->o_bs, d->o_rs};
    for (long s : strides) MRG_REQUIRE(s % 8 == 0, MRG_ERR_SHAPE, "%s: strides must be multiples of 8 elements (16 B)", who);
    MRG_REQUIRE((((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) == 0, MRG_ERR_ARG, "%s: q/k/v must be 16-B aligned", who);
    return 0;
}

void fill(AttnArgs& a, const mrg_attn_desc* d) {
    a.q = (const bf16_t*)d->q; a.k = (const bf16_t*)d->k; a.v = (const bf16_t*)d->v; a.o = (bf16_t*)d->o; a.lse = d->lse;
    a.q_bs = d->q_bs; a.q_rs = d->q_rs; a.k_bs = d->k_bs; a.k_rs = d->k_rs; a.v_bs = d->v_bs; a.v_rs = d->v_rs;
    a.o_bs = d->o_bs; a.o_rs = d->o_rs;
    a.B = d->B; a.H = d->H; a.Nq = d->Nq; a.Nk = d->Nk; a.scale = d->scale;
    a.cs_q = d->colsum_q; a.cs_k = d->colsum_k; a.cs_v = d->colsum_v; a.cs_bs = d->colsum_bs;
    a.d_o = (const bf16_t*)d->d_o; a.do_bs = d->do_bs; a.do_rs = d->do_rs; a.dsum = d->dsum;
    a.dq = (bf16_t*)d->dq; a.dk = (bf16_t*)d->dk; a.dv = (bf16_t*)d->dv;
    a.dq_bs = d->dq_bs; a.dq_rs = d->dq_rs; a.dk_bs = d->dk_bs; a.dk_rs = d->dk_rs; a.dv_bs = d->dv_bs; a.dv_rs = d->dv_rs;
    a.chunk_rows = 0; a.rmax = 0x7fffffff; a.stage_tail = 0; a.tail = 0; a.tail_off = 0;
}

}  // namespace

namespace mrg_detail {       // attention_drop.hip: the plain path with dropout on the probabilities
int attn_drop_fwd(const mrg_attn_desc* d, hipStream_t stream);
int attn_drop_bwd(const mrg_attn_desc* d, hipStream_t stream);
}

extern "C" int mrg_attn_fwd(const mrg_attn_desc* d, hipStream_t stream) {
    int rc = check_common(d, "mrg_attn_fwd");
    if (rc) return rc;
    if (d->drop_rng != nullptr && d->drop_p != 0.f) return mrg_detail::attn_drop_fwd(d, stream);
    AttnArgs a;
    fill(a, d);
    return d->head_dim == 64 ? launch_fwd<64>(a, stream) : launch_fwd<32>(a, stream);
}

extern "C" int mrg_attn_bwd(const mrg_attn_desc* d, hipStream_t stream) {
    int rc = check_common(d, "mrg_attn_bwd");
    if (rc) return rc;
    MRG_REQUIRE(d->d_o && d->lse && d->dsum && d->dq && d->dk && d->dv, MRG_ERR_ARG, "mrg_attn_bwd: null tensor");
    MRG_REQUIRE(d->do_bs % 8 == 0 && d->do_rs % 8 == 0, MRG_ERR_SHAPE, "mrg_attn_bwd: dO strides must be multiples of 8");
    if (d->drop_rng != nullptr && d->drop_p != 0.f) return mrg_detail::attn_drop_bwd(d, stream);
    AttnArgs a;
    fill(a, d);
    const int HD = d->head_dim;
    if (HD == 64 && a.Nq == a.Nk && a.Nq > 96 && a.Nq <= 160 && fused_enabled()) {
        if (a.Nq == 129 && fused_gt_enabled()) return launch_bwd_fused_gt<4>(a, stream);
        if (a.Nq == 97 && fused_gt_enabled()) return launch_bwd_fused_gt<3>(a, stream);
        return a.Nq > 128 ? launch_bwd_fused<5>(a, stream) : launch_bwd_fused<4>(a, stream);
    }
    dim3 grid(a.H, a.B);
    {   // dQ: K,V chunk in LDS
        dim3 block(64 * waves_for((a.Nq + 31) / 32));
        AttnArgs b = a;
        b.chunk_rows = ((a.Nk + 31) / 32) * 32;
        if (b.chunk_rows > 288) b.chunk_rows = 256;
        const size_t sh = (size_t)2 * b.chunk_rows * HD * 2;
        if (sh > 65536) (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (HD == 64) hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, grid, block, sh, stream, b);
        else hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, grid, block, sh, stream, b);
        rc = mrg_check_launch("mrg_attn_bwd/dq");
        if (rc) return rc;
    }
    {   // dK,dV: Q,dO chunk in LDS
        dim3 block(64 * waves_for((a.Nk + 31) / 32));
        AttnArgs b = a;
        b.chunk_rows = ((a.Nq + 31) / 32) * 32;
        if (b.chunk_rows > 288) b.chunk_rows = 256;
        const size_t sh = (size_t)2 * b.chunk_rows * HD * 2 + (size_t)2 * b.chunk_rows * 4;
        if (sh > 65536) (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (HD == 64) hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, grid, block, sh, stream, b);
        else hipLaunchKernelGGL(attn_bwd_dkv_kernel<32>, grid, block, sh, stream, b);
        rc = mrg_check_launch("mrg_attn_bwd/dkv");
    }
    return rc;
}
