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
    float* cs_q; float* cs_k; float* cs_v; long cs_bs;   // optional per-(batch, head) column sums of dQ / dK / dV (bias gradients)
};

// ================================================ forward ===============================================
template <int HD, int CB>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 16, DJ = HD / 32, ROWS = CB * 32;
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

    if (resident) {
        load_rows2<HD, SWZ_ROW, SWZ_TR>(ldsK, kp, a.k_rs, ldsV, vp, a.v_rs, lds_rows, a.Nk, tid, nthr);
        __syncthreads();
    }
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
        float m_run = -INFINITY, l_run = 0.f;

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
                if (kb * 32 < kvalid) {
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk)
                        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD, SWZ_ROW>(ldsK, lbase + kb * 32, kk, lane), qf[kk], s[kb], 0, 0, 0);
                }
                if (kb * 32 + 32 > kvalid) {         // only the last, partial key block needs the key-range test
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (kb * 32 + acc_row(r, lane) >= kvalid) s[kb][r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = fast_exp2((m_run - m_new) * sc);   // first chunk: exp2(-inf) = 0
            float ls = 0.f;
#pragma unroll
            for (int kb = 0; kb < CB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = fast_exp2((s[kb][r] - m_new) * sc);
                    s[kb][r] = p;
                    ls += p;
                }
            ls += __shfl_xor(ls, 32, 64);
            l_run = l_run * alpha + ls;
            m_run = m_new;
            if (ch > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float al = __shfl(alpha, acc_row(r, lane), 64);
#pragma unroll
                    for (int dj = 0; dj < DJ; ++dj) o[dj][r] *= al;
                }
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
                            o[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_cols_tr<HD, SWZ_TR>(ldsV_u, lbase + kb * 32, mf, dj, lane), o[dj], 0, 0, 0);
                    }
                }
            }
        }
        if (!active) continue;
        const float linv = 1.f / l_run;
        bf16_t* op = a.o + (long)b * a.o_bs + h * HD;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lane);
            const float li = __shfl(linv, row, 64);
            const int qr = q0 + row;
            if (qr < a.Nq) {
#pragma unroll
                for (int dj = 0; dj < DJ; ++dj) op[(long)qr * a.o_rs + dj * 32 + (lane & 31)] = f2bf(o[dj][r] * li);
            }
        }
        if (a.lse && lane < 32 && q0 + lane < a.Nq)
            a.lse[((long)b * a.H + h) * a.Nq + q0 + lane] = m_run * a.scale + __logf(l_run);
    }
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
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_kernel(const AttnArgs a) {
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

// waves per workgroup so that every wave owns the same number of 32-row tiles: ceil(tiles / ceil(tiles / 8))
static int waves_for(int ntiles) {
    const int rounds = (ntiles + 7) / 8;
    int w = (ntiles + rounds - 1) / rounds;
    return w < 1 ? 1 : w;
}

template <int HD>
int launch_fwd(const AttnArgs& a0, hipStream_t stream) {
    AttnArgs a = a0;
    const int nkb = (a.Nk + 31) / 32;
    dim3 grid(a.H, a.B), block(64 * waves_for((a.Nq + 31) / 32));
    // K and V of the whole row stay in LDS when that leaves room for two workgroups per CU (<= 80 KiB)
#define MRG_FWD(CB_) do { const bool fits = (size_t)2 * nkb * 32 * HD * 2 <= 80 * 1024; \
        a.chunk_rows = fits ? nkb * 32 : CB_ * 32; \
        if (a.chunk_rows < CB_ * 32) a.chunk_rows = CB_ * 32; \
        const size_t sh = (size_t)2 * a.chunk_rows * HD * 2; \
        if (sh > 65536) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<HD, CB_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((attn_fwd_kernel<HD, CB_>), grid, block, sh, stream, a); } while (0)
    // Scores are kept in registers one chunk of CB key blocks at a time with an online-softmax rescale between chunks.  Small
    // chunks win on these short rows: CB = 1 needs 128 VGPRs (4 waves per SIMD) against 256 for CB = 5, and the kernel is bound
    // by latency chains, not MFMA throughput (C2 30.7 -> 23.2 us, C5 400 -> 297 us; tools/bench_attn_fwd.py)
    (void)nkb;
    if (HD == 64) MRG_FWD(1);
    else MRG_FWD(2);
#undef MRG_FWD
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
    a.chunk_rows = 0;
}

}  // namespace

extern "C" int mrg_attn_fwd(const mrg_attn_desc* d, hipStream_t stream) {
    int rc = check_common(d, "mrg_attn_fwd");
    if (rc) return rc;
    AttnArgs a;
    fill(a, d);
    return d->head_dim == 64 ? launch_fwd<64>(a, stream) : launch_fwd<32>(a, stream);
}

extern "C" int mrg_attn_bwd(const mrg_attn_desc* d, hipStream_t stream) {
    int rc = check_common(d, "mrg_attn_bwd");
    if (rc) return rc;
    MRG_REQUIRE(d->d_o && d->lse && d->dsum && d->dq && d->dk && d->dv, MRG_ERR_ARG, "mrg_attn_bwd: null tensor");
    MRG_REQUIRE(d->do_bs % 8 == 0 && d->do_rs % 8 == 0, MRG_ERR_SHAPE, "mrg_attn_bwd: dO strides must be multiples of 8");
    AttnArgs a;
    fill(a, d);
    const int HD = d->head_dim;
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
