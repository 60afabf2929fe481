// Row-local chains of the 256-wide output adapters in ONE launch per chain (round 6).
//
// The adapters' GEMMs contract over K = 256 or produce 256 columns: a 64-row tile holds whole rows, so everything between two attention
// calls is row-local and can stay on the CU:
//   mrg_rowfuse_fwd, a0 != NULL :  x = a0 W0^T + b0 (+ resid)  ->  LayerNorm  ->  fc1 + GELU  ->  fc2 + b2 + x        ("post-attention")
//                                  mirage/output_adapters.py:279-280 (proj, out_norm, mlp), mirage/utils.py:259-262 (proj + residual,
//                                  norm2, mlp + residual of a decoder Block)
//   mrg_rowfuse_fwd, a0 == NULL :  x = resid (fp32 rows)       ->  LayerNorm  ->  y = ln W1^T + b1 (NH = 256 q / 512 kv / 768 qkv)
//                                  mirage/output_adapters.py:275-276 (query_norm / context_norm + q / kv), mirage/utils.py:259 (norm1 + qkv)
// As separate launches these stages are HBM-bound (the 16384 x 1024 hidden activations are written and read again: 217 MB per MLP
// stage against 127 MB here) and launch-latency-bound (4.7 us node to node); fused, the hidden activations exist only as the bf16
// operand of the fc2 k-chunk they feed (and, in training, as the side outputs the weight gradients read later).
//
// One workgroup = 64 rows x 4 waves; wave w owns columns [64 w, 64 w + 64) of every 256-column product (2 x 2 MFMA 32x32x16 tiles,
// operands swapped so that a lane holds ROW lane & 31 and four consecutive columns per accumulator quad: LayerNorm statistics, GELU,
// residual adds and the bf16 repack for the next product all run on the accumulator registers).  Activations (the A operand) live in
// LDS as 4 k-tiles of [64 rows][64 k] in gemm_common.h's swizzled layout; weights stream L2 -> LDS by global_load_lds through a ring
// of four [256 n][32 k] tiles (16 KiB, three in flight), one raw s_barrier per tile, counted s_waitcnt vmcnt with the side-output
// stores included in the count (vmcnt retires in order).  EVERY global store leaves through an LDS staging area as 16 B per lane over
// whole 128-B row segments: a row-per-lane store straight from the accumulator layout touches 32 lines per wave instruction and is
// store-ISSUE-bound (~600 cycles per instruction, MI355X_MICROARCH.md price list) -- the first version of this kernel spent more than
// half of its 64 us there.  154 KiB LDS, one workgroup per CU.
// M must be a multiple of 64 (every BASELINE shape is: 16384, 8256 = 129 x 64, 6336 = 99 x 64); the host falls back to the separate
// launches otherwise.
#include "gemm_common.h"

namespace {

constexpr int RF_A_BYTES = 64 * 256 * 2;                        // 32 KiB: [4 k-tiles][64 rows][64 k] bf16
constexpr int RF_WT_BYTES = 256 * 32 * 2;                       // 16 KiB: one weight tile [256 n][32 k]
constexpr int RF_RING = 4;
constexpr int RF_OFF_A = 0, RF_OFF_H = RF_A_BYTES, RF_OFF_W = 2 * RF_A_BYTES;
constexpr int RF_OFF_G = RF_OFF_W + RF_RING * RF_WT_BYTES;      // 16 KiB staging of gelu'(h): [4 k-tiles][32 rows][64 cols] fp16, one row half at a time
constexpr int RF_OFF_VEC = RF_OFF_G + 16384;                    // fp32 vectors: b0 | gamma | beta | b2 | b1[<= 1024]
constexpr int RF_OFF_RED = RF_OFF_VEC + 2048 * 4;               // two [4 waves][64 rows] fp32 reduction buffers
constexpr int RF_OFF_TS = RF_OFF_RED + 2 * 4 * 64 * 4;          // (debug build -DMRG_RF_TS: 64 phase timestamps)
constexpr int RF_LDS_BYTES = RF_OFF_TS + 512;                   // 158208

struct RowFuseArgs {
    int M, NH; float eps;
    const bf16_t* a0; const bf16_t* w0; const float* b0; const float* resid; float* x_out;
    const float* gamma; const float* beta; float* mean; float* rstd; bf16_t* ln_out;
    const bf16_t* w1; const float* b1; bf16_t* h_out; _Float16* hgrad_out;
    const bf16_t* w2; const float* b2; float* y32; bf16_t* y16;
};

#define RF_WAIT(N_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N_) : "memory")
#define RF_BARRIER() do { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RF_LDS_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); RF_BARRIER(); } while (0)
#ifdef MRG_RF_TS      // debug build: shader-clock timestamps of wave 0 at the phase boundaries, returned in rstd[m0 .. m0 + 63] as cycles since the first
#define RF_TS(i_) do { if (tid == 0) reinterpret_cast<unsigned long long*>(smem + RF_OFF_TS)[i_] = clock64(); } while (0)
#else
#define RF_TS(i_) do { } while (0)
#endif

// PROJ: stage 0 is a projection of a bf16 operand (else the rows come in as fp32); MLP: fc1 + GELU + fc2 (else a plain projection to NH
// columns); SAVE: training -- x, LayerNorm output / statistics, gelu(h) and gelu'(h) are written for the backward pass
template <bool PROJ, bool MLP, bool SAVE>
__global__ __launch_bounds__(256, 1) void rowfuse_fwd_kernel(const RowFuseArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 64;
    const uint32_t sbase = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    char* Areg = smem + RF_OFF_A;
    char* Hreg = smem + RF_OFF_H;
    float* vec = reinterpret_cast<float*>(smem + RF_OFF_VEC);
    float* red = reinterpret_cast<float*>(smem + RF_OFF_RED);
    const int NH = g.NH, NC = NH >> 8;
    const int Ttot = (PROJ ? 8 : 0) + NC * (MLP ? 16 : 8);

    // ---- weight-tile stream (wave-uniform state)
    const uint32_t c16 = (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const uint32_t voff_d = (uint32_t)((lane >> 2) * 512) + c16;               // rows of 256 elements
    const uint32_t voff_h = (uint32_t)((lane >> 2) * NH * 2) + c16;            // rows of NH elements (fc2's weight)
    int t_issue = 0, slot_issue = 0, slot_use = 0;
    // source / destination of the next tile to request (wave-uniform): scalar base of this wave's 64 weight rows, row pitch, LDS slot
    struct Tile { const char* sb; long pitch; uint32_t voff, dst; };
    auto next_tile = [&]() {
        const int T = min(t_issue, Ttot - 1);        // past the end: reload the last tile (keeps the in-flight count constant)
        int s = T >> 3;
        const int kt = T & 7;
        const char* base;
        int ld;
        if (PROJ && s == 0) { base = reinterpret_cast<const char*>(g.w0) + kt * 64; ld = 256; }
        else {
            if (PROJ) s -= 1;
            if (MLP) {
                const int c = s >> 1;
                if (s & 1) { base = reinterpret_cast<const char*>(g.w2) + (c * 256 + kt * 32) * 2; ld = NH; }
                else { base = reinterpret_cast<const char*>(g.w1) + ((long)c * 65536 + kt * 32) * 2; ld = 256; }
            } else { base = reinterpret_cast<const char*>(g.w1) + ((long)s * 65536 + kt * 32) * 2; ld = 256; }
        }
        Tile t;
        t.voff = ld == 256 ? voff_d : voff_h;
        t.dst = sbase + RF_OFF_W + slot_issue * RF_WT_BYTES + wave * 4096;
        t.sb = base + (long)wave * 128 * ld;                                    // this wave's 64 weight rows
        t.pitch = (long)32 * ld;                                                // 16 rows per 1-KiB piece
        ++t_issue;
        slot_issue = slot_issue == RF_RING - 1 ? 0 : slot_issue + 1;
        return t;
    };
    auto dma = [&](const Tile& t, int jj) { glds16_s(uniform_ptr(t.sb + jj * t.pitch), t.voff, t.dst + jj * 1024); };
    auto issue = [&]() {
        const Tile t = next_tile();
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dma(t, jj);
    };
    const uint32_t sw = (uint32_t)((lr >> 2) & 3);
    const uint32_t fb_off[2] = {(uint32_t)(lr * 64) + (((0u + h) ^ sw) << 4), (uint32_t)(lr * 64) + (((2u + h) ^ sw) << 4)};
    // one tile: all fragment reads first, then the eight MFMAs with the four DMA pieces of the tile three ahead between them -- a DMA
    // piece stalls the issuing wave for 60-180 cycles (MI355X_MICROARCH.md price list), which the matrix pipe covers only if it has work
    // queued; requested up front (the first version) the pieces, the LDS reads and the MFMAs of the only wave of a SIMD ran one after
    // the other: ~1600 cycles per tile for 256 cycles of matrix work
    auto step = [&](const char* A, int kt, f32x16 (&acc)[2][2]) {
        const Tile t = next_tile();
        const char* wb = smem + RF_OFF_W + slot_use * RF_WT_BYTES + wave * 4096;
        bf16x8 fa[2][2], fb[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[kk][i] = read_frag<MRG_OP_KCONTIG>(A + (kt >> 1) * 8192, 0u, i * 32, (kt & 1) * 2 + kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[kk][j] = *reinterpret_cast<const bf16x8*>(wb + j * 2048 + fb_off[kk]);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                dma(t, kk * 2 + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        slot_use = slot_use == RF_RING - 1 ? 0 : slot_use + 1;
    };
    // bf16 quad (4 consecutive columns of one row) into an activation region
    auto a_store = [&](char* reg, int row, int j, int q, uint32_t lo, uint32_t hi) {
        const int c = j * 4 + q;
        *reinterpret_cast<uint2*>(reg + wave * 8192 + row * 128 + ((c ^ ((row >> 1) & 7)) << 4) + h * 8) = make_uint2(lo, hi);
    };
    // region -> global rows, 16 B per thread and pass, 128-B row segments (pass it = 0..7: k-tile it >> 1, rows 32 (it & 1) ..)
    auto copy_out = [&](const char* reg, bf16_t* dst, long ld, int it) {
        const int kt = it >> 1, u = tid + 256 * (it & 1), row = u >> 3, c = u & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(reg + kt * 8192 + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        *reinterpret_cast<uint4*>(dst + (long)(m0 + row) * ld + kt * 64 + c * 8) = v;
    };
    // fp32 [64 rows][256 cols] staged over Areg + Hreg (64 KiB, 1 KiB per row, 16-B chunk index XOR row & 15: conflict-free both ways):
    // the accumulator layout writes a quad per lane, the copy reads 64 consecutive chunks of a row per wave instruction
    auto f32_stage = [&](int i, int j, int q, float a, float b, float c, float d) {
        const int row = i * 32 + lr, cc = wave * 16 + j * 8 + q * 2 + h;
        *reinterpret_cast<float4*>(smem + row * 1024 + ((cc ^ (row & 15)) << 4)) = make_float4(a, b, c, d);
    };
    auto f32_copy_out = [&](float* dst, int it) {           // it = 0..15: rows 4 it .. 4 it + 3
        const int row = it * 4 + wave, cc = lane;
        const float4 v = *reinterpret_cast<const float4*>(smem + row * 1024 + ((cc ^ (row & 15)) << 4));
        *reinterpret_cast<float4*>(dst + (long)(m0 + row) * 256 + cc * 4) = v;
    };
    // gelu'(h) of one row half (32 rows x 256 columns, fp16) through the 16 KiB area at RF_OFF_G
    auto g_store = [&](int j, int q, uint32_t lo, uint32_t hi) {
        const int c = j * 4 + q;
        *reinterpret_cast<uint2*>(smem + RF_OFF_G + wave * 4096 + lr * 128 + ((c ^ ((lr >> 1) & 7)) << 4) + h * 8) = make_uint2(lo, hi);
    };
    auto g_copy_out = [&](_Float16* dst, int half, int cidx, int it) {      // it = 0..3 = k-tile (64 columns)
        const int row = tid >> 3, c = tid & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(smem + RF_OFF_G + it * 4096 + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        *reinterpret_cast<uint4*>(dst + (long)(m0 + half * 32 + row) * NH + cidx * 256 + it * 64 + c * 8) = v;
    };
    auto zero_acc = [&](f32x16 (&acc)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

    // ---- prologue: stage-0 operand and the first three weight tiles in flight, then the fp32 rows and the small vectors
    RF_TS(0);
    if (PROJ) {
        const char* sa = uniform_ptr(reinterpret_cast<const char*>(g.a0) + ((long)m0 * 256 + wave * 64) * 2);   // k-tile = wave
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
            glds16_s(sa, glds_lane_off<MRG_OP_KCONTIG>(256, m0, g.M, jj, lane), sbase + RF_OFF_A + wave * 8192 + jj * 1024);
    }
#pragma unroll
    for (int p = 0; p < RF_RING - 1; ++p) issue();
    f32x16 x[2][2];
    const bool have_x = !PROJ || g.resid != nullptr;
    if (have_x) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = *reinterpret_cast<const float4*>(g.resid + (long)(m0 + i * 32 + lr) * 256 + wave * 64 + j * 32 + q * 8 + h * 4);
                    x[i][j][4 * q] = t.x; x[i][j][4 * q + 1] = t.y; x[i][j][4 * q + 2] = t.z; x[i][j][4 * q + 3] = t.w;
                }
    } else zero_acc(x);
    {
        const float v0 = PROJ ? g.b0[tid] : 0.f, v1 = g.gamma[tid], v2 = g.beta[tid], v3 = MLP ? g.b2[tid] : 0.f;
        vec[tid] = v0; vec[256 + tid] = v1; vec[512 + tid] = v2; vec[768 + tid] = v3;
        for (int i = tid; i < NH; i += 256) vec[1024 + i] = g.b1[i];
    }
    // every compiler-visible load retires here (one wait for the whole prologue, the DMAs above included): nothing below makes
    // hipcc insert a vmcnt wait of its own, so the counted waits of the tile loop are the only ones
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(x[i][j]));
    RF_WAIT(0);
    RF_BARRIER();
    RF_TS(1);

    f32x16 acc[2][2], acc2[2][2];
    // ---- stage 0: x = a0 W0^T + b0 (+ resid)
    if (PROJ) {
        zero_acc(acc);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k > 0) { RF_WAIT(8); RF_BARRIER(); }
            step(Areg, k, acc);
        }
        RF_TS(2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = wave * 64 + j * 32 + q * 8 + h * 4;
                    const float4 b = *reinterpret_cast<const float4*>(vec + col);
                    x[i][j][4 * q] += acc[i][j][4 * q] + b.x; x[i][j][4 * q + 1] += acc[i][j][4 * q + 1] + b.y;
                    x[i][j][4 * q + 2] += acc[i][j][4 * q + 2] + b.z; x[i][j][4 * q + 3] += acc[i][j][4 * q + 3] + b.w;
                }
        if (SAVE) {
            RF_LDS_SYNC();       // every wave is done with the stage-0 operand in Areg: Areg + Hreg become the fp32 staging of x
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) f32_stage(i, j, q, x[i][j][4 * q], x[i][j][4 * q + 1], x[i][j][4 * q + 2], x[i][j][4 * q + 3]);
        }
    }
    // ---- LayerNorm over the 256 columns of each row (two passes over the registers, like ln_fwd_small_kernel)
    RF_TS(3);
    float mu[2], rs[2];
    {
        float s[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) t += (x[i][j][4 * q] + x[i][j][4 * q + 1]) + (x[i][j][4 * q + 2] + x[i][j][4 * q + 3]);
            s[i] = t + __shfl_xor(t, 32, 64);
        }
        if (h == 0) { red[wave * 64 + lr] = s[0]; red[wave * 64 + 32 + lr] = s[1]; }
        RF_LDS_SYNC();           // (also: every wave is done with the stage-0 operand in Areg; the staged x is complete)
        if (PROJ && SAVE) {
#pragma unroll
            for (int it = 0; it < 16; ++it) f32_copy_out(g.x_out, it);
        }
        RF_TS(4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = i * 32 + lr;
            mu[i] = ((red[r] + red[64 + r]) + (red[128 + r] + red[192 + r])) * (1.f / 256.f);
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float d = x[i][j][e] - mu[i]; t = fmaf(d, d, t); }
            s[i] = t + __shfl_xor(t, 32, 64);
        }
        if (h == 0) { red[256 + wave * 64 + lr] = s[0]; red[256 + wave * 64 + 32 + lr] = s[1]; }
        RF_LDS_SYNC();           // (the copy of the staged x has read Areg / Hreg: they may be overwritten from here on)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = i * 32 + lr;
            rs[i] = rsqrtf(((red[256 + r] + red[320 + r]) + (red[384 + r] + red[448 + r])) * (1.f / 256.f) + g.eps);
        }
        if (SAVE && h == 0 && (lr >> 3) == wave) {       // every wave stores a quarter of the rows
            g.mean[m0 + lr] = mu[0]; g.mean[m0 + 32 + lr] = mu[1];
            g.rstd[m0 + lr] = rs[0]; g.rstd[m0 + 32 + lr] = rs[1];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = wave * 64 + j * 32 + q * 8 + h * 4;
                    const float4 gm = *reinterpret_cast<const float4*>(vec + 256 + col), bt = *reinterpret_cast<const float4*>(vec + 512 + col);
                    const float o0 = (x[i][j][4 * q] - mu[i]) * rs[i] * gm.x + bt.x, o1 = (x[i][j][4 * q + 1] - mu[i]) * rs[i] * gm.y + bt.y;
                    const float o2 = (x[i][j][4 * q + 2] - mu[i]) * rs[i] * gm.z + bt.z, o3 = (x[i][j][4 * q + 3] - mu[i]) * rs[i] * gm.w + bt.w;
                    a_store(Areg, i * 32 + lr, j, q, pack_bf2(o0, o1), pack_bf2(o2, o3));
                }
    }
    // ---- the products over the LayerNorm output, 256 output / hidden columns at a time.  vmcnt budget of a tile wait: 8 (the two
    // younger tiles' DMAs) + the stores issued behind the awaited tile's DMA, i.e. in the two iterations before this one (an iteration
    // issues its stores first, then -- between its MFMAs -- the DMA of the tile three ahead)
    RF_TS(5);
    if (MLP) zero_acc(acc2);
    uint32_t gq[2][2][4][2];          // gelu'(h) of the chunk as packed fp16 pairs, until both row halves have gone through the staging area
    for (int c = 0; c < NC; ++c) {
        zero_acc(acc);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // !MLP: 8 stores in k = 0 (the previous chunk's output) count for the waits of k = 1, 2
            if (!MLP && c > 0 && (k == 1 || k == 2)) RF_WAIT(16); else RF_WAIT(8);
            RF_BARRIER();
            if (k == 0) {
                if (c == 0) {
                    if (SAVE) {
#pragma unroll
                        for (int it = 0; it < 8; ++it) copy_out(Areg, g.ln_out, 256, it);
                    }
                } else if (!MLP) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) copy_out(Hreg, g.h_out + (c - 1) * 256, NH, it);
                }
            }
            step(Areg, k, acc);
        }
        RF_TS(6 + 4 * c);
        // epilogue of the chunk: + b1 -> (GELU) -> bf16 into Hreg
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = wave * 64 + j * 32 + q * 8 + h * 4;
                    const float4 b = *reinterpret_cast<const float4*>(vec + 1024 + c * 256 + col);
                    float v[4] = {acc[i][j][4 * q] + b.x, acc[i][j][4 * q + 1] + b.y, acc[i][j][4 * q + 2] + b.z, acc[i][j][4 * q + 3] + b.w};
                    if (MLP) {
                        if (SAVE) {
                            float gd[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float cd, pd;
                                gelu_parts(v[e], cd, pd);
                                gd[e] = cd + v[e] * pd;
                                v[e] = v[e] * cd;
                            }
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                            const h2 lo = {(_Float16)gd[0], (_Float16)gd[1]}, hi = {(_Float16)gd[2], (_Float16)gd[3]};
                            gq[i][j][q][0] = *reinterpret_cast<const uint32_t*>(&lo);
                            gq[i][j][q][1] = *reinterpret_cast<const uint32_t*>(&hi);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
                        }
                    }
                    a_store(Hreg, i * 32 + lr, j, q, pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                }
        RF_TS(7 + 4 * c);
        if (MLP) {
            if (SAVE) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) g_store(j, q, gq[0][j][q][0], gq[0][j][q][1]);
            }
            // fc2 over this hidden chunk.  SAVE: k = 0 copies gelu(h) (8 stores) and the first row half of gelu'(h) (4), k = 1 stages the
            // second half, k = 2 copies it (4)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (SAVE) {
                    if (k == 1 || k == 2) RF_WAIT(20); else if (k == 3 || k == 4) RF_WAIT(12); else RF_WAIT(8);
                } else RF_WAIT(8);
                RF_BARRIER();
                if (SAVE) {
                    if (k == 0) {
#pragma unroll
                        for (int it = 0; it < 8; ++it) copy_out(Hreg, g.h_out + c * 256, NH, it);
#pragma unroll
                        for (int it = 0; it < 4; ++it) g_copy_out(g.hgrad_out, 0, c, it);
                    } else if (k == 1) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) g_store(j, q, gq[1][j][q][0], gq[1][j][q][1]);
                    } else if (k == 2) {
#pragma unroll
                        for (int it = 0; it < 4; ++it) g_copy_out(g.hgrad_out, 1, c, it);
                    }
                }
                step(Hreg, k, acc2);
                if (k == 3) RF_TS(8 + 4 * c);
            }
            RF_TS(9 + 4 * c);
        }
    }
    RF_TS(22);
    RF_WAIT(0);                  // the tail's dummy tiles have landed: the ring is free (and nothing lands after the workgroup has left)
    RF_BARRIER();                // every wave is done with Areg / Hreg
    if (MLP) {
        // y = fc2 + b2 + x: fp32 through Areg + Hreg, the bf16 copy through the ring area
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = wave * 64 + j * 32 + q * 8 + h * 4;
                    const float4 b = *reinterpret_cast<const float4*>(vec + 768 + col);
                    const float o0 = acc2[i][j][4 * q] + b.x + x[i][j][4 * q], o1 = acc2[i][j][4 * q + 1] + b.y + x[i][j][4 * q + 1];
                    const float o2 = acc2[i][j][4 * q + 2] + b.z + x[i][j][4 * q + 2], o3 = acc2[i][j][4 * q + 3] + b.w + x[i][j][4 * q + 3];
                    f32_stage(i, j, q, o0, o1, o2, o3);
                    if (g.y16) a_store(smem + RF_OFF_W, i * 32 + lr, j, q, pack_bf2(o0, o1), pack_bf2(o2, o3));
                }
        RF_TS(23);
        RF_LDS_SYNC();
#pragma unroll
        for (int it = 0; it < 16; ++it) f32_copy_out(g.y32, it);
        if (g.y16) {
#pragma unroll
            for (int it = 0; it < 8; ++it) copy_out(smem + RF_OFF_W, g.y16, 256, it);
        }
        RF_TS(24);
#ifdef MRG_RF_TS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RF_TS(25);
        RF_LDS_SYNC();
        if (SAVE && tid < 64) {
            const unsigned long long* t = reinterpret_cast<const unsigned long long*>(smem + RF_OFF_TS);
            g.rstd[m0 + tid] = (float)(long long)(t[tid] - t[0]);
        }
#endif
    } else {
#pragma unroll
        for (int it = 0; it < 8; ++it) copy_out(Hreg, g.h_out + (NC - 1) * 256, NH, it);
    }
}


}  // namespace

extern "C" int mrg_rowfuse_fwd(const mrg_rowfuse_desc* d, hipStream_t stream) {
    MRG_REQUIRE(d != nullptr, MRG_ERR_ARG, "mrg_rowfuse_fwd: null descriptor");
    MRG_REQUIRE(d->M > 0 && d->M % 64 == 0, MRG_ERR_SHAPE, "mrg_rowfuse_fwd: M must be a positive multiple of 64 (M=%d)", d->M);
    MRG_REQUIRE(d->NH >= 256 && d->NH <= 1024 && d->NH % 256 == 0, MRG_ERR_SHAPE, "mrg_rowfuse_fwd: NH must be 256, 512, 768 or 1024 (NH=%d)", d->NH);
    MRG_REQUIRE(d->gamma && d->beta && d->w1 && d->b1, MRG_ERR_ARG, "mrg_rowfuse_fwd: LayerNorm parameters and w1 / b1 are required");
    const bool proj = d->a0 != nullptr, mlp = d->w2 != nullptr;
    MRG_REQUIRE(proj == mlp, MRG_ERR_UNSUPPORTED, "mrg_rowfuse_fwd: either (a0, w0, b0, w2, b2) for the post-attention chain or none of them");
    const bool save = d->ln_out != nullptr;
    if (proj) {
        MRG_REQUIRE(d->w0 && d->b0 && d->b2 && d->y32, MRG_ERR_ARG, "mrg_rowfuse_fwd: w0 / b0 / b2 / y32 are required with a0");
        MRG_REQUIRE(!save || (d->x_out && d->mean && d->rstd && d->h_out && d->hgrad_out), MRG_ERR_ARG,
                    "mrg_rowfuse_fwd: ln_out (training) needs x_out, mean, rstd, h_out and hgrad_out as well");
    } else {
        MRG_REQUIRE(d->resid && d->h_out, MRG_ERR_ARG, "mrg_rowfuse_fwd: resid (the fp32 rows) and h_out are required without a0");
        MRG_REQUIRE(!save || (d->mean && d->rstd), MRG_ERR_ARG, "mrg_rowfuse_fwd: ln_out (training) needs mean and rstd as well");
    }
    RowFuseArgs g;
    g.M = d->M; g.NH = d->NH; g.eps = d->eps;
    g.a0 = (const bf16_t*)d->a0; g.w0 = (const bf16_t*)d->w0; g.b0 = d->b0; g.resid = d->resid; g.x_out = d->x_out;
    g.gamma = d->gamma; g.beta = d->beta; g.mean = d->mean; g.rstd = d->rstd; g.ln_out = (bf16_t*)d->ln_out;
    g.w1 = (const bf16_t*)d->w1; g.b1 = d->b1; g.h_out = (bf16_t*)d->h_out; g.hgrad_out = (_Float16*)d->hgrad_out;
    g.w2 = (const bf16_t*)d->w2; g.b2 = d->b2; g.y32 = d->y32; g.y16 = (bf16_t*)d->y16;
    const dim3 grid(d->M / 64), block(256);
#define MRG_RF(P_, M_, S_) do { const void* fn = (const void*)rowfuse_fwd_kernel<P_, M_, S_>; \
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, RF_LDS_BYTES); \
        hipLaunchKernelGGL((rowfuse_fwd_kernel<P_, M_, S_>), grid, block, (size_t)RF_LDS_BYTES, stream, g); } while (0)
    if (proj) { if (save) MRG_RF(true, true, true); else MRG_RF(true, true, false); }
    else { if (save) MRG_RF(false, false, true); else MRG_RF(false, false, false); }
#undef MRG_RF
    return mrg_check_launch("mrg_rowfuse_fwd");
}
