// Shared device/host helpers for libmirage_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MRG_OK 0
#define MRG_ERR_ARG (-1)
#define MRG_ERR_SHAPE (-2)
#define MRG_ERR_UNSUPPORTED (-3)

// error plumbing (api.cpp)
extern "C" const char* mrg_last_error_string();
int mrg_set_error(int code, const char* fmt, ...);
int mrg_check_launch(const char* what);

#define MRG_REQUIRE(cond, code, ...) do { if (!(cond)) return mrg_set_error((code), __VA_ARGS__); } while (0)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16 with the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN-safe); emitted by the compiler
// from a vector fptrunc so that it also places the VALU->MFMA operand wait states itself.
typedef __bf16 mrg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mrg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const mrg_f32x2 v = {lo, hi};
    const mrg_bf16x2 r = __builtin_convertvector(v, mrg_bf16x2);
    return *reinterpret_cast<const uint32_t*>(&r);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
// 2^x on the raw transcendental unit (v_exp_f32): callers guarantee x <= ~0 or a bounded range; denormal results flush
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-form GELU and its derivative (reference: nn.GELU() default = erf form, mirage/utils.py:142,156).
// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32-roundoff class and far below the bf16 output
// resolution): one v_exp + one v_rcp + 6 FMAs instead of the branchy libm erff in a GEMM epilogue over [M, 4D] elements.
// The Gaussian factor exp(-(x/sqrt2)^2) = exp(-x^2/2) is shared with the derivative's pdf term.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x) * 0.70710678118654752f;                 // |x| / sqrt(2)
    const float e = __expf(-0.5f * x * x);                            // exp(-x^2/2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // v_rcp_f32 (1 ulp): __frcp_rn expands to the 10-instruction IEEE division
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erfv = copysignf(1.0f - poly * t * e, x);             // exp(-(|x|/sqrt2)^2) = exp(-x^2/2) = e
    cdf = 0.5f * (1.0f + erfv);
    pdf = 0.3989422804014327f * e;
}
// GELU alone (inference forward: no derivative wanted, so no Gaussian factor either): Phi(x) - 0.5 = x P(x^2) / Q(x^2) on |x| <= 5,
// clamped beyond (Phi(-5) = 2.9e-7); |error| <= 6.1e-7 in exact arithmetic, 8.9e-7 as evaluated here in fp32 (tools/fit_gelu_cdf.py,
// tests/test_host_cpu.py::test_gelu_cdf_rational_coefficients) -- the class of the erf formula above, with ONE quarter-rate instruction
// (v_rcp) and 17 issue slots per element instead of two (v_exp, v_rcp) and 24: the GELU epilogue is VALU-issue-bound and its cost adds to
// the main loop's whatever the schedule (profiles/r05_gemm_epilogue.md), so fewer instructions is the one lever it has.
#define MRG_GELU_P0 0.39894565560789313f
#define MRG_GELU_P1 0.015060430819950503f
#define MRG_GELU_P2 0.003054712685341789f
#define MRG_GELU_P3 (-8.205621222599123e-05f)
#define MRG_GELU_P4 (-1.049850072496286e-06f)
#define MRG_GELU_Q0 0.20446006730028407f
#define MRG_GELU_Q1 0.016672357899504955f
#define MRG_GELU_Q2 0.0004785680281318415f
#define MRG_GELU_Q3 (-3.6057791006505383e-05f)
__device__ __forceinline__ float gelu_cdf_rational(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -5.0f, 5.0f);
    const float u = xc * xc;
    float p = fmaf(MRG_GELU_P4, u, MRG_GELU_P3);
    p = fmaf(p, u, MRG_GELU_P2); p = fmaf(p, u, MRG_GELU_P1); p = fmaf(p, u, MRG_GELU_P0);
    float q = fmaf(MRG_GELU_Q3, u, MRG_GELU_Q2);
    q = fmaf(q, u, MRG_GELU_Q1); q = fmaf(q, u, MRG_GELU_Q0); q = fmaf(q, u, 1.0f);
    return fmaf(xc * p, __builtin_amdgcn_rcpf(q), 0.5f);
}
// the factor on the negative side is the CLAMPED argument: x * Phi(-5) would grow linearly below -5 (-2.9e-4 at x = -1e3, -inf at -inf)
// where gelu tends to 0; now |gelu_f(x)| <= 5 (Phi(-5) + 8.9e-7) = 6e-6 for every x <= -5.  (The training epilogue, GELU_SAVE_GRAD, keeps the erf form of gelu_parts: it
// needs the Gaussian factor for the derivative anyway, so train and eval forward differ by the two formulas' < 1e-6.)
__device__ __forceinline__ float gelu_f(float x) { return fmaxf(x, -5.0f) * gelu_cdf_rational(x); }
__device__ __forceinline__ float gelu_grad_f(float x) { float c, p; gelu_parts(x, c, p); return c + x * p; }

// ---- one AdamW element update (torch.optim.AdamW, mutils/optim_factory.py:180), shared by the flat-arena kernel (optim.hip) and the
// weight-gradient epilogue that applies the update to the tile it has just produced (gemm8p.hip, epi_kind 9).  No contraction: both
// callers must round identically (a fused and an unfused step are bit-identical).  hyper-parameters: lr, decay (0 for the no-decay
// segment), 1 / (1 - beta1^t), 1 / sqrt(1 - beta2^t).
__device__ __forceinline__ void adamw_elem(float& p, const float g, float& m, float& v, const float lr, const float decay, const float b1,
                                           const float b2, const float ibc1, const float isbc2, const float eps) {
#pragma clang fp contract(off)
    p *= (1.f - lr * decay);
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    const float denom = sqrtf(v) * isbc2 + eps;
    p -= lr * ibc1 * (m / denom);
}

// ---- dropout (mirage/utils.py:152,158,172,184-187: nn.Dropout after proj / fc2) ------------------------------------------
// Counter-based masks: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator behind
// torch's CUDA dropout as well) keyed by the run's seed, counted by (element quad, launch stream id, step).  One call yields the four
// 32-bit draws of 4 consecutive columns; an element is KEPT when its draw >= thresh = floor(p * 2^32).  The mask is a pure
// function of (seed, step, stream, row, col), so the backward pass regenerates it instead of storing M x N bytes, and the CPU
// oracle restates it bit for bit (oracle/mirage_oracle.py philox4x32_10).
struct MrgDropout {
    const unsigned long long* rng;   // device [2]: seed, step (nullptr: no dropout)
    uint32_t thresh;                 // floor(p * 2^32)
    float scale;                     // 1 / (1 - p)
    uint32_t stream;                 // distinguishes the launches of one step (2 * block + branch)
};
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// multiplies the 4 values of columns col..col+3 (col % 4 == 0) of row `row` of an [*, N] matrix by their mask factor (0 or scale)
__device__ __forceinline__ void dropout_apply4(const MrgDropout& d, long row, int col, int N, float (&v)[4]) {
    const unsigned long long seed = d.rng[0], step = d.rng[1];
    const unsigned long long q = ((unsigned long long)row * (unsigned long long)N + (unsigned long long)col) >> 2;
    uint32_t r[4];
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), d.stream, (uint32_t)step, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r[e] >= d.thresh ? v[e] * d.scale : 0.f;
}

// LDS pointer -> 32-bit LDS byte address
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// ds_read_b64_tr_b16: within each 16-lane group the lanes' 8-byte reads form a [4][16] bf16 block
// (lane i supplies row i>>2, cols 4*(i&3)..+3); lane l receives column (l&15), rows 0..3.
// (semantics verified on hardware by tools/probe/probe_layout.hip)
__device__ __forceinline__ s16x4 lds_read_tr16(uint32_t byte_addr) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)byte_addr);
}

// same with a compile-time byte offset (multiple of 8) that is folded into the instruction's offset field (pointer indexing
// carries the no-wrap guarantee the integer add lacks)
__device__ __forceinline__ s16x4 lds_read_tr16_off(uint32_t byte_addr, int byte_off) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)byte_addr + (byte_off >> 3));
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
