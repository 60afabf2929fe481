// Random token masking of MIRAGEModel.generate_random_masks (mirage/model.py:168-239) as a function of its three random
// inputs: Dirichlet shares p [B, ndom], per-modality noise u [B, T], global tie-break noise g [B, T].
// One workgroup per sample.  The reference's argsorts are stable sorts of <= 2048 keys; here every key's position is its RANK
// (the number of keys that precede it in (key, index) order), counted by the key's own thread over broadcast LDS reads:
// three barriers per sample instead of the ~120 barrier-separated stages of the bitonic network this replaces (39 -> ~6 us at
// C2, where the launch sits alone at the head of the step).  Lexicographic (key, index) order == stable sort, so the integer
// outputs are bit-exact w.r.t. the reference's argsort.  Outputs keep the reference's int64 dtype.
//
// Two noise sources: explicit (p, u, g) tensors (the parity path), or drawn in the kernel from Philox4x32-10 keyed by a device
// {seed, step} pair (the production path: one launch instead of seven torch RNG kernels, nothing for a hipGraph to patch): u and
// g are 24-bit uniforms like torch.rand, p = normalised Gamma(conc) draws (Marsaglia & Tsang 2000) like torch's Dirichlet.
#include "common.h"
#include "mirage_hip.h"

namespace {

constexpr int MAXDOM = MRG_MAX_DOMAINS;
constexpr int MAXT = 2048;

struct MaskArgs {
    const float* p; const float* u; const float* g;     // explicit noise (rng == nullptr)
    const float* conc;                                  // rng path: Dirichlet concentrations [B, ndom]
    unsigned long long* rng;                            // rng path: device [3] = seed, step, finished-workgroup counter
    float* p_out; float* u_out; float* g_out;           // rng path, optional: the noise that was drawn
    int B, T, K, ndom;
    int dom_start[MAXDOM + 1];
    long* task_mask; long* ids_keep; long* ids_restore;
};

__device__ __forceinline__ float u24(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }            // [0, 1)
__device__ __forceinline__ float u24_open(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }   // (0, 1]

// Gamma(alpha, 1) by Marsaglia & Tsang's squeeze-free acceptance test; alpha < 1 through Gamma(alpha + 1) * U^(1 / alpha).
// Philox stream 2, one counter per (sample, domain, attempt); 16 attempts accept with probability > 1 - 1e-20.
__device__ float gamma_draw(float alpha, unsigned long long seed, uint32_t step, uint32_t sample_dom) {
    const float a = alpha < 1.0f ? alpha + 1.0f : alpha;
    const float d = a - (1.0f / 3.0f), c = rsqrtf(9.0f * d);
    float out = d, boost = 1.0f;
    for (int attempt = 0; attempt < 16; ++attempt) {
        uint32_t r[4];
        philox4x32_10(sample_dom * 16u + (uint32_t)attempt, 0u, 2u, step, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        const float u1 = u24_open(r[0]), u2 = u24(r[1]);
        const float x = sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);      // Box-Muller
        const float t = 1.0f + c * x, v = t * t * t;
        if (attempt == 0 && alpha < 1.0f) boost = __powf(u24_open(r[3]), 1.0f / alpha);
        if (v > 0.0f && __logf(u24_open(r[2])) < 0.5f * x * x + d - d * v + d * __logf(v)) { out = d * v; break; }
    }
    return out * boost;
}

// (key, index) as ONE unsigned 64-bit sort key: the float's bits mapped to an order-preserving unsigned integer in the high word,
// the index in the low word -- a rank is then a count of single 64-bit compares instead of three compares and two logic ops per pair
// (the ranks are VALU-bound: T^2 pairs per sample on one CU).
typedef unsigned long long u64;
__device__ __forceinline__ u64 sort_key(float k, int i) {
    const uint32_t bits = __float_as_uint(k);
    const uint32_t ord = bits ^ ((bits & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
    return ((u64)ord << 32) | (uint32_t)i;
}
// number of keys of key[lo .. hi) below `mine`; every lane of a wave walks the same addresses (LDS broadcast reads)
__device__ __forceinline__ int rank_of(const u64* key, int lo, int hi, u64 mine) {
    int r = 0, j = lo;
    if (j < hi && (j & 1)) { r += key[j] < mine; ++j; }
    for (; j + 8 <= hi; j += 8) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 k0 = *reinterpret_cast<const u64x2*>(key + j), k1 = *reinterpret_cast<const u64x2*>(key + j + 2);
        const u64x2 k2 = *reinterpret_cast<const u64x2*>(key + j + 4), k3 = *reinterpret_cast<const u64x2*>(key + j + 6);
        r += (k0.x < mine) + (k0.y < mine) + (k1.x < mine) + (k1.y < mine) + (k2.x < mine) + (k2.y < mine) + (k3.x < mine) + (k3.y < mine);
    }
    for (; j < hi; ++j) r += key[j] < mine;
    return r;
}

template <bool RNG>
__global__ __launch_bounds__(1024) void maskgen_kernel(const MaskArgs a) {
    __shared__ __attribute__((aligned(16))) u64 key[MAXT];
    __shared__ float tie[MAXT];
    __shared__ int perm[MAXT];
    __shared__ float share[MAXDOM];
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned long long seed = 0;
    uint32_t step = 0;
    if (RNG) { seed = a.rng[0]; step = (uint32_t)a.rng[1]; }
    // ---- the three noise tensors
    if (RNG) {
        const int quads = (a.T + 3) >> 2;                                   // per-sample counter range: quads of tokens
        for (int q = tid; q < quads; q += blockDim.x) {
            uint32_t ru[4], rg[4];
            const unsigned long long ctr = (unsigned long long)b * quads + q;
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, step, (uint32_t)seed, (uint32_t)(seed >> 32), ru);
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, step, (uint32_t)seed, (uint32_t)(seed >> 32), rg);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = q * 4 + e;
                if (t < a.T) {
                    const float uu = u24(ru[e]), gg = u24(rg[e]);
                    key[t] = sort_key(uu, t); tie[t] = gg;
                    if (a.u_out) a.u_out[(long)b * a.T + t] = uu;
                    if (a.g_out) a.g_out[(long)b * a.T + t] = gg;
                }
            }
        }
        if (tid < a.ndom) share[tid] = fmaxf(gamma_draw(a.conc[(long)b * a.ndom + tid], seed, step, (uint32_t)(b * MAXDOM + tid)), 1e-30f);
    } else {
        for (int t = tid; t < a.T; t += blockDim.x) { key[t] = sort_key(a.u[(long)b * a.T + t], t); tie[t] = a.g[(long)b * a.T + t]; }
        if (tid < a.ndom) share[tid] = a.p[(long)b * a.ndom + tid];
    }
    __syncthreads();
    // ---- steps 1 + 2, per modality: perm = argsort(u_d); mask[j] = perm[j] < q ? 0 : 1            (model.py:209-221)
    for (int t = tid; t < a.T; t += blockDim.x) {
        int d = 0;
        for (int i = 1; i < a.ndom; ++i) d += t >= a.dom_start[i] ? 1 : 0;
        const int s = a.dom_start[d];
        perm[s + rank_of(key, s, a.dom_start[d + 1], key[t])] = t - s;
    }
    float tot = 0.f;
    for (int d = 0; d < a.ndom; ++d) tot += share[d];                        // rng path: Gamma draws -> Dirichlet shares
    __syncthreads();
    if (RNG && tid < a.ndom) {
        share[tid] = share[tid] / tot;
        if (a.p_out) a.p_out[(long)b * a.ndom + tid] = share[tid];
    }
    __syncthreads();
    // ---- step 3: ids_shuffle = argsort(mask_all + g)                                            (model.py:224-227)
    for (int t = tid; t < a.T; t += blockDim.x) {
        int d = 0;
        for (int i = 1; i < a.ndom; ++i) d += t >= a.dom_start[i] ? 1 : 0;
        const int q = __float2int_rn(share[d] * (float)a.K);                 // torch.round: half to even
        key[t] = sort_key((perm[t] < q ? 0.0f : 1.0f) + tie[t], t);
    }
    __syncthreads();
    // ---- step 4: ids_restore = inverse permutation = the rank itself; final mask = ids_restore >= K   (model.py:226-237)
    for (int t = tid; t < a.T; t += blockDim.x) {
        const int r = rank_of(key, 0, a.T, key[t]);
        a.ids_restore[(long)b * a.T + t] = r;
        a.task_mask[(long)b * a.T + t] = r < a.K ? 0 : 1;
        if (r < a.K) a.ids_keep[(long)b * a.K + r] = t;
    }
    // ---- the workgroup that finishes last advances the step (every workgroup has read it by then)
    if (RNG && tid == 0) {
        __threadfence();
        if (atomicAdd(&a.rng[2], 1ull) == (unsigned long long)(a.B - 1)) { a.rng[2] = 0ull; a.rng[1] = a.rng[1] + 1ull; }
    }
}

int check_shape(const char* who, int B, int T, int K, int ndom, const int* dom_start_host) {
    MRG_REQUIRE(dom_start_host, MRG_ERR_ARG, "%s: null pointer", who);
    MRG_REQUIRE(ndom >= 1 && ndom <= MAXDOM, MRG_ERR_UNSUPPORTED, "%s: ndom out of range", who);
    MRG_REQUIRE(T >= 1 && T <= MAXT && K >= 0 && K <= T && B > 0, MRG_ERR_SHAPE, "%s: need 1 <= T <= %d, 0 <= K <= T", who, MAXT);
    MRG_REQUIRE(dom_start_host[0] == 0 && dom_start_host[ndom] == T, MRG_ERR_SHAPE, "%s: dom_start must span [0, T]", who);
    for (int i = 0; i < ndom; ++i)
        MRG_REQUIRE(dom_start_host[i] <= dom_start_host[i + 1], MRG_ERR_SHAPE, "%s: dom_start must not decrease", who);
    return MRG_OK;
}

}  // namespace

extern "C" int mrg_maskgen(const float* p, const float* u, const float* g, int B, int T, int K, int ndom, const int* dom_start_host,
                           long* task_mask, long* ids_keep, long* ids_restore, hipStream_t stream) {
    MRG_REQUIRE(p && u && g && task_mask && ids_keep && ids_restore, MRG_ERR_ARG, "mrg_maskgen: null pointer");
    if (const int rc = check_shape("mrg_maskgen", B, T, K, ndom, dom_start_host)) return rc;
    MaskArgs a = {};
    a.p = p; a.u = u; a.g = g; a.B = B; a.T = T; a.K = K; a.ndom = ndom;
    for (int i = 0; i <= MAXDOM; ++i) a.dom_start[i] = i <= ndom ? dom_start_host[i] : T;
    a.task_mask = task_mask; a.ids_keep = ids_keep; a.ids_restore = ids_restore;
    hipLaunchKernelGGL(maskgen_kernel<false>, dim3(B), dim3(T > 512 ? 1024 : 512), 0, stream, a);      // one token per thread (the rank loops are O(T) per token)
    return mrg_check_launch("mrg_maskgen");
}

extern "C" int mrg_maskgen_rng(const float* conc, unsigned long long* rng, int B, int T, int K, int ndom, const int* dom_start_host,
                               long* task_mask, long* ids_keep, long* ids_restore, float* p_out, float* u_out, float* g_out,
                               hipStream_t stream) {
    MRG_REQUIRE(conc && rng && task_mask && ids_keep && ids_restore, MRG_ERR_ARG, "mrg_maskgen_rng: null pointer");
    if (const int rc = check_shape("mrg_maskgen_rng", B, T, K, ndom, dom_start_host)) return rc;
    MaskArgs a = {};
    a.conc = conc; a.rng = rng; a.p_out = p_out; a.u_out = u_out; a.g_out = g_out;
    a.B = B; a.T = T; a.K = K; a.ndom = ndom;
    for (int i = 0; i <= MAXDOM; ++i) a.dom_start[i] = i <= ndom ? dom_start_host[i] : T;
    a.task_mask = task_mask; a.ids_keep = ids_keep; a.ids_restore = ids_restore;
    hipLaunchKernelGGL(maskgen_kernel<true>, dim3(B), dim3(T > 512 ? 1024 : 512), 0, stream, a);
    return mrg_check_launch("mrg_maskgen_rng");
}
