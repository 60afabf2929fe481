// Gather-then-embed input adapters.
//
// The reference projects ALL T patch tokens of every modality (Conv2d k=s=P, mirage/input_adapters.py:101,229) and
// then throws 75-87 % of them away with torch.gather (mirage/model.py:384-387).  Here only the K kept tokens per
// sample are ever read: a plan kernel sorts the kept (sample, slot) pairs by modality, im2col kernels pack just
// those patches as bf16 GEMM rows (the layer-map adapter fuses the nn.Embedding(13,64) lookup,
// input_adapters.py:226, into the packing), and the patch GEMM's epilogue adds bias + positional embedding and
// scatters each row straight to its slot of the encoder input [B, K+ng, D] (global tokens last,
// mirage/model.py:390-391).  Counts per modality are data-dependent and stay on the device (m_dev / k_dev of
// mrg_gemm), so the step needs no host synchronisation and is hipGraph-capturable.
#include "common.h"
#include "mirage_hip.h"

namespace {

constexpr int MAXDOM = MRG_MAX_DOMAINS;

struct PlanArgs {
    const long* ids_keep;   // [B, K]
    int B, K, N, ndom;
    int dom_start[MAXDOM + 1];
    int* rows;    // [ndom][cap] destination row b*N + k
    int* toks;    // [ndom][cap] token index inside the modality grid
    int* srcb;    // [ndom][cap] sample index
    int* counts;  // [ndom]
};

__device__ __forceinline__ int find_dom(const int* dom_start, int ndom, int t) {
    int d = 0;
#pragma unroll
    for (int i = 1; i < MAXDOM; ++i) d += (i < ndom && t >= dom_start[i]) ? 1 : 0;
    return d;
}

// single block of 16 waves: deterministic (flat (b, k) order) compaction per modality.  The flat list is cut into chunks of 64
// entries; a round covers 16 waves x PLAN_IT chunks = 8192 entries whose ids every lane requests up front (one memory round trip,
// kept in registers for both passes).  A chunk's per-modality counts and every entry's position inside its chunk are wave ballots,
// the chunk bases one 128-wide exclusive scan: three barriers per round.  (The version this replaces -- per-thread counts, a
// 10-step Hillis-Steele scan, two dependent passes over global memory -- took 17 us at the head of every step.)
constexpr int PLAN_IT = 8, PLAN_CHUNKS = 16 * PLAN_IT;

__global__ __launch_bounds__(1024) void embed_plan_kernel(const PlanArgs a) {
    __shared__ int cnt[MAXDOM][PLAN_CHUNKS];      // chunk counts, then exclusive chunk bases
    __shared__ int wave_tot[MAXDOM][2];
    __shared__ int carry[MAXDOM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = a.B * a.K, cap = total;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (tid < MAXDOM) carry[tid] = 0;
    for (int round0 = 0; round0 < total; round0 += PLAN_CHUNKS * 64) {
        int tok[PLAN_IT], dom[PLAN_IT];
#pragma unroll
        for (int it = 0; it < PLAN_IT; ++it) {
            const int i = round0 + (wave + 16 * it) * 64 + lane;
            tok[it] = i < total ? (int)a.ids_keep[i] : -1;
        }
        __syncthreads();                              // the previous round is done with cnt / wave_tot; carry is visible
        // (1) per-chunk counts
#pragma unroll
        for (int it = 0; it < PLAN_IT; ++it) {
            dom[it] = tok[it] >= 0 ? find_dom(a.dom_start, a.ndom, tok[it]) : -1;
#pragma unroll
            for (int d = 0; d < MAXDOM; ++d) {
                const unsigned long long m = __ballot(dom[it] == d);
                if (lane == 0) cnt[d][wave + 16 * it] = __popcll(m);
            }
        }
        __syncthreads();
        // (2) exclusive scan over the 128 chunks by the first two waves
        int mine[MAXDOM], incl[MAXDOM];
        if (tid < PLAN_CHUNKS) {
#pragma unroll
            for (int d = 0; d < MAXDOM; ++d) {
                mine[d] = cnt[d][tid];
                int v = mine[d];
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(v, o, 64); if (lane >= o) v += up; }
                incl[d] = v;
                if (lane == 63) wave_tot[d][wave] = v;
            }
        }
        __syncthreads();
        if (tid < PLAN_CHUNKS) {
#pragma unroll
            for (int d = 0; d < MAXDOM; ++d) cnt[d][tid] = carry[d] + (wave == 1 ? wave_tot[d][0] : 0) + incl[d] - mine[d];
        }
        __syncthreads();
        // (3) positions
#pragma unroll
        for (int it = 0; it < PLAN_IT; ++it) {
            const int c = wave + 16 * it;
            int p = 0;
#pragma unroll
            for (int d = 0; d < MAXDOM; ++d) {
                const unsigned long long m = __ballot(dom[it] == d);
                if (dom[it] == d) p = cnt[d][c] + __popcll(m & lt);
            }
            if (dom[it] >= 0) {
                const int i = round0 + c * 64 + lane;
                const int b = i / a.K, k = i - b * a.K, dm = dom[it];
                a.rows[dm * cap + p] = b * a.N + k;
                a.toks[dm * cap + p] = tok[it] - a.dom_start[dm];
                a.srcb[dm * cap + p] = b;
            }
        }
        __syncthreads();
        if (tid < MAXDOM) carry[tid] += wave_tot[tid][0] + wave_tot[tid][1];
    }
    __syncthreads();
    if (tid < a.ndom) a.counts[tid] = carry[tid];
}

// image patches: img fp32 [B, C, H, W]; row i = patch (srcb[i], toks[i]) flattened (c, ph, pw) -> bf16 [cap, C*P*P]
__global__ __launch_bounds__(256) void im2col_image_kernel(const float* __restrict__ img, const void* const* __restrict__ slot,
                                                           const int* __restrict__ srcb,
                                                           const int* __restrict__ toks, const int* __restrict__ count,
                                                           int C, int H, int W, int P, int G, bf16_t* __restrict__ out) {
    if (slot) img = static_cast<const float*>(*slot);        // input slot: the batch is read where it lies (mirage_hip.h, "Input slots")
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= *count) return;
    const int b = srcb[i], t = toks[i];
    const int ty = t / G, tx = t - ty * G;
    const int PP = P * P, L = C * PP;
    bf16_t* o = out + (long)i * L;
    for (int e = lane * 4; e < L; e += 256) {
        const int c = e / PP, r = e - c * PP;
        const int ph = r / P, pw = r - ph * P;
        const float4 v = *reinterpret_cast<const float4*>(img + (((long)b * C + c) * H + ty * P + ph) * W + tx * P + pw);
        *reinterpret_cast<uint2*>(o + e) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
}

// layer-map patches: labels int64 [B, Hs, Ws]; out[i][ch*PP + ph*P + pw] = class_emb[label][ch]
__global__ __launch_bounds__(256) void im2col_layermap_kernel(const long* __restrict__ labels, const void* const* __restrict__ slot,
                                                              const float* __restrict__ class_emb,
                                                              const int* __restrict__ srcb, const int* __restrict__ toks,
                                                              const int* __restrict__ count, int ncls, int E, int Hs, int Ws,
                                                              int P, int G, bf16_t* __restrict__ out) {
    if (slot) labels = static_cast<const long*>(*slot);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* emb = reinterpret_cast<float*>(smem);           // [ncls][E]
    int* lab = reinterpret_cast<int*>(emb + ncls * E);     // [P*P]
    const int i = blockIdx.x;
    if (i >= *count) return;
    const int PP = P * P;
    for (int j = threadIdx.x; j < ncls * E; j += 256) emb[j] = class_emb[j];
    const int b = srcb[i], t = toks[i];
    const int ty = t / G, tx = t - ty * G;
    for (int j = threadIdx.x; j < PP; j += 256) {
        const int ph = j / P, pw = j - ph * P;
        lab[j] = (int)labels[((long)b * Hs + ty * P + ph) * Ws + tx * P + pw];
    }
    __syncthreads();
    const int L = E * PP;
    bf16_t* o = out + (long)i * L;
    for (int e = threadIdx.x * 4; e < L; e += 1024) {
        const int ch = e / PP, px = e - ch * PP;
        const float v0 = emb[lab[px] * E + ch], v1 = emb[lab[px + 1] * E + ch];
        const float v2 = emb[lab[px + 2] * E + ch], v3 = emb[lab[px + 3] * E + ch];
        *reinterpret_cast<uint2*>(o + e) = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
    }
}

// d class_emb[c][ch] = sum over rows i < count, pixels px with label == c of dA[i][ch*PP + px].
// grid = (E, nslices): one block per (embedding channel, slice of the kept rows); deterministic tree reduction per block into
// partials[slice][c][ch], summed over slices by class_emb_grad_finish_kernel.
__global__ __launch_bounds__(256) void class_emb_grad_kernel(const bf16_t* __restrict__ dA, const long* __restrict__ labels,
                                                             const void* const* __restrict__ slot,
                                                             const int* __restrict__ srcb, const int* __restrict__ toks,
                                                             const int* __restrict__ count, int ncls, int E, int Hs, int Ws,
                                                             int P, int G, float* __restrict__ out_all) {
    if (slot) labels = static_cast<const long*>(*slot);
    __shared__ float red[256];
    const int ch = blockIdx.x;
    const int PP = P * P;
    const int groups = 256 / PP > 0 ? 256 / PP : 1;
    const int px = threadIdx.x % PP, grp = threadIdx.x / PP;
    const int ntot = *count;
    const int per = (ntot + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * per;
    const int n = min(ntot, lo + per);
    float* out = out_all + (long)blockIdx.y * ncls * E;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    if (grp < groups && PP <= 256) {
        const int ph = px / P, pw = px - ph * P;
        for (int i = lo + grp; i < n; i += groups) {
            const int b = srcb[i], t = toks[i];
            const int ty = t / G, tx = t - ty * G;
            const int lb = (int)labels[((long)b * Hs + ty * P + ph) * Ws + tx * P + pw];
            const float v = bf2f(dA[(long)i * (E * PP) + ch * PP + px]);
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] += (lb == c) ? v : 0.f;
        }
    }
    for (int c = 0; c < ncls; ++c) {
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) mine = (k == c) ? acc[k] : mine;
        red[threadIdx.x] = mine;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[c * E + ch] = red[0];
        __syncthreads();
    }
}

__global__ void class_emb_grad_finish_kernel(const float* __restrict__ partials, int nslices, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < nslices; ++z) s += partials[(long)z * n + i];
    out[i] = s;
}

// x[b*N + K + g][:] = global_tokens[g][:]
__global__ void fill_global_kernel(float* __restrict__ x, const float* __restrict__ glob, int B, int N, int K, int ng, int D) {
    const long total = (long)B * ng * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int g = (int)((i / D) % ng);
        const int b = (int)(i / ((long)D * ng));
        x[((long)b * N + K + g) * D + c] = glob[g * D + c];
    }
}

// dst[i][:] = i < *count ? src[map[i]][:] : 0 for i < cap: the kept-token rows of one modality, compacted (bf16, D % 8 == 0).  Lets the
// patch-embedding weight gradient run on the LDS-DMA GEMM kernels: a gathered (row-mapped) operand whose rows are the CONTRACTION
// index can only be fetched by the generic per-lane loader (64 TF/s on these shapes)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, long src_ld, const int* __restrict__ map,
                                                          const int* __restrict__ count_dev, int cap, int D, bf16_t* __restrict__ dst) {
    const int cpr = D >> 3;                     // 16-byte chunks per row
    const int count = min(*count_dev, cap);
    const long total = (long)cap * cpr;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int row = (int)(q / cpr), c = (int)(q - (long)row * cpr);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < count) v = *reinterpret_cast<const uint4*>(src + (long)map[row] * src_ld + c * 8);
        *reinterpret_cast<uint4*>(dst + (long)row * D + c * 8) = v;
    }
}

}  // namespace

extern "C" int mrg_gather_rows_bf16(const void* src_bf16, long src_ld, const int* map, const int* count_dev, int cap, int D, void* dst_bf16,
                                    hipStream_t stream) {
    MRG_REQUIRE(src_bf16 && map && count_dev && dst_bf16, MRG_ERR_ARG, "mrg_gather_rows_bf16: null pointer");
    MRG_REQUIRE(cap > 0 && D > 0 && D % 8 == 0 && src_ld % 8 == 0, MRG_ERR_SHAPE, "mrg_gather_rows_bf16: D and src_ld must be multiples of 8");
    MRG_REQUIRE((((uintptr_t)src_bf16 | (uintptr_t)dst_bf16) & 15) == 0, MRG_ERR_ARG, "mrg_gather_rows_bf16: misaligned");
    const long total = (long)cap * (D / 8);
    const unsigned nb = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nb), dim3(256), 0, stream, (const bf16_t*)src_bf16, src_ld, map, count_dev, cap, D, (bf16_t*)dst_bf16);
    return mrg_check_launch("mrg_gather_rows_bf16");
}

extern "C" int mrg_embed_plan(const long* ids_keep, int B, int K, int N, int ndom, const int* dom_start_host, int* rows, int* toks,
                              int* srcb, int* counts, hipStream_t stream) {
    MRG_REQUIRE(ids_keep && dom_start_host && rows && toks && srcb && counts, MRG_ERR_ARG, "mrg_embed_plan: null pointer");
    MRG_REQUIRE(ndom >= 1 && ndom <= MAXDOM, MRG_ERR_UNSUPPORTED, "mrg_embed_plan: ndom=%d not in 1..%d", ndom, MAXDOM);
    MRG_REQUIRE(B > 0 && K > 0 && N >= K, MRG_ERR_SHAPE, "mrg_embed_plan: bad shape");
    PlanArgs a;
    a.ids_keep = ids_keep; a.B = B; a.K = K; a.N = N; a.ndom = ndom;
    for (int i = 0; i <= MAXDOM; ++i) a.dom_start[i] = i <= ndom ? dom_start_host[i] : 0x7fffffff;
    a.rows = rows; a.toks = toks; a.srcb = srcb; a.counts = counts;
    hipLaunchKernelGGL(embed_plan_kernel, dim3(1), dim3(1024), 0, stream, a);
    return mrg_check_launch("mrg_embed_plan");
}

namespace {
struct SlotValues { const void* v[8]; };
__global__ void set_slots_kernel(const void** slots, const SlotValues vals, int n) {
    if ((int)threadIdx.x < n) slots[threadIdx.x] = vals.v[threadIdx.x];
}
}  // namespace

extern "C" int mrg_set_slots(const void** slots_dev, const void* const* values_host, int n, hipStream_t stream) {
    MRG_REQUIRE(slots_dev && values_host, MRG_ERR_ARG, "mrg_set_slots: null pointer");
    MRG_REQUIRE(n >= 1 && n <= 8, MRG_ERR_SHAPE, "mrg_set_slots: 1 <= n <= 8");
    SlotValues vals = {};
    for (int i = 0; i < n; ++i) {
        MRG_REQUIRE(values_host[i] && ((uintptr_t)values_host[i] & 15) == 0, MRG_ERR_ARG, "mrg_set_slots: address %d is null or not 16-B aligned", i);
        vals.v[i] = values_host[i];
    }
    hipLaunchKernelGGL(set_slots_kernel, dim3(1), dim3(64), 0, stream, slots_dev, vals, n);
    return mrg_check_launch("mrg_set_slots");
}

extern "C" int mrg_im2col_image(const float* img, const int* srcb, const int* toks, const int* count_dev, int cap, int C, int H,
                                int W, int P, void* out_bf16, const void* const* img_slot, hipStream_t stream) {
    MRG_REQUIRE((img || img_slot) && srcb && toks && count_dev && out_bf16, MRG_ERR_ARG, "mrg_im2col_image: null pointer");
    MRG_REQUIRE(P > 0 && H % P == 0 && W % P == 0 && P % 4 == 0 && W % 4 == 0, MRG_ERR_SHAPE,
                "mrg_im2col_image: need H,W divisible by P and P %% 4 == 0 (H=%d W=%d P=%d)", H, W, P);
    MRG_REQUIRE(((uintptr_t)img & 15) == 0, MRG_ERR_ARG, "mrg_im2col_image: image must be 16-B aligned");   // a slot's target: the caller's promise
    hipLaunchKernelGGL(im2col_image_kernel, dim3(cdiv(cap, 4)), dim3(256), 0, stream, img, img_slot, srcb, toks, count_dev, C, H, W, P, W / P,
                       (bf16_t*)out_bf16);
    return mrg_check_launch("mrg_im2col_image");
}

extern "C" int mrg_im2col_layermap(const long* labels, const float* class_emb, const int* srcb, const int* toks, const int* count_dev,
                                   int cap, int ncls, int E, int Hs, int Ws, int P, void* out_bf16, const void* const* labels_slot,
                                   hipStream_t stream) {
    MRG_REQUIRE((labels || labels_slot) && class_emb && srcb && toks && count_dev && out_bf16, MRG_ERR_ARG, "mrg_im2col_layermap: null pointer");
    MRG_REQUIRE(P > 0 && Hs % P == 0 && Ws % P == 0 && (P * P) % 4 == 0, MRG_ERR_SHAPE, "mrg_im2col_layermap: bad patch geometry");
    MRG_REQUIRE(ncls <= 16, MRG_ERR_UNSUPPORTED, "mrg_im2col_layermap: at most 16 classes");
    const size_t sh = (size_t)ncls * E * 4 + (size_t)P * P * 4;
    hipLaunchKernelGGL(im2col_layermap_kernel, dim3(cap), dim3(256), sh, stream, labels, labels_slot, class_emb, srcb, toks, count_dev, ncls, E, Hs,
                       Ws, P, Ws / P, (bf16_t*)out_bf16);
    return mrg_check_launch("mrg_im2col_layermap");
}

extern "C" int mrg_class_emb_grad_slices(void) { return 16; }

extern "C" int mrg_class_emb_grad(const void* dA_bf16, const long* labels, const int* srcb, const int* toks, const int* count_dev,
                                  int ncls, int E, int Hs, int Ws, int P, float* partials, float* out, const void* const* labels_slot,
                                  hipStream_t stream) {
    MRG_REQUIRE(dA_bf16 && (labels || labels_slot) && srcb && toks && count_dev && partials && out, MRG_ERR_ARG, "mrg_class_emb_grad: null pointer");
    MRG_REQUIRE(ncls <= 16 && P * P <= 256 && 256 % (P * P) == 0, MRG_ERR_UNSUPPORTED,
                "mrg_class_emb_grad: need <= 16 classes and P*P dividing 256");
    const int ns = mrg_class_emb_grad_slices();
    hipLaunchKernelGGL(class_emb_grad_kernel, dim3(E, ns), dim3(256), 0, stream, (const bf16_t*)dA_bf16, labels, labels_slot, srcb, toks, count_dev,
                       ncls, E, Hs, Ws, P, Ws / P, partials);
    int rc = mrg_check_launch("mrg_class_emb_grad");
    if (rc) return rc;
    hipLaunchKernelGGL(class_emb_grad_finish_kernel, dim3(cdiv(ncls * E, 256)), dim3(256), 0, stream, partials, ns, ncls * E, out);
    return mrg_check_launch("mrg_class_emb_grad/finish");
}

// ---- gradient of a learnable positional embedding (mirage/input_adapters.py:72,185: nn.Parameter(pos_emb, requires_grad=learnable_pos_emb);
// x = proj(patch) + pos_emb[token], so d pos_emb[:, token] = sum over the samples that kept the token of the embedding gradient row).
// Deterministic: the compact (sample, token) -> row lists of mrg_embed_plan are first inverted into inv[b][t] (each pair occurs at most
// once), then every token sums its rows in sample order.  The parameter is laid out [1, D, h, w]: out[d * T + t].
namespace {
__global__ void posemb_inv_kernel(const int* __restrict__ rows, const int* __restrict__ toks, const int* __restrict__ srcb,
                                  const int* __restrict__ count_dev, int cap, int T, int* __restrict__ inv) {
    const int n = count_dev ? min(*count_dev, cap) : cap;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) inv[(long)srcb[e] * T + toks[e]] = rows[e];
}
__global__ __launch_bounds__(256) void posemb_grad_kernel(const float* __restrict__ dx, long ld, int D, const int* __restrict__ inv, int B, int T,
                                                          float* __restrict__ out) {
    const int t = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const int r = inv ? inv[(long)b * T + t] : b * T + t;
            if (r >= 0) acc += dx[(long)r * ld + d];
        }
        out[(long)d * T + t] = acc;
    }
}
}  // namespace

extern "C" int mrg_posemb_grad(const float* dx, long ld, int D, const int* rows, const int* toks, const int* srcb, const int* count_dev,
                               int cap, int B, int T, int* inv_ws, float* out, hipStream_t stream) {
    MRG_REQUIRE(dx && out && D > 0 && B > 0 && T > 0, MRG_ERR_ARG, "mrg_posemb_grad: bad arguments");
    if (rows != nullptr) {
        MRG_REQUIRE(toks && srcb && inv_ws && cap > 0, MRG_ERR_ARG, "mrg_posemb_grad: the compact lists need toks, srcb and inv_ws[B * T]");
        hipError_t e = hipMemsetAsync(inv_ws, 0xFF, sizeof(int) * (size_t)B * T, stream);
        MRG_REQUIRE(e == hipSuccess, MRG_ERR_ARG, "mrg_posemb_grad: memset failed: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(posemb_inv_kernel, dim3((unsigned)((cap + 255) / 256 < 1024 ? (cap + 255) / 256 : 1024)), dim3(256), 0, stream,
                           rows, toks, srcb, count_dev, cap, T, inv_ws);
        const int rc = mrg_check_launch("mrg_posemb_grad/invert");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(posemb_grad_kernel, dim3(T), dim3(256), 0, stream, dx, ld, D, rows ? inv_ws : nullptr, B, T, out);
    return mrg_check_launch("mrg_posemb_grad");
}

extern "C" int mrg_fill_global(float* x, const float* global_tokens, int B, int N, int K, int ng, int D, hipStream_t stream) {
    MRG_REQUIRE(x && global_tokens, MRG_ERR_ARG, "mrg_fill_global: null pointer");
    MRG_REQUIRE(N == K + ng, MRG_ERR_SHAPE, "mrg_fill_global: N != K + ng");
    const long total = (long)B * ng * D;
    hipLaunchKernelGGL(fill_global_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0,
                       stream, x, global_tokens, B, N, K, ng, D);
    return mrg_check_launch("mrg_fill_global");
}
