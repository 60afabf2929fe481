// Random token masking of MIRAGEModel.generate_random_masks (mirage/model.py:168-239) as a pure function of its
// three random inputs: Dirichlet shares p [B, ndom], per-modality noise u [B, T], global
// tie-break noise g [B, T].  One workgroup per sample; argsorts are bitonic sorts of (key, index) pairs in LDS
// (lexicographic compare == stable sort, so integer outputs are bit-exact w.r.t. the reference's argsort for
// distinct keys).  Outputs keep the reference's int64 dtype.
#include "common.h"
#include "mirage_hip.h"

namespace {

constexpr int MAXDOM = MRG_MAX_DOMAINS;
constexpr int MAXT = 2048;

struct MaskArgs {
    const float* p; const float* u; const float* g;
    int B, T, K, ndom;
    int dom_start[MAXDOM + 1];
    long* task_mask; long* ids_keep; long* ids_restore;
};

__device__ __forceinline__ bool pair_less(float ka, int ia, float kb, int ib) { return ka < kb || (ka == kb && ia < ib); }

// bitonic sort of n2 (power of two) pairs in shared memory, ascending; all threads of the block participate
__device__ void bitonic_sort(float* key, int* idx, int n2) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = ((i & k) == 0);
                    const float ka = key[i], kb = key[ixj];
                    const int ia = idx[i], ib = idx[ixj];
                    const bool sw = up ? pair_less(kb, ib, ka, ia) : pair_less(ka, ia, kb, ib);
                    if (sw) { key[i] = kb; key[ixj] = ka; idx[i] = ib; idx[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(512) void maskgen_kernel(const MaskArgs a) {
    __shared__ float key[MAXT];
    __shared__ int idx[MAXT];
    __shared__ int mask_all[MAXT];
    const int b = blockIdx.x;
    // step 1+2: per modality: perm = argsort(u_d); mask[j] = perm[j] < q ? 0 : 1        (model.py:209-221)
    for (int d = 0; d < a.ndom; ++d) {
        const int s = a.dom_start[d], n = a.dom_start[d + 1] - s;
        int n2 = 1;
        while (n2 < n) n2 <<= 1;
        for (int i = threadIdx.x; i < n2; i += blockDim.x) {
            key[i] = i < n ? a.u[(long)b * a.T + s + i] : INFINITY;
            idx[i] = i;
        }
        __syncthreads();
        bitonic_sort(key, idx, n2);
        const int q = __float2int_rn(a.p[(long)b * a.ndom + d] * (float)a.K);   // torch.round: half to even
        for (int i = threadIdx.x; i < n; i += blockDim.x) mask_all[s + i] = idx[i] < q ? 0 : 1;
        __syncthreads();
    }
    // step 3: ids_shuffle = argsort(mask_all + g)                                        (model.py:224-227)
    int n2 = 1;
    while (n2 < a.T) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        key[i] = i < a.T ? (float)mask_all[i] + a.g[(long)b * a.T + i] : INFINITY;
        idx[i] = i;
    }
    __syncthreads();
    bitonic_sort(key, idx, n2);
    // step 4: ids_restore = inverse permutation; final mask = ids_restore >= K           (model.py:226-237)
    for (int i = threadIdx.x; i < a.T; i += blockDim.x) {
        const int src = idx[i];
        a.ids_restore[(long)b * a.T + src] = i;
        a.task_mask[(long)b * a.T + src] = i < a.K ? 0 : 1;
        if (i < a.K) a.ids_keep[(long)b * a.K + i] = src;
    }
}

}  // namespace

extern "C" int mrg_maskgen(const float* p, const float* u, const float* g, int B, int T, int K, int ndom, const int* dom_start_host,
                           long* task_mask, long* ids_keep, long* ids_restore, hipStream_t stream) {
    MRG_REQUIRE(p && u && g && dom_start_host && task_mask && ids_keep && ids_restore, MRG_ERR_ARG, "mrg_maskgen: null pointer");
    MRG_REQUIRE(ndom >= 1 && ndom <= MAXDOM, MRG_ERR_UNSUPPORTED, "mrg_maskgen: ndom out of range");
    MRG_REQUIRE(T >= 1 && T <= MAXT && K >= 0 && K <= T && B > 0, MRG_ERR_SHAPE, "mrg_maskgen: need 1 <= T <= %d, 0 <= K <= T", MAXT);
    MRG_REQUIRE(dom_start_host[0] == 0 && dom_start_host[ndom] == T, MRG_ERR_SHAPE, "mrg_maskgen: dom_start must span [0, T]");
    MaskArgs a;
    a.p = p; a.u = u; a.g = g; a.B = B; a.T = T; a.K = K; a.ndom = ndom;
    for (int i = 0; i <= MAXDOM; ++i) a.dom_start[i] = i <= ndom ? dom_start_host[i] : T;
    a.task_mask = task_mask; a.ids_keep = ids_keep; a.ids_restore = ids_restore;
    hipLaunchKernelGGL(maskgen_kernel, dim3(B), dim3(512), 0, stream, a);
    return mrg_check_launch("mrg_maskgen");
}
