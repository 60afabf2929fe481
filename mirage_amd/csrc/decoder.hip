// Decoder query / context assembly of SpatialOutputAdapter (mirage/output_adapters.py:164-246), forward + backward.
//
// The reference materialises [B, T, 256] three times (mask-token fill + unshuffle gather, + context embeddings,
// re-gather of the kept tokens).  Since ids_restore[b, ids_keep[b,k]] == k, the whole thing collapses to two row
// kernels that read the projected encoder tokens once:
//   context[b,k]  = ctx[b,k] + task_emb[dom(t)] + pos[t - start(dom(t))],  t = ids_keep[b,k]   (k <  K)
//   context[b,K+g] = ctx[b,K+g]                                             (global tokens, no embedding)
//   queries[b,j]  = (r < K ? ctx[b,r] : mask_token) + task_emb[qdom] + pos[j],  r = ids_restore[b, q_start + j]
#include "common.h"
#include "mirage_hip.h"

namespace {

constexpr int MAXDOM = MRG_MAX_DOMAINS;

struct AsmArgs {
    const float* ctx;          // [B, N, Dd]
    const float* mask_token;   // [Dd]
    const float* task_emb[MAXDOM];  // per input domain [Dd] or null
    const float* pos;          // [Tq, Dd] token-major positional table of this adapter
    const long* ids_keep;      // [B, K]
    const long* ids_restore;   // [B, T]
    int dom_start[MAXDOM + 1];
    int ndom, qdom;            // qdom = index of this adapter's task among the input domains, or -1
    const float* q_task_emb;   // task embedding used for queries when qdom < 0 (may be null)
    int B, N, K, T, Tq, Dd;
    float* queries;            // [B, Tq, Dd]
    float* context;            // [B, N, Dd]
    // backward
    const float* d_queries; const float* d_context;
    float* d_ctx;              // [B, N, Dd]
    float* partials;           // [nblocks][MAXDOM + 1][Dd] : d task_emb[0..ndom-1], d mask_token
};

__device__ __forceinline__ int find_dom(const int* dom_start, int ndom, int t) {
    int d = 0;
#pragma unroll
    for (int i = 1; i < MAXDOM; ++i) d += (i < ndom && t >= dom_start[i]) ? 1 : 0;
    return d;
}

// one wave per output row; rows [0, B*N) are context rows, rows [B*N, B*N + B*Tq) are query rows
__global__ __launch_bounds__(256) void dec_assemble_fwd_kernel(const AsmArgs a) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nctx = (long)a.B * a.N, nq = (long)a.B * a.Tq;
    if (row >= nctx + nq) return;
    const int nv = a.Dd >> 2;
    const float4* base; const float4* te = nullptr; const float4* pe = nullptr;
    float4* out;
    if (row < nctx) {
        const int b = (int)(row / a.N), k = (int)(row - (long)b * a.N);
        base = reinterpret_cast<const float4*>(a.ctx + row * a.Dd);
        out = reinterpret_cast<float4*>(a.context + row * a.Dd);
        if (k < a.K) {
            const int t = (int)a.ids_keep[(long)b * a.K + k];
            const int dm = find_dom(a.dom_start, a.ndom, t);
            const float* tp = nullptr;
#pragma unroll
            for (int d = 0; d < MAXDOM; ++d) if (d == dm) tp = a.task_emb[d];
            te = reinterpret_cast<const float4*>(tp);
            pe = reinterpret_cast<const float4*>(a.pos + (long)(t - a.dom_start[dm]) * a.Dd);
        }
    } else {
        const long qr = row - nctx;
        const int b = (int)(qr / a.Tq), j = (int)(qr - (long)b * a.Tq);
        out = reinterpret_cast<float4*>(a.queries + qr * a.Dd);
        pe = reinterpret_cast<const float4*>(a.pos + (long)j * a.Dd);
        if (a.qdom >= 0) {
            const int r = (int)a.ids_restore[(long)b * a.T + a.dom_start[a.qdom] + j];
            base = r < a.K ? reinterpret_cast<const float4*>(a.ctx + ((long)b * a.N + r) * a.Dd)
                           : reinterpret_cast<const float4*>(a.mask_token);
            const float* tp = nullptr;
#pragma unroll
            for (int d = 0; d < MAXDOM; ++d) if (d == a.qdom) tp = a.task_emb[d];
            te = reinterpret_cast<const float4*>(tp);
        } else {
            base = reinterpret_cast<const float4*>(a.mask_token);
            te = reinterpret_cast<const float4*>(a.q_task_emb);
        }
    }
    for (int c = lane; c < nv; c += 64) {
        float4 v = base[c];
        if (te) { const float4 t = te[c]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (pe) { const float4 p = pe[c]; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
        out[c] = v;
    }
}

// d_ctx + column partial sums.  Thread = column (Dd <= 1024, looped); block = strided set of rows.
__global__ __launch_bounds__(256) void dec_assemble_bwd_kernel(const AsmArgs a) {
    const int nb = gridDim.x;
    for (int c = threadIdx.x; c < a.Dd; c += 256) {
        float acc[MAXDOM + 1];
#pragma unroll
        for (int d = 0; d <= MAXDOM; ++d) acc[d] = 0.f;
        // context rows
        for (long row = blockIdx.x; row < (long)a.B * a.N; row += nb) {
            const int b = (int)(row / a.N), k = (int)(row - (long)b * a.N);
            float g = a.d_context[row * a.Dd + c];
            if (k < a.K) {
                const int t = (int)a.ids_keep[(long)b * a.K + k];
                const int dm = find_dom(a.dom_start, a.ndom, t);
#pragma unroll
                for (int d = 0; d < MAXDOM; ++d) acc[d] += (d == dm) ? g : 0.f;
                if (a.qdom >= 0 && dm == a.qdom) g += a.d_queries[((long)b * a.Tq + (t - a.dom_start[dm])) * a.Dd + c];
            }
            a.d_ctx[row * a.Dd + c] = g;
        }
        // query rows: task embedding of the query domain + mask token
        for (long qr = blockIdx.x; qr < (long)a.B * a.Tq; qr += nb) {
            const int b = (int)(qr / a.Tq), j = (int)(qr - (long)b * a.Tq);
            const float g = a.d_queries[qr * a.Dd + c];
            bool masked = true;
            if (a.qdom >= 0) {
                masked = (int)a.ids_restore[(long)b * a.T + a.dom_start[a.qdom] + j] >= a.K;
#pragma unroll
                for (int d = 0; d < MAXDOM; ++d) acc[d] += (d == a.qdom) ? g : 0.f;
            }
            acc[MAXDOM] += masked ? g : 0.f;
        }
#pragma unroll
        for (int d = 0; d <= MAXDOM; ++d) a.partials[((long)blockIdx.x * (MAXDOM + 1) + d) * a.Dd + c] = acc[d];
    }
}

struct FinArgs { const float* partials; int nblocks, Dd; float* out[MAXDOM + 1]; };

// grid (ceil(Dd/64), MAXDOM+1), 1024 threads: lane = column, 16 waves split the partial rows
__global__ __launch_bounds__(1024) void dec_assemble_fin_kernel(const FinArgs f) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int q = blockIdx.y;
    float* o = nullptr;
#pragma unroll
    for (int d = 0; d <= MAXDOM; ++d) if (d == q) o = f.out[d];
    if (!o) return;
    float s = 0.f;
    if (c < f.Dd) {
        const float* p = f.partials + (long)q * f.Dd + c;
        const long st = (long)(MAXDOM + 1) * f.Dd;
        int b = wave;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; b + 48 < f.nblocks; b += 64) {      // four independent loads in flight per lane (the rows are one round trip each)
            s += p[b * st]; s1 += p[(b + 16) * st]; s2 += p[(b + 32) * st]; s3 += p[(b + 48) * st];
        }
        for (; b < f.nblocks; b += 16) s += p[b * st];
        s += (s1 + s2) + s3;
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < f.Dd) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        o[c] = t;
    }
}

int fill(AsmArgs& a, const mrg_dec_assemble_desc* d, const char* who) {
    MRG_REQUIRE(d != nullptr, MRG_ERR_ARG, "%s: null descriptor", who);
    MRG_REQUIRE(d->ndom >= 1 && d->ndom <= MAXDOM, MRG_ERR_UNSUPPORTED, "%s: ndom out of range", who);
    MRG_REQUIRE(d->Dd % 4 == 0 && d->Dd <= 1024, MRG_ERR_SHAPE, "%s: Dd must be a multiple of 4, <= 1024", who);
    MRG_REQUIRE(d->mask_token && d->pos && d->ids_keep && d->ids_restore, MRG_ERR_ARG, "%s: null pointer", who);
    MRG_REQUIRE(d->qdom < d->ndom, MRG_ERR_ARG, "%s: bad qdom", who);
    a.ctx = d->ctx; a.mask_token = d->mask_token; a.pos = d->pos; a.ids_keep = d->ids_keep; a.ids_restore = d->ids_restore;
    for (int i = 0; i < MAXDOM; ++i) a.task_emb[i] = i < d->ndom ? d->task_emb[i] : nullptr;
    for (int i = 0; i <= MAXDOM; ++i) a.dom_start[i] = i <= d->ndom ? d->dom_start[i] : 0x7fffffff;
    a.ndom = d->ndom; a.qdom = d->qdom; a.q_task_emb = d->q_task_emb;
    a.B = d->B; a.N = d->N; a.K = d->K; a.T = d->T; a.Tq = d->Tq; a.Dd = d->Dd;
    a.queries = d->queries; a.context = d->context; a.d_queries = d->d_queries; a.d_context = d->d_context; a.d_ctx = d->d_ctx;
    a.partials = d->partials;
    return 0;
}

}  // namespace

extern "C" int mrg_dec_assemble_fwd(const mrg_dec_assemble_desc* d, hipStream_t stream) {
    AsmArgs a;
    int rc = fill(a, d, "mrg_dec_assemble_fwd");
    if (rc) return rc;
    MRG_REQUIRE(d->ctx && d->queries && d->context, MRG_ERR_ARG, "mrg_dec_assemble_fwd: null pointer");
    const long rows = (long)a.B * (a.N + a.Tq);
    hipLaunchKernelGGL(dec_assemble_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a);
    return mrg_check_launch("mrg_dec_assemble_fwd");
}

extern "C" int mrg_dec_assemble_bwd_nblocks(int B, int N, int Tq) {
    long rows = (long)B * (N > Tq ? N : Tq);
    int nb = (int)((rows + 15) / 16);
    if (nb > 1024) nb = 1024;      // each block walks its rows one dependent round trip at a time: 4 x more blocks, 4 x shorter chains
    if (nb < 1) nb = 1;
    return nb;
}

extern "C" int mrg_dec_assemble_bwd(const mrg_dec_assemble_desc* d, hipStream_t stream) {
    AsmArgs a;
    int rc = fill(a, d, "mrg_dec_assemble_bwd");
    if (rc) return rc;
    MRG_REQUIRE(d->d_queries && d->d_context && d->d_ctx && d->partials, MRG_ERR_ARG, "mrg_dec_assemble_bwd: null pointer");
    const int nb = mrg_dec_assemble_bwd_nblocks(a.B, a.N, a.Tq);
    hipLaunchKernelGGL(dec_assemble_bwd_kernel, dim3(nb), dim3(256), 0, stream, a);
    rc = mrg_check_launch("mrg_dec_assemble_bwd");
    if (rc) return rc;
    FinArgs f;
    f.partials = d->partials; f.nblocks = nb; f.Dd = a.Dd;
    for (int i = 0; i < MAXDOM; ++i) f.out[i] = i < d->ndom ? d->d_task_emb[i] : nullptr;
    f.out[MAXDOM] = d->d_mask_token;
    hipLaunchKernelGGL(dec_assemble_fin_kernel, dim3(cdiv(a.Dd, 64), MAXDOM + 1), dim3(1024), 0, stream, f);
    return mrg_check_launch("mrg_dec_assemble_bwd/fin");
}
