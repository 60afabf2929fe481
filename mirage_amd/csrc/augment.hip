// GPU-side pretraining augmentation: the per-sample transform of DataAugmentationForMIRAGE.__call__
// (mutils/datasets_pretrain.py:33-81) fused into one pass over the uint8 sample:
//   horizontal flip (:41-42) -> intensity shift + clip to [0,1] (:43-48, image tasks only) -> TF.affine(..., BILINEAR, fill=0)
//   (:57-64) which for tensors is torchvision's _gen_affine_grid + grid_sample(bilinear, zeros, align_corners=False) + the
//   "ones channel" fill blend out = img * mask + (1 - mask) * fill (torchvision 0.20.1 _functional_tensor.py
//   _apply_grid_transform).  Label maps go through the same BILINEAR path and are rounded back to integers
//   (_cast_squeeze_out: torch.round = round-half-even), exactly as the reference does (interpolation=affine.interpolation).
// The host supplies, per sample, the 2x3 inverse matrix already divided by (0.5 W, 0.5 H) as torchvision does before its
// bmm, the intensity shift and the flip flag; the arithmetic below keeps torchvision's order of operations
// (normalised grid -> unnormalise -> floor -> corner weights) so that results agree to float rounding.
// HBM-bound: 1 B read (uint8, ~4 taps from L2) and 4 B (image) or 8 B (label) written per output pixel.
#include "common.h"
#include "mirage_hip.h"

namespace {

struct AugArgs {
    const uint8_t* src;      // [B, H, W]
    void* dst;               // image: float [B, H, W]; label: long [B, H, W]
    const float* params;     // [B, 8]: t00 t01 t02 t10 t11 t12 (theta / (0.5 W | 0.5 H)), shift, flip
    int B, H, W;
    int do_affine;
};

template <bool LABEL>
__device__ __forceinline__ float fetch(const uint8_t* __restrict__ img, int H, int W, int y, int x, bool flip, float shift) {
    // value of the flipped / shifted source at (y, x); caller guarantees the coordinates are in range
    const int xs = flip ? W - 1 - x : x;
    const float raw = (float)img[(long)y * W + xs];
    if (LABEL) return raw;
    const float v = raw / 255.0f;                       // np.load(...).astype(np.float32) / 255.0 (datasets_pretrain.py:185)
    return shift != 0.0f ? fminf(fmaxf(v + shift, 0.0f), 1.0f) : v;
}

template <bool LABEL>
__global__ __launch_bounds__(256) void augment_kernel(const AugArgs a) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const float* p = a.params + b * 8;
    const bool flip = p[7] != 0.0f;
    const float shift = LABEL ? 0.0f : p[6];
    const uint8_t* img = a.src + (long)b * a.H * a.W;
    float out;
    if (!a.do_affine) {
        out = fetch<LABEL>(img, a.H, a.W, y, x, flip, shift);
    } else {
        // base grid (torchvision _gen_affine_grid): x_j = -W/2 + 0.5 + j, y_i = -H/2 + 0.5 + i
        const float bx = (float)x + (0.5f - 0.5f * (float)a.W);
        const float by = (float)y + (0.5f - 0.5f * (float)a.H);
        const float gx = bx * p[0] + by * p[1] + p[2];
        const float gy = bx * p[3] + by * p[4] + p[5];
        // grid_sample, align_corners=False: ((g + 1) * size - 1) / 2
        const float ix = ((gx + 1.0f) * (float)a.W - 1.0f) * 0.5f;
        const float iy = ((gy + 1.0f) * (float)a.H - 1.0f) * 0.5f;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
        const float w_nw = (fx1 - ix) * (fy1 - iy), w_ne = (ix - fx0) * (fy1 - iy);
        const float w_sw = (fx1 - ix) * (iy - fy0), w_se = (ix - fx0) * (iy - fy0);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const bool xin0 = x0 >= 0 && x0 < a.W, xin1 = x1 >= 0 && x1 < a.W;
        const bool yin0 = y0 >= 0 && y0 < a.H, yin1 = y1 >= 0 && y1 < a.H;
        float val = 0.0f, mask = 0.0f;
        if (xin0 && yin0) { val += fetch<LABEL>(img, a.H, a.W, y0, x0, flip, shift) * w_nw; mask += w_nw; }
        if (xin1 && yin0) { val += fetch<LABEL>(img, a.H, a.W, y0, x1, flip, shift) * w_ne; mask += w_ne; }
        if (xin0 && yin1) { val += fetch<LABEL>(img, a.H, a.W, y1, x0, flip, shift) * w_sw; mask += w_sw; }
        if (xin1 && yin1) { val += fetch<LABEL>(img, a.H, a.W, y1, x1, flip, shift) * w_se; mask += w_se; }
        out = val * mask + (1.0f - mask) * 0.0f;        // fill = 0
    }
    const long o = ((long)b * a.H + y) * a.W + x;
    if (LABEL) reinterpret_cast<long*>(a.dst)[o] = (long)rintf(out);
    else reinterpret_cast<float*>(a.dst)[o] = out;
}

// ---- TF.resize branch (datasets_pretrain.py:66-81): raw data whose size differs from args.input_size[task] is resized AFTER the
// affine transform -- images with BILINEAR (torchvision >= 0.17 resizes tensors with antialias=True: aten
// _upsample_bilinear2d_aa), label maps with NEAREST (aten upsample_nearest2d, legacy index rule floor(dst * in / out)).
// Bilinear-antialias weights as aten/native/cuda/UpSampleBilinear2d.cu computes them: scale = in / out, support = max(scale, 1),
// center = scale * (i + 0.5), taps [int(center - support + 0.5), int(center + support + 0.5)) clipped to the image, triangle
// filter of width support around center - 0.5, normalised to sum 1; the 2-D result is the separable product.
__device__ __forceinline__ void aa_taps(int i, float scale, int in_size, int& lo, int& n, float& total, float& center, float& invscale) {
    const float support = scale >= 1.0f ? scale : 1.0f;
    invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
    center = scale * ((float)i + 0.5f);
    lo = max((int)(center - support + 0.5f), 0);
    n = min((int)(center + support + 0.5f), in_size) - lo;
    float tot = 0.0f;
    for (int j = 0; j < n; ++j) {
        const float x = fabsf(((float)(j + lo) - center + 0.5f) * invscale);
        tot += x < 1.0f ? 1.0f - x : 0.0f;
    }
    total = tot;
}

__global__ __launch_bounds__(256) void resize_bilinear_aa_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int Hi, int Wi,
                                                                 int Ho, int Wo) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wo || y >= Ho) return;
    const float sx = (float)Wi / (float)Wo, sy = (float)Hi / (float)Ho;
    int x0, nx, y0, ny;
    float itx, cx, isx, ity, cy, isy;
    aa_taps(x, sx, Wi, x0, nx, itx, cx, isx);
    aa_taps(y, sy, Hi, y0, ny, ity, cy, isy);
    const float* img = src + (long)b * Hi * Wi;
    float acc = 0.0f;
    for (int jy = 0; jy < ny; ++jy) {
        const float dy = fabsf(((float)(jy + y0) - cy + 0.5f) * isy);
        float wy = dy < 1.0f ? 1.0f - dy : 0.0f;
        if (ity != 0.0f) wy /= ity;                                      // aten: weights are divided by their sum
        float row = 0.0f;
        for (int jx = 0; jx < nx; ++jx) {
            const float dx = fabsf(((float)(jx + x0) - cx + 0.5f) * isx);
            float wx = dx < 1.0f ? 1.0f - dx : 0.0f;
            if (itx != 0.0f) wx /= itx;
            row += wx * img[(long)(y0 + jy) * Wi + x0 + jx];
        }
        acc += wy * row;
    }
    dst[((long)b * Ho + y) * Wo + x] = acc;
}

__global__ __launch_bounds__(256) void resize_nearest_kernel(const long* __restrict__ src, long* __restrict__ dst, int B, int Hi, int Wi, int Ho, int Wo) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wo || y >= Ho) return;
    const float sx = (float)Wi / (float)Wo, sy = (float)Hi / (float)Ho;
    const int xs = min((int)floorf((float)x * sx), Wi - 1), ys = min((int)floorf((float)y * sy), Hi - 1);
    dst[((long)b * Ho + y) * Wo + x] = src[((long)b * Hi + ys) * Wi + xs];
}

int launch(const uint8_t* src, void* dst, const float* params, int B, int H, int W, int do_affine, bool label, hipStream_t stream,
           const char* who) {
    MRG_REQUIRE(src && dst && params, MRG_ERR_ARG, "%s: null pointer", who);
    MRG_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535, MRG_ERR_SHAPE, "%s: bad shape B=%d H=%d W=%d", who, B, H, W);
    AugArgs a{src, dst, params, B, H, W, do_affine};
    dim3 grid(cdiv(W, 64), cdiv(H, 4), B);
    if (label) hipLaunchKernelGGL(augment_kernel<true>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(augment_kernel<false>, grid, dim3(256), 0, stream, a);
    return mrg_check_launch(who);
}

}  // namespace

extern "C" int mrg_augment_image(const uint8_t* src_u8, float* dst, const float* params, int B, int H, int W, int do_affine,
                                 hipStream_t stream) {
    return launch(src_u8, dst, params, B, H, W, do_affine, false, stream, "mrg_augment_image");
}

extern "C" int mrg_augment_label(const uint8_t* src_u8, long* dst, const float* params, int B, int H, int W, int do_affine,
                                 hipStream_t stream) {
    return launch(src_u8, dst, params, B, H, W, do_affine, true, stream, "mrg_augment_label");
}

extern "C" int mrg_resize_image(const float* src, float* dst, int B, int Hi, int Wi, int Ho, int Wo, hipStream_t stream) {
    MRG_REQUIRE(src && dst, MRG_ERR_ARG, "mrg_resize_image: null pointer");
    MRG_REQUIRE(B > 0 && B <= 65535 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, MRG_ERR_SHAPE, "mrg_resize_image: bad shape");
    hipLaunchKernelGGL(resize_bilinear_aa_kernel, dim3(cdiv(Wo, 64), cdiv(Ho, 4), B), dim3(256), 0, stream, src, dst, B, Hi, Wi, Ho, Wo);
    return mrg_check_launch("mrg_resize_image");
}

extern "C" int mrg_resize_label(const long* src, long* dst, int B, int Hi, int Wi, int Ho, int Wo, hipStream_t stream) {
    MRG_REQUIRE(src && dst, MRG_ERR_ARG, "mrg_resize_label: null pointer");
    MRG_REQUIRE(B > 0 && B <= 65535 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, MRG_ERR_SHAPE, "mrg_resize_label: bad shape");
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(cdiv(Wo, 64), cdiv(Ho, 4), B), dim3(256), 0, stream, src, dst, B, Hi, Wi, Ho, Wo);
    return mrg_check_launch("mrg_resize_label");
}
