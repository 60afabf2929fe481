// Hardware-semantics probe for gfx950 (run once on the GPU box; not part of the product):
//   * v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 operand + result lane maps
//   * ds_read_b64_tr_b16 transpose-read semantics (the wgrad/dgrad GEMM operand path)
//   * __builtin_amdgcn_global_load_lds (16 B) lane-linear destination
// Prints PASS/FAIL per assumption plus the raw lane maps so a wrong assumption can be re-derived.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// A: [32][16] row-major (k contiguous); B: [16][32] row-major (j contiguous); D: [32][32]
__global__ void k_mfma32(const uint16_t* A, const uint16_t* B, float* D) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[(l & 31) * 16 + 8 * (l >> 5) + j];
        b[j] = (short)B[(8 * (l >> 5) + j) * 32 + (l & 31)];
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        int col = l & 31;
        D[row * 32 + col] = c[r];
    }
}

// 16x16x32: A [16][32], B [32][16], D [16][16]
__global__ void k_mfma16(const uint16_t* A, const uint16_t* B, float* D) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[(l & 15) * 32 + 8 * (l >> 4) + j];
        b[j] = (short)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    }
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        int col = l & 15;
        D[row * 16 + col] = c[r];
    }
}

// tr read: LDS holds u16 index values. Lane i supplies byte offset addr[i]. Output 4 u16 per lane.
__global__ void k_tr(const int* addr, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    lds_s16x4* p = (lds_s16x4*)((__attribute__((address_space(3))) char*)lds + addr[l]);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

// global_load_lds 16B: each lane supplies its own global pointer; LDS destination = uniform base + lane*16.
__global__ void k_glds(const uint16_t* src, const int* src_off, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
    int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = 0xffff;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + src_off[l]),
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
    int fails = 0;
    // ---- MFMA 32x32x16
    {
        std::vector<uint16_t> A(32 * 16), B(16 * 32);
        std::vector<float> Af(32 * 16), Bf(16 * 32), D(32 * 32), R(32 * 32, 0.f);
        srand(1);
        for (int i = 0; i < 32 * 16; ++i) { float v = (rand() % 17 - 8) / 4.0f; A[i] = f2bf(v); Af[i] = bf2f(A[i]); }
        for (int i = 0; i < 16 * 32; ++i) { float v = (rand() % 13 - 6) / 2.0f; B[i] = f2bf(v); Bf[i] = bf2f(B[i]); }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += Af[i * 16 + k] * Bf[k * 32 + j]; R[i * 32 + j] = s; }
        uint16_t *dA, *dB; float* dD;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, D.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        k_mfma32<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        double md = 0; for (int i = 0; i < 1024; ++i) md = fmax(md, fabs(D[i] - R[i]));
        printf("[mfma_32x32x16_bf16 layout] maxdiff=%g %s\n", md, md < 1e-3 ? "PASS" : "FAIL"); if (md >= 1e-3) ++fails;
    }
    // ---- MFMA 16x16x32
    {
        std::vector<uint16_t> A(16 * 32), B(32 * 16);
        std::vector<float> Af(16 * 32), Bf(32 * 16), D(256), R(256, 0.f);
        srand(2);
        for (int i = 0; i < 512; ++i) { float v = (rand() % 17 - 8) / 4.0f; A[i] = f2bf(v); Af[i] = bf2f(A[i]); }
        for (int i = 0; i < 512; ++i) { float v = (rand() % 13 - 6) / 2.0f; B[i] = f2bf(v); Bf[i] = bf2f(B[i]); }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += Af[i * 32 + k] * Bf[k * 16 + j]; R[i * 16 + j] = s; }
        uint16_t *dA, *dB; float* dD;
        CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        k_mfma16<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        double md = 0; for (int i = 0; i < 256; ++i) md = fmax(md, fabs(D[i] - R[i]));
        printf("[mfma_16x16x32_bf16 layout] maxdiff=%g %s\n", md, md < 1e-3 ? "PASS" : "FAIL"); if (md >= 1e-3) ++fails;
    }
    // ---- ds_read_b64_tr_b16
    {
        // Case 1: lane i -> byte offset i*8 (linear). Expect lane l elem j = (l&15) + j*16 + (l>>4)*64.
        std::vector<int> addr(64); std::vector<uint16_t> out(256);
        int* dAddr; uint16_t* dOut; CK(hipMalloc(&dAddr, 256)); CK(hipMalloc(&dOut, 512));
        for (int i = 0; i < 64; ++i) addr[i] = i * 8;
        CK(hipMemcpy(dAddr, addr.data(), 256, hipMemcpyHostToDevice));
        k_tr<<<1, 64>>>(dAddr, dOut); CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dOut, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (out[l * 4 + j] != (l & 15) + j * 16 + (l >> 4) * 64) ++bad;
        printf("[tr_b16 linear] %s\n", bad ? "FAIL" : "PASS"); if (bad) ++fails;
        printf("  raw (lane: e0 e1 e2 e3):");
        for (int l = 0; l < 64; ++l) { if (l % 8 == 0) printf("\n   "); printf(" %2d:%4d %4d %4d %4d |", l, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]); }
        printf("\n");
        // Case 2: strided rows. 16-lane group g, lane i in group: row = i>>2 (stride 200 elems), col chunk = 4*(i&3), group col offset 16*g.
        // Hypothesis: lane l elem j = row j, col 16*g + (l&15)  => value j*200 + 16*g + (l&15).
        for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; addr[l] = ((i >> 2) * 200 + 16 * g + 4 * (i & 3)) * 2; }
        CK(hipMemcpy(dAddr, addr.data(), 256, hipMemcpyHostToDevice));
        k_tr<<<1, 64>>>(dAddr, dOut); CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dOut, 512, hipMemcpyDeviceToHost));
        bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (out[l * 4 + j] != j * 200 + 16 * (l >> 4) + (l & 15)) ++bad;
        printf("[tr_b16 strided-rows hypothesis] %s\n", bad ? "FAIL" : "PASS"); if (bad) ++fails;
        if (bad) { for (int l = 0; l < 64; ++l) { if (l % 8 == 0) printf("\n   "); printf(" %2d:%4d %4d %4d %4d |", l, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]); } printf("\n"); }
    }
    // ---- global_load_lds 16B
    {
        std::vector<uint16_t> src(4096); for (int i = 0; i < 4096; ++i) src[i] = (uint16_t)i;
        std::vector<int> off(64); for (int l = 0; l < 64; ++l) off[l] = ((l * 37) % 64) * 8 + 1024;  // permuted 16B chunks
        uint16_t *dS, *dO; int* dOff; std::vector<uint16_t> out(1024);
        CK(hipMalloc(&dS, 8192)); CK(hipMalloc(&dO, 2048)); CK(hipMalloc(&dOff, 256));
        CK(hipMemcpy(dS, src.data(), 8192, hipMemcpyHostToDevice)); CK(hipMemcpy(dOff, off.data(), 256, hipMemcpyHostToDevice));
        k_glds<<<1, 64>>>(dS, dOff, dO); CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dO, 2048, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) if (out[l * 8 + j] != off[l] + j) ++bad;
        for (int i = 512; i < 1024; ++i) if (out[i] != 0xffff) ++bad;
        printf("[global_load_lds 16B lane-linear dest] %s\n", bad ? "FAIL" : "PASS"); if (bad) ++fails;
    }
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s CUs=%d clock=%d MHz LDS/block=%zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, p.sharedMemPerBlock);
    printf("probe: %d failures\n", fails);
    return fails ? 1 : 0;
}
