// Issue rate of v_mfma_f32_16x16x32_f16 with the operand placement of k_mid_f16x3 (A in AGPRs or VGPRs, accumulators in
// VGPRs or AGPRs), one wave per SIMD, 16 independent accumulators.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const h8* src, float* out, unsigned long long* cyc, int iters) {
    h8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[threadIdx.x + 256 * i];
    for (int i = 0; i < 2; ++i) b[i] = src[threadIdx.x + 256 * (4 + i)];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0)      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(a[i & 3]), "v"(b[i & 1]));
            else if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[i & 1]));
            else if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & 3]), "v"(b[i & 1]));
            else                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                                             : "+v"(acc[i]) : "a"(a[i & 3]), "v"(b[i & 1]));    // dependent triple
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    h8* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 256 * 6 * sizeof(h8)); hipMemset(src, 0x3c, 256 * 6 * sizeof(h8));     // fp16 ~1.06: real data, real power
    hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    const int iters = 2000;
    const char* names[4] = {"A in AGPR, acc in VGPR", "A in VGPR, acc in VGPR", "A in VGPR, acc in AGPR", "A in AGPR, acc VGPR, dependent triples"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k<0><<<256, 256>>>(src, out, cyc, iters);
            if (mode == 1) k<1><<<256, 256>>>(src, out, cyc, iters);
            if (mode == 2) k<2><<<256, 256>>>(src, out, cyc, iters);
            if (mode == 3) k<3><<<256, 256>>>(src, out, cyc, iters);
            hipDeviceSynchronize();
        }
        unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const double n = (double)iters * 16 * (mode == 3 ? 3 : 1);
        printf("%-42s %.2f cycles per MFMA (workgroup 0), %.2f (workgroup 128)\n", names[mode], h[0] / n, h[128] / n);
    }
    return 0;
}
