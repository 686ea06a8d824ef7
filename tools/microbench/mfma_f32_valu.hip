// Does vector-ALU work hide under the fp32 MFMA on gfx950?  One wave per SIMD, 16 independent accumulators of
// v_mfma_f32_16x16x4_f32 (32 cycles each), with K independent v_pk_add_f32 (or v_mfma_f32_16x16x32_f16 + K v_pk_add_f32
// for comparison) issued between consecutive MFMAs.   hipcc --offload-arch=gfx950 -O3 mfma_f32_valu.hip -o mfma_f32_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int K, bool F16>
__global__ __launch_bounds__(256, 1) void k(const float* src, float* out, unsigned long long* cyc, int iters) {
    float a = src[threadIdx.x], b = src[threadIdx.x + 256];
    h8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)src[threadIdx.x + j]; hb[j] = (_Float16)src[threadIdx.x + 8 + j]; }
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{src[threadIdx.x + i], src[threadIdx.x + 2 * i]};
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ha), "v"(hb));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < K; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[(i + j) & 7]) : "v"(v[(i + j + 3) & 7]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, bool F16> void run(const float* src, float* out, unsigned long long* cyc) {
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) { k<K, F16><<<256, 256>>>(src, out, cyc, iters); hipDeviceSynchronize(); }
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s + %d v_pk_add_f32 per MFMA: %.1f cycles per MFMA\n", F16 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_16x16x4_f32 ", K, h[7] / (1000.0 * 16));
}

int main() {
    float* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 4096 * sizeof(float));
    float hsrc[4096]; for (int i = 0; i < 4096; ++i) hsrc[i] = 1.0f + 1e-3f * i;
    hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    run<0, false>(src, out, cyc); run<1, false>(src, out, cyc); run<2, false>(src, out, cyc); run<4, false>(src, out, cyc); run<6, false>(src, out, cyc);
    run<0, true>(src, out, cyc); run<1, true>(src, out, cyc); run<2, true>(src, out, cyc); run<4, true>(src, out, cyc);
    return 0;
}
