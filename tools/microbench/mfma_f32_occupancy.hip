// f32 MFMA rate against waves per SIMD, by wall clock (hipEvents) and by s_memtime of every wave: 256 workgroups (one per CU)
// of 256 / 512 / 1024 threads, every wave a bare chain of v_mfma_f32_16x16x4_f32 (or 32x32x2) on 8 independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 mfma_f32_occupancy.hip -o mfma_f32_occupancy && ./mfma_f32_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int THREADS, bool BIG, bool AG = false>
__global__ __launch_bounds__(THREADS, 1) void k(const float* src, float* out, unsigned long long* cyc, int iters) {
    float a = src[threadIdx.x], b = src[threadIdx.x + 256];
    float s = 0;
    unsigned long long t0, t1;
    if (!BIG) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i & 7]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a), "v"(b));
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int THREADS, bool BIG, bool AG = false> void run(const float* src, float* out, unsigned long long* cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<THREADS, BIG, AG><<<256, THREADS>>>(src, out, cyc, iters);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
    (void)hipMemset(cyc, 0, 256 * 16 * 8);
    (void)hipEventRecord(e0);
    k<THREADS, BIG, AG><<<256, THREADS>>>(src, out, cyc, iters);
    (void)hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 16]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int waves = THREADS / 64;
    const double flop = 256.0 * waves * iters * 32.0 * (BIG ? 32 * 32 * 2 * 2 : 16 * 16 * 4 * 2);
    unsigned long long mn = ~0ull, mx = 0;
    for (int w = 0; w < waves; ++w) { mn = h[7 * 16 + w] < mn ? h[7 * 16 + w] : mn; mx = h[7 * 16 + w] > mx ? h[7 * 16 + w] : mx; }
    printf("%-30s %d waves / SIMD: %.3f ms  %.1f TFLOP/s   cycles per own MFMA of the waves of workgroup 7: %.2f .. %.2f\n",
           BIG ? "32x32x2 " : AG ? "16x16x4 (AGPR accumulators)" : "16x16x4 ", waves / 4, ms, flop / ms * 1e-9, mn / (iters * 32.0), mx / (iters * 32.0));
}

int main() {
    float* src; float* out; unsigned long long* cyc;
    (void)hipMalloc(&src, 8192 * sizeof(float));
    static float hsrc[8192]; for (int i = 0; i < 8192; ++i) hsrc[i] = 1.0f + 1e-3f * (i % 997);
    (void)hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 1024 * sizeof(float)); (void)hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
    run<256, false>(src, out, cyc); run<512, false>(src, out, cyc); run<1024, false>(src, out, cyc);
    run<256, false, true>(src, out, cyc); run<512, false, true>(src, out, cyc);
    run<256, true>(src, out, cyc); run<512, true>(src, out, cyc); run<1024, true>(src, out, cyc);
    return 0;
}
