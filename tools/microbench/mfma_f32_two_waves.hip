// Does a second wave on the SIMD hide vector-ALU work under v_mfma_f32_16x16x4_f32?  (One wave cannot: mfma_f32_fillers.hip.)
// Every wave runs: [block of NV v_pk_fma_f32] [NL ds_read_b64] 48 MFMAs, repeated; a 512-thread workgroup puts two waves on every SIMD; each wave starts at
// another phase of the iteration.  Prints shader cycles per MFMA *of the SIMD* (all waves' MFMAs over the time of the slowest wave).
//   hipcc --offload-arch=gfx950 -O3 mfma_f32_two_waves.hip -o mfma_f32_two_waves && ./mfma_f32_two_waves
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[(i) & 15]) : "v"(a), "v"(b))
#define VPKFMA(j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[(j) & 7]) : "v"(p[((j) + 3) & 7]), "v"(p[((j) + 5) & 7]))
#define DSR(j) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[(j) & 7]) : "v"(laddr), "n"(((j) & 15) * 512) : "memory")

template <int THREADS, int NV, int NL, bool SPREAD>
__global__ __launch_bounds__(THREADS, 1) void k(const float* src, float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += THREADS) lds[i] = src[i & 4095];
    __syncthreads();
    float a = src[threadIdx.x], b = src[threadIdx.x + 256];
    f32x2 p[8], d[8];
    for (int i = 0; i < 8; ++i) { p[i] = f32x2{src[threadIdx.x + i], src[threadIdx.x + 2 * i]}; d[i] = f32x2{0.f, 0.f}; }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + 8u * (threadIdx.x & 63);
    for (int w = threadIdx.x >> 6; w > 0; --w) {                  // every wave starts at another phase of the iteration
#pragma unroll
        for (int i = 0; i < 6; ++i) MFMA(i);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (!SPREAD) {
#pragma unroll
            for (int j = 0; j < NL; ++j) DSR(j);
#pragma unroll
            for (int j = 0; j < NV; ++j) VPKFMA(j);
        }
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            MFMA(i);
            if (SPREAD) {                                          // the conv kernel's form: a block of VALU every 8th gap, LDS ops singly
                if (i % 8 == 2) {
#pragma unroll
                    for (int j = 0; j < NV / 6; ++j) VPKFMA(i + j);
                } else if (i % 8 != 6 && (i / 8) * 7 + (i % 8) < NL + 6) DSR(i);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + d[i].x + d[i].y;
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[8 * blockIdx.x + (threadIdx.x >> 6)] = t1 - t0;
}

template <int THREADS, int NV, int NL, bool SPREAD> void run(const char* name, const float* src, float* out, unsigned long long* cyc) {
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) { k<THREADS, NV, NL, SPREAD><<<256, THREADS>>>(src, out, cyc, iters); hipDeviceSynchronize(); }
    static unsigned long long h[2048]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int waves = THREADS / 256;
    // the SIMD's arbiter serves its oldest wave first (mfma_f32_occupancy.hip): the workgroup is done when its slowest wave is
    unsigned long long t = 0;
    for (int w = 0; w < THREADS / 64; ++w) t = h[8 * 7 + w] > t ? h[8 * 7 + w] : t;
    printf("%-64s %d wave(s) / SIMD: %.2f cycles per MFMA of the SIMD\n", name, waves, t / (400.0 * 48 * waves));
}

int main() {
    float* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 8192 * sizeof(float));
    static float hsrc[8192]; for (int i = 0; i < 8192; ++i) hsrc[i] = 1.0f + 1e-3f * (i % 997);
    hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 2048 * sizeof(unsigned long long));
    // (bare MFMA rates against waves per SIMD: mfma_f32_occupancy.hip -- 155 TFLOP/s whatever the occupancy)
    run<256, 36, 0, false>("block of 36 v_pk_fma_f32 / 48 MFMA", src, out, cyc);
    run<512, 36, 0, false>("block of 36 v_pk_fma_f32 / 48 MFMA", src, out, cyc);
    run<256, 72, 0, false>("block of 72 v_pk_fma_f32 / 48 MFMA", src, out, cyc);
    run<512, 72, 0, false>("block of 72 v_pk_fma_f32 / 48 MFMA", src, out, cyc);
    run<256, 36, 32, false>("block of 32 ds_read_b64 + 36 v_pk_fma_f32 / 48 MFMA", src, out, cyc);
    run<512, 36, 32, false>("block of 32 ds_read_b64 + 36 v_pk_fma_f32 / 48 MFMA", src, out, cyc);
    run<256, 36, 32, true>("conv-kernel mix spread: 6 x 6 v_pk_fma_f32, 32 ds_read_b64 / 48", src, out, cyc);
    run<512, 36, 32, true>("conv-kernel mix spread: 6 x 6 v_pk_fma_f32, 32 ds_read_b64 / 48", src, out, cyc);
    run<256, 144, 32, false>("epilogue-like: block of 144 v_pk_fma_f32 + 32 ds / 48 MFMA", src, out, cyc);
    run<512, 144, 32, false>("epilogue-like: block of 144 v_pk_fma_f32 + 32 ds / 48 MFMA", src, out, cyc);
    return 0;
}
