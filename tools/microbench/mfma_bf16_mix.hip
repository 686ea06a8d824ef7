// Projection for a 3 x bf16 split conv: what does the instruction mix of such a kernel cost per
// v_mfma_f32_16x16x32_bf16 at one wave per SIMD?  Per 48 MFMAs: NV vector-ALU instructions (plain f32 FMAs / packed
// converts), NL ds_read_b128, NG global_load_dwordx4 (1 KiB per wave), spread over the gaps.
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16_mix.hip -o mfma_bf16_mix && ./mfma_bf16_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NL, int NG>
__global__ __launch_bounds__(256, 1) void k(const float* src, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = src[i & 4095];
    __syncthreads();
    f32x4 a4 = *(const f32x4*)(src + 4 * threadIdx.x), b4 = *(const f32x4*)(src + 1024 + 4 * threadIdx.x);
    bf16x8 a = __builtin_bit_cast(bf16x8, a4), b = __builtin_bit_cast(bf16x8, b4);
    float v[16];
    f32x4 d[8], g[8];
    for (int i = 0; i < 16; ++i) v[i] = src[threadIdx.x + i];
    for (int i = 0; i < 8; ++i) { d[i] = f32x4{0.f, 0.f, 0.f, 0.f}; g[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + 16u * (threadIdx.x & 63);
    const f32x4* gp = (const f32x4*)src + (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        int nv = 0, nl = 0, ng = 0;
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i & 15]) : "v"(a), "v"(b));
            // this gap's share of the fillers
#pragma unroll
            for (; nv < (NV * (i + 1)) / 48; ++nv)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[nv & 15]) : "v"(v[(nv + 5) & 15]), "v"(v[(nv + 9) & 15]));
#pragma unroll
            for (; nl < (NL * (i + 1)) / 48; ++nl)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[nl & 7]) : "v"(laddr), "n"((nl & 15) * 1024) : "memory");
#pragma unroll
            for (; ng < (NG * (i + 1)) / 48; ++ng)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(g[ng & 7]) : "v"(gp), "n"((ng & 3) * 1024) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3] + v[i];
    for (int i = 0; i < 8; ++i) s += d[i].x + d[i].w + g[i].x + g[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int NL, int NG> void run(const char* name, const float* src, float* out, unsigned long long* cyc) {
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) { k<NV, NL, NG><<<256, 256>>>(src, out, cyc, iters); (void)hipDeviceSynchronize(); }
    unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double c = h[7] / (400.0 * 48);
    printf("%-66s %6.2f cycles per MFMA -> %5.1f k cycles per 864 MFMAs (one 8 x 64 region)\n", name, c, c * 864 / 1000);
}

int main() {
    float* src; float* out; unsigned long long* cyc;
    (void)hipMalloc(&src, 8192 * sizeof(float));
    static float hsrc[8192]; for (int i = 0; i < 8192; ++i) hsrc[i] = 1.0f + 1e-3f * (i % 997);
    (void)hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 256 * sizeof(float)); (void)hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    run<0, 0, 0>("bare v_mfma_f32_16x16x32_bf16", src, out, cyc);
    run<183, 0, 0>("+ 183 vector-ALU instructions per 48 MFMAs (3.3 k per region)", src, out, cyc);
    run<183, 24, 0>("+ 24 ds_read_b128 (432 per region)", src, out, cyc);
    run<183, 24, 12>("+ 12 global_load_dwordx4 (216 weight loads per region)", src, out, cyc);
    run<183, 24, 16>("+ 16 global_load_dwordx4 (weights + DMA + stores: 296 per region)", src, out, cyc);
    run<92, 24, 16>("half the vector-ALU work (92 per 48 MFMAs)", src, out, cyc);
    return 0;
}
