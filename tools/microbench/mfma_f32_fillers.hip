// What hides under v_mfma_f32_16x16x4_f32 (32 cycles/SIMD) on gfx950 at one wave per SIMD?  16 independent
// accumulators, a filler pattern between consecutive MFMAs.  Prints shader cycles per MFMA for each pattern.
//   hipcc --offload-arch=gfx950 -O3 mfma_f32_fillers.hip -o mfma_f32_fillers && ./mfma_f32_fillers
// Patterns:  add K   : K plain v_add_f32 after every MFMA
//            sub2    : v_sub_f32 + v_add_f32 pairs (the scalar form of the Winograd B^T d)
//            pk K    : K v_pk_add_f32 after every MFMA (the form k_mid_wino used in round 1)
//            ds K/M  : K ds_read2_b32 after every M-th MFMA
//            mixS    : per 48 MFMAs 10 ds_read2_b32 + 20 v_add_f32 + 1 LDS-DMA piece, SPREAD one per gap
//            mixB    : the same work as ONE block in front of the 48 MFMAs (round-1 k_mid_wino schedule)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { P_ADD, P_PK, P_DS, P_MIXS, P_MIXB, P_MIXS2, P_FMA, P_PKFMA, P_PKFMA_S, P_BLOCK_FMA, P_BLOCK_PKFMA };

#define MFMA(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[(i) & 15]) : "a"(a), "v"(b))
#define VADD(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(j) & 15]) : "v"(v[((j) + 5) & 15]))
#define VSUB(j) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[(j) & 15]) : "v"(v[((j) + 5) & 15]))
#define VPK(j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(j) & 7]) : "v"(p[((j) + 3) & 7]))
#define VFMA(j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[(j) & 15]) : "v"(v[((j) + 5) & 15]), "v"(v[((j) + 9) & 15]))
#define VPKFMA(j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[(j) & 7]) : "v"(p[((j) + 3) & 7]), "v"(p[((j) + 5) & 7]))
#define VPKFMAS(j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[(j) & 7]) : "s"(kc), "v"(p[((j) + 5) & 7]))
#define DSR(j) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(d[(j) & 15]) : "v"(laddr), "n"(((j) & 15) * 2), "n"(((j) & 15) * 2 + 1) : "memory")

template <int PAT, int K, int M>
__global__ __launch_bounds__(256, 1) void k(const float* src, float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = src[i & 4095];
    __syncthreads();
    float a = src[threadIdx.x], b = src[threadIdx.x + 256];
    float v[16];
    f32x2 p[8], d[16];
    for (int i = 0; i < 16; ++i) { v[i] = src[threadIdx.x + i]; d[i] = f32x2{0.f, 0.f}; }
    for (int i = 0; i < 8; ++i) p[i] = f32x2{src[threadIdx.x + i], src[threadIdx.x + 2 * i]};
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + 4u * (threadIdx.x & 63) * 2u;
    const float* gsrc = src + (threadIdx.x & 63) * 4;
    float* dma_dst = lds + 8192 + (threadIdx.x >> 6) * 256;
    const unsigned long long kc = 0x4080000040800000ull;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (PAT == P_BLOCK_FMA) {
#pragma unroll
            for (int j = 0; j < 36; ++j) VFMA(j);
        }
        if (PAT == P_BLOCK_PKFMA) {
#pragma unroll
            for (int j = 0; j < 18; ++j) VPKFMA(j);
        }
        if (PAT == P_MIXB) {
#pragma unroll
            for (int j = 0; j < 10; ++j) DSR(j);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                             (__attribute__((address_space(3))) void*)dma_dst, 16, 0, 0);
#pragma unroll
            for (int j = 0; j < 20; ++j) { if (j & 1) VSUB(j); else VADD(j); }
        }
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            MFMA(i);
            if (PAT == P_ADD) {
#pragma unroll
                for (int j = 0; j < K; ++j) { if (j & 1) VSUB(i + j); else VADD(i + j); }
            } else if (PAT == P_FMA) {
#pragma unroll
                for (int j = 0; j < K; ++j) VFMA(i + j);
            } else if (PAT == P_PKFMA) {
#pragma unroll
                for (int j = 0; j < K; ++j) VPKFMA(i + j);
            } else if (PAT == P_PKFMA_S) {
#pragma unroll
                for (int j = 0; j < K; ++j) VPKFMAS(i + j);
            } else if (PAT == P_PK) {
#pragma unroll
                for (int j = 0; j < K; ++j) VPK(i + j);
            } else if (PAT == P_DS) {
                if (i % M == 0) {
#pragma unroll
                    for (int j = 0; j < K; ++j) DSR(i + j);
                }
            } else if (PAT == P_MIXS) {
                // 31 fillers over 48 gaps: gaps 0..9 a ds_read2, gap 10 the DMA, gaps 12..31 one add each
                if (i < 10) DSR(i);
                else if (i == 10)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                                     (__attribute__((address_space(3))) void*)dma_dst, 16, 0, 0);
                else if (i >= 12 && i < 32) { if (i & 1) VSUB(i); else VADD(i); }
            } else if (PAT == P_MIXS2) {
                // two fillers in every other gap
                if (i < 20 && (i & 1) == 0) { DSR(i >> 1); VADD(i); }
                else if (i == 21)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                                     (__attribute__((address_space(3))) void*)dma_dst, 16, 0, 0);
                else if (i >= 22 && i < 42 && (i & 1) == 0) { VSUB(i); VADD(i + 1); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3] + v[i] + d[i].x + d[i].y;
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[8192 + threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PAT, int K, int M> void run(const char* name, const float* src, float* out, unsigned long long* cyc) {
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) { k<PAT, K, M><<<256, 256>>>(src, out, cyc, iters); hipDeviceSynchronize(); }
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s %.2f cycles per MFMA\n", name, h[7] / (400.0 * 48));
}

int main() {
    float* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 8192 * sizeof(float));
    static float hsrc[8192]; for (int i = 0; i < 8192; ++i) hsrc[i] = 1.0f + 1e-3f * (i % 997);
    hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    run<P_ADD, 0, 1>("bare", src, out, cyc);
    run<P_ADD, 1, 1>("1 v_add_f32 / gap", src, out, cyc);
    run<P_ADD, 2, 1>("2 v_add/sub_f32 / gap", src, out, cyc);
    run<P_ADD, 4, 1>("4 v_add/sub_f32 / gap", src, out, cyc);
    run<P_ADD, 6, 1>("6 v_add/sub_f32 / gap", src, out, cyc);
    run<P_ADD, 8, 1>("8 v_add/sub_f32 / gap", src, out, cyc);
    run<P_PK, 1, 1>("1 v_pk_add_f32 / gap", src, out, cyc);
    run<P_PK, 2, 1>("2 v_pk_add_f32 / gap", src, out, cyc);
    run<P_FMA, 1, 1>("1 v_fma_f32 / gap", src, out, cyc);
    run<P_FMA, 2, 1>("2 v_fma_f32 / gap", src, out, cyc);
    run<P_PKFMA, 1, 1>("1 v_pk_fma_f32 / gap", src, out, cyc);
    run<P_PKFMA, 2, 1>("2 v_pk_fma_f32 / gap", src, out, cyc);
    run<P_PKFMA_S, 1, 1>("1 v_pk_fma_f32 (SGPR const) / gap", src, out, cyc);
    run<P_BLOCK_FMA, 0, 1>("block of 36 v_fma_f32 / 48 MFMA", src, out, cyc);
    run<P_BLOCK_PKFMA, 0, 1>("block of 18 v_pk_fma_f32 / 48 MFMA", src, out, cyc);
    run<P_DS, 1, 1>("1 ds_read2_b32 / gap", src, out, cyc);
    run<P_DS, 1, 2>("1 ds_read2_b32 / 2 gaps", src, out, cyc);
    run<P_DS, 2, 1>("2 ds_read2_b32 / gap", src, out, cyc);
    run<P_DS, 1, 4>("1 ds_read2_b32 / 4 gaps", src, out, cyc);
    run<P_MIXS, 0, 1>("mix spread (1 filler per gap)", src, out, cyc);
    run<P_MIXS2, 0, 1>("mix spread (2 per other gap)", src, out, cyc);
    run<P_MIXB, 0, 1>("mix as one block (round 1)", src, out, cyc);
    return 0;
}
