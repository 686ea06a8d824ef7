// csmri_fused.hip -- one whole inner iteration of pnp_svrg on CSMRI in ONE kernel (f32, 256 x 256):
//
//     z <- prox_TV( alpha * Re ifft2( sel o fft2(a - b) ) + beta * c1 + gamma * c2 )        [+ noise estimate, PSNR error]
//
// i.e. reference algorithms/pnp_svrg.py:52-80 (minibatch SVRG direction via problems/CSMRI.py:83-89, step,
// estimate_sigma, TVDenoiser.denoise, Problem.PSNR) with a = z, b = w, c1 = z, c2 = mu.  The four streaming kernels of
// csmri.hip / prox.hip move the half spectrum through HBM three times and the stepped image once (3.3 MB per problem-
// iteration against 2.4 MB algorithmic); here ONE 1024-thread workgroup owns one image and keeps it in registers from
// the first load to the last store -- 64 VGPRs x 1024 threads is exactly a 256 x 256 f32 image -- so HBM sees only the
// operands: a, b, c1, c2, xrec in, z out (1.5 MB).
//
// Phases (all data movement between them is through the CU's LDS):
//   1  rows forward   : 64 lane-groups x 2 passes; a group packs two real rows of (a - b) into one complex FFT-256
//                       (fft.h layout: lane + 16 * register)
//   2  columns        : the 256 KiB raw spectrum does not fit LDS, so it crosses in two halves of 128 k-space columns
//                       chosen so that kx and W - kx travel together (the split of the packed transforms needs both):
//                       row side writes [kx][row pair]; every group takes one column pair, splits it into the true half-
//                       spectrum column (256 points), FFT -> selector weights (bit-packed mask o minibatch) -> inverse
//                       FFT in registers, re-packs and writes back; row side reads its entries back
//   3  rows inverse   : one complex inverse FFT per row pair = two real rows; epilogue alpha*g + beta*c1 + gamma*c2
//   4  re-layout      : row-pair layout -> the prox's column layout (4 lanes x 64 rows per column), two halves of 128
//                       image columns through LDS
//   5  prox           : prox_tv.h -- per-column MAD noise estimate, Haar BayesShrink, squared error, store
// The FFT scratch of a lane group is private to it and the group lies inside one wavefront, so the in-FFT exchanges need
// no workgroup barrier (a wavefront's LDS operations complete in order); barriers separate only the phases that hand data
// between wavefronts.  DENOISE = false stops after the noise estimate and stores the stepped image (the DnCNN prox takes
// over from there).
#include "fft.h"
#include "prox_tv.h"

namespace pnp {

constexpr int FN = 256;                                   // image side
constexpr int F_SCR = 16 * 17;                            // complex elements of one group's FFT scratch
constexpr int F_RS = 129;                                 // row stride (complex) of the transposition buffer [128 kx][129]
constexpr int F_CS = 257;                                 // row stride (floats) of the re-layout buffer [128 cols][257]
constexpr size_t F_LDS_BYTES = (size_t)64 * F_SCR * sizeof(cx<float>);   // 139 264 B >= 128*129*8 and 128*257*4

// FFT-256 of one lane group (16 lanes x 16 registers, element lane + 16 r, natural order in and out); twiddles from an
// LDS table; group-private scratch; no workgroup barrier (see header).
template <bool INV>
__device__ __forceinline__ void fft256(cx<float> (&v)[16], const cx<float>* twl, cx<float>* scr, int lane) {
    dft_reg<float, 16, INV>(v);
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        const cx<float> tw = twl[(lane * r) & 255];
        v[r] = cmul(v[r], INV ? cconj(tw) : tw);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[r * 17 + lane] = v[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = scr[lane * 17 + r];
    __builtin_amdgcn_wave_barrier();
    dft_reg<float, 16, INV>(v);
}

__device__ __forceinline__ float dpp_xor1(float v) {       // value of lane ^ 1 (quad_perm [1,0,3,2]), VALU only
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// which half a k-space column travels in, and its row in the transposition buffer
__device__ __forceinline__ bool in_half(int kx, int half) {
    const bool a = kx <= 63 || kx == 128 || kx >= 193;
    return half == 0 ? a : !a;
}
__device__ __forceinline__ int kx_local(int kx, int half) {
    if (half == 0) return kx <= 63 ? kx : (kx == 128 ? 64 : kx - 128);           // 0..63, 64, 65..127
    return kx <= 127 ? kx - 64 : kx - 65;                                        // 64..127 -> 0..63, 129..192 -> 64..127
}

template <bool DENOISE>
__global__ __launch_bounds__(1024) void k_svrg_iter(const float* a, const float* __restrict__ b,
                                                    const uint32_t* __restrict__ bitsT, const cx<float>* __restrict__ twtab,
                                                    float scale, const float* __restrict__ alpha_vec, float beta, const float* c1,
                                                    float gamma, const float* __restrict__ c2, float* out,
                                                    float sigma_modifier, float fallback_sigma, const float* __restrict__ xrec,
                                                    double* __restrict__ sse_out, float* __restrict__ sigma_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cx<float>* ldc = reinterpret_cast<cx<float>*>(lds_raw);
    float* ldf = reinterpret_cast<float*>(lds_raw);
    __shared__ cx<float> twl[FN];
    __shared__ uint32_t sbits[64][16];
    __shared__ double red[16];
    __shared__ float sig_sh;
    const int t = threadIdx.x, g = t >> 4, l = t & 15, wv = t >> 6, lane64 = t & 63;
    const int prob = blockIdx.x;
    const size_t img = (size_t)prob * FN * FN;
    cx<float>* scr = ldc + g * F_SCR;
    if (t < FN) twl[t] = twtab[t];
    if (alpha_vec != nullptr) scale *= alpha_vec[prob];
    __syncthreads();

    // ------------------------------------------------------------------ 1: rows forward
    cx<float> Z[2][16];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const size_t ra = img + (size_t)(2 * (p * 64 + g)) * FN, rb = ra + FN;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = l + 16 * r;
            float va = a[ra + w], vb = a[rb + w];
            if (b != nullptr) { va -= b[ra + w]; vb -= b[rb + w]; }
            Z[p][r] = {va, vb};
        }
        fft256<false>(Z[p], twl, scr, l);
    }

    // ------------------------------------------------------------------ 2: columns, two halves
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                    // FFT scratch / previous half's reads are done
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kx = l + 16 * r;
                if (in_half(kx, half)) ldc[kx_local(kx, half) * F_RS + p * 64 + g] = Z[p][r];
            }
        // selector bits of this group's column pair (row kx and row W - kx of the transposed bit mask)
        const int ca = half == 0 ? g : 64 + g;              // the pair's first column; half 0, g == 0: columns 0 and 128
        const int cb = (half == 0 && g == 0) ? 128 : FN - ca;
        sbits[g][l] = bitsT[((size_t)prob * FN + (l < 8 ? ca : cb)) * 8 + (l & 7)];
        __syncthreads();
        const int la = kx_local(ca, half), lb = kx_local(cb, half);
        cx<float> v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // h = l + 16 r: row pair rp = h >> 1; even lanes fetch column ca, odd lanes column cb, then trade
            const int rp = (l + 16 * r) >> 1;
            const cx<float> own = ldc[((l & 1) ? lb : la) * F_RS + rp];
            const cx<float> oth = {dpp_xor1(own.x), dpp_xor1(own.y)};
            const cx<float> zk = (l & 1) ? oth : own, zm = (l & 1) ? own : oth;
            if (half == 0 && g == 0) {
                // packed column: (kx = 0, kx = 128) of row 2rp as (re, im) on even lanes, of row 2rp+1 on odd lanes
                v[r] = (l & 1) ? cx<float>{zk.y, zm.y} : cx<float>{zk.x, zm.x};
            } else {
                // split of the two packed real rows: A (row 2rp) on even lanes, B (row 2rp+1) on odd lanes
                v[r] = (l & 1) ? cx<float>{0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x)}
                               : cx<float>{0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};
            }
        }
        __syncthreads();                                    // every group has its column: the buffer becomes FFT scratch
        fft256<false>(v, twl, scr, l);                      // along h: element ky = l + 16 r
        auto bit = [&](int slot, int ky) -> float { return (float)((sbits[g][slot * 8 + (ky >> 5)] >> (ky & 31)) & 1u); };
        if (half == 0 && g == 0) {
            // the packed column holds two real-input transforms: separate, weight, re-pack (k_cols, blockIdx.x == 0)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[l + 16 * r] = v[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ky = l + 16 * r, km = (FN - ky) & (FN - 1);
                const cx<float> pk = v[r], pm = scr[km];
                const cx<float> A = {0.5f * (pk.x + pm.x), 0.5f * (pk.y - pm.y)};
                const cx<float> B = {0.5f * (pk.y + pm.y), -0.5f * (pk.x - pm.x)};
                const float wA = 0.5f * (bit(0, ky) + bit(0, km)), wB = 0.5f * (bit(1, ky) + bit(1, km));
                v[r] = {wA * A.x - wB * B.y, wA * A.y + wB * B.x};
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ky = l + 16 * r, km = (FN - ky) & (FN - 1);
                const float wgt = 0.5f * (bit(0, ky) + bit(1, km));
                v[r] = {wgt * v[r].x, wgt * v[r].y};
            }
        }
        fft256<true>(v, twl, scr, l);                       // back to h = l + 16 r
        __syncthreads();                                    // all FFT scratch use is over: the buffer carries data again
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // Hermitian re-expansion (k_rows_inv): even lane has q.a,q.b (row 2rp), odd lane q.c,q.d (row 2rp+1)
            const int rp = (l + 16 * r) >> 1;
            const cx<float> own = v[r], oth = {dpp_xor1(v[r].x), dpp_xor1(v[r].y)};
            cx<float> o;
            if (half == 0 && g == 0) o = (l & 1) ? cx<float>{oth.y, own.y} : cx<float>{own.x, oth.x};   // zp[128] | zp[0]
            else o = (l & 1) ? cx<float>{oth.x + own.y, own.x - oth.y}                                  // zp[W - kx] = {a + d, c - b}
                             : cx<float>{own.x - oth.y, own.y + oth.x};                                 // zp[kx]     = {a - d, b + c}
            ldc[((l & 1) ? lb : la) * F_RS + rp] = o;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kx = l + 16 * r;
                if (in_half(kx, half)) Z[p][r] = ldc[kx_local(kx, half) * F_RS + p * 64 + g];
            }
    }
    __syncthreads();

    // ------------------------------------------------------------------ 3: rows inverse + epilogue (in place in Z)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        fft256<true>(Z[p], twl, scr, l);
        const size_t ra = img + (size_t)(2 * (p * 64 + g)) * FN, rb = ra + FN;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = l + 16 * r;
            float oa = scale * Z[p][r].x, ob = scale * Z[p][r].y;
            if (c1 != nullptr) { oa += beta * c1[ra + w]; ob += beta * c1[rb + w]; }
            if (c2 != nullptr) { oa += gamma * c2[ra + w]; ob += gamma * c2[rb + w]; }
            Z[p][r] = {oa, ob};
        }
    }

    // ------------------------------------------------------------------ 4: row-pair layout -> column layout
    // wave wv owns image columns [16 wv, 16 wv + 16); lane = column + 16 * chunk keeps rows [64 chunk, 64 chunk + 64)
    float x[64];
    const int cl = lane64 & 15, q = lane64 >> 4;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const cx<float> val = half == 0 ? Z[p][r] : Z[p][r + 8];
                float* dst = ldf + (l + 16 * r) * F_CS + 2 * (p * 64 + g);
                dst[0] = val.x;
                dst[1] = val.y;
            }
        __syncthreads();
        if ((wv >> 3) == half) {
            const float* src = ldf + (16 * (wv & 7) + cl) * F_CS + 64 * q;
#pragma unroll
            for (int i = 0; i < 64; ++i) x[i] = src[i];
        }
    }

    // ------------------------------------------------------------------ 5: noise estimate, prox, error, store
    const size_t base = img + (size_t)(q * 64) * FN + 16 * wv + cl;
    prox_tv_regs<float, FN, DENOISE>(x, prob, FN, base, wv, lane64, q, 16, nullptr, sigma_modifier, fallback_sigma, xrec, out,
                                     sse_out, sigma_out, red, &sig_sh);
}

}  // namespace pnp

// plan internals live in csmri.hip (pnp_csmri_svrg_step); the kernel only needs the plan's twiddle table
namespace pnp {
int csmri_fused_launch(int batch, const void* twtab, const void* a, const void* b, const uint32_t* bitsT,
                                      double alpha, const void* alpha_vec, double beta, const void* c1, double gamma,
                                      const void* c2, void* out, int denoise, double sigma_modifier, double fallback_sigma,
                                      const void* xrec, double* sse_out, void* sigma_out, void* stream) {
    const float scale = (float)(alpha / ((double)FN * (double)FN));
    hipStream_t s = (hipStream_t)stream;
    // > 64 KiB of dynamic LDS needs the opt-in, once per device (the attribute is per device)
    static unsigned long long attr_done = 0;
    int dev = 0;
    PNP_CHECK_HIP(hipGetDevice(&dev));
    if (!((attr_done >> (dev & 63)) & 1ull)) {
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        attr_done |= 1ull << (dev & 63);
    }
    if (denoise) {
        k_svrg_iter<true><<<batch, 1024, F_LDS_BYTES, s>>>((const float*)a, (const float*)b, bitsT, (const cx<float>*)twtab, scale,
                                                           (const float*)alpha_vec, (float)beta, (const float*)c1, (float)gamma,
                                                           (const float*)c2, (float*)out, (float)sigma_modifier, (float)fallback_sigma,
                                                           (const float*)xrec, sse_out, (float*)sigma_out);
    } else {
        k_svrg_iter<false><<<batch, 1024, F_LDS_BYTES, s>>>((const float*)a, (const float*)b, bitsT, (const cx<float>*)twtab, scale,
                                                            (const float*)alpha_vec, (float)beta, (const float*)c1, (float)gamma,
                                                            (const float*)c2, (float*)out, (float)sigma_modifier, (float)fallback_sigma,
                                                            (const float*)xrec, sse_out, (float*)sigma_out);
    }
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}
}  // namespace pnp
