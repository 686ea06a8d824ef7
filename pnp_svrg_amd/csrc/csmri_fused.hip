// csmri_fused.hip -- one whole inner iteration of pnp_svrg on CSMRI in ONE kernel (f32, 256 x 256):
//
//     z <- prox_TV( alpha * Re ifft2( sel o fft2(a - b) ) + beta * c1 + gamma * c2 )        [+ noise estimate, PSNR error]
//
// i.e. reference algorithms/pnp_svrg.py:52-80 (minibatch SVRG direction via problems/CSMRI.py:83-89, step,
// estimate_sigma, TVDenoiser.denoise, Problem.PSNR) with a = z, b = w, c1 = z, c2 = mu.  The four streaming kernels of
// csmri.hip / prox.hip move the half spectrum through HBM three times and the stepped image once (3.3 MB per problem-
// iteration against 2.4 MB algorithmic); here ONE workgroup owns one image and keeps it in registers from the first load
// to the last store, so HBM sees only the operands: a, b, c1, c2, xrec in, z out (1.5 MB).
//
// 512 threads = 8 wavefronts = two per SIMD, i.e. 256 VGPRs per lane: 128 of them hold the image (a 256 x 256 f32 image
// is 128 registers x 512 lanes), the rest is working space -- at 1024 threads (128 VGPRs) the same kernel spilled ~300
// registers per lane to scratch and the spill traffic alone exceeded the HBM traffic it was meant to save (measured).
//
// Phases (all data movement between them is through the CU's LDS):
//   1  rows forward   : 32 lane-groups x 4 passes; a group packs two real rows of (a - b) into one complex FFT-256
//                       (fft.h layout: lane + 16 * register)
//   2  columns        : the 256 KiB raw spectrum does not fit LDS, so it crosses in two halves of 128 k-space columns
//                       chosen so that kx and W - kx travel together (the split of the packed transforms needs both):
//                       row side writes [kx][row pair]; every group takes two column pairs, splits each into the true
//                       half-spectrum column (256 points), FFT -> selector weights (bit-packed mask o minibatch) -> inverse
//                       FFT in registers, re-packs and writes back; row side reads its entries back
//   3  rows inverse   : one complex inverse FFT per row pair = two real rows; epilogue alpha*g + beta*c1 + gamma*c2
//   4  re-layout      : row-pair layout -> the prox's column layout (4 lanes x 64 rows per column; every lane ends up
//                       with column c and column c + 128), two halves of 128 image columns through LDS
//   5  prox           : prox_tv.h -- per-column MAD noise estimate, Haar BayesShrink, squared error, store
// The FFT scratch of a lane group is private to it and the group lies inside one wavefront, so the in-FFT exchanges need
// no workgroup barrier (a wavefront's LDS operations complete in order); barriers separate only the phases that hand data
// between wavefronts.  DENOISE = false stops after the noise estimate and stores the stepped image (the DnCNN prox takes
// over from there).
#include "fft.h"
#include <cstdlib>

// Diagnostic build (-DPNP_FUSED_CLOCK, tools/fused_clock.py): thread 0 of every workgroup stamps the shader clock at the phase
// boundaries, so the phases can be timed in steady state (workgroups of a full launch are out of step with each other, unlike
// the PNP_FUSED_STOP builds where every CU is in the same phase and the loads of all of them saturate HBM together).
#ifdef PNP_FUSED_CLOCK
#define PNP_STAMP_MAXB 4096
__device__ unsigned long long g_fused_stamps[PNP_STAMP_MAXB * 16];
#define PNP_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < PNP_STAMP_MAXB) g_fused_stamps[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" int pnp_debug_fused_stamps(unsigned long long* host_out, int nblocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fused_stamps), (size_t)nblocks * 16 * sizeof(unsigned long long));
}
#else
#define PNP_STAMP(k) do { } while (0)
#endif

namespace pnp {

constexpr int FN = 256;                                   // image side
constexpr int FT = 512;                                   // threads per workgroup
constexpr int FG = FT / 16;                               // 32 lane groups
constexpr int FP = (FN / 2) / FG;                         // 4 row-pair passes per group
constexpr int F_SCR = 16 * 17;                            // complex elements of one group's FFT scratch
constexpr int F_RS = 129;                                 // row stride (complex) of the transposition buffer [128 kx][129]
constexpr int F_CS = 257;                                 // row stride (floats) of the re-layout buffer [128 cols][257]
constexpr size_t F_LDS_BYTES = (size_t)128 * F_RS * sizeof(cx<float>);   // 132 096 B >= 32 scratches and 128*257*4

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
// the same for kx = l + 16 r with the row index r a compile-time constant: fourteen of the sixteen r belong to one half
// for every lane (only kx = 128 and kx = 192, lane 0 of r = 8 and r = 12, sit on the other side), and said so the
// compiler sees that those registers of Z are dead between the hand-over and the reload -- with a per-lane predicate
// on every r it kept all 128 alive through both halves and spilled 80 registers around the column transforms
__device__ __forceinline__ bool in_half_r(int r, int l, int half) {
    if (r == 8) return (l == 0) == (half == 0);
    if (r == 12) return (l == 0) == (half == 1);
    return (r < 4 || r > 12) == (half == 0);
}
__device__ __forceinline__ int kx_local(int kx, int half) {
    if (half == 0) return kx <= 63 ? kx : (kx == 128 ? 64 : kx - 128);           // 0..63, 64, 65..127
    return kx <= 127 ? kx - 64 : kx - 65;                                        // 64..127 -> 0..63, 129..192 -> 64..127
}

// One column pair (ca, cb = W - ca; packed: ca = 0, cb = W/2) of the raw spectrum in the transposition buffer: split
// into the true half-spectrum column, FFT along h, selector weights, inverse FFT, Hermitian re-expansion.  `col_load` /
// `col_store` touch the buffer (the caller puts workgroup barriers around the part between them).
struct ColPair { int la, lb; bool packed; };

__device__ __forceinline__ void col_load(cx<float> (&v)[16], const cx<float>* ldc, const ColPair& c, int l) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // h = l + 16 r: row pair rp = h >> 1; even lanes fetch column ca, odd lanes column cb, then trade
        const int rp = (l + 16 * r) >> 1;
        const cx<float> own = ldc[((l & 1) ? c.lb : c.la) * F_RS + rp];
        const cx<float> oth = {dpp_xor1(own.x), dpp_xor1(own.y)};
        const cx<float> zk = (l & 1) ? oth : own, zm = (l & 1) ? own : oth;
        if (c.packed) {
            // packed column: (kx = 0, kx = 128) of row 2rp as (re, im) on even lanes, of row 2rp+1 on odd lanes
            v[r] = (l & 1) ? cx<float>{zk.y, zm.y} : cx<float>{zk.x, zm.x};
        } else {
            // split of the two packed real rows: A (row 2rp) on even lanes, B (row 2rp+1) on odd lanes
            v[r] = (l & 1) ? cx<float>{0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x)}
                           : cx<float>{0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};
        }
    }
}

__device__ __forceinline__ void col_transform(cx<float> (&v)[16], const cx<float>* twl, cx<float>* scr, const uint32_t* sb,
                                              const ColPair& c, int l, const cx<float>* __restrict__ yhc) {
    fft256<false>(v, twl, scr, l);                          // along h: element ky = l + 16 r
    auto bit = [&](int slot, int ky) -> float { return (float)((sb[slot * 8 + (ky >> 5)] >> (ky & 31)) & 1u); };
    if (c.packed) {
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
            v[r] = {fma_(wA, A.x, -(wB * B.y)), fma_(wA, A.y, wB * B.x)};
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
    if (yhc != nullptr) {                                   // packed data term of this half-spectrum column (pnp_csmri_pack_y)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = csub(v[r], yhc[l + 16 * r]);
    }
    fft256<true>(v, twl, scr, l);                           // back to h = l + 16 r
}

__device__ __forceinline__ void col_store(const cx<float> (&v)[16], cx<float>* ldc, const ColPair& c, int l) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // Hermitian re-expansion (k_rows_inv): even lane has q.a,q.b (row 2rp), odd lane q.c,q.d (row 2rp+1)
        const int rp = (l + 16 * r) >> 1;
        const cx<float> own = v[r], oth = {dpp_xor1(v[r].x), dpp_xor1(v[r].y)};
        cx<float> o;
        if (c.packed) o = (l & 1) ? cx<float>{oth.y, own.y} : cx<float>{own.x, oth.x};       // zp[128] | zp[0]
        else o = (l & 1) ? cx<float>{oth.x + own.y, own.x - oth.y}                            // zp[W - kx] = {a + d, c - b}
                         : cx<float>{own.x - oth.y, own.y + oth.x};                           // zp[kx]     = {a - d, b + c}
        ldc[((l & 1) ? c.lb : c.la) * F_RS + rp] = o;
    }
}

// phases 1-3: the gradient step; leaves the stepped image in Z (row-pair layout: Z[p][r] = rows 2rp, 2rp+1 at column
// l + 16 r, rp = p * 32 + g)
// a, b, c1, c2: THIS image's arrays (wave-uniform pointers -> scalar base + 32-bit lane offset addressing; 64-bit per-lane
// addresses for five arrays would cost dozens of registers).  No __restrict__ on them: the batches below are ordered by
// memory clobbers, which the compiler may ignore for loads it knows to be invariant.
// OUTER (the outer-loop refresh folded into the first inner iteration, algorithms/pnp_svrg.py:32-57 at j = 0): the
// scaled transform IS mu = grad_full(z); the epilogue stores it, stores w = z (the operand c1 it has to load anyway) and
// leaves z + gamma * mu in Z -- what the plain form computes from an all-zero difference z - w plus mu, bit for bit.
template <int STOP = 0, bool OUTER = false>
__device__ __forceinline__ void fused_gradient(cx<float> (&Z)[FP][16], const float* a, const float* b,
                                               const uint32_t* __restrict__ bits, const cx<float>* __restrict__ twtab, cx<float>* twl, cx<float>* ldc,
                                               uint32_t (*sbits)[FG][2][16], const cx<float>* __restrict__ yh, float scale, float beta,
                                               const float* c1, float gamma, const float* c2, int g, int l,
                                               float* w_out = nullptr, float* mu_out = nullptr) {
    cx<float>* scr = ldc + g * F_SCR;
    // selector bits of this group's four column pairs (two per half): fetched now, under the operand loads of phase 1 --
    // inside phase 2 their round trip to L2 sat exposed between two workgroup barriers, once per half
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = k * FG + g;
            const int ca = half == 0 ? idx : 64 + idx;
            const int cb = (half == 0 && idx == 0) ? 128 : FN - ca;
            sbits[half][g][k][l] = bits[(size_t)(l < 8 ? ca : cb) * 8 + (l & 7)];
        }
    cx<float> twv = {0.f, 0.f};
    if ((int)threadIdx.x < FN) twv = twtab[threadIdx.x];
    unsigned off[FP];                                       // element offset of (row 2rp, column l); row 2rp+1 is +FN
#pragma unroll
    for (int p = 0; p < FP; ++p) off[p] = (unsigned)(2 * (p * FG + g)) * FN + l;
    // ------------------------------------------------------------------ 1: rows forward
    // Register budget (256 per lane): the image is 128, one FFT needs ~64 of working space.  So the operands arrive in
    // batches that never coexist with a transform: `a` of all four row pairs (128 loads in flight per lane), then `b` two
    // row pairs at a time, then the four transforms.  (Left to itself the compiler issues all 256 loads at once and
    // spills; the memory clobbers pin the batches.)
#pragma unroll
    for (int p = 0; p < FP; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) Z[p][r] = {a[off[p] + 16 * r], a[off[p] + FN + 16 * r]};
    asm volatile("" ::: "memory");
    if (b != nullptr) {
#pragma unroll
        for (int p0 = 0; p0 < FP; p0 += 2) {
            cx<float> tb[2][16];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[k][r] = {b[off[p0 + k] + 16 * r], b[off[p0 + k] + FN + 16 * r]};
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) Z[p0 + k][r] = csub(Z[p0 + k][r], tb[k][r]);
            asm volatile("" ::: "memory");
        }
    }
    // twiddles: requested before the operands (twv above), landed with them; the barrier also orders the selector bits
    if ((int)threadIdx.x < FN) twl[threadIdx.x] = twv;
    __syncthreads();
    PNP_STAMP(1);
    if (STOP == 10) return;                                 // (diagnostic: the operand loads of phase 1 alone)
#pragma unroll
    for (int p = 0; p < FP; ++p) {
        fft256<false>(Z[p], twl, scr, l);
        asm volatile("" ::: "memory");
    }
    PNP_STAMP(2);
    if (STOP == 1) return;

    // ------------------------------------------------------------------ 2: columns, two halves
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                    // FFT scratch / previous half's reads are done
#pragma unroll
        for (int p = 0; p < FP; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kx = l + 16 * r;
                if (in_half_r(r, l, half)) ldc[kx_local(kx, half) * F_RS + p * FG + g] = Z[p][r];
            }
        // this group's two column pairs and their selector bits (rows ca and W - ca of the transposed bit mask)
        ColPair cp[2];
        const cx<float>* yhc[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = k * FG + g;                     // 0..63
            const int ca = half == 0 ? idx : 64 + idx;
            yhc[k] = yh != nullptr ? yh + (size_t)ca * FN : nullptr;
            const bool packed = half == 0 && idx == 0;      // columns 0 and 128
            const int cb = packed ? 128 : FN - ca;
            cp[k] = {kx_local(ca, half), kx_local(cb, half), packed};
        }
        __syncthreads();
        cx<float> v0[16], v1[16];
        col_load(v0, ldc, cp[0], l);
        col_load(v1, ldc, cp[1], l);
        __syncthreads();                                    // every group has its columns: the buffer becomes FFT scratch
        col_transform(v0, twl, scr, sbits[half][g][0], cp[0], l, yhc[0]);
        asm volatile("" ::: "memory");                      // one transform's working registers at a time
        col_transform(v1, twl, scr, sbits[half][g][1], cp[1], l, yhc[1]);
        __syncthreads();                                    // all FFT scratch use is over: the buffer carries data again
        col_store(v0, ldc, cp[0], l);
        col_store(v1, ldc, cp[1], l);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < FP; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kx = l + 16 * r;
                if (in_half_r(r, l, half)) Z[p][r] = ldc[kx_local(kx, half) * F_RS + p * FG + g];
            }
    }
    __syncthreads();
    PNP_STAMP(3);
    if (STOP == 2) return;

    // ------------------------------------------------------------------ 3: rows inverse + epilogue (in place in Z)
    // same budget: the four inverse transforms first, then the epilogue operands two row pairs at a time
#pragma unroll
    for (int p = 0; p < FP; ++p) {
        fft256<true>(Z[p], twl, scr, l);
        asm volatile("" ::: "memory");
    }
    PNP_STAMP(4);
#pragma unroll
    for (int p = 0; p < FP; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) Z[p][r] = {scale * Z[p][r].x, scale * Z[p][r].y};
    if (OUTER) {
#pragma unroll
        for (int p0 = 0; p0 < FP; p0 += 2) {
            cx<float> u[2][16];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) u[k][r] = {c1[off[p0 + k] + 16 * r], c1[off[p0 + k] + FN + 16 * r]};
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    mu_out[off[p0 + k] + 16 * r] = Z[p0 + k][r].x;
                    mu_out[off[p0 + k] + FN + 16 * r] = Z[p0 + k][r].y;
                }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    w_out[off[p0 + k] + 16 * r] = u[k][r].x;
                    w_out[off[p0 + k] + FN + 16 * r] = u[k][r].y;
                    Z[p0 + k][r] = {u[k][r].x + gamma * Z[p0 + k][r].x, u[k][r].y + gamma * Z[p0 + k][r].y};
                }
            asm volatile("" ::: "memory");
        }
        return;
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float* src = which == 0 ? c1 : c2;
        const float cf = which == 0 ? beta : gamma;
        if (src == nullptr) continue;
#pragma unroll
        for (int p0 = 0; p0 < FP; p0 += 2) {
            cx<float> u[2][16];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) u[k][r] = {src[off[p0 + k] + 16 * r], src[off[p0 + k] + FN + 16 * r]};
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Z[p0 + k][r] = {Z[p0 + k][r].x + cf * u[k][r].x, Z[p0 + k][r].y + cf * u[k][r].y};
            asm volatile("" ::: "memory");
        }
    }
}

}  // namespace pnp

// everything from here on follows pywt / skimage product for product: no FMA contraction (exact zeros in the wavelet
// coefficients are semantically significant); the FFT code above keeps the compiler's default (contraction on)
#pragma clang fp contract(off)
#include "prox_tv.h"

namespace pnp {

// MODE: 0 = the whole iteration; 1 = stop after the noise estimate and store the stepped image (another prox follows);
//       2 = the gradient only (phases 1-3, stored in row order: grad_full with its data term, or any other use of
//           pnp_csmri_grad_sel that fits this kernel).
// STOP (diagnostic builds, PNP_FUSED_STOP): leave after phase STOP (10 = after the operand loads of phase 1) with a checksum
// store, to time the phases one by one
enum { FUSED_FULL = 0, FUSED_NO_DENOISE = 1, FUSED_GRAD = 2 };
template <int MODE, int STOP, bool OUTER = false>
__global__ __launch_bounds__(FT) void k_svrg_iter(const float* a, const float* b,
                                                  const uint32_t* __restrict__ bitsT, const cx<float>* __restrict__ yh,
                                                  const cx<float>* __restrict__ twtab,
                                                  float scale, const float* __restrict__ alpha_vec, float beta, const float* c1,
                                                  float gamma, const float* c2, float* out,
                                                  float sigma_modifier, float fallback_sigma, const float* __restrict__ xrec,
                                                  double* __restrict__ sse_out, float* __restrict__ sigma_out,
                                                  float* w_out, float* mu_out) {
    constexpr bool DENOISE = MODE == FUSED_FULL;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cx<float>* ldc = reinterpret_cast<cx<float>*>(lds_raw);
    float* ldf = reinterpret_cast<float*>(lds_raw);
    __shared__ cx<float> twl[FN];
    __shared__ uint32_t sbits[2][FG][2][16];
    __shared__ double red[8];
    __shared__ float sig_sh;
    const int t = threadIdx.x, g = t >> 4, l = t & 15, wv = t >> 6, lane64 = t & 63;
    const int prob = blockIdx.x;
    const size_t img = (size_t)prob * FN * FN;
    // (the twiddle table goes to LDS inside fused_gradient, under the operand loads of phase 1)
    if (alpha_vec != nullptr) scale *= alpha_vec[prob];

    PNP_STAMP(0);
    cx<float> Z[FP][16];
    fused_gradient<(STOP == 1 || STOP == 2 || STOP == 10) ? STOP : 0, OUTER>(Z, a + img, b != nullptr ? b + img : nullptr, bitsT + (size_t)prob * FN * 8, twtab, twl, ldc, sbits,
                   yh != nullptr ? yh + (size_t)prob * (FN / 2) * FN : nullptr, scale, beta,
                   c1 != nullptr ? c1 + img : nullptr, gamma, c2 != nullptr ? c2 + img : nullptr, g, l,
                   OUTER ? w_out + img : nullptr, OUTER ? mu_out + img : nullptr);
    if (MODE == FUSED_GRAD) {
        float* oi = out + img;
#pragma unroll
        for (int p = 0; p < FP; ++p) {
            const unsigned o = (unsigned)(2 * (p * FG + g)) * FN + l;
#pragma unroll
            for (int r = 0; r < 16; ++r) { oi[o + 16 * r] = Z[p][r].x; oi[o + FN + 16 * r] = Z[p][r].y; }
        }
        return;
    }
    if (STOP == 3 || STOP == 1 || STOP == 2 || STOP == 10) {
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < FP; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc += Z[p][r].x + Z[p][r].y;
        out[img + t] = acc;
        return;
    }

    PNP_STAMP(5);
    // ------------------------------------------------------------------ 4: row-pair layout -> column layout
    // wave wv owns image columns [16 wv, 16 wv + 16) and [128 + 16 wv, ...); lane = column + 16 * chunk keeps rows
    // [64 chunk, 64 chunk + 64) of both
    float x[2][64];
    const int cl = lane64 & 15, q = lane64 >> 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < FP; ++p)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const cx<float> val = Z[p][r + 8 * half];
                float* dst = ldf + (l + 16 * r) * F_CS + 2 * (p * FG + g);
                dst[0] = val.x;
                dst[1] = val.y;
            }
        __syncthreads();
        const float* src = ldf + (16 * wv + cl) * F_CS + 64 * q;
#pragma unroll
        for (int i = 0; i < 64; ++i) x[half][i] = src[i];
    }
    if (STOP == 4) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) acc += x[0][i] + x[1][i];
        out[img + t] = acc;
        return;
    }

    // ------------------------------------------------------------------ 5: noise estimate, prox, error, store
    const unsigned base0 = (unsigned)(q * 64) * FN + 16 * wv + cl, base1 = base0 + 128;   // inside this image
    float* oi = out + img;
    const float* xri = xrec != nullptr ? xrec + img : nullptr;
    const bool want_err = DENOISE && xri != nullptr;
    // The ground truth for the error sum arrives by LDS-DMA while the noise estimate computes (the LDS is idle from here
    // on): columns [0, 128) of all 256 rows = 128 KiB as [row][128]; one wave instruction moves two rows (1 KiB).
    auto dma_xrec = [&](int colbase) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int pc = wv + 8 * k;                          // 128 pieces, 16 per wave
            const float* src = xri + (unsigned)(2 * pc + (lane64 >> 5)) * FN + colbase + 4 * (lane64 & 31);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ldf + pc * 256), 16, 0, 0);
        }
    };
    __syncthreads();                                            // the re-layout reads of every wave are done
    PNP_STAMP(6);
    if (want_err) dma_xrec(0);
    // sigma_est = mean over the 256 columns of the per-column MAD estimate
    {
        const float s0 = column_sigma<float, 64>(x[0], q), s1 = column_sigma<float, 64>(x[1], q);
        double part = q == 0 ? (double)s0 + (double)s1 : 0.0;
        part = wave_sum(part);
        if (lane64 == 0) red[wv] = part;
        __syncthreads();
        if (t == 0) {
            double s = 0;
            for (int i = 0; i < FT / 64; ++i) s += red[i];
            sig_sh = (float)(s / (double)FN);
        }
        __syncthreads();
    }
    const float sigma_est = sig_sh;
    PNP_STAMP(7);
    if (sigma_out != nullptr && t == 0) sigma_out[prob] = sigma_est;
    if (DENOISE) {
        const float sigma = sigma_est > 0.f ? sigma_est * sigma_modifier : fallback_sigma;
        haar_bayes_shrink<float, FN>(x[0], sigma * sigma);
        haar_bayes_shrink<float, FN>(x[1], sigma * sigma);
    }
    PNP_STAMP(8);
    double err = 0.0;
    const float* xl = ldf + (64 * q) * 128 + 16 * wv + cl;      // this lane's column in the staged [row][128] block
    if (want_err) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                        // columns [0, 128) of the ground truth have landed
        err = (double)column_sq_err<float, 64>(x[0], xl, 128);
        __syncthreads();
        dma_xrec(128);                                          // columns [128, 256) under the first half's stores
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) oi[base0 + (unsigned)i * FN] = x[0][i];
    if (want_err) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        err += (double)column_sq_err<float, 64>(x[1], xl, 128);
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) oi[base1 + (unsigned)i * FN] = x[1][i];
    if (sse_out != nullptr && want_err) {
        err = wave_sum(err);
        __syncthreads();
        if (lane64 == 0) red[wv] = err;
        __syncthreads();
        if (t == 0) {
            double s = 0;
            for (int i = 0; i < FT / 64; ++i) s += red[i];
            sse_out[prob] = s;
        }
    }
#ifdef PNP_FUSED_CLOCK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PNP_STAMP(9);
}

// plan internals live in csmri.hip (pnp_csmri_svrg_step / pnp_csmri_grad_sel); the kernel only needs the plan's twiddle table.
// mode: FUSED_FULL / FUSED_NO_DENOISE / FUSED_GRAD
int csmri_fused_launch(int batch, const void* twtab, const void* a, const void* b, const uint32_t* bitsT, const void* yh,
                       double alpha, const void* alpha_vec, double beta, const void* c1, double gamma, const void* c2, void* out,
                       int mode, double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_out, void* sigma_out,
                       void* stream, void* w_out, void* mu_out) {
    const float scale = (float)(alpha / ((double)FN * (double)FN));
    hipStream_t s = (hipStream_t)stream;
    const char* stop_env = getenv("PNP_FUSED_STOP");
    const int stop = stop_env ? atoi(stop_env) : 0;
    // > 64 KiB of dynamic LDS needs the opt-in, once per device (the attribute is per device)
    static unsigned long long attr_done = 0;
    int dev = 0;
    PNP_CHECK_HIP(hipGetDevice(&dev));
    if (!((attr_done >> (dev & 63)) & 1ull)) {
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<0, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        attr_done |= 1ull << (dev & 63);
    }
#define PNP_FUSED_LAUNCH(MD, ST, ...)                                                                                     \
    k_svrg_iter<MD, ST, ##__VA_ARGS__><<<batch, FT, F_LDS_BYTES, s>>>((const float*)a, (const float*)b, bitsT, (const cx<float>*)yh, \
                                                       (const cx<float>*)twtab, scale, (const float*)alpha_vec, (float)beta,    \
                                                       (const float*)c1, (float)gamma, (const float*)c2, (float*)out,         \
                                                       (float)sigma_modifier, (float)fallback_sigma, (const float*)xrec,      \
                                                       sse_out, (float*)sigma_out, (float*)w_out, (float*)mu_out)
    if (w_out != nullptr) {                                     // the outer refresh folded into the first inner iteration
        if (mode == FUSED_FULL) PNP_FUSED_LAUNCH(0, 0, true);
        else PNP_FUSED_LAUNCH(1, 0, true);
    } else if (mode == FUSED_GRAD) PNP_FUSED_LAUNCH(2, 0);
    else if (stop == 1) PNP_FUSED_LAUNCH(0, 1);
    else if (stop == 2) PNP_FUSED_LAUNCH(0, 2);
    else if (stop == 10) PNP_FUSED_LAUNCH(0, 10);
    else if (stop == 3) PNP_FUSED_LAUNCH(0, 3);
    else if (stop == 4) PNP_FUSED_LAUNCH(0, 4);
    else if (mode == FUSED_FULL) PNP_FUSED_LAUNCH(0, 0);
    else PNP_FUSED_LAUNCH(1, 0);
#undef PNP_FUSED_LAUNCH
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

}  // namespace pnp
