// prox.hip -- noise estimate + "TV" (per-column Haar BayesShrink) prox + PSNR error sum, fused.
//
// Replaces, per inner iteration of every reference loop (e.g. algorithms/pnp_svrg.py:70-80):
//   estimate_sigma(z0, multichannel=True, average_sigmas=True)   [skimage; SURVEY F3, a17]
//   TVDenoiser.denoise(z0, sigma_est)                            [denoisers/TV.py:21-26; SURVEY F2, a18]
//   Problem.PSNR(z0)                                             [problems/problem.py:33-35]
//
// One workgroup owns one image.  A wavefront owns 16 adjacent columns x 4 row-chunks
// (lane = col16 + 16*chunk); every lane keeps its H/4 rows of one column in VGPRs, so the
// whole column pipeline -- db2 detail coefficients, the MAD median (bitwise radix select on the
// IEEE bit patterns), the multi-level Haar analysis, per-level BayesShrink thresholds, synthesis
// and the squared error against the ground truth -- runs out of registers with only
// 16-lane-stride wave shuffles between the four lanes of a column.  The image is read once and
// written once.
#include "common.h"
#include "reduce.h"
#include "keys.h"

namespace pnp {

// Reductions over the 4 lanes {l, l^16, l^32, l^48} that share a column, on gfx950's v_permlane16_swap /
// v_permlane32_swap (VALU): swapping a register with a copy of itself leaves the even-row (lower-half) value in one
// result and the odd-row (upper-half) value in the other, for both lanes of a pair.  A ds_bpermute shuffle costs ~32
// LDS-pipe cycles per wave instruction, and the median's radix select does two of these reductions per bit.
// Same pairing as (v + v^16) + (v^32 + v^48), so sums are bit-identical to the shuffle form.
template <typename T> __device__ __forceinline__ void pair16(T v, T& a, T& b) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte values");
    if constexpr (sizeof(T) == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
        a = __builtin_bit_cast(T, (unsigned)r[0]); b = __builtin_bit_cast(T, (unsigned)r[1]);
    } else {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
        a = __builtin_bit_cast(T, ((unsigned long long)hi[0] << 32) | lo[0]);
        b = __builtin_bit_cast(T, ((unsigned long long)hi[1] << 32) | lo[1]);
    }
}
template <typename T> __device__ __forceinline__ void pair32(T v, T& a, T& b) {
    if constexpr (sizeof(T) == 4) {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
        a = __builtin_bit_cast(T, (unsigned)r[0]); b = __builtin_bit_cast(T, (unsigned)r[1]);
    } else {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)u, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
        a = __builtin_bit_cast(T, ((unsigned long long)hi[0] << 32) | lo[0]);
        b = __builtin_bit_cast(T, ((unsigned long long)hi[1] << 32) | lo[1]);
    }
}
template <typename T> __device__ __forceinline__ T col_sum(T v) {
    T a, b;
    pair16(v, a, b); v = a + b;
    pair32(v, a, b); v = a + b;
    return v;
}
template <typename K> __device__ __forceinline__ K col_min(K v) {
    K a, b;
    pair16(v, a, b); v = b < a ? b : a;
    pair32(v, a, b); v = b < a ? b : a;
    return v;
}

// MAD sigma of one column from this lane's chunk x[0..RPC) (all 4 lanes of the column get it).
template <typename T, int RPC>
__device__ __forceinline__ T column_sigma(const T (&x)[RPC], int q) {
    using K = typename KeyOf<T>::type;
    constexpr int NC = RPC / 2 + 1;
    K key[NC];
    // halo: the two samples above this chunk (symmetric extension at the top edge)
    T up1 = __shfl_up(x[RPC - 1], 16, 64), up2 = __shfl_up(x[RPC - 2], 16, 64);
    const T m1 = q == 0 ? x[0] : up1;      // x[-1]
    const T m2 = q == 0 ? x[1] : up2;      // x[-2]
    bool has_nan = false;
    int n = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        // no FMA contraction: exact zeros must stay exact zeros (they are masked out of the median)
#pragma clang fp contract(off)
        T d;
        if (i == 0)              d = ((Db2<T>::h0 * x[1] + Db2<T>::h1 * x[0]) + Db2<T>::h2 * m1) + Db2<T>::h3 * m2;
        else if (i < RPC / 2)    d = ((Db2<T>::h0 * x[2 * i + 1] + Db2<T>::h1 * x[2 * i]) + Db2<T>::h2 * x[2 * i - 1]) + Db2<T>::h3 * x[2 * i - 2];
        else                     d = ((Db2<T>::h0 * x[RPC - 2] + Db2<T>::h1 * x[RPC - 1]) + Db2<T>::h2 * x[RPC - 1]) + Db2<T>::h3 * x[RPC - 2];
        if (i == RPC / 2 && q != 3) d = (T)0;          // only the bottom chunk owns the extra coefficient
        d = d < 0 ? -d : d;
        has_nan |= (d != d);
        const bool nz = d != (T)0;
        n += nz ? 1 : 0;
        key[i] = nz ? to_key(d) : ~(K)0;               // zeros are masked out of the median
    }
    n = col_sum(n);
    has_nan = col_sum((int)has_nan) != 0;
    // k-th smallest by bitwise radix select over the (monotone) bit patterns of |d|
    const int k = (n - 1) >> 1;
    K pfx = 0;
#pragma unroll 1
    for (int bit = KeyOf<T>::BITS - 1; bit >= 0; --bit) {
        const K cand = pfx | ((K)1 << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) c += key[i] < cand ? 1 : 0;
        c = col_sum(c);
        if (c <= k) pfx = cand;
    }
    T med = from_key(pfx);
    if ((n & 1) == 0) {                                // even count: mean of the two middle values
        int cle = 0;
        K nxt = ~(K)0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            cle += key[i] <= pfx ? 1 : 0;
            if (key[i] > pfx && key[i] < nxt) nxt = key[i];
        }
        cle = col_sum(cle);
        nxt = col_min(nxt);
        const T hi = cle > k + 1 ? med : from_key(nxt);
        med = (med + hi) * (T)0.5;
    }
    if (n == 0 || has_nan) med = (T)NAN;               // np.median([]) / NaN input
    return med / (T)0.6744897501960817;
}

template <int H> struct HaarLevels { static constexpr int value = (H >= 256 ? 5 : H >= 128 ? 4 : H >= 64 ? 3 : H >= 32 ? 2 : 1); };

// MODE bit 0: denoise (else estimate only)
template <typename T, int H, bool DENOISE>
__global__ __launch_bounds__(1024) void k_prox_tv(const T* __restrict__ zin, T* __restrict__ zout, int W,
                                                  const T* __restrict__ sigma_in, T sigma_modifier, T fallback_sigma,
                                                  const T* __restrict__ xrec, double* __restrict__ sse_out,
                                                  T* __restrict__ sigma_out) {
    constexpr int RPC = H / 4;
    constexpr int L = HaarLevels<H>::value;
    constexpr T HA = (T)0.7071067811865476;
    __shared__ double red[16];
    __shared__ T sig_sh;
    const int prob = blockIdx.x;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 15, q = lane >> 4;
    const int nwaves = blockDim.x >> 6;
    const int col = wv * 16 + cl;
    const size_t base = (size_t)prob * H * W + (size_t)(q * RPC) * W + col;

    T x[RPC];
#pragma unroll
    for (int i = 0; i < RPC; ++i) x[i] = zin[base + (size_t)i * W];

    // ---------------- sigma_est = mean over columns of the per-column MAD estimate
    T sigma_est;
    if (sigma_in != nullptr) {
        sigma_est = sigma_in[prob];
    } else {
        T sc = column_sigma<T, RPC>(x, q);
        double part = q == 0 ? (double)sc : 0.0;
        part = wave_sum(part);
        if (lane == 0) red[wv] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0;
            for (int i = 0; i < nwaves; ++i) s += red[i];
            sig_sh = (T)(s / (double)W);
        }
        __syncthreads();
        sigma_est = sig_sh;
    }
    if (sigma_out != nullptr && threadIdx.x == 0) sigma_out[prob] = sigma_est;
    if (!DENOISE) return;

    // ---------------- per-column Haar BayesShrink (TV.py:22-26)
    const T sigma = sigma_est > (T)0 ? sigma_est * sigma_modifier : fallback_sigma;
    const T var = sigma * sigma;
    T thr[L];
#pragma unroll
    for (int lev = 0; lev < L; ++lev) {
        const int s = 1 << lev;
        T ss = 0;
#pragma unroll
        for (int j = 0; j < RPC / (2 * s); ++j) {
            const T ev = x[2 * s * j], od = x[2 * s * j + s];
            const T d = -HA * od + HA * ev;
            x[2 * s * j] = HA * od + HA * ev;
            x[2 * s * j + s] = d;
            ss += d * d;
        }
        ss = col_sum(ss);
        const T dvar = ss / (T)(H >> (lev + 1));
        T den = dvar - var;
        den = den > (T)2.220446049250313e-16 ? den : (T)2.220446049250313e-16;
        thr[lev] = var / sqrt(den);
    }
#pragma unroll
    for (int lev = L - 1; lev >= 0; --lev) {
        const int s = 1 << lev;
#pragma unroll
        for (int j = 0; j < RPC / (2 * s); ++j) {
            const T a = x[2 * s * j];
            T d = x[2 * s * j + s];
            const T mag = d < 0 ? -d : d;
            T shr = (T)1 - thr[lev] / mag;
            shr = shr < (T)0 ? (T)0 : shr;             // keeps NaN (0/0) like numpy clip
            d = d * shr;
            x[2 * s * j] = HA * a + HA * d;
            x[2 * s * j + s] = HA * a - HA * d;
        }
    }

    // ---------------- store + squared error against the ground truth
    double err = 0.0;
    if (xrec != nullptr) {
        T e = 0;
#pragma unroll
        for (int i = 0; i < RPC; ++i) {
            const T df = xrec[base + (size_t)i * W] - x[i];
            e += df * df;
        }
        err = (double)e;
    }
#pragma unroll
    for (int i = 0; i < RPC; ++i) zout[base + (size_t)i * W] = x[i];
    if (sse_out != nullptr) {
        err = wave_sum(err);
        __syncthreads();
        if (lane == 0) red[wv] = err;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0;
            for (int i = 0; i < nwaves; ++i) s += red[i];
            sse_out[prob] = s;
        }
    }
}

// ------------------------------------------------------------------------------- reductions
template <typename T>
__global__ __launch_bounds__(256) void k_sse(const T* __restrict__ z, const T* __restrict__ xr, int n, double* __restrict__ out) {
    __shared__ double red[4];
    const size_t base = (size_t)blockIdx.x * n;
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double d = (double)xr[base + i] - (double)z[base + i];
        acc += d * d;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ void k_axpbypcz(T a, const T* x, T b, const T* y, T c, const T* w, T* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = a * x[i];
        if (y != nullptr) v += b * y[i];
        if (w != nullptr) v += c * w[i];
        out[i] = v;
    }
}

template <typename T, int H, bool DENOISE>
int launch_prox(const void* zin, void* zout, int W, int batch, const void* sigma_in, double mod, double fb,
                const void* xrec, double* sse, void* sigma_out, hipStream_t s) {
    k_prox_tv<T, H, DENOISE><<<batch, (W / 16) * 64, 0, s>>>((const T*)zin, (T*)zout, W, (const T*)sigma_in, (T)mod,
                                                          (T)fb, (const T*)xrec, sse, (T*)sigma_out);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

template <bool DENOISE>
int dispatch_prox(const void* zin, void* zout, int H, int W, int batch, int dtype, const void* sigma_in, double mod,
                  double fb, const void* xrec, double* sse, void* sigma_out, hipStream_t s) {
    PNP_CHECK_ARG(zin != nullptr && batch >= 1, "null input / empty batch");
    PNP_CHECK_ARG(H == 16 || H == 32 || H == 64 || H == 128 || H == 256, "H must be 16, 32, 64, 128 or 256");
    PNP_CHECK_ARG(W % 16 == 0 && W >= 16 && W <= 256, "W must be a multiple of 16 in [16, 256]");
    PNP_CHECK_ARG(dtype == PNP_F32 || dtype == PNP_F64, "bad dtype");
#define PNP_PROX_CASE(TT, HH) return launch_prox<TT, HH, DENOISE>(zin, zout, W, batch, sigma_in, mod, fb, xrec, sse, sigma_out, s)
    if (dtype == PNP_F32) {
        if (H == 256) PNP_PROX_CASE(float, 256);
        if (H == 128) PNP_PROX_CASE(float, 128);
        if (H == 64) PNP_PROX_CASE(float, 64);
        if (H == 32) PNP_PROX_CASE(float, 32);
        PNP_PROX_CASE(float, 16);
    }
    if (H == 256) PNP_PROX_CASE(double, 256);
    if (H == 128) PNP_PROX_CASE(double, 128);
    if (H == 64) PNP_PROX_CASE(double, 64);
    if (H == 32) PNP_PROX_CASE(double, 32);
    PNP_PROX_CASE(double, 16);
#undef PNP_PROX_CASE
}

}  // namespace pnp

using namespace pnp;

extern "C" int pnp_sigma_est(const void* z, int H, int W, int batch, int dtype, void* sigma_out, void* stream) {
    PNP_CHECK_ARG(sigma_out != nullptr, "null output");
    return dispatch_prox<false>(z, nullptr, H, W, batch, dtype, nullptr, 1.0, 0.0, nullptr, nullptr, sigma_out,
                                (hipStream_t)stream);
}

extern "C" int pnp_prox_tv(const void* z_in, void* z_out, int H, int W, int batch, int dtype, const void* sigma_in,
                           double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_out,
                           void* sigma_out, void* stream) {
    PNP_CHECK_ARG(z_out != nullptr, "null output");
    return dispatch_prox<true>(z_in, z_out, H, W, batch, dtype, sigma_in, sigma_modifier, fallback_sigma, xrec, sse_out,
                               sigma_out, (hipStream_t)stream);
}

extern "C" int pnp_sse(const void* z, const void* xrec, int n, int batch, int dtype, double* sse_out, void* stream) {
    PNP_CHECK_ARG(z && xrec && sse_out && n > 0 && batch > 0, "bad argument");
    if (dtype == PNP_F32) k_sse<float><<<batch, 256, 0, (hipStream_t)stream>>>((const float*)z, (const float*)xrec, n, sse_out);
    else if (dtype == PNP_F64) k_sse<double><<<batch, 256, 0, (hipStream_t)stream>>>((const double*)z, (const double*)xrec, n, sse_out);
    else PNP_CHECK_ARG(false, "bad dtype");
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_minmax(const void* z, int n, int batch, int dtype, void* out, void* stream) {
    PNP_CHECK_ARG(z && out && n > 0 && batch > 0, "bad argument");
    if (dtype == PNP_F32) k_minmax<float><<<batch, 256, 0, (hipStream_t)stream>>>((const float*)z, n, (float*)out);
    else if (dtype == PNP_F64) k_minmax<double><<<batch, 256, 0, (hipStream_t)stream>>>((const double*)z, n, (double*)out);
    else PNP_CHECK_ARG(false, "bad dtype");
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

// ---- device-resident step counter + log ring: what lets a whole outer iteration be captured in a hipGraph
__global__ void k_counter_add(uint32_t* ctr, uint32_t inc) { if (threadIdx.x == 0 && blockIdx.x == 0) *ctr += inc; }

__global__ void k_log_append(const double* __restrict__ src, int n, double* __restrict__ log, int n_log,
                             const uint32_t* __restrict__ step_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) log[(size_t)(*step_dev % (uint32_t)n_log) * n + i] = src[i];
}

extern "C" int pnp_counter_add(uint32_t* counter, uint32_t inc, void* stream) {
    PNP_CHECK_ARG(counter != nullptr, "null counter");
    k_counter_add<<<1, 64, 0, (hipStream_t)stream>>>(counter, inc);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_log_append(const double* src, int n, double* log, int n_log, const uint32_t* step_dev, void* stream) {
    PNP_CHECK_ARG(src && log && step_dev && n >= 1 && n_log >= 1, "bad argument");
    k_log_append<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(src, n, log, n_log, step_dev);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_axpbypcz(double a, const void* x, double b, const void* y, double c, const void* w, void* out,
                            size_t n, int dtype, void* stream) {
    PNP_CHECK_ARG(x && out, "null argument");
    if (n == 0) return PNP_OK;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (dtype == PNP_F32)
        k_axpbypcz<float><<<grid, 256, 0, (hipStream_t)stream>>>((float)a, (const float*)x, (float)b, (const float*)y, (float)c, (const float*)w, (float*)out, n);
    else if (dtype == PNP_F64)
        k_axpbypcz<double><<<grid, 256, 0, (hipStream_t)stream>>>(a, (const double*)x, b, (const double*)y, c, (const double*)w, (double*)out, n);
    else PNP_CHECK_ARG(false, "bad dtype");
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}
