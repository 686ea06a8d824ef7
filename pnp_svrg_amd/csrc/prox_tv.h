// prox_tv.h -- the register-level column pipeline of the "TV" prox (noise estimate + per-column Haar BayesShrink +
// PSNR error), shared by prox.hip (k_prox_tv) and csmri_fused.hip (whole-iteration kernel).  Arithmetic follows
// pywt / skimage product for product: translation units that include this header are compiled with
// -ffp-contract=off (exact zeros in the wavelet coefficients are semantically significant).
#pragma once
#include "common.h"
#include "keys.h"
#pragma clang fp contract(off)

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

// In-register transpose of a 32 x 32 bit matrix: on return bit i of a[b] is what bit b of a[i] was.  Five butterfly
// stages (J = 16, 8, 4, 2, 1): rows k and k + J trade the low row's high part for the high row's low part.
template <int J> __device__ __forceinline__ void transpose32_stage(uint32_t (&a)[32]) {
    constexpr uint32_t m = J == 16 ? 0x0000FFFFu : J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu : J == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (k & J) continue;
        const uint32_t lo = a[k], hi = a[k + J];
        a[k] = (lo & m) | ((hi & m) << J);
        a[k + J] = ((lo >> J) & m) | (hi & ~m);
    }
}
__device__ __forceinline__ void transpose32(uint32_t (&a)[32]) {
    transpose32_stage<16>(a);
    transpose32_stage<8>(a);
    transpose32_stage<4>(a);
    transpose32_stage<2>(a);
    transpose32_stage<1>(a);
}

template <int N> struct IntC { static constexpr int value = N; };
// f(IntC<BIT>{}) for BIT = FROM, FROM - 1, ..., 0
template <int FROM, typename F> __device__ __forceinline__ void radix_bits(F&& f) {
    f(IntC<FROM>{});
    if constexpr (FROM > 0) radix_bits<FROM - 1>(f);
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
    if constexpr (sizeof(T) == 4 && NC == 33) {
        // Bit-sliced form (H = 256, f32): keys 0..31 are transposed into 32 bit planes (plane b, bit i = bit b of key i:
        // a 32 x 32 bit-matrix transpose in registers, five butterfly stages), after which one radix step is a mask, a
        // population count and the column reduction -- ~20 vector instructions per bit instead of two per key and bit
        // (66 + reduction).  The 33rd key (the bottom chunk's extra coefficient) rides along as a flag.  Same result:
        // the (k+1)-th smallest key; masked keys (all ones) are never below a candidate.
        uint32_t pl[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) pl[i] = key[i];
        transpose32(pl);
        const uint32_t k32 = key[32];
        uint32_t alive = 0xFFFFFFFFu, alive32 = 1u;
        int kk = k;
        // (unrolled by recursion: planes are registers, so the bit index must be a compile-time constant; clang declines
        // `#pragma unroll` on this loop because of the convergent cross-lane reductions in its body)
        radix_bits<30>([&](auto BIT) {
            constexpr int bit = decltype(BIT)::value;
            const uint32_t zeros = alive & ~pl[bit];
            const uint32_t b32 = (k32 >> bit) & 1u, z32 = alive32 & (b32 ^ 1u);
            const int c = col_sum((int)(__builtin_popcount(zeros) + z32));     // alive keys of the column with this bit clear
            const bool lower = kk < c;                                       // the wanted key is among them
            alive = lower ? zeros : (alive & pl[bit]);
            alive32 = lower ? z32 : (alive32 & b32);
            kk = lower ? kk : kk - c;
            pfx |= lower ? 0u : (1u << bit);
        });
    } else {
#pragma unroll 1
        for (int bit = KeyOf<T>::BITS - 1; bit >= 0; --bit) {
            const K cand = pfx | ((K)1 << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < NC; ++i) c += key[i] < cand ? 1 : 0;
            c = col_sum(c);
            if (c <= k) pfx = cand;
        }
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

// per-column multi-level Haar BayesShrink in place (denoisers/TV.py:22-26 -> skimage denoise_wavelet on one column):
// this lane holds RPC = H/4 consecutive rows of its column in x[]; var = sigma^2
template <typename T, int H>
__device__ __forceinline__ void haar_bayes_shrink(T (&x)[H / 4], T var) {
    constexpr int RPC = H / 4;
    constexpr int L = HaarLevels<H>::value;
    constexpr T HA = (T)0.7071067811865476;
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
}

// sum over this lane's rows of (xrec - x)^2; xr points at the element of row 0 of the chunk, rows are `stride` apart
template <typename T, int RPC>
__device__ __forceinline__ T column_sq_err(const T (&x)[RPC], const T* __restrict__ xr, int stride) {
    T e = 0;
#pragma unroll
    for (int i = 0; i < RPC; ++i) {
        const T df = xr[(size_t)i * stride] - x[i];
        e += df * df;
    }
    return e;
}

// The column pipeline on a register-resident image: this lane holds rows [q*RPC, (q+1)*RPC) of column `col` in x[]
// (lane = col16 + 16*q inside wave wv, which owns columns [16 wv, 16 wv + 16)); base = element offset of x[0].
// Whole-workgroup collective (block barriers): every thread of the one-image workgroup must call it.
template <typename T, int H, bool DENOISE>
__device__ __forceinline__ void prox_tv_regs(T (&x)[H / 4], int prob, int W, size_t base, int wv, int lane, int q, int nwaves,
                                             const T* __restrict__ sigma_in, T sigma_modifier, T fallback_sigma,
                                             const T* __restrict__ xrec, T* zout, double* __restrict__ sse_out,
                                             T* __restrict__ sigma_out, double* red, T* sig_sh) {
    constexpr int RPC = H / 4;
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
            *sig_sh = (T)(s / (double)W);
        }
        __syncthreads();
        sigma_est = *sig_sh;
    }
    if (sigma_out != nullptr && threadIdx.x == 0) sigma_out[prob] = sigma_est;
    if (DENOISE) {
        const T sigma = sigma_est > (T)0 ? sigma_est * sigma_modifier : fallback_sigma;
        haar_bayes_shrink<T, H>(x, sigma * sigma);
    }
    if (zout == nullptr) return;

    // ---------------- store + squared error against the ground truth
    double err = 0.0;
    if (xrec != nullptr) err = (double)column_sq_err<T, RPC>(x, xrec + base, W);
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

}  // namespace pnp
