// csmri.hip -- masked-FFT data-fidelity gradient for CSMRI on gfx950.
//
// Replaces reference problems/CSMRI.py:76-81 (grad_full) and :83-89 (grad_stoch):
//     g = Re ifft2( sel o fft2(x) - sel o Y )
// The real part of an inverse FFT only sees the Hermitian part of its argument, so the
// whole thing runs on the half spectrum with real<->complex transforms:
//
//   k_rows_fwd   two real image rows -> one complex FFT-W -> split -> packed half spectrum,
//                written TRANSPOSED ([kx][h]) through an LDS tile (256-B segments)
//   k_cols       per kx column: FFT-H -> symmetrised selector (and data term) -> inverse FFT-H,
//                in one kernel, in place (forward and inverse share the axis)
//   k_rows_inv   transposed read -> Hermitian re-expansion -> one complex inverse FFT-W gives two
//                real rows -> fused epilogue  out = alpha*g + beta*c1 + gamma*c2
//
// "Packed": column kx=0 stores (X[.,0], X[.,W/2]) as (re,im) -- both are real after the row
// pass -- so the half spectrum is exactly [W/2][H] complex = the bytes of the real image.
#include "fft.h"
#include "keys.h"
#include <vector>
#include <cstdlib>
#include <cmath>

namespace pnp {

template <typename T> struct alignas(4 * sizeof(T)) vec4 { T a, b, c, d; };

// RA x LA = the register x lane split of one length-N transform (fft.h: fft_gen); RA == LA for N = 64, 256,
// <8,16> for N = 128.  A lane group is LG = max(RA, LA) lanes.
template <typename T, int RA, int LA> struct FftSmem {
    static constexpr int N = RA * LA;
    static constexpr int LG = RA > LA ? RA : LA;
    static constexpr int G = 256 / LG;                       // lane groups per 256-thread block
    static constexpr int TILE = G * (N + 1);                 // [group][N+1] complex
    static constexpr int SCR = G * LG * (LG + 1);            // [group][LG][LG+1] complex
    static constexpr int ELEMS = TILE > SCR ? TILE : SCR;
};

// ------------------------------------------------------------------------------- rows forward
template <typename T, int RA, int LA>
__global__ __launch_bounds__(256) void k_rows_fwd(const T* __restrict__ a, const T* __restrict__ b,
                                                  cx<T>* __restrict__ S1T, const cx<T>* __restrict__ twtab, int H) {
    using S = FftSmem<T, RA, LA>;
    constexpr int N = S::N, G = S::G, LG = S::LG;
    __shared__ cx<T> smem[S::ELEMS];
    const int t = threadIdx.x, g = t / LG, lane = t % LG;
    const int prob = blockIdx.y, h0 = blockIdx.x * 2 * G;
    const size_t img = (size_t)prob * H * N;
    const size_t ra = img + (size_t)(h0 + 2 * g) * N, rb = ra + N;

    cx<T> v[LG], tw[LG];
    load_twiddles_gen<T, LG>(tw, twtab, lane, N);
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int w = (lane < LA ? lane : 0) + LA * r;
        T va = a[ra + w], vb = a[rb + w];
        if (b != nullptr) { va -= b[ra + w]; vb -= b[rb + w]; }
        v[r] = {va, vb};
    }
    fft_gen<T, RA, LA, false>(v, tw, smem + g * LG * (LG + 1), lane);
    __syncthreads();
    if (lane < RA) {
#pragma unroll
        for (int r = 0; r < LA; ++r) smem[g * (N + 1) + lane + RA * r] = v[r];
    }
    __syncthreads();

    // split the two interleaved real transforms and store transposed
    const int p = t % G;
    const cx<T>* zp = smem + p * (N + 1);
    for (int kx = t / G; kx < N / 2; kx += 256 / G) {
        const cx<T> zk = zp[kx], zm = zp[(N - kx) & (N - 1)];
        vec4<T> o;
        if (kx == 0) {
            const cx<T> zn = zp[N / 2];
            o = {zk.x, zn.x, zk.y, zn.y};
        } else {
            o = {(T)0.5 * (zk.x + zm.x), (T)0.5 * (zk.y - zm.y), (T)0.5 * (zk.y + zm.y), (T)-0.5 * (zk.x - zm.x)};
        }
        *reinterpret_cast<vec4<T>*>(S1T + ((size_t)prob * (N / 2) + kx) * H + h0 + 2 * p) = o;
    }
}

// ------------------------------------------------------------------------------- columns
template <typename T, int RA, int LA>
__global__ __launch_bounds__(256) void k_cols(cx<T>* __restrict__ S1T, const uint8_t* __restrict__ selT,
                                              const cx<T>* __restrict__ yh, const cx<T>* __restrict__ twtab, int W) {
    using S = FftSmem<T, RA, LA>;
    constexpr int N = S::N, G = S::G, LG = S::LG;            // N = H
    __shared__ cx<T> smem[S::SCR];
    const int t = threadIdx.x, g = t / LG, lane = t % LG;
    const int prob = blockIdx.y, c = blockIdx.x * G + g;
    cx<T>* col = S1T + ((size_t)prob * (W / 2) + c) * N;
    cx<T>* scr = smem + g * LG * (LG + 1);
    const bool act = lane < RA;                              // lanes that hold spectrum values after the forward pass
    const int ln = act ? lane : 0;

    cx<T> v[LG], tw[LG];
    load_twiddles_gen<T, LG>(tw, twtab, lane, N);
#pragma unroll
    for (int r = 0; r < RA; ++r) v[r] = col[(lane < LA ? lane : 0) + LA * r];
    fft_gen<T, RA, LA, false>(v, tw, scr, lane);

    const uint8_t* sp = selT + (size_t)prob * W * N;
    if (blockIdx.x == 0) {
        // the packed column c == 0 holds two real-input transforms: separate, weight, re-pack
        __syncthreads();
        if (act) {
#pragma unroll
            for (int r = 0; r < LA; ++r) scr[lane + RA * r] = v[r];
        }
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < LA; ++r) {
                const int ky = ln + RA * r, km = (N - ky) & (N - 1);
                const cx<T> pk = v[r], pm = scr[km];
                const cx<T> A = {(T)0.5 * (pk.x + pm.x), (T)0.5 * (pk.y - pm.y)};
                const cx<T> B = {(T)0.5 * (pk.y + pm.y), (T)-0.5 * (pk.x - pm.x)};
                const T wA = (T)0.5 * (T)(sp[ky] + sp[km]);
                const T wB = (T)0.5 * (T)(sp[(size_t)(W / 2) * N + ky] + sp[(size_t)(W / 2) * N + km]);
                v[r] = {wA * A.x - wB * B.y, wA * A.y + wB * B.x};
            }
        }
    }
    if (c != 0) {
        const uint8_t* s1 = sp + (size_t)c * N;
        const uint8_t* s2 = sp + (size_t)(W - c) * N;
#pragma unroll
        for (int r = 0; r < LA; ++r) {
            const int ky = ln + RA * r, km = (N - ky) & (N - 1);
            const T wgt = (T)0.5 * (T)(s1[ky] + s2[km]);
            v[r] = {wgt * v[r].x, wgt * v[r].y};
        }
    }
    if (yh != nullptr) {
        const cx<T>* yc = yh + ((size_t)prob * (W / 2) + c) * N;
#pragma unroll
        for (int r = 0; r < LA; ++r) v[r] = csub(v[r], yc[ln + RA * r]);
    }
    fft_gen<T, LA, RA, true>(v, tw, scr, lane);
    if (lane < LA) {
#pragma unroll
        for (int r = 0; r < RA; ++r) col[lane + LA * r] = v[r];
    }
}

// ------------------------------------------------------------------------------- rows inverse + epilogue
template <typename T, int RA, int LA>
__global__ __launch_bounds__(256) void k_rows_inv(const cx<T>* __restrict__ S1T, const cx<T>* __restrict__ twtab, int H,
                                                  T alpha, T beta, const T* c1, T gamma, const T* c2, T* out) {
    using S = FftSmem<T, RA, LA>;
    constexpr int N = S::N, G = S::G, LG = S::LG;
    __shared__ cx<T> smem[S::ELEMS];
    const int t = threadIdx.x, g = t / LG, lane = t % LG;
    const int prob = blockIdx.y, h0 = blockIdx.x * 2 * G;

    const int p = t % G;
    cx<T>* zp = smem + p * (N + 1);
    for (int kx = t / G; kx < N / 2; kx += 256 / G) {
        const vec4<T> q = *reinterpret_cast<const vec4<T>*>(S1T + ((size_t)prob * (N / 2) + kx) * H + h0 + 2 * p);
        if (kx == 0) {
            zp[0] = {q.a, q.c};
            zp[N / 2] = {q.b, q.d};
        } else {
            zp[kx] = {q.a - q.d, q.b + q.c};                 // A + iB
            zp[N - kx] = {q.a + q.d, q.c - q.b};             // conj(A) + i conj(B)
        }
    }
    __syncthreads();
    cx<T> v[LG], tw[LG];
    load_twiddles_gen<T, LG>(tw, twtab, lane, N);
#pragma unroll
    for (int r = 0; r < LA; ++r) v[r] = smem[g * (N + 1) + (lane < RA ? lane : 0) + RA * r];
    fft_gen<T, LA, RA, true>(v, tw, smem + g * LG * (LG + 1), lane);

    const size_t ra = (size_t)prob * H * N + (size_t)(h0 + 2 * g) * N, rb = ra + N;
    if (lane < LA) {
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int w = lane + LA * r;
            T oa = alpha * v[r].x, ob = alpha * v[r].y;
            if (c1 != nullptr) { oa += beta * c1[ra + w]; ob += beta * c1[rb + w]; }
            if (c2 != nullptr) { oa += gamma * c2[ra + w]; ob += gamma * c2[rb + w]; }
            out[ra + w] = oa;
            out[rb + w] = ob;
        }
    }
}

// ------------------------------------------------------------------------------- rows inverse + line-wise TV prox
// The last pass of the inverse transform leaves every NL-lane group holding two complete storage rows (element
// lane + NL*r in register r).  When the caller keeps its images TRANSPOSED (storage row = image column) those rows
// are exactly the lines the "TV" prox works along (per-column db2 noise estimate + multi-level Haar BayesShrink,
// denoisers/TV.py:21-26 and estimate_sigma, see prox.hip), so the gradient step, the noise estimate, the prox and
// the PSNR error run in ONE kernel: the stepped image never goes to HBM before it is denoised.  One workgroup owns
// one image (the noise estimate is a mean over all of its lines).  Cross-lane steps are shuffles inside the NL-lane
// group; arithmetic follows prox.hip product for product (no FMA contraction in the prox part).
// MEASURED SLOWER than k_rows_inv + k_prox_tv (B = 256, 256 x 256: 281 vs 115 us) and therefore opt-in only: with
// one 1024-thread workgroup per CU the phases run in lock step (no other workgroup hides a phase's memory latency:
// the bare inverse pass takes 98 us here against 56 us as a streaming kernel), and the group shuffles compile to
// ds_bpermute_b32 at ~32 LDS-pipe cycles per wave instruction (the Haar levels alone cost 118 us; DPP row shifts
// would bring that to ~15 us, still short of break-even).  Kept as the measured negative result behind DESIGN 3.3.
template <typename T, int NL> __device__ __forceinline__ T group_sum(T v) {
#pragma unroll
    for (int o = NL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, NL);
    return v;
}
template <typename K, int NL> __device__ __forceinline__ K group_min(K v) {
#pragma unroll
    for (int o = NL / 2; o > 0; o >>= 1) { const K u = __shfl_xor(v, o, NL); v = u < v ? u : v; }
    return v;
}

// MAD noise estimate of one line (all lanes of the group get it).  Odd lanes own the db2 detail coefficient of the
// sample pair (2i, 2i+1) that ends on them; the coefficient past the end (symmetric extension) sits on the last lane.
template <typename T, int NL>
__device__ __forceinline__ T line_sigma(const T (&x)[NL], int lane) {
#pragma clang fp contract(off)
    using K = typename KeyOf<T>::type;
    K key[NL + 1];
    bool has_nan = false;
    int n = 0;
    const bool odd = (lane & 1) != 0;
#pragma unroll
    for (int r = 0; r <= NL; ++r) {
        T d;
        bool use;
        if (r < NL) {
            const T m1 = __shfl_up(x[r], 1, NL);
            T m2 = __shfl_up(x[r], 2, NL), m3 = __shfl_up(x[r], 3, NL);
            if (r > 0) {                                       // lane 1 reaches back into the previous register
                const T w1 = __shfl(x[r - 1], NL - 1, NL), w2 = __shfl(x[r - 1], NL - 2, NL);
                if (lane == 1) { m2 = w1; m3 = w2; }
            } else if (lane == 1) {                            // start of the line: x[-1] = x[0], x[-2] = x[1]
                m2 = m1; m3 = x[0];
            }
            d = ((Db2<T>::h0 * x[r] + Db2<T>::h1 * m1) + Db2<T>::h2 * m2) + Db2<T>::h3 * m3;
            use = odd;
        } else {                                               // end of the line: x[N] = x[N-1], x[N+1] = x[N-2]
            const T own = x[NL - 1], m1 = __shfl_up(x[NL - 1], 1, NL);
            d = ((Db2<T>::h0 * m1 + Db2<T>::h1 * own) + Db2<T>::h2 * own) + Db2<T>::h3 * m1;
            use = lane == NL - 1;
        }
        d = d < 0 ? -d : d;
        has_nan |= use && (d != d);
        const bool nz = use && d != (T)0;
        n += nz ? 1 : 0;
        key[r] = nz ? to_key(d) : ~(K)0;                       // zeros (and non-owning lanes) are masked out
    }
    n = group_sum<int, NL>(n);
    has_nan = group_sum<int, NL>((int)has_nan) != 0;
    const int k = (n - 1) >> 1;
    K pfx = 0;
#pragma unroll 1
    for (int bit = KeyOf<T>::BITS - 1; bit >= 0; --bit) {
        const K cand = pfx | ((K)1 << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i <= NL; ++i) c += key[i] < cand ? 1 : 0;
        c = group_sum<int, NL>(c);
        if (c <= k) pfx = cand;
    }
    T med = from_key(pfx);
    if ((n & 1) == 0) {
        int cle = 0;
        K nxt = ~(K)0;
#pragma unroll
        for (int i = 0; i <= NL; ++i) {
            cle += key[i] <= pfx ? 1 : 0;
            if (key[i] > pfx && key[i] < nxt) nxt = key[i];
        }
        cle = group_sum<int, NL>(cle);
        nxt = group_min<K, NL>(nxt);
        const T hi = cle > k + 1 ? med : from_key(nxt);
        med = (med + hi) * (T)0.5;
    }
    if (n == 0 || has_nan) med = (T)NAN;
    return med / (T)0.6744897501960817;
}

// Multi-level Haar BayesShrink of one line in place (levels as prox.hip: 5 for 256, 3 for 64).
template <typename T, int NL>
__device__ __forceinline__ void line_haar_shrink(T (&x)[NL], int lane, T var) {
#pragma clang fp contract(off)
    constexpr int N = NL * NL;
    constexpr int L = N >= 256 ? 5 : N >= 128 ? 4 : N >= 64 ? 3 : N >= 32 ? 2 : 1;
    constexpr T HA = (T)0.7071067811865476;
    T thr[L];
#pragma unroll
    for (int lev = 0; lev < L; ++lev) {
        const int s = 1 << lev;
        T ss = 0;
        if (s < NL) {                                          // partner sample lives s lanes away, same register
            const bool ev = (lane & (2 * s - 1)) == 0, od = (lane & (2 * s - 1)) == s;
#pragma unroll
            for (int r = 0; r < NL; ++r) {
                const T p = __shfl_xor(x[r], s, NL);
                if (ev) x[r] = HA * p + HA * x[r];
                else if (od) { const T d = -HA * x[r] + HA * p; x[r] = d; ss += d * d; }
            }
        } else if (lane == 0) {                                // partner is another register of lane 0
            const int s2 = s / NL;
#pragma unroll
            for (int j = 0; j < NL / (2 * s2); ++j) {
                const T e = x[2 * s2 * j], o = x[2 * s2 * j + s2];
                const T d = -HA * o + HA * e;
                x[2 * s2 * j] = HA * o + HA * e;
                x[2 * s2 * j + s2] = d;
                ss += d * d;
            }
        }
        ss = group_sum<T, NL>(ss);
        const T dvar = ss / (T)(N >> (lev + 1));
        T den = dvar - var;
        den = den > (T)2.220446049250313e-16 ? den : (T)2.220446049250313e-16;
        thr[lev] = var / sqrt(den);
    }
#pragma unroll
    for (int lev = L - 1; lev >= 0; --lev) {
        const int s = 1 << lev;
        if (s < NL) {
            const bool ev = (lane & (2 * s - 1)) == 0, od = (lane & (2 * s - 1)) == s;
#pragma unroll
            for (int r = 0; r < NL; ++r) {
                T val = x[r];
                if (od) {
                    const T mag = val < 0 ? -val : val;
                    T shr = (T)1 - thr[lev] / mag;
                    shr = shr < (T)0 ? (T)0 : shr;             // keeps NaN (0/0) like numpy clip
                    val = val * shr;
                }
                const T p = __shfl_xor(val, s, NL);
                if (ev) x[r] = HA * val + HA * p;
                else if (od) x[r] = HA * p - HA * val;
            }
        } else if (lane == 0) {
            const int s2 = s / NL;
#pragma unroll
            for (int j = 0; j < NL / (2 * s2); ++j) {
                const T a = x[2 * s2 * j];
                T d = x[2 * s2 * j + s2];
                const T mag = d < 0 ? -d : d;
                T shr = (T)1 - thr[lev] / mag;
                shr = shr < (T)0 ? (T)0 : shr;
                d = d * shr;
                x[2 * s2 * j] = HA * a + HA * d;
                x[2 * s2 * j + s2] = HA * a - HA * d;
            }
        }
    }
}

template <int NL> struct FusedGeom {
    static constexpr int N = NL * NL;                          // square images: H == W == N
    static constexpr int THREADS = (N / 2) * NL > 1024 ? 1024 : (N / 2) * NL;
    static constexpr int G = THREADS / NL;                     // complex rows in flight
    static constexpr int IT = (N / 2) / G;                     // passes over the image
    static constexpr int TILE = G * (N + 1), SCR = G * NL * (NL + 1);
    static constexpr int ELEMS = TILE > SCR ? TILE : SCR;      // complex elements of dynamic LDS
};

template <typename T, int NL>
__global__ __launch_bounds__(1024) void k_rows_inv_prox(const cx<T>* __restrict__ S1T, const cx<T>* __restrict__ twtab,
                                                        T alpha, T beta, const T* c1, T gamma, const T* c2, T* out,
                                                        T sigma_modifier, T fallback_sigma, const T* __restrict__ xrec,
                                                        double* __restrict__ sse_out, T* __restrict__ sigma_out) {
    using F = FusedGeom<NL>;
    constexpr int N = F::N, G = F::G, IT = F::IT, H = F::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cx<T>* smem = reinterpret_cast<cx<T>*>(smem_raw);
    __shared__ double red[16];
    __shared__ T sig_sh;
    const int t = threadIdx.x, g = t / NL, lane = t % NL, wv = t >> 6;
    const int prob = blockIdx.x, p = t % G;
    constexpr int NW = F::THREADS / 64;

    T x[IT][2][NL];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int h0 = it * 2 * G;
        if (it > 0) __syncthreads();
        cx<T>* zp = smem + p * (N + 1);
        for (int kx = t / G; kx < N / 2; kx += NL) {
            const vec4<T> q = *reinterpret_cast<const vec4<T>*>(S1T + ((size_t)prob * (N / 2) + kx) * H + h0 + 2 * p);
            if (kx == 0) {
                zp[0] = {q.a, q.c};
                zp[N / 2] = {q.b, q.d};
            } else {
                zp[kx] = {q.a - q.d, q.b + q.c};
                zp[N - kx] = {q.a + q.d, q.c - q.b};
            }
        }
        __syncthreads();
        cx<T> v[NL], tw[NL];
        load_twiddles_gen<T, NL>(tw, twtab, lane, N);
#pragma unroll
        for (int r = 0; r < NL; ++r) v[r] = smem[g * (N + 1) + lane + NL * r];
        fft_gen<T, NL, NL, true>(v, tw, smem + g * NL * (NL + 1), lane);
        const size_t ra = (size_t)prob * H * N + (size_t)(h0 + 2 * g) * N, rb = ra + N;
#pragma unroll
        for (int r = 0; r < NL; ++r) {
            const int w = lane + NL * r;
            T oa = alpha * v[r].x, ob = alpha * v[r].y;
            if (c1 != nullptr) { oa += beta * c1[ra + w]; ob += beta * c1[rb + w]; }
            if (c2 != nullptr) { oa += gamma * c2[ra + w]; ob += gamma * c2[rb + w]; }
            x[it][0][r] = oa;
            x[it][1][r] = ob;
        }
    }

    // ---------------- noise estimate: mean over the image's lines of the per-line MAD estimate
    double part = 0.0;
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const T sc = line_sigma<T, NL>(x[it][j], lane);
            part += lane == 0 ? (double)sc : 0.0;
        }
    part = wave_sum(part);
    __syncthreads();
    if ((t & 63) == 0) red[wv] = part;
    __syncthreads();
    if (t == 0) {
        double s = 0;
        for (int i = 0; i < NW; ++i) s += red[i];
        sig_sh = (T)(s / (double)H);
    }
    __syncthreads();
    const T sigma_est = sig_sh;
    if (sigma_out != nullptr && t == 0) sigma_out[prob] = sigma_est;
    const T sigma = sigma_est > (T)0 ? sigma_est * sigma_modifier : fallback_sigma;
    const T var = sigma * sigma;

    // ---------------- prox, error sum, store
    double err = 0.0;
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            line_haar_shrink<T, NL>(x[it][j], lane, var);
            const size_t row = (size_t)prob * H * N + (size_t)(it * 2 * G + 2 * g + j) * N;
            if (xrec != nullptr) {
                T e = 0;
#pragma unroll
                for (int r = 0; r < NL; ++r) {
                    const T df = xrec[row + lane + NL * r] - x[it][j][r];
                    e += df * df;
                }
                err += (double)e;
            }
#pragma unroll
            for (int r = 0; r < NL; ++r) out[row + lane + NL * r] = x[it][j][r];
        }
    if (sse_out != nullptr) {
        err = wave_sum(err);
        __syncthreads();
        if ((t & 63) == 0) red[wv] = err;
        __syncthreads();
        if (t == 0) {
            double s = 0;
            for (int i = 0; i < NW; ++i) s += red[i];
            sse_out[prob] = s;
        }
    }
}

// ------------------------------------------------------------------------------- data term
template <typename T>
__global__ void k_pack_y(const cx<T>* __restrict__ YT, const uint8_t* __restrict__ selT, cx<T>* __restrict__ yh, int H, int W) {
    const int prob = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (W / 2) * H) return;
    const int c = i / H, ky = i % H;
    const cx<T>* Y = YT + (size_t)prob * W * H;
    const uint8_t* S = selT + (size_t)prob * W * H;
    auto term = [&](int cc) -> cx<T> {
        const size_t i1 = (size_t)cc * H + ky, i2 = (size_t)((W - cc) % W) * H + ((H - ky) % H);
        const T s1 = (T)S[i1], s2 = (T)S[i2];
        const cx<T> y1 = Y[i1], y2 = Y[i2];
        return {(T)0.5 * (s1 * y1.x + s2 * y2.x), (T)0.5 * (s1 * y1.y - s2 * y2.y)};
    };
    cx<T> o;
    if (c == 0) {
        const cx<T> A = term(0), B = term(W / 2);
        o = {A.x - B.y, A.y + B.x};
    } else {
        o = term(c);
    }
    yh[(size_t)prob * (W / 2) * H + i] = o;
}

__global__ void k_sel_scatter(const int32_t* __restrict__ idx, int n, uint8_t* __restrict__ selT, int H, int W) {
    const int prob = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int i = idx[(size_t)prob * n + j];
    if (i < 0 || i >= H * W) return;
    selT[(size_t)prob * H * W + (size_t)(i % W) * H + i / W] = 1;
}

__global__ void k_sel_transpose(const uint8_t* __restrict__ sel, uint8_t* __restrict__ selT, int H, int W) {
    __shared__ uint8_t tile[32][33];
    const int prob = blockIdx.z;
    const uint8_t* s = sel + (size_t)prob * H * W;
    uint8_t* d = selT + (size_t)prob * H * W;
    int x = blockIdx.x * 32 + threadIdx.x, y = blockIdx.y * 32 + threadIdx.y;
    for (int j = 0; j < 32; j += 8)
        if (x < W && y + j < H) tile[threadIdx.y + j][threadIdx.x] = s[(size_t)(y + j) * W + x] != 0;
    __syncthreads();
    x = blockIdx.y * 32 + threadIdx.x;
    y = blockIdx.x * 32 + threadIdx.y;
    for (int j = 0; j < 32; j += 8)
        if (x < H && y + j < W) d[(size_t)(y + j) * H + x] = tile[threadIdx.x][threadIdx.y + j];
}

// ------------------------------------------------------------------------------- minibatch draw
// problems/CSMRI.py:66-74 (`np.random.choice(flatnonzero(mask), size, replace=False)`) as a device kernel:
// every sampled k-space location gets an i.i.d. 32-bit key from a counter-based hash of
// (seed, step, problem, position); the `mb` smallest keys are the minibatch (uniform without replacement).
// The threshold key is found by a 4-pass 8-bit radix select with an LDS histogram; equal keys at the
// threshold (probability ~2^-32 per pair) are resolved by position, so a draw is deterministic.
// The kernel also clears and fills the transposed selector, replacing memset + scatter.
// NOT the NumPy legacy stream: reference-identical draws still come from the host (CSMRI.select_mb).
__device__ __forceinline__ uint32_t mb_hash(uint64_t seed, uint32_t step, uint32_t prob, uint32_t j) {
    uint64_t x = seed ^ ((uint64_t)step << 40) ^ ((uint64_t)prob << 20) ^ (uint64_t)j;
    x += 0x9E3779B97F4A7C15ull;                                 // splitmix64 finaliser
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}

// CACHE: the keys are hashed once into dynamic LDS (M0 words) and the five sweeps read them back; otherwise
// (M0 too large for LDS) every sweep re-hashes.  Same keys either way, so the draw does not depend on it.
template <bool CACHE>
__global__ __launch_bounds__(1024) void k_draw_mb(const int32_t* __restrict__ mask_idx, int M0, int mb, uint64_t seed,
                                                  uint32_t step, const uint32_t* __restrict__ step_dev,
                                                  uint8_t* __restrict__ selT, int H, int W) {
    if (step_dev != nullptr) step += *step_dev;                  // device-resident counter (hipGraph replays)
    extern __shared__ uint32_t keys[];
    __shared__ int hist[256];
    __shared__ int s_bin, s_before, s_ntie;
    __shared__ int tie[64];
    const int prob = blockIdx.x, tid = threadIdx.x;
    const int32_t* idx = mask_idx + (size_t)prob * M0;
    uint8_t* out = selT + (size_t)prob * H * W;
    for (int i = tid; i < H * W / 16; i += 1024) reinterpret_cast<uint4*>(out)[i] = make_uint4(0, 0, 0, 0);
    if (CACHE)
        for (int j = tid; j < M0; j += 1024) keys[j] = mb_hash(seed, step, prob, j);   // visible after the first barrier below

    uint32_t prefix = 0;
    int k = mb;                                                  // rank (1-based) still to locate inside the prefix bucket
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int j = tid; j < M0; j += 1024) {
            const uint32_t key = CACHE ? keys[j] : mb_hash(seed, step, prob, j);
            if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
        }
        __syncthreads();
        if (tid < 64) {                                          // one wave: 4 bins per lane, inclusive scan
            const int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const int tot = c0 + c1 + c2 + c3;
            int incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (tid >= o) incl += v;
            }
            const int excl = incl - tot;
            if (excl < k && k <= incl) {                         // exactly one lane
                int before = excl, bin = 4 * tid;
                if (k > before + c0) { before += c0; ++bin;
                    if (k > before + c1) { before += c1; ++bin;
                        if (k > before + c2) { before += c2; ++bin; } } }
                s_bin = bin;
                s_before = before;
            }
        }
        __syncthreads();
        prefix = (prefix << 8) | (uint32_t)s_bin;
        k -= s_before;
        __syncthreads();
    }
    // prefix = the mb-th smallest key T; take every key < T and the k first (by position) keys == T
    if (tid == 0) s_ntie = 0;
    __syncthreads();
    for (int j = tid; j < M0; j += 1024) {
        const uint32_t key = CACHE ? keys[j] : mb_hash(seed, step, prob, j);
        if (key < prefix) {
            const int i = idx[j];
            out[(size_t)(i % W) * H + i / W] = 1;
        } else if (key == prefix) {
            const int t = atomicAdd(&s_ntie, 1);
            if (t < 64) tie[t] = j;
        }
    }
    __syncthreads();
    const int ntie = s_ntie < 64 ? s_ntie : 64;
    if (tid < ntie) {
        const int j = tie[tid];
        int rank = 0;
        for (int t = 0; t < ntie; ++t) rank += tie[t] < j ? 1 : 0;
        if (rank < k) {
            const int i = idx[j];
            out[(size_t)(i % W) * H + i / W] = 1;
        }
    }
}

template <typename T> void fill_twiddles(std::vector<cx<T>>& tab, int N) {
    tab.resize(N);
    for (int j = 0; j < N; ++j) {
        const double ang = -2.0 * 3.14159265358979323846 * j / N;
        tab[j] = {(T)cos(ang), (T)sin(ang)};
    }
}

}  // namespace pnp

using namespace pnp;

struct pnp_csmri_plan {
    int H, W, batch, dtype, NL;              // NL: 16 -> N = 256, 8 -> N = 64, 12 -> N = 128 (8 x 16 split)
    void* work;     // [batch][W/2][H] complex
    void* twtab;    // [N] complex
};

extern "C" int pnp_csmri_plan_create(pnp_csmri_plan** out, int H, int W, int batch, int dtype) {
    PNP_CHECK_ARG(out != nullptr, "null plan pointer");
    PNP_CHECK_ARG(H == W && (H == 64 || H == 128 || H == 256), "supported sizes: H == W in {64, 128, 256}");
    PNP_CHECK_ARG(batch >= 1, "batch must be >= 1");
    PNP_CHECK_ARG(dtype == PNP_F32 || dtype == PNP_F64, "dtype must be PNP_F32 or PNP_F64");
    auto* p = new pnp_csmri_plan{H, W, batch, dtype, H == 256 ? 16 : H == 128 ? 12 : 8, nullptr, nullptr};
    const size_t esz = dtype == PNP_F32 ? 8 : 16;
    hipError_t e = hipMalloc(&p->work, (size_t)batch * (W / 2) * H * esz);
    if (e == hipSuccess) e = hipMalloc(&p->twtab, (size_t)H * esz);
    if (e == hipSuccess) {
        if (dtype == PNP_F32) {
            std::vector<cx<float>> tab; fill_twiddles(tab, H);
            e = hipMemcpy(p->twtab, tab.data(), H * esz, hipMemcpyHostToDevice);
        } else {
            std::vector<cx<double>> tab; fill_twiddles(tab, H);
            e = hipMemcpy(p->twtab, tab.data(), H * esz, hipMemcpyHostToDevice);
        }
    }
    if (e != hipSuccess) {
        set_error(std::string("pnp_csmri_plan_create: ") + hipGetErrorString(e));
        if (p->work) (void)hipFree(p->work);
        if (p->twtab) (void)hipFree(p->twtab);
        delete p;
        return PNP_ERR_HIP;
    }
    *out = p;
    return PNP_OK;
}

extern "C" int pnp_csmri_plan_destroy(pnp_csmri_plan* p) {
    if (p == nullptr) return PNP_OK;
    (void)hipFree(p->work);
    (void)hipFree(p->twtab);
    delete p;
    return PNP_OK;
}

extern "C" int pnp_csmri_sel_from_indices(pnp_csmri_plan* p, const int32_t* idx, int n, uint8_t* selT, void* stream) {
    PNP_CHECK_ARG(p && idx && selT && n >= 0, "null argument");
    hipStream_t s = (hipStream_t)stream;
    PNP_CHECK_HIP(hipMemsetAsync(selT, 0, (size_t)p->batch * p->H * p->W, s));
    if (n > 0) {
        k_sel_scatter<<<dim3((n + 255) / 256, p->batch), 256, 0, s>>>(idx, n, selT, p->H, p->W);
        PNP_CHECK_LAUNCH();
    }
    return PNP_OK;
}

extern "C" int pnp_csmri_draw_minibatch(pnp_csmri_plan* p, const int32_t* mask_idx, int M0, int mb, uint64_t seed,
                                        uint32_t step, const uint32_t* step_dev, uint8_t* selT, void* stream) {
    PNP_CHECK_ARG(p && mask_idx && selT, "null argument");
    PNP_CHECK_ARG(M0 >= 1 && M0 <= p->H * p->W && mb >= 1 && mb <= M0, "need 1 <= mb <= M0 <= H*W");
    constexpr int kMaxCachedKeys = 36 * 1024;                   // 144 KiB of the CU's 160 KiB LDS
    if (M0 <= kMaxCachedKeys) {
        static bool attr_set = false;                           // > 64 KiB of dynamic LDS needs the opt-in, once
        if (!attr_set) {
            PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_draw_mb<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              kMaxCachedKeys * (int)sizeof(uint32_t)));
            attr_set = true;
        }
        k_draw_mb<true><<<p->batch, 1024, (size_t)M0 * sizeof(uint32_t), (hipStream_t)stream>>>(mask_idx, M0, mb, seed, step,
                                                                                                step_dev, selT, p->H, p->W);
    } else {
        k_draw_mb<false><<<p->batch, 1024, 0, (hipStream_t)stream>>>(mask_idx, M0, mb, seed, step, step_dev, selT, p->H, p->W);
    }
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_csmri_sel_from_dense(pnp_csmri_plan* p, const uint8_t* sel, uint8_t* selT, void* stream) {
    PNP_CHECK_ARG(p && sel && selT, "null argument");
    k_sel_transpose<<<dim3((p->W + 31) / 32, (p->H + 31) / 32, p->batch), dim3(32, 8), 0, (hipStream_t)stream>>>(
        sel, selT, p->H, p->W);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_csmri_pack_y(pnp_csmri_plan* p, const void* YT, const uint8_t* selT, void* yh, void* stream) {
    PNP_CHECK_ARG(p && YT && selT && yh, "null argument");
    const int n = (p->W / 2) * p->H;
    dim3 grid((n + 255) / 256, p->batch);
    if (p->dtype == PNP_F32)
        k_pack_y<float><<<grid, 256, 0, (hipStream_t)stream>>>((const cx<float>*)YT, selT, (cx<float>*)yh, p->H, p->W);
    else
        k_pack_y<double><<<grid, 256, 0, (hipStream_t)stream>>>((const cx<double>*)YT, selT, (cx<double>*)yh, p->H, p->W);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

namespace {
template <typename T, int RA, int LA>
int run_grad(pnp_csmri_plan* p, const void* a, const void* b, const uint8_t* selT, const void* yh, double alpha,
             double beta, const void* c1, double gamma, const void* c2, void* out, hipStream_t s) {
    constexpr int G = FftSmem<T, RA, LA>::G;
    const int H = p->H, W = p->W;
    cx<T>* work = (cx<T>*)p->work;
    const cx<T>* tw = (const cx<T>*)p->twtab;
    const T scale = (T)(alpha / ((double)H * (double)W));
    k_rows_fwd<T, RA, LA><<<dim3(H / (2 * G), p->batch), 256, 0, s>>>((const T*)a, (const T*)b, work, tw, H);
    PNP_CHECK_LAUNCH();
    k_cols<T, RA, LA><<<dim3((W / 2) / G, p->batch), 256, 0, s>>>(work, selT, (const cx<T>*)yh, tw, W);
    PNP_CHECK_LAUNCH();
    k_rows_inv<T, RA, LA><<<dim3(H / (2 * G), p->batch), 256, 0, s>>>(work, tw, H, scale, (T)beta, (const T*)c1, (T)gamma,
                                                                 (const T*)c2, (T*)out);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}
}  // namespace

extern "C" int pnp_csmri_grad(pnp_csmri_plan* p, const void* a, const void* b, const uint8_t* selT, const void* yh,
                              double alpha, double beta, const void* c1, double gamma, const void* c2, void* out,
                              void* stream) {
    PNP_CHECK_ARG(p && a && selT && out, "null argument");
    hipStream_t s = (hipStream_t)stream;
    if (p->dtype == PNP_F32) {
        if (p->NL == 16) return run_grad<float, 16, 16>(p, a, b, selT, yh, alpha, beta, c1, gamma, c2, out, s);
        if (p->NL == 12) return run_grad<float, 8, 16>(p, a, b, selT, yh, alpha, beta, c1, gamma, c2, out, s);
        return run_grad<float, 8, 8>(p, a, b, selT, yh, alpha, beta, c1, gamma, c2, out, s);
    }
    if (p->NL == 16) return run_grad<double, 16, 16>(p, a, b, selT, yh, alpha, beta, c1, gamma, c2, out, s);
    if (p->NL == 12) return run_grad<double, 8, 16>(p, a, b, selT, yh, alpha, beta, c1, gamma, c2, out, s);
    return run_grad<double, 8, 8>(p, a, b, selT, yh, alpha, beta, c1, gamma, c2, out, s);
}

namespace {
template <int NL>
int run_grad_prox(pnp_csmri_plan* p, const void* a, const void* b, const uint8_t* selT, const void* yh, double alpha,
                  double beta, const void* c1, double gamma, const void* c2, void* out, double sigma_modifier,
                  double fallback_sigma, const void* xrec, double* sse_out, void* sigma_out, hipStream_t s) {
    using T = float;
    using F = FusedGeom<NL>;
    constexpr int G = FftSmem<T, NL, NL>::G;
    const int H = p->H, W = p->W;
    cx<T>* work = (cx<T>*)p->work;
    const cx<T>* tw = (const cx<T>*)p->twtab;
    const T scale = (T)(alpha / ((double)H * (double)W));
    k_rows_fwd<T, NL, NL><<<dim3(H / (2 * G), p->batch), 256, 0, s>>>((const T*)a, (const T*)b, work, tw, H);
    PNP_CHECK_LAUNCH();
    k_cols<T, NL, NL><<<dim3((W / 2) / G, p->batch), 256, 0, s>>>(work, selT, (const cx<T>*)yh, tw, W);
    PNP_CHECK_LAUNCH();
    constexpr size_t lds = (size_t)F::ELEMS * sizeof(cx<T>);
    static bool attr_set = false;                               // > 64 KiB of dynamic LDS needs the opt-in, once
    if (!attr_set) {
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_rows_inv_prox<T, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    k_rows_inv_prox<T, NL><<<p->batch, F::THREADS, lds, s>>>(work, tw, scale, (T)beta, (const T*)c1, (T)gamma, (const T*)c2,
                                                            (T*)out, (T)sigma_modifier, (T)fallback_sigma, (const T*)xrec,
                                                            sse_out, (T*)sigma_out);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}
}  // namespace

extern "C" int pnp_csmri_grad_prox_tv(pnp_csmri_plan* p, const void* a, const void* b, const uint8_t* selT, const void* yh,
                                      double alpha, double beta, const void* c1, double gamma, const void* c2, void* out,
                                      double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_out,
                                      void* sigma_out, void* stream) {
    PNP_CHECK_ARG(p && a && selT && out, "null argument");
    PNP_CHECK_ARG(!(sse_out && !xrec), "sse_out needs xrec");
    PNP_CHECK_ARG(p->dtype == PNP_F32 && (p->NL == 16 || p->NL == 8), "fused gradient + prox: f32 plans of 64 x 64 or 256 x 256");
    hipStream_t s = (hipStream_t)stream;
    if (p->NL == 16)
        return run_grad_prox<16>(p, a, b, selT, yh, alpha, beta, c1, gamma, c2, out, sigma_modifier, fallback_sigma, xrec,
                                 sse_out, sigma_out, s);
    return run_grad_prox<8>(p, a, b, selT, yh, alpha, beta, c1, gamma, c2, out, sigma_modifier, fallback_sigma, xrec, sse_out,
                            sigma_out, s);
}
