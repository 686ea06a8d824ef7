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
#include "draw.h"
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
// Selector forms of the column pass:
//   SEL_U8   explicit transposed uint8 selector [W][H] (host-drawn minibatches: reference-identical index lists)
//   SEL_BITS a bit-packed transposed selector: word [kx][ky >> 5], bit ky & 31 -- the sampling mask itself (grad_full),
//            or mask o device-drawn minibatch as k_draw_thr emits it (8 KiB per problem instead of a 64 KiB byte
//            selector written by one kernel and read by the next)
enum { SEL_U8 = 0, SEL_BITS = 1 };

template <typename T, int RA, int LA, int SEL>
__global__ __launch_bounds__(256) void k_cols(cx<T>* __restrict__ S1T, const uint8_t* __restrict__ selT,
                                              const uint32_t* __restrict__ bitsT,
                                              const cx<T>* __restrict__ yh, const cx<T>* __restrict__ YT,
                                              const cx<T>* __restrict__ twtab, int W) {
    using S = FftSmem<T, RA, LA>;
    constexpr int N = S::N, G = S::G, LG = S::LG;            // N = H
    constexpr int WPR = N / 32;                              // mask words per k-space column
    __shared__ cx<T> smem[S::SCR];
    __shared__ uint32_t sb[2][G][WPR];                       // mask bits of this block's columns c (slot 0) and W - c (slot 1)
    const int t = threadIdx.x, g = t / LG, lane = t % LG;
    const int prob = blockIdx.y, c = blockIdx.x * G + g;
    cx<T>* col = S1T + ((size_t)prob * (W / 2) + c) * N;
    cx<T>* scr = smem + g * LG * (LG + 1);
    const bool act = lane < RA;                              // lanes that hold spectrum values after the forward pass
    const int ln = act ? lane : 0;

    if (SEL != SEL_U8 && t < 2 * G * WPR) {
        const int slot = t / (G * WPR), gg = (t / WPR) % G, wd = t % WPR, cc = blockIdx.x * G + gg;
        const int kx = cc == 0 ? (slot == 0 ? 0 : W / 2) : (slot == 0 ? cc : W - cc);     // packed column 0 = kx 0 and W/2
        sb[slot][gg][wd] = bitsT[((size_t)prob * W + kx) * WPR + wd];
    }
    // selector value at (kx, ky); slot says in which staged row kx sits
    auto sel_at = [&](int slot, int kx, int ky, const uint8_t* row) -> T {
        if (SEL == SEL_U8) return (T)row[ky];
        return (T)((sb[slot][g][ky >> 5] >> (ky & 31)) & 1u);
    };

    cx<T> v[LG], tw[LG];
    load_twiddles_gen<T, LG>(tw, twtab, lane, N);
#pragma unroll
    for (int r = 0; r < RA; ++r) v[r] = col[(lane < LA ? lane : 0) + LA * r];
    fft_gen<T, RA, LA, false>(v, tw, scr, lane);             // (its barriers also publish sb)

    const uint8_t* sp = SEL == SEL_U8 ? selT + (size_t)prob * W * N : nullptr;
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
                const T a1 = sel_at(0, 0, ky, sp), a2 = sel_at(0, 0, km, sp);
                const T b1 = sel_at(1, W / 2, ky, sp + (size_t)(W / 2) * N), b2 = sel_at(1, W / 2, km, sp + (size_t)(W / 2) * N);
                const T wA = (T)0.5 * (a1 + a2), wB = (T)0.5 * (b1 + b2);
                v[r] = {wA * A.x - wB * B.y, wA * A.y + wB * B.x};
                if (YT != nullptr) {                             // data term of this selector, built on the fly (k_pack_y)
                    const cx<T>* y0 = YT + (size_t)prob * W * N;
                    const cx<T>* yn = y0 + (size_t)(W / 2) * N;
                    const cx<T> p1 = y0[ky], p2 = y0[km], q1 = yn[ky], q2 = yn[km];
                    const cx<T> ya = {(T)0.5 * (a1 * p1.x + a2 * p2.x), (T)0.5 * (a1 * p1.y - a2 * p2.y)};
                    const cx<T> yb = {(T)0.5 * (b1 * q1.x + b2 * q2.x), (T)0.5 * (b1 * q1.y - b2 * q2.y)};
                    v[r] = csub(v[r], cx<T>{ya.x - yb.y, ya.y + yb.x});
                }
            }
        }
    }
    if (c != 0) {
        const uint8_t* s1 = sp + (size_t)c * N;
        const uint8_t* s2 = sp + (size_t)(W - c) * N;
#pragma unroll
        for (int r = 0; r < LA; ++r) {
            const int ky = ln + RA * r, km = (N - ky) & (N - 1);
            const T w1 = sel_at(0, c, ky, s1), w2 = sel_at(1, W - c, km, s2);
            const T wgt = (T)0.5 * (w1 + w2);
            v[r] = {wgt * v[r].x, wgt * v[r].y};
            if (YT != nullptr) {
                const cx<T> y1 = YT[((size_t)prob * W + c) * N + ky], y2 = YT[((size_t)prob * W + (W - c)) * N + km];
                v[r] = csub(v[r], cx<T>{(T)0.5 * (w1 * y1.x + w2 * y2.x), (T)0.5 * (w1 * y1.y - w2 * y2.y)});
            }
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
                                                  T alpha, const T* __restrict__ alpha_vec, T beta, const T* c1, T gamma,
                                                  const T* c2, T* out) {
    using S = FftSmem<T, RA, LA>;
    constexpr int N = S::N, G = S::G, LG = S::LG;
    __shared__ cx<T> smem[S::ELEMS];
    const int t = threadIdx.x, g = t / LG, lane = t % LG;
    const int prob = blockIdx.y, h0 = blockIdx.x * 2 * G;
    if (alpha_vec != nullptr) alpha *= alpha_vec[prob];        // per-problem 1/M0 of a mixed-mask batch

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

// ------------------------------------------------------------------------------- mask bits
// uint8 transposed selector [W][H] -> bit-packed [W][H/32] (bit ky & 31 of word [kx][ky >> 5])
__global__ void k_pack_bits(const uint8_t* __restrict__ selT, uint32_t* __restrict__ bitsT, int nwords) {
    const int prob = blockIdx.y;
    const int wd = blockIdx.x * blockDim.x + threadIdx.x;
    if (wd >= nwords) return;
    const uint4* src = reinterpret_cast<const uint4*>(selT + ((size_t)prob * nwords + wd) * 32);
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint4 v = src[q];
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) m |= (((w4[j] >> (8 * k)) & 0xFFu) != 0 ? 1u : 0u) << (16 * q + 4 * j + k);
    }
    bitsT[(size_t)prob * nwords + wd] = m;
}

// materialise mask o minibatch as a transposed uint8 selector (tests, explicit-selector callers)
__global__ void k_sel_from_thr(const uint32_t* __restrict__ bitsT, const MbDesc* __restrict__ mbd, uint8_t* __restrict__ selT,
                               int H, int W) {
    const int prob = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;         // transposed flat index kx*H + ky
    if (j >= H * W) return;
    const int kx = j / H, ky = j - kx * H;
    const uint32_t bit = (bitsT[((size_t)prob * W + kx) * (H / 32) + (ky >> 5)] >> (ky & 31)) & 1u;
    const MbDesc d = mbd[prob];
    selT[(size_t)prob * H * W + j] = (bit && mb_member(d, (uint32_t)(ky * W + kx))) ? 1 : 0;
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
    void* mbd;      // [batch] MbDesc scratch of pnp_csmri_draw_minibatch
    int fused_min_batch;   // batches at least this large take the one-kernel gradient (env PNP_CSMRI_FUSED_MIN_BATCH)
};

extern "C" int pnp_csmri_plan_create(pnp_csmri_plan** out, int H, int W, int batch, int dtype) {
    PNP_CHECK_ARG(out != nullptr, "null plan pointer");
    PNP_CHECK_ARG(H == W && (H == 64 || H == 128 || H == 256), "supported sizes: H == W in {64, 128, 256}");
    PNP_CHECK_ARG(batch >= 1, "batch must be >= 1");
    PNP_CHECK_ARG(dtype == PNP_F32 || dtype == PNP_F64, "dtype must be PNP_F32 or PNP_F64");
    auto* p = new pnp_csmri_plan{H, W, batch, dtype, H == 256 ? 16 : H == 128 ? 12 : 8, nullptr, nullptr, nullptr, 192};
    if (const char* ev = getenv("PNP_CSMRI_FUSED_MIN_BATCH")) p->fused_min_batch = atoi(ev);
    const size_t esz = dtype == PNP_F32 ? 8 : 16;
    hipError_t e = hipMalloc(&p->work, (size_t)batch * (W / 2) * H * esz);
    if (e == hipSuccess) e = hipMalloc(&p->twtab, (size_t)H * esz);
    if (e == hipSuccess) e = hipMalloc(&p->mbd, (size_t)batch * sizeof(MbDesc));
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
        if (p->mbd) (void)hipFree(p->mbd);
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
    (void)hipFree(p->mbd);
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

extern "C" int pnp_csmri_pack_mask(pnp_csmri_plan* p, const uint8_t* selT, uint32_t* bitsT, void* stream) {
    PNP_CHECK_ARG(p && selT && bitsT, "null argument");
    const int nwords = p->W * (p->H / 32);
    k_pack_bits<<<dim3((nwords + 255) / 256, p->batch), 256, 0, (hipStream_t)stream>>>(selT, bitsT, nwords);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_csmri_draw_thresholds(pnp_csmri_plan* p, const uint32_t* bitsT, int mb, uint64_t seed, uint32_t step0,
                                         int nsteps, const uint32_t* step_dev, void* mbd, uint32_t* selbits, void* stream) {
    PNP_CHECK_ARG(p && bitsT && mbd, "null argument");
    PNP_CHECK_ARG(mb >= 1 && mb <= p->H * p->W && nsteps >= 1 && nsteps <= 65535, "need 1 <= mb <= H*W, 1 <= nsteps <= 65535");
    k_draw_thr<true><<<dim3(p->batch, nsteps), 256, 0, (hipStream_t)stream>>>(bitsT, p->H, p->W, mb, seed, step0, step_dev, (MbDesc*)mbd, selbits, draw_fast_path());
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_csmri_sel_from_thresholds(pnp_csmri_plan* p, const uint32_t* bitsT, const void* mbd, uint8_t* selT,
                                             void* stream) {
    PNP_CHECK_ARG(p && bitsT && mbd && selT, "null argument");
    k_sel_from_thr<<<dim3((p->H * p->W + 255) / 256, p->batch), 256, 0, (hipStream_t)stream>>>(bitsT, (const MbDesc*)mbd, selT,
                                                                                                p->H, p->W);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_csmri_draw_minibatch(pnp_csmri_plan* p, const uint32_t* bitsT, int mb, uint64_t seed, uint32_t step,
                                        const uint32_t* step_dev, uint8_t* selT, void* stream) {
    int rc = pnp_csmri_draw_thresholds(p, bitsT, mb, seed, step, 1, step_dev, p ? p->mbd : nullptr, nullptr, stream);
    if (rc != PNP_OK) return rc;
    return pnp_csmri_sel_from_thresholds(p, bitsT, p->mbd, selT, stream);
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

namespace pnp {
int csmri_fused_launch(int batch, const void* twtab, const void* a, const void* b, const uint32_t* bitsT, const void* yh,
                       double alpha, const void* alpha_vec, double beta, const void* c1, double gamma, const void* c2, void* out,
                       int mode, double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_out, void* sigma_out,
                       void* stream, void* w_out = nullptr, void* mu_out = nullptr);
int csmri_fused_outer_launch(int batch, const void* twtab, void* z, void* w, void* mu, const uint32_t* mask_bits, const void* yh,
                             const void* alpha_vec, const uint32_t* selbits, int T2, double lr, int mini_batch_size,
                             double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_log, int log_row0, int n_log,
                             void* sigma_out, void* stream);
}

namespace {
template <typename T, int RA, int LA>
int run_grad(pnp_csmri_plan* p, const void* a, const void* b, const uint8_t* selT, const uint32_t* bitsT,
             const void* yh, const void* YT, double alpha, const void* alpha_vec, double beta, const void* c1, double gamma, const void* c2,
             void* out, hipStream_t s) {
    constexpr int G = FftSmem<T, RA, LA>::G;
    const int H = p->H, W = p->W;
    cx<T>* work = (cx<T>*)p->work;
    const cx<T>* tw = (const cx<T>*)p->twtab;
    const T scale = (T)(alpha / ((double)H * (double)W));
    k_rows_fwd<T, RA, LA><<<dim3(H / (2 * G), p->batch), 256, 0, s>>>((const T*)a, (const T*)b, work, tw, H);
    PNP_CHECK_LAUNCH();
    const dim3 cg((W / 2) / G, p->batch);
    if (selT != nullptr)
        k_cols<T, RA, LA, SEL_U8><<<cg, 256, 0, s>>>(work, selT, nullptr, (const cx<T>*)yh, (const cx<T>*)YT, tw, W);
    else
        k_cols<T, RA, LA, SEL_BITS><<<cg, 256, 0, s>>>(work, nullptr, bitsT, (const cx<T>*)yh, (const cx<T>*)YT, tw, W);
    PNP_CHECK_LAUNCH();
    k_rows_inv<T, RA, LA><<<dim3(H / (2 * G), p->batch), 256, 0, s>>>(work, tw, H, scale, (const T*)alpha_vec, (T)beta,
                                                                 (const T*)c1, (T)gamma, (const T*)c2, (T*)out);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}
}  // namespace

extern "C" int pnp_csmri_grad_sel(pnp_csmri_plan* p, const void* a, const void* b, const uint8_t* selT, const uint32_t* bitsT,
                                  const void* yh, const void* YT, double alpha, const void* alpha_vec,
                                  double beta, const void* c1, double gamma, const void* c2, void* out, void* stream) {
    PNP_CHECK_ARG(p && a && out, "null argument");
    PNP_CHECK_ARG(!(yh != nullptr && YT != nullptr), "pass the packed data term (yh) or the raw data (YT), not both");
    PNP_CHECK_ARG((selT != nullptr) != (bitsT != nullptr), "pass either an explicit selector (selT) or mask bits (bitsT)");
    // large f32 256 x 256 batches with a bit-packed selector: the three passes in ONE kernel, the spectrum never in HBM
    // (csmri_fused.hip; one workgroup per image, so it needs about a workgroup per CU to pay off)
    if (p->dtype == PNP_F32 && p->H == 256 && bitsT != nullptr && YT == nullptr && p->batch >= p->fused_min_batch)
        return csmri_fused_launch(p->batch, p->twtab, a, b, bitsT, yh, alpha, alpha_vec, beta, c1, gamma, c2, out, 2, 1.0, 0.0,
                                  nullptr, nullptr, nullptr, stream);
    hipStream_t s = (hipStream_t)stream;
#define PNP_CS_ARGS p, a, b, selT, bitsT, yh, YT, alpha, alpha_vec, beta, c1, gamma, c2, out, s
    if (p->dtype == PNP_F32) {
        if (p->NL == 16) return run_grad<float, 16, 16>(PNP_CS_ARGS);
        if (p->NL == 12) return run_grad<float, 8, 16>(PNP_CS_ARGS);
        return run_grad<float, 8, 8>(PNP_CS_ARGS);
    }
    if (p->NL == 16) return run_grad<double, 16, 16>(PNP_CS_ARGS);
    if (p->NL == 12) return run_grad<double, 8, 16>(PNP_CS_ARGS);
    return run_grad<double, 8, 8>(PNP_CS_ARGS);
#undef PNP_CS_ARGS
}

extern "C" int pnp_csmri_grad(pnp_csmri_plan* p, const void* a, const void* b, const uint8_t* selT, const void* yh,
                              double alpha, double beta, const void* c1, double gamma, const void* c2, void* out,
                              void* stream) {
    PNP_CHECK_ARG(selT != nullptr, "null selector");
    return pnp_csmri_grad_sel(p, a, b, selT, nullptr, yh, nullptr, alpha, nullptr, beta, c1, gamma, c2, out, stream);
}

// ---- whole inner iteration in one kernel (csmri_fused.hip)
extern "C" int pnp_csmri_svrg_step(pnp_csmri_plan* p, const void* a, const void* b, const uint32_t* bitsT, double alpha,
                                   const void* alpha_vec, double beta, const void* c1, double gamma, const void* c2, void* out,
                                   int denoise, double sigma_modifier, double fallback_sigma, const void* xrec,
                                   double* sse_out, void* sigma_out, void* stream) {
    PNP_CHECK_ARG(p && a && bitsT && out, "null argument");
    PNP_CHECK_ARG(p->dtype == PNP_F32 && p->H == 256 && p->W == 256, "the one-kernel iteration exists for f32 plans of 256 x 256");
    PNP_CHECK_ARG(!(sse_out && !xrec), "sse_out needs xrec");
    return csmri_fused_launch(p->batch, p->twtab, a, b, bitsT, nullptr, alpha, alpha_vec, beta, c1, gamma, c2, out, denoise ? 0 : 1,
                              sigma_modifier, fallback_sigma, xrec, sse_out, sigma_out, stream);
}

// ---- the same with the outer-loop refresh folded in (first inner iteration of an outer iteration)
extern "C" int pnp_csmri_svrg_outer_step(pnp_csmri_plan* p, const void* z, const uint32_t* mask_bitsT, const void* yh,
                                         const void* alpha_vec, double lr, void* w_out, void* mu_out, void* out, int denoise,
                                         double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_out,
                                         void* sigma_out, void* stream) {
    PNP_CHECK_ARG(p && z && mask_bitsT && yh && w_out && mu_out && out, "null argument");
    PNP_CHECK_ARG(p->dtype == PNP_F32 && p->H == 256 && p->W == 256, "the one-kernel iteration exists for f32 plans of 256 x 256");
    PNP_CHECK_ARG(!(sse_out && !xrec), "sse_out needs xrec");
    PNP_CHECK_ARG(w_out != z && mu_out != z && w_out != mu_out && w_out != out && mu_out != out, "w_out and mu_out must be buffers of their own");
    return csmri_fused_launch(p->batch, p->twtab, z, nullptr, mask_bitsT, yh, 1.0, alpha_vec, 1.0, z, -lr, nullptr, out,
                              denoise ? 0 : 1, sigma_modifier, fallback_sigma, xrec, sse_out, sigma_out, stream, w_out, mu_out);
}

// ---- a whole outer iteration (refresh + T2 inner iterations, TV prox) in one launch
extern "C" int pnp_csmri_svrg_outer_iteration(pnp_csmri_plan* p, void* z, void* w, void* mu, const uint32_t* mask_bitsT, const void* yh,
                                              const void* alpha_vec, const uint32_t* selbits, int T2, double lr, int mini_batch_size,
                                              double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_log,
                                              int log_row0, int n_log, void* sigma_out, void* stream) {
    PNP_CHECK_ARG(p && z && w && mu && mask_bitsT && yh && alpha_vec && xrec && sse_log && sigma_out, "null argument");
    PNP_CHECK_ARG(p->dtype == PNP_F32 && p->H == 256 && p->W == 256, "the one-kernel iteration exists for f32 plans of 256 x 256");
    PNP_CHECK_ARG(T2 >= 1 && (T2 == 1 || selbits != nullptr) && mini_batch_size >= 1 && n_log >= 1 && log_row0 >= 0, "bad T2 / selbits / mini_batch_size / log");
    PNP_CHECK_ARG(z != w && z != mu && w != mu, "z, w and mu must be buffers of their own");
    return csmri_fused_outer_launch(p->batch, p->twtab, z, w, mu, mask_bitsT, yh, alpha_vec, selbits, T2, lr, mini_batch_size,
                                    sigma_modifier, fallback_sigma, xrec, sse_log, log_row0, n_log, sigma_out, stream);
}
