// fft.h -- N = NL*NL point complex FFT spread over NL lanes x NL registers (gfx950).
//
// Layout ("lane + NL*reg"): lane l of a NL-lane group holds element l + NL*r in v[r].
// The transform is the classic four-step split N = NL x NL:
//   1. in-register DFT-NL over r            (radix-2 recursion, compile-time twiddles)
//   2. twiddle by W_N^(lane*r)              (table held in registers)
//   3. NL x NL transpose between lanes      (one LDS round trip, padded rows)
//   4. in-register DFT-NL again
// Input and output are both natural order in the same layout, so a forward transform, a
// pointwise k-space operation and the inverse transform chain with no reshuffle: that is
// what lets the CSMRI column pass (FFT -> selector -> inverse FFT) run as one kernel.
#pragma once
#include "common.h"

namespace pnp {

__device__ constexpr double kCos16[8] = {1.0, 0.92387953251128674, 0.70710678118654752, 0.38268343236508977,
                                         0.0, -0.38268343236508977, -0.70710678118654752, -0.92387953251128674};
__device__ constexpr double kSin16[8] = {0.0, 0.38268343236508977, 0.70710678118654752, 0.92387953251128674,
                                         1.0, 0.92387953251128674, 0.70710678118654752, 0.38268343236508977};

// In-register DFT of R points (R in {1,2,4,8,16}); INV selects exp(+i..) (unnormalised).
template <typename T, int R, bool INV>
__device__ __forceinline__ void dft_reg(cx<T> (&v)[R]) {
    if constexpr (R == 2) {
        cx<T> a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else if constexpr (R > 2) {
        cx<T> e[R / 2], o[R / 2];
#pragma unroll
        for (int i = 0; i < R / 2; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
        dft_reg<T, R / 2, INV>(e);
        dft_reg<T, R / 2, INV>(o);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            cx<T> t;
            if (k == 0) {
                t = o[k];
            } else if (4 * k == R) {                       // -i (forward) / +i (inverse)
                t = INV ? cx<T>{-o[k].y, o[k].x} : cx<T>{o[k].y, -o[k].x};
            } else {
                const T c = (T)kCos16[k * (16 / R)];
                const T s = (T)kSin16[k * (16 / R)];
                t = cmul(o[k], cx<T>{c, INV ? s : -s});
            }
            v[k] = cadd(e[k], t);
            v[k + R / 2] = csub(e[k], t);
        }
    }
}

// Load the NL inter-stage twiddles of this lane: tw[r] = W_N^(lane*r), table[j] = exp(-2*pi*i*j/N).
template <typename T, int NL>
__device__ __forceinline__ void load_twiddles(cx<T> (&tw)[NL], const cx<T>* __restrict__ table, int lane) {
#pragma unroll
    for (int r = 0; r < NL; ++r) tw[r] = table[(lane * r) & (NL * NL - 1)];
}

// Whole-block collective (contains __syncthreads): every thread of the block must call it.
// scr: this group's private LDS scratch, NL*(NL+1) complex.
template <typename T, int NL, bool INV>
__device__ __forceinline__ void fft_group(cx<T> (&v)[NL], const cx<T> (&tw)[NL], cx<T>* scr, int lane) {
    dft_reg<T, NL, INV>(v);
#pragma unroll
    for (int r = 0; r < NL; ++r) v[r] = cmul(v[r], INV ? cconj(tw[r]) : tw[r]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NL; ++r) scr[r * (NL + 1) + lane] = v[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NL; ++r) v[r] = scr[lane * (NL + 1) + r];
    dft_reg<T, NL, INV>(v);
}

// General (rectangular) form: N = RA * LA points; input spread over LA lanes x RA registers (element
// lane + LA*r), output over RA lanes x LA registers (element lane + RA*r), both natural order.  The lane
// group is max(RA, LA) wide; after the call only lanes < RA hold results.  A forward <RA, LA> followed by an
// inverse <LA, RA> returns to the input layout, so FFT -> pointwise -> inverse FFT still chains in registers.
// RA == LA is the square case above (N = 64, 256); <8,16>/<16,8> give N = 128.
// Whole-block collective (contains __syncthreads).  scr: group-private LDS, MX*(MX+1) complex, MX = max(RA, LA).
// tw[r] = W_N^(lane*r) for r < MX (load_twiddles_gen).
template <typename T, int RA, int LA, bool INV>
__device__ __forceinline__ void fft_gen(cx<T> (&v)[(RA > LA ? RA : LA)], const cx<T> (&tw)[(RA > LA ? RA : LA)], cx<T>* scr, int lane) {
    {
        cx<T> a[RA];
#pragma unroll
        for (int r = 0; r < RA; ++r) a[r] = v[r];
        dft_reg<T, RA, INV>(a);
#pragma unroll
        for (int r = 0; r < RA; ++r) v[r] = cmul(a[r], INV ? cconj(tw[r]) : tw[r]);
    }
    __syncthreads();
    if (lane < LA) {
#pragma unroll
        for (int r = 0; r < RA; ++r) scr[r * (LA + 1) + lane] = v[r];
    }
    __syncthreads();
    {
        cx<T> b[LA];
        const int ln = lane < RA ? lane : 0;                 // idle lanes read a valid slot (results unused)
#pragma unroll
        for (int r = 0; r < LA; ++r) b[r] = scr[ln * (LA + 1) + r];
        dft_reg<T, LA, INV>(b);
#pragma unroll
        for (int r = 0; r < LA; ++r) v[r] = b[r];
    }
}

template <typename T, int MX>
__device__ __forceinline__ void load_twiddles_gen(cx<T> (&tw)[MX], const cx<T>* __restrict__ table, int lane, int N) {
#pragma unroll
    for (int r = 0; r < MX; ++r) tw[r] = table[(lane * r) & (N - 1)];
}

}  // namespace pnp
