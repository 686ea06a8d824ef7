// reduce.h -- small per-problem reductions shared by prox.hip and dncnn.hip.
#pragma once
#include "common.h"

namespace pnp {

// per-problem min / max (RealSN_DnCNN.py:20-22): out[2b] = min, out[2b+1] = max
template <typename T>
__global__ __launch_bounds__(256) void k_minmax(const T* __restrict__ z, int n, T* __restrict__ out) {
    __shared__ T rmin[4], rmax[4];
    const size_t base = (size_t)blockIdx.x * n;
    T lo = z[base], hi = lo;
    for (int i = threadIdx.x; i < n; i += 256) {
        const T v = z[base + i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) { rmin[threadIdx.x >> 6] = lo; rmax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { lo = rmin[i] < lo ? rmin[i] : lo; hi = rmax[i] > hi ? rmax[i] : hi; }
        out[2 * blockIdx.x] = lo;
        out[2 * blockIdx.x + 1] = hi;
    }
}


}  // namespace pnp
