// common.h -- shared helpers for the gfx950 kernels (error plumbing, complex type, reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <string>
#include "../../include/pnp_hip.h"

namespace pnp {

void set_error(const std::string& msg);

#define PNP_CHECK_ARG(cond, msg)                                          \
    do { if (!(cond)) { pnp::set_error(std::string(__func__) + ": " + (msg)); return PNP_ERR_ARG; } } while (0)

#define PNP_CHECK_HIP(expr)                                               \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) {                  \
        pnp::set_error(std::string(__func__) + ": " #expr ": " + hipGetErrorString(e_)); \
        return PNP_ERR_HIP; } } while (0)

#define PNP_CHECK_LAUNCH() PNP_CHECK_HIP(hipGetLastError())

template <typename T> struct cx { T x, y; };

template <typename T> __device__ __forceinline__ cx<T> cadd(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> __device__ __forceinline__ cx<T> csub(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
// Complex product with the fused multiply-adds SPELLED OUT: under the compiler's default contraction, which of the two
// products of `a.x * b.x - a.y * b.y` gets fused is decided per call site, so two instantiations of the same transform could
// differ in the last bit (seen between the gradient-only and the whole-iteration forms of k_svrg_iter); written this way
// every kernel rounds the same way.
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
    return {fma_(a.x, b.x, -(a.y * b.y)), fma_(a.x, b.y, a.y * b.x)};
}
template <typename T> __device__ __forceinline__ cx<T> cconj(cx<T> a) { return {a.x, -a.y}; }

// 64-lane wavefront helpers (gfx950: warpSize == 64)
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T> __device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o, 64); v = u < v ? u : v; }
    return v;
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
    return v;
}

}  // namespace pnp
