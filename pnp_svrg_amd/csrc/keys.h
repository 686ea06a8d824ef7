// keys.h -- order-preserving bit patterns of non-negative IEEE values (for the radix-select median) and the db2
// high-pass taps of the noise estimate; shared by prox.hip (column-owning lanes) and csmri.hip (line-owning groups).
#pragma once
#include "common.h"

namespace pnp {

template <typename T> struct KeyOf;
template <> struct KeyOf<float> { using type = uint32_t; static constexpr int BITS = 31; };
template <> struct KeyOf<double> { using type = uint64_t; static constexpr int BITS = 63; };

__device__ __forceinline__ uint32_t to_key(float v) { return __float_as_uint(v); }
__device__ __forceinline__ uint64_t to_key(double v) { return (uint64_t)__double_as_longlong(v); }
__device__ __forceinline__ float from_key(uint32_t k) { return __uint_as_float(k); }
__device__ __forceinline__ double from_key(uint64_t k) { return __longlong_as_double((long long)k); }

template <typename T> struct Db2 {
    static constexpr T h0 = (T)-0.48296291314453416, h1 = (T)0.8365163037378079,
                       h2 = (T)-0.2241438680420134, h3 = (T)-0.12940952255126037;
};

}  // namespace pnp
