// vr.hip -- problem-independent pieces of the variance-reduced loops on gfx950:
//   * minibatch draws over a length-M measurement vector (problems/problem.py:110-117: `np.random.choice(M, size,
//     replace=False)` -> 0/1 indicator), as thresholds (draw.h) and, when a caller needs them, as an indicator
//     (Deblur) or an ascending row list (PhaseRetrieval);
//   * the SAGA gradient-table update (algorithms/pnp_saga.py:43-57) as ONE pass over the vectors.
#include "draw.h"

namespace pnp {

__global__ void k_indicator_from_thr(const MbDesc* __restrict__ mbd, uint8_t* __restrict__ sel, int M) {
    const int prob = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    sel[(size_t)prob * M + i] = mb_member(mbd[prob], (uint32_t)i) ? 1 : 0;
}

__global__ void k_indicator_scatter(const int32_t* __restrict__ idx, int n, uint8_t* __restrict__ sel, int M) {
    const int prob = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int i = idx[(size_t)prob * n + j];
    if (i >= 0 && i < M) sel[(size_t)prob * M + i] = 1;
}

// ascending list of the minibatch members of one problem (deterministic: block-wide prefix sum, no atomics)
__global__ __launch_bounds__(256) void k_rows_from_thr(const MbDesc* __restrict__ mbd, int32_t* __restrict__ rows, int M, int mb) {
    __shared__ int wtot[4];
    const int prob = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const MbDesc d = mbd[prob];
    const int per = (M + 255) / 256, i0 = tid * per, i1 = i0 + per < M ? i0 + per : M;
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += mb_member(d, (uint32_t)i) ? 1 : 0;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    int pos = incl - cnt;
    for (int q = 0; q < wv; ++q) pos += wtot[q];
    int32_t* out = rows + (size_t)prob * mb;
    for (int i = i0; i < i1; ++i)
        if (mb_member(d, (uint32_t)i)) { if (pos < mb) out[pos] = i; ++pos; }
}

// SAGA step (pnp_saga.py:45-57 with the table sum kept incrementally, which is algebraically sum(table)):
//   old = slot; sum += g - old; v = g - prev + sum / hist; z -= lr * v; slot = g
// `prev` is the slot written by the previous step (it may be this very slot: read before write, same thread).
template <typename T>
__global__ void k_saga_update(T* z, const T* __restrict__ g, T* slot, const T* prev, T* sum, T lr, T inv_hist, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T gi = g[i], old = slot[i], pv = prev[i];
        const T s = sum[i] + gi - old;
        z[i] -= lr * ((gi - pv) + s * inv_hist);
        slot[i] = gi;
        sum[i] = s;
    }
}

}  // namespace pnp

using namespace pnp;

extern "C" int pnp_draw_thresholds(int M, int batch, int mb, uint64_t seed, uint32_t step0, int nsteps,
                                   const uint32_t* step_dev, void* mbd, void* stream) {
    PNP_CHECK_ARG(mbd != nullptr, "null argument");
    PNP_CHECK_ARG(M >= 1 && batch >= 1 && mb >= 1 && nsteps >= 1 && nsteps <= 65535, "need M, batch, mb >= 1 and 1 <= nsteps <= 65535");
    k_draw_thr<false><<<dim3(batch, nsteps), 256, 0, (hipStream_t)stream>>>(nullptr, 1, M, mb, seed, step0, step_dev, (MbDesc*)mbd, nullptr, draw_fast_path());
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_indicator_from_thresholds(int M, int batch, const void* mbd, uint8_t* sel, void* stream) {
    PNP_CHECK_ARG(mbd && sel && M >= 1 && batch >= 1, "bad argument");
    k_indicator_from_thr<<<dim3((M + 255) / 256, batch), 256, 0, (hipStream_t)stream>>>((const MbDesc*)mbd, sel, M);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_indicator_from_indices(const int32_t* idx, int n, int M, int batch, uint8_t* sel, void* stream) {
    PNP_CHECK_ARG(idx && sel && n >= 0 && M >= 1 && batch >= 1, "bad argument");
    PNP_CHECK_HIP(hipMemsetAsync(sel, 0, (size_t)batch * M, (hipStream_t)stream));
    if (n > 0) {
        k_indicator_scatter<<<dim3((n + 255) / 256, batch), 256, 0, (hipStream_t)stream>>>(idx, n, sel, M);
        PNP_CHECK_LAUNCH();
    }
    return PNP_OK;
}

extern "C" int pnp_rows_from_thresholds(int M, int batch, int mb, const void* mbd, int32_t* rows, void* stream) {
    PNP_CHECK_ARG(mbd && rows && M >= 1 && batch >= 1 && mb >= 1 && mb <= M, "bad argument");
    k_rows_from_thr<<<batch, 256, 0, (hipStream_t)stream>>>((const MbDesc*)mbd, rows, M, mb);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_saga_table_update(void* z, const void* g, void* slot, const void* prev, void* sum, double lr,
                                     double inv_hist, size_t n, int dtype, void* stream) {
    PNP_CHECK_ARG(z && g && slot && prev && sum, "null argument");
    if (n == 0) return PNP_OK;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == PNP_F32)
        k_saga_update<float><<<grid, 256, 0, (hipStream_t)stream>>>((float*)z, (const float*)g, (float*)slot, (const float*)prev,
                                                                      (float*)sum, (float)lr, (float)inv_hist, n);
    else if (dtype == PNP_F64)
        k_saga_update<double><<<grid, 256, 0, (hipStream_t)stream>>>((double*)z, (const double*)g, (double*)slot, (const double*)prev,
                                                                       (double*)sum, lr, inv_hist, n);
    else PNP_CHECK_ARG(false, "bad dtype");
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}
