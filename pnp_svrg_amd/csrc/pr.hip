// pr.hip -- phase-retrieval (amplitude-flow) gradients, reference problems/PR.py:75-87:
//     t = A_sel w ;  u = ((|t| - y_sel) / |t|) * t ;  g = A_sel^T u      [/ M for grad_full]
// A is a dense M x N Gaussian matrix: both products stream A once -> HBM-bound GEMV pair.
//   k_pr_rows : one wavefront per selected row, 16-byte loads along the row, shuffle reduction,
//               the amplitude weight fused into the epilogue
//   k_pr_cols : column-parallel A^T u over a chunk of rows per block (coalesced across columns),
//               deterministic two-stage sum (partials + reduce; no float atomics)
#include "common.h"

namespace pnp {

// SPECTRAL: the row weight is y[m] (u = y o (A w)): one application of A^T diag(y) A, the matrix whose leading
// eigenvector is the spectral initialisation (PR.py:50-63) -- never formed, the power iteration streams A twice.
template <typename T, bool SPECTRAL = false>
__global__ __launch_bounds__(256) void k_pr_rows(const T* __restrict__ A, const T* __restrict__ w, const T* __restrict__ y,
                                                 const int32_t* __restrict__ rows, int nsel, int N, T* __restrict__ u) {
    const int wv = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wv >= nsel) return;
    const int m = rows ? rows[wv] : wv;
    const T* a = A + (size_t)m * N;
    T acc = 0;
    for (int n = lane; n < N; n += 64) acc += a[n] * w[n];
    acc = wave_sum(acc);
    if (lane == 0) {
        const T mag = acc < 0 ? -acc : acc;
        u[wv] = SPECTRAL ? y[m] * acc : ((mag - y[m]) / mag) * acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_pr_cols(const T* __restrict__ A, const T* __restrict__ u,
                                                 const int32_t* __restrict__ rows, int nsel, int N, int rows_per_chunk,
                                                 T* __restrict__ part) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * rows_per_chunk, j1 = j0 + rows_per_chunk < nsel ? j0 + rows_per_chunk : nsel;
    if (n >= N) return;
    T acc = 0;
    for (int j = j0; j < j1; ++j) {
        const int m = rows ? rows[j] : j;
        acc += A[(size_t)m * N + n] * u[j];
    }
    part[(size_t)blockIdx.y * N + n] = acc;
}

template <typename T>
__global__ void k_pr_reduce(const T* __restrict__ part, int nchunks, int N, T scale, T* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    T acc = 0;
    for (int c = 0; c < nchunks; ++c) acc += part[(size_t)c * N + n];
    out[n] = scale * acc;
}

template <typename T>
int run_pr(const T* A, const T* w, const T* y, const int32_t* rows, int nsel, int M, int N, double scale, T* ws, T* out,
           hipStream_t s, bool spectral = false) {
    const int nchunks = 64;
    T* u = ws;                       // [nsel]
    T* part = ws + M;                // [nchunks][N]
    if (spectral) k_pr_rows<T, true><<<(nsel * 64 + 255) / 256, 256, 0, s>>>(A, w, y, rows, nsel, N, u);
    else k_pr_rows<T><<<(nsel * 64 + 255) / 256, 256, 0, s>>>(A, w, y, rows, nsel, N, u);
    PNP_CHECK_LAUNCH();
    const int rpc = (nsel + nchunks - 1) / nchunks;
    k_pr_cols<T><<<dim3((N + 255) / 256, nchunks), 256, 0, s>>>(A, u, rows, nsel, N, rpc, part);
    PNP_CHECK_LAUNCH();
    k_pr_reduce<T><<<(N + 255) / 256, 256, 0, s>>>(part, nchunks, N, (T)scale, out);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

}  // namespace pnp

using namespace pnp;

extern "C" size_t pnp_pr_workspace_elems(int M, int N) { return (size_t)M + (size_t)64 * N; }

// out = scale * A_sel^T ( ((|A_sel w| - y_sel) / |A_sel w|) o A_sel w );  rows: int32 [nsel] selected row ids
// (NULL = all M rows).  workspace: pnp_pr_workspace_elems(M, N) elements of `dtype`.
extern "C" int pnp_pr_grad(const void* A, const void* w, const void* y, const int32_t* rows, int nsel, int M, int N,
                           int dtype, double scale, void* workspace, void* out, void* stream) {
    PNP_CHECK_ARG(A && w && y && workspace && out, "null argument");
    PNP_CHECK_ARG(M >= 1 && N >= 1 && nsel >= 0 && nsel <= M, "bad sizes");
    if (rows == nullptr) nsel = M;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PNP_F32)
        return run_pr<float>((const float*)A, (const float*)w, (const float*)y, rows, nsel, M, N, scale, (float*)workspace, (float*)out, s);
    if (dtype == PNP_F64)
        return run_pr<double>((const double*)A, (const double*)w, (const double*)y, rows, nsel, M, N, scale, (double*)workspace, (double*)out, s);
    PNP_CHECK_ARG(false, "bad dtype");
}

// out = scale * A^T ( y o (A v) )  -- one power-iteration step of the spectral initialisation (PR.py:53,59 without
// the N x N matrix D = A^T diag(y) A / M).  Same workspace as pnp_pr_grad.
extern "C" int pnp_pr_spectral_apply(const void* A, const void* v, const void* y, int M, int N, int dtype, double scale,
                                     void* workspace, void* out, void* stream) {
    PNP_CHECK_ARG(A && v && y && workspace && out, "null argument");
    PNP_CHECK_ARG(M >= 1 && N >= 1, "bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PNP_F32)
        return run_pr<float>((const float*)A, (const float*)v, (const float*)y, nullptr, M, M, N, scale, (float*)workspace, (float*)out, s, true);
    if (dtype == PNP_F64)
        return run_pr<double>((const double*)A, (const double*)v, (const double*)y, nullptr, M, M, N, scale, (double*)workspace, (double*)out, s, true);
    PNP_CHECK_ARG(false, "bad dtype");
}
