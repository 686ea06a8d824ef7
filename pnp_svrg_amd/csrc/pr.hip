// pr.hip -- phase-retrieval (amplitude-flow) gradients, reference problems/PR.py:75-87:
//     t = A_sel w ;  u = ((|t| - y_sel) / |t|) * t ;  g = A_sel^T u      [/ M for grad_full]
// A is a dense M x N Gaussian matrix: both products stream A once -> HBM-bound GEMV pair.
//   k_pr_rows : one wavefront per selected row, 16-byte loads along the row (float4 / double2; scalar when N is
//               not a multiple of the vector length), shuffle reduction, the amplitude weight fused into the epilogue
//   k_pr_cols : column-parallel A^T u over a chunk of rows per block (coalesced across columns),
//               deterministic two-stage sum (partials + reduce; no float atomics)
#include "common.h"

namespace pnp {

template <typename T> struct Vec16;                     // 16-byte vector of T
template <> struct Vec16<float> { using type = float4; static constexpr int n = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int n = 2; };

// SPECTRAL: the row weight is y[m] (u = y o (A w)): one application of A^T diag(y) A, the matrix whose leading
// eigenvector is the spectral initialisation (PR.py:50-63) -- never formed, the power iteration streams A twice.
// One wavefront per selected row; 16-byte loads along the row when N allows it (N % (16 / sizeof(T)) == 0 keeps every
// row 16-byte aligned), scalar loads otherwise.  blockIdx.y = problem of the batch (A, w, y, rows, u strided).
template <typename T, bool SPECTRAL = false>
__global__ __launch_bounds__(256) void k_pr_rows(const T* __restrict__ A, const T* __restrict__ w, const T* __restrict__ y,
                                                 const int32_t* __restrict__ rows, int nsel, int M, int N, T* __restrict__ u) {
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::n;
    const int prob = blockIdx.y;
    const int wv = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wv >= nsel) return;
    const int m = rows ? rows[(size_t)prob * nsel + wv] : wv;
    const T* a = A + ((size_t)prob * M + m) * N;
    const T* wp = w + (size_t)prob * N;
    T acc = 0;
    if (N % VN == 0) {
        const V* a4 = reinterpret_cast<const V*>(a);
        const V* w4 = reinterpret_cast<const V*>(wp);
        for (int n = lane; n < N / VN; n += 64) {
            const V av = a4[n], wq = w4[n];
            if constexpr (VN == 4) acc += (av.x * wq.x + av.y * wq.y) + (av.z * wq.z + av.w * wq.w);
            else acc += av.x * wq.x + av.y * wq.y;
        }
    } else {
        for (int n = lane; n < N; n += 64) acc += a[n] * wp[n];
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        const T mag = acc < 0 ? -acc : acc;
        const T ym = y[(size_t)prob * M + m];
        u[(size_t)prob * M + wv] = SPECTRAL ? ym * acc : ((mag - ym) / mag) * acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_pr_cols(const T* __restrict__ A, const T* __restrict__ u,
                                                 const int32_t* __restrict__ rows, int nsel, int M, int N, int rows_per_chunk,
                                                 int nchunks, T* __restrict__ part) {
    const int prob = blockIdx.z;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * rows_per_chunk, j1 = j0 + rows_per_chunk < nsel ? j0 + rows_per_chunk : nsel;
    if (n >= N) return;
    const T* Ap = A + (size_t)prob * M * N;
    const T* up = u + (size_t)prob * M;
    const int32_t* rp = rows ? rows + (size_t)prob * nsel : nullptr;
    T acc = 0;
    for (int j = j0; j < j1; ++j) {
        const int m = rp ? rp[j] : j;
        acc += Ap[(size_t)m * N + n] * up[j];
    }
    part[((size_t)prob * nchunks + blockIdx.y) * N + n] = acc;
}

template <typename T>
__global__ void k_pr_reduce(const T* __restrict__ part, int nchunks, int N, T scale, T* __restrict__ out) {
    const int prob = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    T acc = 0;
    for (int c = 0; c < nchunks; ++c) acc += part[((size_t)prob * nchunks + c) * N + n];
    out[(size_t)prob * N + n] = scale * acc;
}

template <typename T>
int run_pr(const T* A, const T* w, const T* y, const int32_t* rows, int nsel, int M, int N, int batch, double scale, T* ws,
           T* out, hipStream_t s, bool spectral = false) {
    const int nchunks = 64;
    T* u = ws;                               // [batch][M] (first nsel of each used)
    T* part = ws + (size_t)batch * M;        // [batch][nchunks][N]
    const dim3 rg((nsel * 64 + 255) / 256, batch);
    if (spectral) k_pr_rows<T, true><<<rg, 256, 0, s>>>(A, w, y, rows, nsel, M, N, u);
    else k_pr_rows<T><<<rg, 256, 0, s>>>(A, w, y, rows, nsel, M, N, u);
    PNP_CHECK_LAUNCH();
    const int rpc = (nsel + nchunks - 1) / nchunks;
    k_pr_cols<T><<<dim3((N + 255) / 256, nchunks, batch), 256, 0, s>>>(A, u, rows, nsel, M, N, rpc, nchunks, part);
    PNP_CHECK_LAUNCH();
    k_pr_reduce<T><<<dim3((N + 255) / 256, batch), 256, 0, s>>>(part, nchunks, N, (T)scale, out);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

}  // namespace pnp

using namespace pnp;

extern "C" size_t pnp_pr_workspace_elems(int M, int N) { return (size_t)M + (size_t)64 * N; }

// Batched form: A [batch][M][N], w [batch][N], y [batch][M], rows [batch][nsel] (or NULL = all M rows), out [batch][N];
// workspace: batch * pnp_pr_workspace_elems(M, N) elements of `dtype`.
extern "C" int pnp_pr_grad_batch(const void* A, const void* w, const void* y, const int32_t* rows, int nsel, int M, int N,
                                 int batch, int dtype, double scale, void* workspace, void* out, void* stream) {
    PNP_CHECK_ARG(A && w && y && workspace && out, "null argument");
    PNP_CHECK_ARG(M >= 1 && N >= 1 && batch >= 1 && nsel >= 0 && nsel <= M, "bad sizes");
    if (rows == nullptr) nsel = M;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PNP_F32)
        return run_pr<float>((const float*)A, (const float*)w, (const float*)y, rows, nsel, M, N, batch, scale, (float*)workspace, (float*)out, s);
    if (dtype == PNP_F64)
        return run_pr<double>((const double*)A, (const double*)w, (const double*)y, rows, nsel, M, N, batch, scale, (double*)workspace, (double*)out, s);
    PNP_CHECK_ARG(false, "bad dtype");
}

// out = scale * A_sel^T ( ((|A_sel w| - y_sel) / |A_sel w|) o A_sel w );  rows: int32 [nsel] selected row ids
// (NULL = all M rows).  workspace: pnp_pr_workspace_elems(M, N) elements of `dtype`.
extern "C" int pnp_pr_grad(const void* A, const void* w, const void* y, const int32_t* rows, int nsel, int M, int N,
                           int dtype, double scale, void* workspace, void* out, void* stream) {
    return pnp_pr_grad_batch(A, w, y, rows, nsel, M, N, 1, dtype, scale, workspace, out, stream);
}

// out = scale * A^T ( y o (A v) )  -- one power-iteration step of the spectral initialisation (PR.py:53,59 without
// the N x N matrix D = A^T diag(y) A / M).  Same workspace as pnp_pr_grad.
extern "C" int pnp_pr_spectral_apply(const void* A, const void* v, const void* y, int M, int N, int dtype, double scale,
                                     void* workspace, void* out, void* stream) {
    PNP_CHECK_ARG(A && v && y && workspace && out, "null argument");
    PNP_CHECK_ARG(M >= 1 && N >= 1, "bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PNP_F32)
        return run_pr<float>((const float*)A, (const float*)v, (const float*)y, nullptr, M, M, N, 1, scale, (float*)workspace, (float*)out, s, true);
    if (dtype == PNP_F64)
        return run_pr<double>((const double*)A, (const double*)v, (const double*)y, nullptr, M, M, N, 1, scale, (double*)workspace, (double*)out, s, true);
    PNP_CHECK_ARG(false, "bad dtype");
}
