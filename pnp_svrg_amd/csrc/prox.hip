// prox.hip -- noise estimate + "TV" (per-column Haar BayesShrink) prox + PSNR error sum, fused.
//
// Replaces, per inner iteration of every reference loop (e.g. algorithms/pnp_svrg.py:70-80):
//   estimate_sigma(z0, multichannel=True, average_sigmas=True)   [skimage; SURVEY F3, a17]
//   TVDenoiser.denoise(z0, sigma_est)                            [denoisers/TV.py:21-26; SURVEY F2, a18]
//   Problem.PSNR(z0)                                             [problems/problem.py:33-35]
//
// One workgroup owns one image.  A wavefront owns 16 adjacent columns x 4 row-chunks
// (lane = col16 + 16*chunk); every lane keeps its H/4 rows of one column in VGPRs, so the
// whole column pipeline -- db2 detail coefficients, the MAD median (bitwise radix select on the
// IEEE bit patterns), the multi-level Haar analysis, per-level BayesShrink thresholds, synthesis
// and the squared error against the ground truth -- runs out of registers with only
// 16-lane-stride wave shuffles between the four lanes of a column.  The image is read once and
// written once.
#include "common.h"
#include <mutex>
#include <vector>
#include "reduce.h"
#include "prox_tv.h"
#include <cstdlib>

namespace pnp {

// DENOISE: estimate + prox, else estimate only.  zin / zout may alias (in-place prox): no __restrict__ on them.
template <typename T, int H, bool DENOISE>
__global__ __launch_bounds__(1024) void k_prox_tv(const T* zin, T* zout, int W,
                                                  const T* __restrict__ sigma_in, T sigma_modifier, T fallback_sigma,
                                                  const T* __restrict__ xrec, double* __restrict__ sse_out,
                                                  T* __restrict__ sigma_out) {
    constexpr int RPC = H / 4;
    __shared__ double red[16];
    __shared__ T sig_sh;
    const int prob = blockIdx.x;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 15, q = lane >> 4;
    const int nwaves = blockDim.x >> 6;
    const int col = wv * 16 + cl;
    const size_t base = (size_t)prob * H * W + (size_t)(q * RPC) * W + col;

    T x[RPC];
#pragma unroll
    for (int i = 0; i < RPC; ++i) x[i] = zin[base + (size_t)i * W];
    prox_tv_regs<T, H, DENOISE>(x, prob, W, base, wv, lane, q, nwaves, sigma_in, sigma_modifier, fallback_sigma, xrec,
                                DENOISE ? zout : nullptr, sse_out, sigma_out, red, &sig_sh);
}

// ------------------------------------------------------------------------------- small batches: one wave per 16 columns
// k_prox_tv gives an image to ONE workgroup, i.e. one CU: at B = 1 (the drop-in loops) the whole chip waits for it.  For
// small batches the same pipeline runs as two launches of W/16 single-wave workgroups per image -- (1) per-column noise
// estimates, (2) mean, shrink, store, error -- with exactly the summation trees of k_prox_tv (a wave here is a wave there;
// the per-wave partial sums are combined in wave order by the last workgroup to finish), so results are bit-identical.
template <typename T, int H>
__global__ __launch_bounds__(64) void k_sigma_cols(const T* __restrict__ zin, int W, T* __restrict__ sig_cols) {
    constexpr int RPC = H / 4;
    const int prob = blockIdx.y, wv = blockIdx.x, lane = threadIdx.x, cl = lane & 15, q = lane >> 4;
    const size_t base = (size_t)prob * H * W + (size_t)(q * RPC) * W + wv * 16 + cl;
    T x[RPC];
#pragma unroll
    for (int i = 0; i < RPC; ++i) x[i] = zin[base + (size_t)i * W];
    const T sc = column_sigma<T, RPC>(x, q);
    if (q == 0) sig_cols[(size_t)prob * W + wv * 16 + cl] = sc;
}

template <typename T, int H, bool DENOISE>
__global__ __launch_bounds__(64) void k_shrink_cols(const T* zin, T* zout, int W, const T* __restrict__ sig_cols,
                                                    const T* __restrict__ sigma_in, T sigma_modifier, T fallback_sigma,
                                                    const T* __restrict__ xrec, double* __restrict__ sse_out,
                                                    T* __restrict__ sigma_out, double* __restrict__ partial,
                                                    unsigned* __restrict__ counter) {
    constexpr int RPC = H / 4;
    const int prob = blockIdx.y, wv = blockIdx.x, lane = threadIdx.x, cl = lane & 15, q = lane >> 4;
    const int nwaves = W / 16;
    T sigma_est;
    if (sigma_in != nullptr) {
        sigma_est = sigma_in[prob];
    } else {
        double s = 0;
        for (int v = 0; v < nwaves; ++v) {                      // k_prox_tv: wave_sum over a wave's 16 columns, waves in order
            double part = lane < 16 ? (double)sig_cols[(size_t)prob * W + v * 16 + lane] : 0.0;
            s += wave_sum(part);
        }
        sigma_est = (T)(s / (double)W);
    }
    if (sigma_out != nullptr && wv == 0 && lane == 0) sigma_out[prob] = sigma_est;
    if (!DENOISE) return;
    const size_t base = (size_t)prob * H * W + (size_t)(q * RPC) * W + wv * 16 + cl;
    T x[RPC];
#pragma unroll
    for (int i = 0; i < RPC; ++i) x[i] = zin[base + (size_t)i * W];
    const T sigma = sigma_est > (T)0 ? sigma_est * sigma_modifier : fallback_sigma;
    haar_bayes_shrink<T, H>(x, sigma * sigma);
    double err = 0.0;
    if (xrec != nullptr) err = (double)column_sq_err<T, RPC>(x, xrec + base, W);
#pragma unroll
    for (int i = 0; i < RPC; ++i) zout[base + (size_t)i * W] = x[i];
    if (sse_out != nullptr) {
        err = wave_sum(err);
        __shared__ bool last;
        if (lane == 0) {
            partial[(size_t)prob * 16 + wv] = err;
            __threadfence();
            last = atomicAdd(&counter[prob], 1u) == (unsigned)(nwaves - 1);
        }
        __syncthreads();
        if (last && lane == 0) {
            __threadfence();
            double s = 0;
            for (int i = 0; i < nwaves; ++i) s += partial[(size_t)prob * 16 + i];
            sse_out[prob] = s;
            counter[prob] = 0;                                  // ready for the next call
        }
    }
}

constexpr int kSmallBatch = 32;                                // up to this many images take the split form
struct SmallProxScratch { void* sig_cols; double* partial; unsigned* counter; };

// scratch of the split form, one set per (device, stream) -- two streams running a small-batch prox at the same time would
// otherwise race on sig_cols and on the last-workgroup counter.  Nothing can be allocated while a stream is being
// captured, and a capture stream is a new stream: a CAPTURED call takes the device's default set, which every eager call makes
// sure exists (captured callers run one eager warm-up call first, as every graph in this code base does) -- graphs of one
// device therefore share a set and must not be replayed concurrently on two streams.  Sets are never freed or moved (graphs
// bake the pointers in); an entry is committed only once all of it exists.
static int small_prox_scratch(SmallProxScratch** out, hipStream_t stream) {
    struct Entry { int dev; hipStream_t stream; SmallProxScratch s; };
    static std::vector<Entry*> tab;
    static std::mutex mu;
    static const hipStream_t kDefault = (hipStream_t)(intptr_t)-1;
    int dev = 0;
    PNP_CHECK_HIP(hipGetDevice(&dev));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream != nullptr) (void)hipStreamIsCapturing(stream, &cap);
    std::lock_guard<std::mutex> lock(mu);
    auto find = [&](hipStream_t st) -> SmallProxScratch* {
        for (Entry* e : tab)
            if (e->dev == dev && e->stream == st) return &e->s;
        return nullptr;
    };
    auto create = [&](hipStream_t st, SmallProxScratch** res) -> int {
        SmallProxScratch n = {};
        hipError_t err = hipMalloc(&n.sig_cols, (size_t)kSmallBatch * 256 * sizeof(double));
        if (err == hipSuccess) err = hipMalloc((void**)&n.partial, (size_t)kSmallBatch * 16 * sizeof(double));
        if (err == hipSuccess) err = hipMalloc((void**)&n.counter, (size_t)kSmallBatch * sizeof(unsigned));
        if (err == hipSuccess) err = hipMemset(n.counter, 0, (size_t)kSmallBatch * sizeof(unsigned));
        if (err != hipSuccess) {
            if (n.sig_cols) (void)hipFree(n.sig_cols);
            if (n.partial) (void)hipFree(n.partial);
            if (n.counter) (void)hipFree(n.counter);
            set_error(std::string("small_prox_scratch: ") + hipGetErrorString(err));
            return PNP_ERR_HIP;
        }
        tab.push_back(new Entry{dev, st, n});
        *res = &tab.back()->s;
        return PNP_OK;
    };
    if (cap != hipStreamCaptureStatusNone) {
        SmallProxScratch* d = find(kDefault);
        if (d == nullptr) {
            set_error("small_prox_scratch: a small-batch prox was captured in a hipGraph before any eager call on this device (run one eager warm-up call first)");
            return PNP_ERR_HIP;
        }
        *out = d;
        return PNP_OK;
    }
    SmallProxScratch* d = find(kDefault);
    if (d == nullptr) { const int rc = create(kDefault, &d); if (rc != PNP_OK) return rc; }
    SmallProxScratch* e = find(stream);
    if (e == nullptr) { const int rc = create(stream, &e); if (rc != PNP_OK) return rc; }
    *out = e;
    return PNP_OK;
}

// ------------------------------------------------------------------------------- reductions
template <typename T>
__global__ __launch_bounds__(256) void k_sse(const T* __restrict__ z, const T* __restrict__ xr, int n, double* __restrict__ out) {
    __shared__ double red[4];
    const size_t base = (size_t)blockIdx.x * n;
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double d = (double)xr[base + i] - (double)z[base + i];
        acc += d * d;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ void k_axpbypcz(T a, const T* x, T b, const T* y, T c, const T* w, T* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = a * x[i];
        if (y != nullptr) v += b * y[i];
        if (w != nullptr) v += c * w[i];
        out[i] = v;
    }
}

template <typename T, int H, bool DENOISE>
int launch_prox(const void* zin, void* zout, int W, int batch, const void* sigma_in, double mod, double fb,
                const void* xrec, double* sse, void* sigma_out, hipStream_t s) {
    if (batch <= kSmallBatch && getenv("PNP_PROX_NO_SPLIT") == nullptr) {
        SmallProxScratch* sc = nullptr;
        const int rc = small_prox_scratch(&sc, s);
        if (rc != PNP_OK) return rc;
        const dim3 grid(W / 16, batch);
        if (sigma_in == nullptr) {
            k_sigma_cols<T, H><<<grid, 64, 0, s>>>((const T*)zin, W, (T*)sc->sig_cols);
            PNP_CHECK_LAUNCH();
        }
        k_shrink_cols<T, H, DENOISE><<<DENOISE ? grid : dim3(1, batch), 64, 0, s>>>(
            (const T*)zin, (T*)zout, W, (const T*)sc->sig_cols, (const T*)sigma_in, (T)mod, (T)fb, (const T*)xrec, sse,
            (T*)sigma_out, sc->partial, sc->counter);
        PNP_CHECK_LAUNCH();
        return PNP_OK;
    }
    k_prox_tv<T, H, DENOISE><<<batch, (W / 16) * 64, 0, s>>>((const T*)zin, (T*)zout, W, (const T*)sigma_in, (T)mod,
                                                          (T)fb, (const T*)xrec, sse, (T*)sigma_out);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

template <bool DENOISE>
int dispatch_prox(const void* zin, void* zout, int H, int W, int batch, int dtype, const void* sigma_in, double mod,
                  double fb, const void* xrec, double* sse, void* sigma_out, hipStream_t s) {
    PNP_CHECK_ARG(zin != nullptr && batch >= 1, "null input / empty batch");
    PNP_CHECK_ARG(H == 16 || H == 32 || H == 64 || H == 128 || H == 256, "H must be 16, 32, 64, 128 or 256");
    PNP_CHECK_ARG(W % 16 == 0 && W >= 16 && W <= 256, "W must be a multiple of 16 in [16, 256]");
    PNP_CHECK_ARG(dtype == PNP_F32 || dtype == PNP_F64, "bad dtype");
#define PNP_PROX_CASE(TT, HH) return launch_prox<TT, HH, DENOISE>(zin, zout, W, batch, sigma_in, mod, fb, xrec, sse, sigma_out, s)
    if (dtype == PNP_F32) {
        if (H == 256) PNP_PROX_CASE(float, 256);
        if (H == 128) PNP_PROX_CASE(float, 128);
        if (H == 64) PNP_PROX_CASE(float, 64);
        if (H == 32) PNP_PROX_CASE(float, 32);
        PNP_PROX_CASE(float, 16);
    }
    if (H == 256) PNP_PROX_CASE(double, 256);
    if (H == 128) PNP_PROX_CASE(double, 128);
    if (H == 64) PNP_PROX_CASE(double, 64);
    if (H == 32) PNP_PROX_CASE(double, 32);
    PNP_PROX_CASE(double, 16);
#undef PNP_PROX_CASE
}

}  // namespace pnp

using namespace pnp;

extern "C" int pnp_sigma_est(const void* z, int H, int W, int batch, int dtype, void* sigma_out, void* stream) {
    PNP_CHECK_ARG(sigma_out != nullptr, "null output");
    return dispatch_prox<false>(z, nullptr, H, W, batch, dtype, nullptr, 1.0, 0.0, nullptr, nullptr, sigma_out,
                                (hipStream_t)stream);
}

extern "C" int pnp_prox_tv(const void* z_in, void* z_out, int H, int W, int batch, int dtype, const void* sigma_in,
                           double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_out,
                           void* sigma_out, void* stream) {
    PNP_CHECK_ARG(z_out != nullptr, "null output");
    return dispatch_prox<true>(z_in, z_out, H, W, batch, dtype, sigma_in, sigma_modifier, fallback_sigma, xrec, sse_out,
                               sigma_out, (hipStream_t)stream);
}

extern "C" int pnp_sse(const void* z, const void* xrec, int n, int batch, int dtype, double* sse_out, void* stream) {
    PNP_CHECK_ARG(z && xrec && sse_out && n > 0 && batch > 0, "bad argument");
    if (dtype == PNP_F32) k_sse<float><<<batch, 256, 0, (hipStream_t)stream>>>((const float*)z, (const float*)xrec, n, sse_out);
    else if (dtype == PNP_F64) k_sse<double><<<batch, 256, 0, (hipStream_t)stream>>>((const double*)z, (const double*)xrec, n, sse_out);
    else PNP_CHECK_ARG(false, "bad dtype");
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_minmax(const void* z, int n, int batch, int dtype, void* out, void* stream) {
    PNP_CHECK_ARG(z && out && n > 0 && batch > 0, "bad argument");
    if (dtype == PNP_F32) k_minmax<float><<<batch, 256, 0, (hipStream_t)stream>>>((const float*)z, n, (float*)out);
    else if (dtype == PNP_F64) k_minmax<double><<<batch, 256, 0, (hipStream_t)stream>>>((const double*)z, n, (double*)out);
    else PNP_CHECK_ARG(false, "bad dtype");
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

// ---- device-resident step counter + log ring: what lets a whole outer iteration be captured in a hipGraph
__global__ void k_counter_add(uint32_t* ctr, uint32_t inc) { if (threadIdx.x == 0 && blockIdx.x == 0) *ctr += inc; }

__global__ void k_log_append(const double* __restrict__ src, int n, double* __restrict__ log, int n_log,
                             const uint32_t* __restrict__ step_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) log[(size_t)(*step_dev % (uint32_t)n_log) * n + i] = src[i];
}

// log[(*counter % n_log)][0..n) = src[0..n), then ++*counter: one block, so the increment follows every read of the counter
__global__ __launch_bounds__(256) void k_log_append_inc(const double* __restrict__ src, int n, double* __restrict__ log, int n_log,
                                                        uint32_t* __restrict__ counter) {
    const uint32_t row = *counter % (uint32_t)n_log;
    for (int i = threadIdx.x; i < n; i += 256) log[(size_t)row * n + i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) *counter += 1;
}

extern "C" int pnp_log_append_inc(const double* src, int n, double* log, int n_log, uint32_t* counter, void* stream) {
    PNP_CHECK_ARG(src && log && counter && n >= 1 && n_log >= 1, "bad argument");
    k_log_append_inc<<<1, 256, 0, (hipStream_t)stream>>>(src, n, log, n_log, counter);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_counter_add(uint32_t* counter, uint32_t inc, void* stream) {
    PNP_CHECK_ARG(counter != nullptr, "null counter");
    k_counter_add<<<1, 64, 0, (hipStream_t)stream>>>(counter, inc);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_log_append(const double* src, int n, double* log, int n_log, const uint32_t* step_dev, void* stream) {
    PNP_CHECK_ARG(src && log && step_dev && n >= 1 && n_log >= 1, "bad argument");
    k_log_append<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(src, n, log, n_log, step_dev);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_axpbypcz(double a, const void* x, double b, const void* y, double c, const void* w, void* out,
                            size_t n, int dtype, void* stream) {
    PNP_CHECK_ARG(x && out, "null argument");
    if (n == 0) return PNP_OK;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (dtype == PNP_F32)
        k_axpbypcz<float><<<grid, 256, 0, (hipStream_t)stream>>>((float)a, (const float*)x, (float)b, (const float*)y, (float)c, (const float*)w, (float*)out, n);
    else if (dtype == PNP_F64)
        k_axpbypcz<double><<<grid, 256, 0, (hipStream_t)stream>>>(a, (const double*)x, b, (const double*)y, c, (const double*)w, (double*)out, n);
    else PNP_CHECK_ARG(false, "bad dtype");
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}
