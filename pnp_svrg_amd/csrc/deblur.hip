// deblur.hip -- Deblur / super-resolution data-fidelity gradient, reference problems/DeblurSR.py:119-147:
//     fft_blur(a, B) = Re ifft(fft(a) * fft(B)) * sqrt(N)        (1-D circular convolution of the RAVELED image)
//     grad = fft_blur( Bop^H ( sel o (Bop fft_blur(z, B) - Y) ), roll(flip(B), 1) )  [/ M for grad_full]
// The adjoint kernel's spectrum is conj(fft(B)) exactly, and fft(B) is computed ONCE
// at plan creation (the reference recomputes both on every call: 4 forward + 2 inverse FFTs per gradient;
// here 2 + 2).
//
// Length-N (= n*n, n = 256 or 64) FFT as the four-step algorithm over the row-major n x n view
// x[a*n + b] of the vector, built from the same lane+NL*reg register FFT as the CSMRI kernels (fft.h):
//   forward : X[k1 + n*k2] = sum_b W_n^(b k2) * W_N^(b k1) * [ sum_a x[a n + b] W_n^(a k1) ]
//             k_colpass (DFT over the strided index a, twiddle, output Y[k1][b])  then a row DFT over b
//   inverse : x[n*m1 + m2] = sum_k1 W_n^-(k1 m1) * W_N^-(k1 m2) * [ sum_k2 X[k1][k2] W_n^-(k2 m2) ]
//             a row DFT over k2, twiddle, then k_colpass (DFT over the strided index k1)
// The spectrum lives in "[k1][k2]" (digit-swapped) order; that is irrelevant for a convolution as long
// as fft(B) is stored in the same order.  The forward row DFT, the spectrum product, the inverse row
// DFT and the inverse twiddle all act on the same contiguous line -> ONE kernel (k_rowpass), so
// a blur is 3 launches: k_colpass -> k_rowpass -> k_colpass (its real epilogue carries "- Y", the
// minibatch mask and the scaling).
#include "fft.h"
#include "draw.h"
#include <vector>
#include <cmath>

namespace pnp {

// RA x LA = register x lane split of one length-n line transform (fft.h: fft_gen): <8,8> n = 64, <8,16> n = 128,
// <16,16> n = 256.
template <typename T, int RA, int LA> struct LineSmem {
    static constexpr int N = RA * LA;
    static constexpr int LG = RA > LA ? RA : LA;
    static constexpr int G = 256 / LG;
    static constexpr int TILE = G * (N + 1);
    static constexpr int SCR = G * LG * (LG + 1);
    static constexpr int ELEMS = TILE > SCR ? TILE : SCR;
};

// DFT along the STRIDED index of the n x n view, G columns per block (loads/stores go through an LDS
// tile so that global accesses are row segments of G elements).
//   IN_REAL : input real T [n][n]           else complex
//   TWIDDLE : multiply output k1 of column b by W_N^(+-b*k1)
//   OUT_REAL: out real = sel ? alpha*Re(v) + beta*c : 0      else complex
template <typename T, int RA, int LA, bool INV, bool IN_REAL, bool TWIDDLE, bool OUT_REAL>
__global__ __launch_bounds__(256) void k_colpass(const void* __restrict__ in_, void* __restrict__ out_,
                                                 const cx<T>* __restrict__ tw_line, const cx<T>* __restrict__ tw_big,
                                                 T alpha, T beta, const T* __restrict__ c, const uint8_t* __restrict__ sel,
                                                 const MbDesc* __restrict__ mbd) {
    using S = LineSmem<T, RA, LA>;
    constexpr int N = S::N, G = S::G, LG = S::LG;
    __shared__ cx<T> smem[S::ELEMS];
    const int t = threadIdx.x, g = t / LG, lane = t % LG;
    const int prob = blockIdx.y, b0 = blockIdx.x * G;
    const size_t base = (size_t)prob * N * N;
    const int p = t % G;
    for (int a = t / G; a < N; a += 256 / G) {
        const size_t i = base + (size_t)a * N + b0 + p;
        cx<T> val;
        if (IN_REAL) val = {((const T*)in_)[i], (T)0};
        else val = ((const cx<T>*)in_)[i];
        smem[p * (N + 1) + a] = val;
    }
    __syncthreads();
    cx<T> v[LG], tw[LG];
    load_twiddles_gen<T, LG>(tw, tw_line, lane, N);
#pragma unroll
    for (int r = 0; r < RA; ++r) v[r] = smem[g * (N + 1) + (lane < LA ? lane : 0) + LA * r];
    fft_gen<T, RA, LA, INV>(v, tw, smem + g * LG * (LG + 1), lane);
    __syncthreads();
    if (lane < RA) {
#pragma unroll
        for (int r = 0; r < LA; ++r) {
            const int k1 = lane + RA * r;
            cx<T> o = v[r];
            if (TWIDDLE) {
                cx<T> w = tw_big[(size_t)(b0 + g) * k1];
                if (INV) w = cconj(w);
                o = cmul(o, w);
            }
            smem[g * (N + 1) + k1] = o;
        }
    }
    __syncthreads();
    for (int k1 = t / G; k1 < N; k1 += 256 / G) {
        const size_t i = base + (size_t)k1 * N + b0 + p;
        const cx<T> val = smem[p * (N + 1) + k1];
        if (OUT_REAL) {
            T o = alpha * val.x;
            if (c != nullptr) o += beta * c[i];
            if (sel != nullptr && sel[i] == 0) o = (T)0;
            if (mbd != nullptr && !mb_member(mbd[prob], (uint32_t)(k1 * N + b0 + p))) o = (T)0;   // device-drawn minibatch
            ((T*)out_)[i] = o;
        } else {
            ((cx<T>*)out_)[i] = val;
        }
    }
}

// Contiguous lines (one per k1): forward DFT -> [x mul or conj(mul)] -> (SPECTRUM_ONLY: store) else
// inverse DFT -> x conj(W_N^(k1*m2)) -> store.  In place allowed.
template <typename T, int RA, int LA, bool CONJ, bool SPECTRUM_ONLY>
__global__ __launch_bounds__(256) void k_rowpass(const cx<T>* in, cx<T>* out, const cx<T>* __restrict__ tw_line,
                                                 const cx<T>* __restrict__ tw_big, const cx<T>* __restrict__ mul) {
    using S = LineSmem<T, RA, LA>;
    constexpr int N = S::N, G = S::G, LG = S::LG;
    __shared__ cx<T> smem[S::SCR];
    const int t = threadIdx.x, g = t / LG, lane = t % LG;
    const int prob = blockIdx.y, k1 = blockIdx.x * G + g;
    const size_t base = (size_t)prob * N * N + (size_t)k1 * N;
    cx<T> v[LG], tw[LG];
    load_twiddles_gen<T, LG>(tw, tw_line, lane, N);
#pragma unroll
    for (int r = 0; r < RA; ++r) v[r] = in[base + (lane < LA ? lane : 0) + LA * r];
    fft_gen<T, RA, LA, false>(v, tw, smem + g * LG * (LG + 1), lane);     // -> lanes < RA, element lane + RA*r
    const int ln = lane < RA ? lane : 0;
    if (mul != nullptr) {
#pragma unroll
        for (int r = 0; r < LA; ++r) {
            cx<T> m = mul[(size_t)k1 * N + ln + RA * r];
            if (CONJ) m = cconj(m);
            v[r] = cmul(v[r], m);
        }
    }
    if (SPECTRUM_ONLY) {
        if (lane < RA) {
#pragma unroll
            for (int r = 0; r < LA; ++r) out[base + lane + RA * r] = v[r];
        }
    } else {
        fft_gen<T, LA, RA, true>(v, tw, smem + g * LG * (LG + 1), lane);   // -> lanes < LA, element lane + LA*r
        if (lane < LA) {
#pragma unroll
            for (int r = 0; r < RA; ++r)
                out[base + lane + LA * r] = cmul(v[r], cconj(tw_big[(size_t)k1 * (lane + LA * r)]));
        }
    }
}

// 4-tap sparse operator (pylops Bilinear forward / its CSR adjoint): out[m] = sum_t w[m][t] * x[idx[m][t]]
template <typename T>
__global__ void k_gather4(const T* __restrict__ x, const int32_t* __restrict__ idx, const T* __restrict__ w, int n_out,
                          int n_in, T beta, const T* __restrict__ c, const uint8_t* __restrict__ sel,
                          const MbDesc* __restrict__ mbd, T* __restrict__ out) {
    const int prob = blockIdx.y;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_out) return;
    const T* xp = x + (size_t)prob * n_in;
    T o = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) o += w[(size_t)m * 4 + t] * xp[idx[(size_t)m * 4 + t]];
    const size_t i = (size_t)prob * n_out + m;
    if (c != nullptr) o += beta * c[i];
    if (sel != nullptr && sel[i] == 0) o = (T)0;
    if (mbd != nullptr && !mb_member(mbd[prob], (uint32_t)m)) o = (T)0;
    out[i] = o;
}

template <typename T>
__global__ void k_csr(const T* __restrict__ y, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                      const T* __restrict__ val, int n_out, int n_in, T* __restrict__ out) {
    const int prob = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_out) return;
    const T* yp = y + (size_t)prob * n_in;
    T o = 0;
    for (int e = rowptr[n]; e < rowptr[n + 1]; ++e) o += val[e] * yp[col[e]];
    out[(size_t)prob * n_out + n] = o;
}

}  // namespace pnp

using namespace pnp;

struct pnp_deblur_plan {
    int n, N, NL, batch, dtype, M;            // N = n*n = H*W; M = number of measurements
    void *tw_line, *tw_big, *FB;              // [n], [N], [N] complex
    void *w0, *r0, *r1;                       // complex [batch][N]; real [batch][N] x2
    // optional bilinear operator
    int32_t *g_idx, *a_rowptr, *a_col;
    void *g_w, *a_val, *down;                 // [M][4], [nnz], real [batch][M]
};

namespace {
// real [batch][N] -> Y[k1][b] (complex, work buffer w0)
template <typename T, int RA, int LA>
int col_fwd(pnp_deblur_plan* p, const T* x, int batch, hipStream_t s) {
    constexpr int G = LineSmem<T, RA, LA>::G;
    k_colpass<T, RA, LA, false, true, true, false><<<dim3(p->n / G, batch), 256, 0, s>>>(
        x, p->w0, (const cx<T>*)p->tw_line, (const cx<T>*)p->tw_big, (T)0, (T)0, nullptr, nullptr, nullptr);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

// one blur: out = sel ? alpha * (x (*) kernel) * sqrt(N)-normalised + beta*c : 0
template <typename T, int RA, int LA>
int blur(pnp_deblur_plan* p, const T* x, bool conj_kernel, T alpha, T beta, const T* c, const uint8_t* sel, T* out,
         hipStream_t s, const MbDesc* mbd = nullptr) {
    constexpr int G = LineSmem<T, RA, LA>::G;
    const int B = p->batch;
    int rc = col_fwd<T, RA, LA>(p, x, B, s);
    if (rc) return rc;
    dim3 grid(p->n / G, B);
    cx<T>* w0 = (cx<T>*)p->w0;
    if (conj_kernel)
        k_rowpass<T, RA, LA, true, false><<<grid, 256, 0, s>>>(w0, w0, (const cx<T>*)p->tw_line, (const cx<T>*)p->tw_big, (const cx<T>*)p->FB);
    else
        k_rowpass<T, RA, LA, false, false><<<grid, 256, 0, s>>>(w0, w0, (const cx<T>*)p->tw_line, (const cx<T>*)p->tw_big, (const cx<T>*)p->FB);
    PNP_CHECK_LAUNCH();
    k_colpass<T, RA, LA, true, false, false, true><<<grid, 256, 0, s>>>(w0, out, (const cx<T>*)p->tw_line, (const cx<T>*)p->tw_big,
                                                                  alpha, beta, c, sel, mbd);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

template <typename T, int RA, int LA>
int run_grad(pnp_deblur_plan* p, const T* z, const T* Y, const uint8_t* sel, double scale, T* out, hipStream_t s,
             const MbDesc* mbd = nullptr) {
    const int N = p->N, B = p->batch;
    const T inv_sqrtN = (T)(1.0 / std::sqrt((double)N));           // Re ifft(.) * sqrt(N); the inverse carries 1/N
    T* res = (T*)p->r0;
    int rc;
    if (p->g_idx == nullptr) {
        rc = blur<T, RA, LA>(p, z, false, inv_sqrtN, (T)-1, Y, sel, res, s, mbd);   // sel o (B z - Y)
        if (rc) return rc;
    } else {
        T* blurred = (T*)p->r1;
        rc = blur<T, RA, LA>(p, z, false, inv_sqrtN, (T)0, nullptr, nullptr, blurred, s);
        if (rc) return rc;
        T* down = (T*)p->down;
        k_gather4<T><<<dim3((p->M + 255) / 256, B), 256, 0, s>>>(blurred, p->g_idx, (const T*)p->g_w, p->M, N, (T)-1, Y, sel, mbd, down);
        PNP_CHECK_LAUNCH();
        k_csr<T><<<dim3((N + 255) / 256, B), 256, 0, s>>>(down, p->a_rowptr, p->a_col, (const T*)p->a_val, N, p->M, res);
        PNP_CHECK_LAUNCH();
    }
    return blur<T, RA, LA>(p, res, true, (T)(scale / std::sqrt((double)N)), (T)0, nullptr, nullptr, out, s);
}

template <typename T, int RA, int LA>
int run_forward(pnp_deblur_plan* p, const T* x, T* out, hipStream_t s) {
    const T inv_sqrtN = (T)(1.0 / std::sqrt((double)p->N));
    if (p->g_idx == nullptr) return blur<T, RA, LA>(p, x, false, inv_sqrtN, (T)0, nullptr, nullptr, out, s);
    T* blurred = (T*)p->r1;
    int rc = blur<T, RA, LA>(p, x, false, inv_sqrtN, (T)0, nullptr, nullptr, blurred, s);
    if (rc) return rc;
    k_gather4<T><<<dim3((p->M + 255) / 256, p->batch), 256, 0, s>>>(blurred, p->g_idx, (const T*)p->g_w, p->M, p->N, (T)0, nullptr, nullptr, nullptr, out);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

// FB = spectrum of the blur kernel in the plan's [k1][k2] order
template <typename T, int RA, int LA>
int make_spectrum(pnp_deblur_plan* p, hipStream_t s) {
    constexpr int G = LineSmem<T, RA, LA>::G;
    int rc = col_fwd<T, RA, LA>(p, (const T*)p->r0, 1, s);
    if (rc) return rc;
    k_rowpass<T, RA, LA, false, true><<<dim3(p->n / G, 1), 256, 0, s>>>((const cx<T>*)p->w0, (cx<T>*)p->FB, (const cx<T>*)p->tw_line,
                                                                   (const cx<T>*)p->tw_big, nullptr);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

template <typename T> void fill_tw(std::vector<cx<T>>& tab, size_t N) {
    tab.resize(N);
    for (size_t j = 0; j < N; ++j) {
        const double ang = -2.0 * 3.14159265358979323846 * (double)j / (double)N;
        tab[j] = {(T)std::cos(ang), (T)std::sin(ang)};
    }
}
}  // namespace

// Bk: the blur kernel vector B (DeblurSR.py:93, already / N), HOST pointer, `dtype` elements.
// Bilinear operator (optional, all HOST pointers; pass M = N and NULLs for scale_percent == 100):
//   g_idx/g_w [M][4]: forward taps; a_rowptr [N+1], a_col/a_val [nnz]: CSR of the adjoint.
extern "C" int pnp_deblur_plan_create(pnp_deblur_plan** out, int H, int W, int batch, int dtype, const void* Bk, int M,
                                      const int32_t* g_idx, const void* g_w, const int32_t* a_rowptr,
                                      const int32_t* a_col, const void* a_val) {
    PNP_CHECK_ARG(out && Bk, "null argument");
    const int N = H * W;
    PNP_CHECK_ARG(N == 65536 || N == 16384 || N == 4096, "H*W must be 65536 (256x256), 16384 (128x128) or 4096 (64x64)");
    PNP_CHECK_ARG(dtype == PNP_F32 || dtype == PNP_F64, "bad dtype");
    PNP_CHECK_ARG(batch >= 1 && M >= 1 && M <= N, "bad batch / M");
    PNP_CHECK_ARG(g_idx == nullptr || (g_w && a_rowptr && a_col && a_val), "bilinear operator needs all five arrays");
    auto* p = new pnp_deblur_plan{};
    p->N = N; p->n = N == 65536 ? 256 : N == 16384 ? 128 : 64; p->NL = N == 65536 ? 16 : N == 16384 ? 12 : 8; p->batch = batch; p->dtype = dtype; p->M = M;
    const size_t rs = dtype == PNP_F32 ? 4 : 8, cs = 2 * rs;
    hipError_t e = hipMalloc(&p->tw_line, p->n * cs);
    if (e == hipSuccess) e = hipMalloc(&p->tw_big, (size_t)N * cs);
    if (e == hipSuccess) e = hipMalloc(&p->FB, (size_t)N * cs);
    if (e == hipSuccess) e = hipMalloc(&p->w0, (size_t)batch * N * cs);
    if (e == hipSuccess) e = hipMalloc(&p->r0, (size_t)batch * N * rs);
    if (e == hipSuccess) e = hipMalloc(&p->r1, (size_t)batch * N * rs);
    if (e == hipSuccess) {
        if (dtype == PNP_F32) {
            std::vector<cx<float>> a, b; fill_tw(a, p->n); fill_tw(b, N);
            e = hipMemcpy(p->tw_line, a.data(), p->n * cs, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(p->tw_big, b.data(), (size_t)N * cs, hipMemcpyHostToDevice);
        } else {
            std::vector<cx<double>> a, b; fill_tw(a, p->n); fill_tw(b, N);
            e = hipMemcpy(p->tw_line, a.data(), p->n * cs, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(p->tw_big, b.data(), (size_t)N * cs, hipMemcpyHostToDevice);
        }
    }
    if (e == hipSuccess && g_idx != nullptr) {
        const int nnz = a_rowptr[N];
        e = hipMalloc(&p->g_idx, (size_t)M * 4 * 4);
        if (e == hipSuccess) e = hipMemcpy(p->g_idx, g_idx, (size_t)M * 16, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&p->g_w, (size_t)M * 4 * rs);
        if (e == hipSuccess) e = hipMemcpy(p->g_w, g_w, (size_t)M * 4 * rs, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&p->a_rowptr, (size_t)(N + 1) * 4);
        if (e == hipSuccess) e = hipMemcpy(p->a_rowptr, a_rowptr, (size_t)(N + 1) * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&p->a_col, (size_t)nnz * 4);
        if (e == hipSuccess) e = hipMemcpy(p->a_col, a_col, (size_t)nnz * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&p->a_val, (size_t)nnz * rs);
        if (e == hipSuccess) e = hipMemcpy(p->a_val, a_val, (size_t)nnz * rs, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&p->down, (size_t)batch * M * rs);
    }
    if (e == hipSuccess) {
        // FB = fft(B): upload B into r0 (as problem 0), transform once
        e = hipMemcpy(p->r0, Bk, (size_t)N * rs, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            int rc;
#define PNP_DB_DISPATCH(FN, ...)                                                                         \
    (p->dtype == PNP_F32 ? (p->NL == 16 ? FN<float, 16, 16>(__VA_ARGS__) : p->NL == 12 ? FN<float, 8, 16>(__VA_ARGS__) : FN<float, 8, 8>(__VA_ARGS__)) \
                         : (p->NL == 16 ? FN<double, 16, 16>(__VA_ARGS__) : p->NL == 12 ? FN<double, 8, 16>(__VA_ARGS__) : FN<double, 8, 8>(__VA_ARGS__)))
            rc = PNP_DB_DISPATCH(make_spectrum, p, (hipStream_t)0);
            if (rc) e = hipErrorUnknown;
            else e = hipDeviceSynchronize();
        }
    }
    if (e != hipSuccess) {
        set_error(std::string("pnp_deblur_plan_create: ") + hipGetErrorString(e));
        for (void* q : {p->tw_line, p->tw_big, p->FB, p->w0, p->r0, p->r1, (void*)p->g_idx, p->g_w, (void*)p->a_rowptr,
                        (void*)p->a_col, p->a_val, p->down})
            if (q) (void)hipFree(q);
        delete p;
        return PNP_ERR_HIP;
    }
    *out = p;
    return PNP_OK;
}

extern "C" int pnp_deblur_plan_destroy(pnp_deblur_plan* p) {
    if (!p) return PNP_OK;
    for (void* q : {p->tw_line, p->tw_big, p->FB, p->w0, p->r0, p->r1, (void*)p->g_idx, p->g_w, (void*)p->a_rowptr,
                    (void*)p->a_col, p->a_val, p->down})
        if (q) (void)hipFree(q);
    delete p;
    return PNP_OK;
}

// out = scale * B^T S^T ( sel o (S B z - Y) );  sel (uint8 [batch][M], may be NULL = all measurements)
extern "C" int pnp_deblur_grad(pnp_deblur_plan* p, const void* z, const void* Y, const uint8_t* sel, double scale,
                               void* out, void* stream) {
    PNP_CHECK_ARG(p && z && Y && out, "null argument");
    hipStream_t s = (hipStream_t)stream;
    if (p->dtype == PNP_F32) {
        const float *zz = (const float*)z, *yy = (const float*)Y;
        float* oo = (float*)out;
        return p->NL == 16 ? run_grad<float, 16, 16>(p, zz, yy, sel, scale, oo, s)
             : p->NL == 12 ? run_grad<float, 8, 16>(p, zz, yy, sel, scale, oo, s) : run_grad<float, 8, 8>(p, zz, yy, sel, scale, oo, s);
    }
    const double *zz = (const double*)z, *yy = (const double*)Y;
    double* oo = (double*)out;
    return p->NL == 16 ? run_grad<double, 16, 16>(p, zz, yy, sel, scale, oo, s)
         : p->NL == 12 ? run_grad<double, 8, 16>(p, zz, yy, sel, scale, oo, s) : run_grad<double, 8, 8>(p, zz, yy, sel, scale, oo, s);
}

// the same with a device-drawn minibatch given as this step's threshold descriptors (pnp_draw_thresholds over M)
extern "C" int pnp_deblur_grad_mb(pnp_deblur_plan* p, const void* z, const void* Y, const void* mbd, double scale,
                                  void* out, void* stream) {
    PNP_CHECK_ARG(p && z && Y && mbd && out, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const MbDesc* d = (const MbDesc*)mbd;
    if (p->dtype == PNP_F32) {
        const float *zz = (const float*)z, *yy = (const float*)Y;
        float* oo = (float*)out;
        return p->NL == 16 ? run_grad<float, 16, 16>(p, zz, yy, nullptr, scale, oo, s, d)
             : p->NL == 12 ? run_grad<float, 8, 16>(p, zz, yy, nullptr, scale, oo, s, d) : run_grad<float, 8, 8>(p, zz, yy, nullptr, scale, oo, s, d);
    }
    const double *zz = (const double*)z, *yy = (const double*)Y;
    double* oo = (double*)out;
    return p->NL == 16 ? run_grad<double, 16, 16>(p, zz, yy, nullptr, scale, oo, s, d)
         : p->NL == 12 ? run_grad<double, 8, 16>(p, zz, yy, nullptr, scale, oo, s, d) : run_grad<double, 8, 8>(p, zz, yy, nullptr, scale, oo, s, d);
}

// forward model S B x (DeblurSR.py:110-112), for problem setup / f(w); out real [batch][M]
extern "C" int pnp_deblur_forward(pnp_deblur_plan* p, const void* x, void* out, void* stream) {
    PNP_CHECK_ARG(p && x && out, "null argument");
    hipStream_t s = (hipStream_t)stream;
    if (p->dtype == PNP_F32) {
        const float* xx = (const float*)x;
        float* oo = (float*)out;
        return p->NL == 16 ? run_forward<float, 16, 16>(p, xx, oo, s) : p->NL == 12 ? run_forward<float, 8, 16>(p, xx, oo, s) : run_forward<float, 8, 8>(p, xx, oo, s);
    }
    const double* xx = (const double*)x;
    double* oo = (double*)out;
    return p->NL == 16 ? run_forward<double, 16, 16>(p, xx, oo, s) : p->NL == 12 ? run_forward<double, 8, 16>(p, xx, oo, s) : run_forward<double, 8, 8>(p, xx, oo, s);
}
