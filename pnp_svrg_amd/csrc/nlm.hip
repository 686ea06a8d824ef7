// nlm.hip -- non-local-means prox, reference denoisers/NLM.py:22-27 -> skimage 0.18
// `_nl_means_denoising_2d` (slow mode, one channel), restated from the compiled kernel's semantics
// (oracle/denoise.py:nl_means_2d is the bit-exact CPU restatement):
//   * patch side s (4 -> 5), search radius d (5): 11 x 11 window clipped at the image border,
//     patches taken from the reflect-padded image;
//   * dist = sum_rows sum_cols w[i][j] * (diff^2 - var), accumulated sequentially in that order; before
//     each patch ROW the running distance is tested: dist > 5 -> weight 0 (early exit);
//   * weight = fast_exp(-max(0, dist)) with Schraudolph's integer trick (NOT exp), then
//     out = sum weight * centre / sum weight, accumulated in window order.
// One thread per output pixel; the (16 + 2d + s - 1)^2 neighbourhood of a 16 x 16 output tile is
// staged in LDS in reflect-padded coordinates.  Arithmetic order is the reference's (no FMA
// contraction in this file), so the f64 instantiation reproduces the CPU kernel bit for bit; the f32
// production instantiation fuses the two multiply-adds of a patch element explicitly (-7 % time, within the
// f32 parity bounds of tests/test_gpu_nlm.py and the config-4 traces).
#include "common.h"
#include <cstdlib>

namespace pnp {

constexpr int NT = 16;                       // output tile side
constexpr int NLM_MAX_SIDE = NT + 2 * 8 + 6; // d <= 8, s <= 7

__device__ __forceinline__ double fast_exp_d(double y) {
    // skimage/_shared/fast_exp.h: high word = (int32)(2^20/ln2 * y) + (1072693248 - 60801), low word 0
    const int hi = (int)(1512775.3951951856938 * y) + 1072632447;
    return __hiloint2double(hi, 0);
}

__device__ __forceinline__ int reflect_idx(int i, int n) {   // np.pad(mode='reflect')
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

template <typename T, int S>
__global__ __launch_bounds__(NT * NT) void k_nlm(const T* __restrict__ zin, T* __restrict__ zout, int H, int W, int d,
                                                 const T* __restrict__ sigma_in, double modifier, double fixed_h,
                                                 const double* __restrict__ w0, double w0_sum,
                                                 const T* __restrict__ xrec, double* __restrict__ sse_part) {
    constexpr int OFF = S / 2;
    __shared__ T tile[NLM_MAX_SIDE * NLM_MAX_SIDE];
    __shared__ double red[4];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * NT, c0 = blockIdx.x * NT;
    const int side = NT + 2 * d + S - 1;
    const T* img = zin + (size_t)b * H * W;
    // tile[y][x] = padded[r0 - d + y][c0 - d + x], padded[u][v] = img[reflect(u - OFF)][reflect(v - OFF)]
    for (int i = threadIdx.x; i < side * side; i += NT * NT) {
        const int y = i / side, x = i - y * side;
        const int u = r0 - d + y - OFF, v = c0 - d + x - OFF;      // image coordinates before reflection
        T val = (T)0;
        if (u >= -OFF && u < H + OFF && v >= -OFF && v < W + OFF) val = img[(size_t)reflect_idx(u, H) * W + reflect_idx(v, W)];
        tile[i] = val;
    }
    __syncthreads();

    // h, var and the patch weights (NLM.py:24-27; non_local_means.py:153)
    T h, var;
    if (sigma_in != nullptr) {
        h = (T)((double)sigma_in[b] * modifier);
        var = (T)2 * (h * h);
    } else {
        h = (T)fixed_h;
        var = (T)0;
    }
    const double hd = (double)h;
    const double scale = 1.0 / (1 * w0_sum * hd * hd);
    T w[S * S];
#pragma unroll
    for (int i = 0; i < S * S; ++i) w[i] = (T)(w0[i] * scale);

    const int ly = threadIdx.x / NT, lx = threadIdx.x % NT;
    const int row = r0 + ly, col = c0 + lx;
    double err = 0.0;
    if (row < H && col < W) {
        // own patch in registers
        T own[S * S];
#pragma unroll
        for (int pi = 0; pi < S; ++pi)
#pragma unroll
            for (int pj = 0; pj < S; ++pj) own[pi * S + pj] = tile[(ly + d + pi) * side + lx + d + pj];
        const int i_lo = row - (d < row ? d : row), i_hi = row + (d + 1 < H - row ? d + 1 : H - row);
        const int j_lo = col - (d < col ? d : col), j_hi = col + (d + 1 < W - col ? d + 1 : W - col);
        T wsum = (T)0, acc = (T)0;
        for (int i = i_lo; i < i_hi; ++i) {
            for (int j = j_lo; j < j_hi; ++j) {
                const T* nb = tile + (i - r0 + d) * side + (j - c0 + d);
                T dist = (T)0;
                bool dead = false;
#pragma unroll
                for (int pi = 0; pi < S; ++pi) {
                    if (dist > (T)5) { dead = true; break; }
#pragma unroll
                    for (int pj = 0; pj < S; ++pj) {
                        const T df = own[pi * S + pj] - nb[pi * side + pj];
                        if constexpr (sizeof(T) == 4) {
                            // f32 production path: two fused multiply-adds per element instead of mul, sub, mul, add
                            // (the kernel is bound by the vector ALU); the f64 parity path stays product for product
                            dist = __builtin_fmaf(w[pi * S + pj], __builtin_fmaf(df, df, -var), dist);
                        } else {
                            dist += w[pi * S + pj] * (df * df - var);
                        }
                    }
                }
                T weight = (T)0;
                if (!dead) {
                    const double dd = (double)dist;
                    weight = (T)fast_exp_d(-(dd > 0.0 ? dd : 0.0));
                }
                wsum += weight;
                acc += weight * nb[OFF * side + OFF];
            }
        }
        const T o = acc / wsum;
        const size_t p = (size_t)b * H * W + (size_t)row * W + col;
        zout[p] = o;
        if (xrec != nullptr) {
            const double df = (double)xrec[p] - (double)o;
            err = df * df;
        }
    }
    if (sse_part != nullptr) {
        err = wave_sum(err);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = err;
        __syncthreads();
        if (threadIdx.x == 0)
            sse_part[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

// The same kernel for the reference's own search radius (D = 5) with the neighbour patches kept in REGISTERS: a strip of
// S rows x (S + 2D) columns holds every patch of one window row; the 2D + 1 column offsets are unrolled, so their patch
// elements are static register indices, and moving to the next window row shifts the strip by one image row (S + 2D new
// LDS reads).  LDS reads per output pixel: (2D + 1)(S + 2D) + (S - 1)(S + 2D) = 225 instead of (2D + 1)^2 S^2 = 3025 --
// k_nlm is bound by its ds_read_b32 stream (25.6 KB per window offset and workgroup against ~150 cycles of arithmetic).
// Arithmetic and its order are those of k_nlm (the running distance is tested before every patch row; a dead candidate
// simply stops counting), so the f64 instantiation stays bit-exact.
template <typename T, int S, int D>
__global__ __launch_bounds__(NT * NT) void k_nlm_strip(const T* __restrict__ zin, T* __restrict__ zout, int H, int W,
                                                       const T* __restrict__ sigma_in, double modifier, double fixed_h,
                                                       const double* __restrict__ w0, double w0_sum,
                                                       const T* __restrict__ xrec, double* __restrict__ sse_part) {
    constexpr int OFF = S / 2, SIDE = NT + 2 * D + S - 1, SW = S + 2 * D;
    __shared__ T tile[SIDE * SIDE];
    __shared__ double red[4];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * NT, c0 = blockIdx.x * NT;
    const T* img = zin + (size_t)b * H * W;
    for (int i = threadIdx.x; i < SIDE * SIDE; i += NT * NT) {
        const int y = i / SIDE, x = i - y * SIDE;
        const int u = r0 - D + y - OFF, v = c0 - D + x - OFF;
        T val = (T)0;
        if (u >= -OFF && u < H + OFF && v >= -OFF && v < W + OFF) val = img[(size_t)reflect_idx(u, H) * W + reflect_idx(v, W)];
        tile[i] = val;
    }
    __syncthreads();

    T h, var;
    if (sigma_in != nullptr) {
        h = (T)((double)sigma_in[b] * modifier);
        var = (T)2 * (h * h);
    } else {
        h = (T)fixed_h;
        var = (T)0;
    }
    const double hd = (double)h;
    const double scale = 1.0 / (1 * w0_sum * hd * hd);
    T w[S * S];
#pragma unroll
    for (int i = 0; i < S * S; ++i) w[i] = (T)(w0[i] * scale);

    const int ly = threadIdx.x / NT, lx = threadIdx.x % NT;
    const int row = r0 + ly, col = c0 + lx;
    double err = 0.0;
    if (row < H && col < W) {
        T own[S * S];
#pragma unroll
        for (int pi = 0; pi < S; ++pi)
#pragma unroll
            for (int pj = 0; pj < S; ++pj) own[pi * S + pj] = tile[(ly + D + pi) * SIDE + lx + D + pj];
        // strip[pi][k] = tile[ly + (di + D) + pi][lx + k]: the patches of window row di; starts at di = -D
        T strip[S][SW];
#pragma unroll
        for (int pi = 0; pi < S - 1; ++pi)
#pragma unroll
            for (int k = 0; k < SW; ++k) strip[pi + 1][k] = tile[(ly + pi) * SIDE + lx + k];
        T wsum = (T)0, acc = (T)0;
#pragma unroll 1
        for (int di = -D; di <= D; ++di) {
            // shift the strip down by one image row
#pragma unroll
            for (int pi = 0; pi < S - 1; ++pi)
#pragma unroll
                for (int k = 0; k < SW; ++k) strip[pi][k] = strip[pi + 1][k];
#pragma unroll
            for (int k = 0; k < SW; ++k) strip[S - 1][k] = tile[(ly + di + D + S - 1) * SIDE + lx + k];
            if (row + di < 0 || row + di >= H) continue;
#pragma unroll
            for (int dj = -D; dj <= D; ++dj) {
                if (col + dj < 0 || col + dj >= W) continue;
                T dist = (T)0;
                bool dead = false;
#pragma unroll
                for (int pi = 0; pi < S; ++pi) {
                    dead = dead || dist > (T)5;
#pragma unroll
                    for (int pj = 0; pj < S; ++pj) {
                        const T df = own[pi * S + pj] - strip[pi][dj + D + pj];
                        if constexpr (sizeof(T) == 4) dist = __builtin_fmaf(w[pi * S + pj], __builtin_fmaf(df, df, -var), dist);
                        else dist += w[pi * S + pj] * (df * df - var);
                    }
                }
                T weight = (T)0;
                if (!dead) {
                    const double dd = (double)dist;
                    weight = (T)fast_exp_d(-(dd > 0.0 ? dd : 0.0));
                }
                wsum += weight;
                acc += weight * strip[OFF][dj + D + OFF];
            }
        }
        const T o = acc / wsum;
        const size_t p = (size_t)b * H * W + (size_t)row * W + col;
        zout[p] = o;
        if (xrec != nullptr) {
            const double df = (double)xrec[p] - (double)o;
            err = df * df;
        }
    }
    if (sse_part != nullptr) {
        err = wave_sum(err);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = err;
        __syncthreads();
        if (threadIdx.x == 0)
            sse_part[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

__global__ void k_sum_parts_nlm(const double* __restrict__ part, int nparts, double* __restrict__ out) {
    double s = 0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[(size_t)blockIdx.x * nparts + i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

template <typename T, int S>
int launch_nlm(const void* zin, void* zout, int H, int W, int batch, int d, const void* sigma_in, double modifier,
               double fixed_h, const double* w0, double w0_sum, const void* xrec, double* sse_out, double* sse_part,
               hipStream_t s) {
    dim3 grid((W + NT - 1) / NT, (H + NT - 1) / NT, batch);
    static const bool generic = getenv("PNP_NLM_GENERIC") != nullptr;      // diagnostic: the LDS-streaming form for every radius
    if constexpr (S == 5) {                                                 // the reference's configuration (denoisers/NLM.py:22-27)
        // (f64, the parity mode, needs 256 VGPRs for the strip: one wave per SIMD, slower than the LDS form on large batches)
        if (d == 5 && !generic && (sizeof(T) == 4 || batch <= 4)) {
            k_nlm_strip<T, 5, 5><<<grid, NT * NT, 0, s>>>((const T*)zin, (T*)zout, H, W, (const T*)sigma_in, modifier, fixed_h, w0,
                                                          w0_sum, (const T*)xrec, sse_out ? sse_part : nullptr);
            PNP_CHECK_LAUNCH();
            if (sse_out) {
                k_sum_parts_nlm<<<batch, 64, 0, s>>>(sse_part, (int)(grid.x * grid.y), sse_out);
                PNP_CHECK_LAUNCH();
            }
            return PNP_OK;
        }
    }
    {
        k_nlm<T, S><<<grid, NT * NT, 0, s>>>((const T*)zin, (T*)zout, H, W, d, (const T*)sigma_in, modifier, fixed_h, w0,
                                             w0_sum, (const T*)xrec, sse_out ? sse_part : nullptr);
    }
    PNP_CHECK_LAUNCH();
    if (sse_out) {
        k_sum_parts_nlm<<<batch, 64, 0, s>>>(sse_part, (int)(grid.x * grid.y), sse_out);
        PNP_CHECK_LAUNCH();
    }
    return PNP_OK;
}

}  // namespace pnp

using namespace pnp;

extern "C" int pnp_nlm2d(const void* z_in, void* z_out, int H, int W, int batch, int dtype, int patch_size,
                         int patch_distance, const void* sigma_in, double sigma_modifier, double fixed_h,
                         const double* w0, double w0_sum, const void* xrec, double* sse_out, double* sse_workspace,
                         void* stream) {
    PNP_CHECK_ARG(z_in && z_out && w0 && batch >= 1, "null argument");
    PNP_CHECK_ARG(z_in != z_out, "NLM cannot run in place (every output reads an 15x15 input neighbourhood)");
    const int s = patch_size % 2 == 0 ? patch_size + 1 : patch_size;          // skimage bumps even sizes
    PNP_CHECK_ARG(s == 3 || s == 5 || s == 7, "patch side must be 3, 5 or 7 (after the even->odd bump)");
    PNP_CHECK_ARG(patch_distance >= 1 && patch_distance <= 8, "patch_distance must be in [1, 8]");
    PNP_CHECK_ARG(!(sse_out && !(xrec && sse_workspace)), "sse_out needs xrec and a workspace");
    PNP_CHECK_ARG(dtype == PNP_F32 || dtype == PNP_F64, "bad dtype");
    hipStream_t st = (hipStream_t)stream;
#define PNP_NLM_CASE(TT, SS) return launch_nlm<TT, SS>(z_in, z_out, H, W, batch, patch_distance, sigma_in, sigma_modifier, fixed_h, w0, w0_sum, xrec, sse_out, sse_workspace, st)
    if (dtype == PNP_F64) {
        if (s == 3) PNP_NLM_CASE(double, 3);
        if (s == 5) PNP_NLM_CASE(double, 5);
        PNP_NLM_CASE(double, 7);
    }
    if (s == 3) PNP_NLM_CASE(float, 3);
    if (s == 5) PNP_NLM_CASE(float, 5);
    PNP_NLM_CASE(float, 7);
#undef PNP_NLM_CASE
}
