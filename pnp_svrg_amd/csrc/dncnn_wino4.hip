// dncnn_wino4.hip -- the 64 -> 64 channel 3x3 layer of the DnCNN prox (reference denoisers/DeepDenoisers/model/models.py:
// 13-17: conv + BatchNorm + ReLU, BN folded by the caller) with the three horizontal taps through the 1-D Winograd
// minimal-filtering transform F(4,3): per output QUAD (x .. x+3) and input row
//     V = B^T d  (d = inputs x-1 .. x+4, 6 values)        U = G g  (g = the 3 horizontal weights, 6 values per (cout, cin, dy))
//     m_xi = sum_{dy,cin} U_xi V_xi  (6 x 3 x 64 multiply-adds per quad)        y = A^T m  (4 outputs)
// i.e. HALF the matrix-core work of the direct form (F(2,3) of k_mid_wino: two thirds) for the same exact-arithmetic
// result; fp32 throughout.  Measured through the whole 17-layer network (tests/test_gpu_dncnn.py, NumPy model in DESIGN
// 3.1): 1.6e-7 from a float64 evaluation, against 1.0e-7 for the direct fp32 form and 1.1e-7 for F(2,3).
//
// Same machine organisation as k_mid_wino (dncnn.hip): one workgroup = 4 waves = one per SIMD, persistent over output
// tiles in the XCD-aware order; wave wv owns output channels [16 wv, 16 wv + 16) and keeps its transformed weights for
// the whole launch -- 6 xi x 3 dy x 16 (channel quad, K-half) = 288 values per lane: 256 in AGPRs (the whole file), 32 in
// VGPRs (an MFMA takes SrcA from either) -- accumulators (4 rows x 6 xi x 4) in VGPRs, first MFMA of an accumulator in
// the constant-zero SrcC form; activations of one K-half (32 channels) as LDS planes [6 halo rows][72 columns] staged by
// LDS-DMA (16-byte pieces), the two K-halves double-buffered.  An M-tile is the 16 quads of a 64-pixel output row, so the
// output tile is 4 rows x 64 columns and a lane's four results of a channel are ONE 16-byte store.
// Per group (channel quad, 3 halo rows): 9 ds_read2_b32, 36 VALU (B^T d of three rows, 12 each), one DMA piece -- all
// in ONE block in front of the group's 36 MFMAs (every excursion from the MFMA stream to the vector ALU costs ~13 cycles
// on top of ~4 per instruction: tools/microbench/mfma_f32_fillers.hip).
#include "common.h"
#include "wino4.h"
#include "tilewalk.h"
#include <vector>

namespace pnp {
namespace w4 {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int C = 64;
constexpr int TR = 4, TC = 64;                        // output tile
constexpr int PR = TR + 2;                            // halo rows
constexpr int PC = 72;                                // LDS row: image columns [tx0 - 4, tx0 + 68) = eighteen 16-byte chunks
constexpr int XOFF = 3;                               // LDS column of image column tx0 - 1
constexpr int PLANE = PR * PC;                        // 432 floats; 432 % 32 == 16: the k-rows of a B operand alternate bank halves
constexpr int HALF_C = 32;
constexpr int CHUNKS = HALF_C * PLANE / 4;            // 3456 16-byte chunks per K-half
constexpr int PIECES = CHUNKS / 64;                   // 54 wave pieces
constexpr int PPW = 14;                               // pieces per wave (56 slots; 54, 55 move zeros)
constexpr int HALF_LDS = 4 * PPW * 256;               // 14336 floats = 56 KB per buffer
constexpr int NU = 2 * (HALF_C / 4) * 3 * 6;          // 288 transformed weights per lane
constexpr int NU_AGPR = 256;
static_assert(PIECES <= 4 * PPW && PLANE % 32 == 16, "geometry");

// ureg index of U_xi[dy] for (K-half, channel quad)
__host__ __device__ constexpr int uidx(int half, int c4, int dy, int xi) { return ((half * (HALF_C / 4) + c4) * 3 + dy) * 6 + xi; }

// 3 halo rows x 6 inputs (d0..d5 at LDS columns base + 0..5, rows 72 dwords apart) as nine ds_read2_b32 off one base
__device__ __forceinline__ void lds_load(f32x2v (&dd)[3][3], unsigned lds_byte_addr) {
    f32x2v r0, r1, r2, r3, r4, r5, r6, r7, r8;
    asm volatile(
        "ds_read2_b32 %0, %9 offset0:0 offset1:1\n"
        "ds_read2_b32 %1, %9 offset0:2 offset1:3\n"
        "ds_read2_b32 %2, %9 offset0:4 offset1:5\n"
        "ds_read2_b32 %3, %9 offset0:72 offset1:73\n"
        "ds_read2_b32 %4, %9 offset0:74 offset1:75\n"
        "ds_read2_b32 %5, %9 offset0:76 offset1:77\n"
        "ds_read2_b32 %6, %9 offset0:144 offset1:145\n"
        "ds_read2_b32 %7, %9 offset0:146 offset1:147\n"
        "ds_read2_b32 %8, %9 offset0:148 offset1:149\n"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8)
        : "v"(lds_byte_addr)
        : "memory");
    dd[0][0] = r0; dd[0][1] = r1; dd[0][2] = r2; dd[1][0] = r3; dd[1][1] = r4; dd[1][2] = r5;
    dd[2][0] = r6; dd[2][1] = r7; dd[2][2] = r8;
}
// the wait takes the nine register pairs as in/out operands: every consumer is data-dependent on it
__device__ __forceinline__ void lds_wait(f32x2v (&dd)[3][3]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(dd[0][0]), "+v"(dd[0][1]), "+v"(dd[0][2]), "+v"(dd[1][0]), "+v"(dd[1][1]), "+v"(dd[1][2]),
                   "+v"(dd[2][0]), "+v"(dd[2][1]), "+v"(dd[2][2])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
static_assert(PC == 72, "lds_load hard-codes the 72-dword LDS row stride");

// hand-issued MFMAs (see k_mid_wino): accumulators in VGPRs, weights in AGPRs ("a") or VGPRs ("v")
template <bool AG> __device__ __forceinline__ void mfma_w(f32x4& acc, float w, float v) {
    if (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(v));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(v));
}
template <bool AG> __device__ __forceinline__ void mfma_w_first(f32x4& acc, float w, float v) {
    if (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(v));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(v));
}

// prologue staging of one K-half (branchy form; the steady state uses the per-lane piece descriptors)
__device__ __forceinline__ void dma_half(const float* __restrict__ in, const float* __restrict__ zeros, float* ldsbuf, int H, int W,
                                         int b, int ty0, int tx0, int half, int tid, bool valid_tile) {
    const int wv = tid >> 6, lane = tid & 63;
#pragma unroll 1
    for (int pc = wv; pc < 4 * PPW; pc += 4) {
        const int q = pc * 64 + lane;
        const int cin = q / (PR * 18), r = q - cin * (PR * 18);
        const int ry = r / 18, cx = r - ry * 18;
        const int y = ty0 - 1 + ry, x = tx0 - 4 + 4 * cx;
        const float* src = zeros;
        if (valid_tile && pc < PIECES && y >= 0 && y < H && x >= 0 && x < W)
            src = in + (((size_t)b * C + half * HALF_C + cin) * H + y) * W + x;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ldsbuf + pc * 256), 16, 0, 0);
    }
}

template <bool LEAKY>
__global__ __launch_bounds__(256, 1) void k_mid_wino4(const float* __restrict__ in, float* __restrict__ out,
                                                      const float* __restrict__ upack, const float* __restrict__ bias,
                                                      const float* __restrict__ zeros, int H, int W, int ntiles, float slope) {
    __shared__ float lds[2 * HALF_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = W / TC, tiles_per_img = tiles_x * (H / TR);

    // ureg[uidx(half, c4, dy, xi)] = U_xi[cout = 16wv + (lane&15)][cin = 32half + 4c4 + (lane>>4)][dy]
    float ureg[NU];
#pragma unroll
    for (int s = 0; s < NU; ++s) ureg[s] = upack[((size_t)wv * NU + s) * 64 + lane];
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[16 * wv + 4 * (lane >> 4) + r];

    // lane (k-row kq = lane>>4, quad j = lane&15) reads d0..d5 at LDS columns XOFF + 4j + {0..5}
    const int lbase = (lane >> 4) * PLANE + 4 * (lane & 15) + XOFF;
    int loff[4];                                               // output offsets: quad 4j of channel 16wv + 4kq + r
#pragma unroll
    for (int r = 0; r < 4; ++r) loff[r] = (16 * wv + 4 * (lane >> 4) + r) * H * W + 4 * (lane & 15);

    // DMA piece descriptors, one register each: bits 0..27 = element offset of the lane's 16-byte chunk inside the half's
    // 32 channel planes, bits 28..31 = which image edge would put the chunk outside (top / bottom halo row, left / right chunk)
    unsigned pdesc[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = (wv + 4 * i) * 64 + lane;
        const int cin = q / (PR * 18), r = q - cin * (PR * 18);
        const int ry = r / 18, cx4 = 4 * (r - ry * 18);
        const unsigned edge = (ry == 0 ? 1u : 0u) | (ry == TR + 1 ? 2u : 0u) | (cx4 == 0 ? 4u : 0u) | (cx4 == TC + 4 ? 8u : 0u);
        pdesc[i] = (unsigned)((cin * H + ry) * W + cx4) | (edge << 28);
    }

    const TileWalk tw_ = tile_walk(ntiles);
    int tile = tw_.first;
    {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        dma_half(in, zeros, lds, H, W, b, (t2 / tiles_x) * TR, (t2 % tiles_x) * TC, 0, tid, tile < tw_.limit);
    }
    __syncthreads();

    for (; tile < tw_.limit; tile += tw_.step) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int ty0 = (t2 / tiles_x) * TR, tx0 = (t2 % tiles_x) * TC;
        f32x4 acc[TR][6];                                       // written first by mfma_w_first (half 0, channel quad 0, dy 0)

#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float* nbuf = lds + (half ^ 1) * HALF_LDS;
            const int nt = tile + tw_.step;
            const int nb = half == 0 ? b : nt / tiles_per_img;
            const int n2 = nt - nb * tiles_per_img;
            const int nty0 = half == 0 ? ty0 : (n2 / tiles_x) * TR, ntx0 = half == 0 ? tx0 : (n2 % tiles_x) * TC;
            const bool nvalid = half == 0 ? true : nt < tw_.limit;
            const float* nsrc0 = in + (((size_t)nb * C + (half ^ 1) * HALF_C) * H + nty0 - 1) * (size_t)W + ntx0 - 4;
            const unsigned nedge = ((nty0 == 0 ? 1u : 0u) | (nty0 + TR == H ? 2u : 0u) | (ntx0 == 0 ? 4u : 0u) | (ntx0 + TC == W ? 8u : 0u)) << 28;

            int xb_off = half * HALF_LDS + lbase;
            asm volatile("" : "+v"(xb_off));

            // group = (channel quad c4, block rb of 3 halo rows)
            constexpr int NG = (HALF_C / 4) * 2;               // 16 groups per half
            f32x2v d[2][3][3];
            const unsigned xb_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + 4u * (unsigned)xb_off;
            lds_load(d[0], xb_addr);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int c4 = g / 2, rb = g % 2;
                lds_wait(d[g & 1]);                             // issued a whole group ago (or just above for g = 0)
                if (g + 1 < NG) {
                    const int c4n = (g + 1) / 2, rbn = (g + 1) % 2;
                    lds_load(d[(g + 1) & 1], xb_addr + 4u * ((4 * c4n) * PLANE + (3 * rbn) * PC));
                }
                if (g < PPW) {
                    const int pc = wv + 4 * g;
                    const bool ok = (nvalid & (pc < PIECES)) & ((pdesc[g] & nedge) == 0u);
                    const float* src = ok ? nsrc0 + (pdesc[g] & 0x0FFFFFFFu) : zeros;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(nbuf + pc * 256), 16, 0, 0);
                }
                // B^T d of the group's three halo rows (F(4,3): 12 VALU per row)
                float V[3][6];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float d0 = d[g & 1][i][0].x, d1 = d[g & 1][i][0].y, d2 = d[g & 1][i][1].x, d3 = d[g & 1][i][1].y,
                                d4 = d[g & 1][i][2].x, d5 = d[g & 1][i][2].y;
                    const float t1 = __builtin_fmaf(-4.f, d2, d4), t2 = __builtin_fmaf(-4.f, d1, d3);
                    const float t3 = d4 - d2, sd = d3 - d1;
                    V[i][0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
                    V[i][1] = t1 + t2;
                    V[i][2] = t1 - t2;
                    V[i][3] = __builtin_fmaf(2.f, sd, t3);
                    V[i][4] = __builtin_fmaf(-2.f, sd, t3);
                    V[i][5] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
                }
                // pin the transform results before the hand-issued MFMAs read them (the hazard recognizer does not see
                // inside inline asm: VALU write -> MFMA read needs wait states)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    asm volatile("" : "+v"(V[i][0]), "+v"(V[i][1]), "+v"(V[i][2]), "+v"(V[i][3]), "+v"(V[i][4]), "+v"(V[i][5]));
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 1" ::: "memory");
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int ry = 3 * rb + i;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int r = ry - dy;
                        if (r >= 0 && r < TR) {
#pragma unroll
                            for (int xi = 0; xi < 6; ++xi) {
                                const int s = uidx(half, c4, dy, xi);
                                const bool first = half == 0 && c4 == 0 && dy == 0;
                                if (s < NU_AGPR) {
                                    if (first) mfma_w_first<true>(acc[r][xi], ureg[s], V[i][xi]);
                                    else mfma_w<true>(acc[r][xi], ureg[s], V[i][xi]);
                                } else {
                                    if (first) mfma_w_first<false>(acc[r][xi], ureg[s], V[i][xi]);
                                    else mfma_w<false>(acc[r][xi], ureg[s], V[i][xi]);
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }

        // epilogue: inverse transform y = A^T m, bias, ReLU; a lane holds pixel quad 4j..4j+3 of 4 channels per row
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // MFMA write -> VALU read distance
        float* ob = out + (size_t)b * C * H * W + ty0 * W + tx0;
#pragma unroll
        for (int r = 0; r < TR; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float m0 = acc[r][0][q], m1 = acc[r][1][q], m2 = acc[r][2][q], m3 = acc[r][3][q], m4 = acc[r][4][q],
                            m5 = acc[r][5][q];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                float4 v;
                v.x = (m0 + s12) + s34 + bv[q];
                v.y = __builtin_fmaf(2.f, d34, d12) + bv[q];
                v.z = __builtin_fmaf(4.f, s34, s12) + bv[q];
                v.w = (__builtin_fmaf(8.f, d34, d12) + m5) + bv[q];
                v.x = v.x > 0.f ? v.x : (LEAKY ? slope * v.x : 0.f);
                v.y = v.y > 0.f ? v.y : (LEAKY ? slope * v.y : 0.f);
                v.z = v.z > 0.f ? v.z : (LEAKY ? slope * v.z : 0.f);
                v.w = v.w > 0.f ? v.w : (LEAKY ? slope * v.w : 0.f);
                *reinterpret_cast<float4*>(ob + loff[q] + r * W) = v;
            }
        }
    }
}

}  // namespace w4

bool wino4_supports(int H, int W) { return H % w4::TR == 0 && W % w4::TC == 0; }

size_t wino4_weight_floats(int n_mid) { return (size_t)n_mid * 4 * w4::NU * 64; }

// w_mid [n_mid][64][64][3][3] (BN folded) -> upack[l][wv][s][lane], s = uidx(half, c4, dy, xi)
void wino4_pack_weights(const float* w_mid, int n_mid, float* out) {
    static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (int l = 0; l < n_mid; ++l)
        for (int wv = 0; wv < 4; ++wv)
            for (int half = 0; half < 2; ++half)
                for (int c4 = 0; c4 < w4::HALF_C / 4; ++c4)
                    for (int dy = 0; dy < 3; ++dy)
                        for (int xi = 0; xi < 6; ++xi)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int cout = 16 * wv + (lane & 15), cin = w4::HALF_C * half + 4 * c4 + (lane >> 4);
                                const float* g = w_mid + (((size_t)l * w4::C + cout) * w4::C + cin) * 9 + dy * 3;
                                const double u = G[xi][0] * g[0] + G[xi][1] * g[1] + G[xi][2] * g[2];
                                out[(((size_t)l * 4 + wv) * w4::NU + w4::uidx(half, c4, dy, xi)) * 64 + lane] = (float)u;
                            }
}

int wino4_layer(const float* in, float* out, const float* upack_layer, const float* bias, const float* zeros, int H, int W,
                int batch, int num_cu, float slope, hipStream_t s) {
    const int ntiles = batch * (H / w4::TR) * (W / w4::TC);
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    if (slope != 0.f) w4::k_mid_wino4<true><<<grid, 256, 0, s>>>(in, out, upack_layer, bias, zeros, H, W, ntiles, slope);
    else w4::k_mid_wino4<false><<<grid, 256, 0, s>>>(in, out, upack_layer, bias, zeros, H, W, ntiles, 0.f);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

}  // namespace pnp
