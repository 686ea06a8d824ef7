// dncnn_f16x3.hip -- OPT-IN 64->64 3x3 conv layer on the fp16 matrix cores with fp32-class accuracy.
//
// The default conv kernels (dncnn.hip) multiply in fp32 on v_mfma_f32_16x16x4_f32, whose rate is the vector rate
// (157 TFLOP/s).  The fp16-input MFMA runs 16x faster.  This kernel keeps fp32 accuracy by SPLITTING every fp32
// operand into two fp16 terms,  x = xh + 2^-11 xl  (xh = fp16(x), xl = fp16(2^11 (x - xh)); 22 significant bits),
// and evaluating  x w = xh wh + 2^-11 (xh wl + xl wh)  -- three fp16 MFMAs, dropping only the 2^-22 term -- with
// fp32 accumulation in two accumulator sets.  The 2^11 scaling keeps the low parts out of fp16's subnormals: the MFMA
// does not flush them, and at O(1) activations one accumulator with unscaled low parts measures the same error and
// is 5 % faster, but at activations of 1e-3 its error is 30x the fp32 kernels' (test_conv_kernels_against_float64).
// Relative error per product <= ~3 * 2^-22, i.e. the size of fp32's own rounding over a 576-term dot product.
// It is NOT the arithmetic of the reference (plain fp32), so it is never the default and bench.py's headline never
// uses it; `pnp_dncnn_set_winograd(plan, 3)` selects it, tests bound its deviation from the fp32 kernels.
//
// Activation format between split layers ("A16", produced by k_first in this mode and by every split layer): per image 2 parts (hi, lo) x 8 channel groups planes of [H][W]
// 16-byte records holding 8 fp16 channels -- exactly the B fragment of one lane of v_mfma_f32_16x16x32_f16
// (k = 8 (lane >> 4) + j), so a lane's operand is ONE ds_read_b128 with an immediate offset; same bytes as fp32 NCHW.
// Tile 8 x 32 pixels per workgroup, 4 waves x 16 output channels, weights stationary in AGPRs (144), two K-halves of
// 32 channels as LDS double buffer filled by LDS-DMA, direct (not Winograd: its input transform is inexact in fp16).
#include "common.h"
#include "tilewalk.h"

namespace pnp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace f16x3 {
constexpr int C = 64, TR = 8, TC = 32;
constexpr int ROWS = TR + 2, PCR = TC + 2;                     // halo tile: 10 rows x 34 records
constexpr int GH = 4;                                          // channel groups (of 8) per K-half
constexpr int RECS = 2 * GH * ROWS * PCR;                      // 2720 records per half (hi and lo parts)
constexpr int PIECES = (RECS + 63) / 64;                       // 43 wave-pieces of 64 records
constexpr int PPW = (PIECES + 3) / 4;                          // 11 per wave
constexpr int HALF_BYTES = PIECES * 64 * 16;                   // 44 032
constexpr int NW = 2 * 9 * 2;                                  // weight fragments per wave: half x tap x part
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;
}  // namespace f16x3

// One pixel block x one tap: hi += wh*xh ; lo += wh*xl ; lo += wl*xh  (hand-issued: weights come from AGPRs, the
// first tap of a tile uses the constant-zero accumulator form so the accumulators are never cleared).
__device__ __forceinline__ void mfma3(f32x4& hi, f32x4& lo, h8 wh_agpr, h8 wl_agpr, h8 xh, h8 xl) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %4, %0\n\t"
                 "v_mfma_f32_16x16x32_f16 %1, %2, %5, %1\n\t"
                 "v_mfma_f32_16x16x32_f16 %1, %3, %4, %1"
                 : "+v"(hi), "+v"(lo) : "a"(wh_agpr), "a"(wl_agpr), "v"(xh), "v"(xl));
}
__device__ __forceinline__ void mfma3_first(f32x4& hi, f32x4& lo, h8 wh_agpr, h8 wl_agpr, h8 xh, h8 xl) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %4, 0\n\t"
                 "v_mfma_f32_16x16x32_f16 %1, %2, %5, 0\n\t"
                 "v_mfma_f32_16x16x32_f16 %1, %3, %4, %1"
                 : "=&v"(hi), "=&v"(lo) : "a"(wh_agpr), "a"(wl_agpr), "v"(xh), "v"(xl));
}

// OUT_F32: write fp32 NCHW (the layer feeding k_last) instead of A16.
template <bool OUT_F32, bool LEAKY, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void k_mid_f16x3(const h8* __restrict__ in, void* __restrict__ outv,
                                                      const h8* __restrict__ wpack, const float* __restrict__ bias,
                                                      const h8* __restrict__ zeros, int H, int W, int ntiles, float slope,
                                                      unsigned long long* __restrict__ stamps) {
    using namespace f16x3;
    // diagnostic (pnp_dncnn_debug_clock): shader cycles and 100 MHz reference ticks of the whole workgroup
    const unsigned long long st_c = __builtin_amdgcn_s_memtime(), st_r = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = W / TC, tiles_per_img = tiles_x * (H / TR);
    const int HW = H * W;

    h8 wreg[NW];                                               // [half][tap][part], AGPRs (mfma_h takes them from there)
#pragma unroll
    for (int i = 0; i < NW; ++i) wreg[i] = wpack[((size_t)wv * NW + i) * 64 + lane];
    float bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = bias[16 * wv + 4 * (lane >> 4) + q];

    // DMA piece descriptors: bits 0..27 record offset inside the half's planes, bits 28..31 image-edge flags
    unsigned pdesc[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = (wv + 4 * i) * 64 + lane;
        const int part = q / (GH * ROWS * PCR), r1 = q - part * (GH * ROWS * PCR);
        const int grp = r1 / (ROWS * PCR), r2 = r1 - grp * (ROWS * PCR);
        const int row = r2 / PCR, col = r2 - row * PCR;
        const unsigned edge = (row == 0 ? 1u : 0u) | (row == ROWS - 1 ? 2u : 0u) | (col == 0 ? 4u : 0u) | (col == PCR - 1 ? 8u : 0u);
        const unsigned off = (unsigned)((part * 8 + grp) * HW + row * W + col);
        // records past the end of the half (LDS padding, never read): fetch an always-valid interior record
        pdesc[i] = q < RECS ? (off | (edge << 28)) : (unsigned)(W + 1);
    }
    // lane's B-fragment base inside a half buffer (bytes): channel group lane>>4, pixel lane&15
    const unsigned lb = (unsigned)(((lane >> 4) * ROWS * PCR + (lane & 15)) * 16);

    auto dma = [&](int i, const h8* src0, unsigned edge28, bool valid, unsigned char* buf) {
        const int pc = wv + 4 * i;
        const unsigned d = pdesc[i];
        const bool ok = (valid & (pc < PIECES)) & ((d & edge28) == 0u);
        const h8* src = ok ? src0 + (d & 0x0FFFFFFFu) : zeros;
        if (pc < PIECES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(buf + pc * 1024), 16, 0, 0);
    };
    auto tile_src = [&](int b, int ty0, int tx0, int half) {
        return in + ((size_t)b * 16 + 4 * half) * HW + (ty0 - 1) * W + (tx0 - 1);
    };
    auto tile_edge = [&](int ty0, int tx0) {
        return ((ty0 == 0 ? 1u : 0u) | (ty0 + TR == H ? 2u : 0u) | (tx0 == 0 ? 4u : 0u) | (tx0 + TC == W ? 8u : 0u)) << 28;
    };

    const TileWalk tw_ = tile_walk(ntiles);
    int tile = tw_.first;
    {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int ty0 = (t2 / tiles_x) * TR, tx0 = (t2 % tiles_x) * TC;
        const h8* s0 = tile_src(b, ty0, tx0, 0);
        const unsigned e = tile_edge(ty0, tx0);
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma(i, s0, e, tile < tw_.limit, lds);
    }
    __syncthreads();

    unsigned long long acc_compute = 0, acc_barrier = 0, acc_epi = 0, tp = 0;
    for (; tile < tw_.limit; tile += tw_.step) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int ty0 = (t2 / tiles_x) * TR, tx0 = (t2 % tiles_x) * TC;
        f32x4 ah[16], al[16];                                   // pixel block pb = 2 r + xh; first written by mfma_h_first
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned char* nbuf = lds + (half ^ 1) * HALF_BYTES;
            const int nt = tile + tw_.step;
            const int nb = half == 0 ? b : nt / tiles_per_img;
            const int n2 = nt - nb * tiles_per_img;
            const int nty0 = half == 0 ? ty0 : (n2 / tiles_x) * TR, ntx0 = half == 0 ? tx0 : (n2 % tiles_x) * TC;
            const bool nvalid = half == 0 ? true : nt < tw_.limit;
            const h8* nsrc0 = tile_src(nb, nty0, ntx0, half ^ 1);
            const unsigned nedge = tile_edge(nty0, ntx0);
            const unsigned char* cur = lds + half * HALF_BYTES + lb;
            if (STAMP) tp = __builtin_amdgcn_s_memtime();

            // fragment s = (rho, xh, dx): halo row rho, 16-pixel half xh, horizontal tap dx -> hi and lo records, as two
            // hand-issued ds_read_b128 with immediate offsets off one base register.  The reads run AHEAD fragments
            // ahead of their MFMAs and are retired by a COUNTED wait (LDS returns in order): left to hipcc every use
            // waits lgkmcnt(0), i.e. also for the reads just issued for the fragments behind it.  The 60 steps are
            // spelled out by macro so that every offset is a constant expression (an asm immediate).
            const unsigned cur_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + half * HALF_BYTES + lb;
            constexpr int NS = ROWS * 6;                        // 60 fragment pairs per half
            constexpr int AHEAD = 2;                            // LDS reads run two fragment pairs ahead of their MFMAs
            h8 fh[AHEAD + 1], fl[AHEAD + 1];
#define F16X3_LOAD(S_)                                                                                                        \
    {                                                                                                                         \
        constexpr int s_ = (S_), imm_ = (((s_ / 6) * PCR) + 16 * ((s_ / 3) % 2) + s_ % 3) * 16;                               \
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                                        \
                     : "=&v"(fh[s_ % (AHEAD + 1)]), "=&v"(fl[s_ % (AHEAD + 1)])                                               \
                     : "v"(cur_addr), "n"(imm_), "n"(imm_ + GH * ROWS * PCR * 16) : "memory");                                \
    }
#define F16X3_STEP(S_)                                                                                                        \
    {                                                                                                                         \
        constexpr int s = (S_);                                                                                               \
        if constexpr (s + AHEAD < NS) F16X3_LOAD(s + AHEAD)                                                                   \
        {   /* fragment s has landed once at most the reads of the fragments issued after it are outstanding */               \
            constexpr int behind = (NS - 1 - s) < AHEAD ? (NS - 1 - s) : AHEAD;                                               \
            if constexpr (behind == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fh[s % 3]), "+v"(fl[s % 3]) :: "memory");   \
            else if constexpr (behind == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[s % 3]), "+v"(fl[s % 3]) :: "memory"); \
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[s % 3]), "+v"(fl[s % 3]) :: "memory");                         \
        }                                                                                                                     \
        /* the next half's DMA goes out early: a half lasts only ~7k cycles, a late piece would be waited for */            \
        if constexpr (s % 2 == 0 && s / 2 < PPW) dma(s / 2, nsrc0, nedge, nvalid, nbuf);                                      \
        constexpr int rho = s / 6, xh = (s / 3) % 2, dx = s % 3;                                                              \
        _Pragma("unroll") for (int dy = 0; dy < 3; ++dy) {                                                                    \
            const int r = rho - dy;                                                                                           \
            if (r >= 0 && r < TR) {                                                                                           \
                const int pb = 2 * r + xh, tap = dy * 3 + dx;                                                                 \
                const h8 wh = wreg[(half * 9 + tap) * 2 + 0], wl = wreg[(half * 9 + tap) * 2 + 1];                            \
                if (half == 0 && tap == 0) mfma3_first(ah[pb], al[pb], wh, wl, fh[s % 3], fl[s % 3]);                         \
                else mfma3(ah[pb], al[pb], wh, wl, fh[s % 3], fl[s % 3]);                                                     \
            }                                                                                                                 \
        }                                                                                                                     \
    }
            static_assert(AHEAD == 2 && NS == 60, "the step list below is written for 60 steps, two fragments ahead");
            F16X3_LOAD(0) F16X3_LOAD(1)
            F16X3_STEP(0) F16X3_STEP(1) F16X3_STEP(2) F16X3_STEP(3) F16X3_STEP(4) F16X3_STEP(5) 
            F16X3_STEP(6) F16X3_STEP(7) F16X3_STEP(8) F16X3_STEP(9) F16X3_STEP(10) F16X3_STEP(11) 
            F16X3_STEP(12) F16X3_STEP(13) F16X3_STEP(14) F16X3_STEP(15) F16X3_STEP(16) F16X3_STEP(17) 
            F16X3_STEP(18) F16X3_STEP(19) F16X3_STEP(20) F16X3_STEP(21) F16X3_STEP(22) F16X3_STEP(23) 
            F16X3_STEP(24) F16X3_STEP(25) F16X3_STEP(26) F16X3_STEP(27) F16X3_STEP(28) F16X3_STEP(29) 
            F16X3_STEP(30) F16X3_STEP(31) F16X3_STEP(32) F16X3_STEP(33) F16X3_STEP(34) F16X3_STEP(35) 
            F16X3_STEP(36) F16X3_STEP(37) F16X3_STEP(38) F16X3_STEP(39) F16X3_STEP(40) F16X3_STEP(41) 
            F16X3_STEP(42) F16X3_STEP(43) F16X3_STEP(44) F16X3_STEP(45) F16X3_STEP(46) F16X3_STEP(47) 
            F16X3_STEP(48) F16X3_STEP(49) F16X3_STEP(50) F16X3_STEP(51) F16X3_STEP(52) F16X3_STEP(53) 
            F16X3_STEP(54) F16X3_STEP(55) F16X3_STEP(56) F16X3_STEP(57) F16X3_STEP(58) F16X3_STEP(59) 
#undef F16X3_STEP
#undef F16X3_LOAD
            if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_compute += t - tp; tp = t; }
            __syncthreads();                                    // next half landed (vmcnt(0)) + everyone done reading
            if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_barrier += t - tp; tp = t; }
        }

        // epilogue: v = hi + 2^-11 lo + bias, activation, then split again (A16) or plain fp32 (layer before k_last)
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");       // MFMA write -> VALU read distance (hand-issued MFMAs)
#pragma unroll
        for (int pb = 0; pb < 16; ++pb) {
            const int y = ty0 + pb / 2, x = tx0 + 16 * (pb % 2) + (lane & 15);
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = (ah[pb][q] + al[pb][q] * LO_INV) + bv[q];
                t = t > 0.f ? t : (LEAKY ? slope * t : 0.f);
                v[q] = t;
            }
            if (OUT_F32) {
                float* o = (float*)outv + ((size_t)b * C + 16 * wv + 4 * (lane >> 4)) * HW + (size_t)y * W + x;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[(size_t)q * HW] = v[q];
            } else {
                h4 hi, lo;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const _Float16 h = (_Float16)v[q];
                    hi[q] = h;
                    lo[q] = (_Float16)((v[q] - (float)h) * LO_SCALE);
                }
                // couts 16 wv + 4 (lane>>4) + q -> group 2 wv + (lane >> 5), half-record (lane >> 4) & 1
                h4* o = (h4*)outv;
                const size_t rec = ((size_t)(b * 2) * 8 + 2 * wv + (lane >> 5)) * HW + (size_t)y * W + x;
                o[rec * 2 + ((lane >> 4) & 1)] = hi;
                o[(rec + (size_t)8 * HW) * 2 + ((lane >> 4) & 1)] = lo;
            }
        }
        if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_epi += t - tp; }
    }
    if (stamps != nullptr && tid == 0) {
        stamps[5 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c;
        stamps[5 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r;
        stamps[5 * blockIdx.x + 2] = acc_compute;               // STAMP builds only: MFMA phases / barrier waits / epilogues
        stamps[5 * blockIdx.x + 3] = acc_barrier;
        stamps[5 * blockIdx.x + 4] = acc_epi;
    }
}

}  // namespace pnp

// ---- host side: called from dncnn.hip's plan code (declarations in f16x3.h)
// wpack16 [n_mid][4 waves][NW][64 lanes] h8: A fragment of lane l = w[cout 16 wv + (l&15)][cin 32 half + 8 (l>>4) + j][tap]
namespace pnp {
void f16x3_pack_weights(const float* w_mid, int n_mid, void* out_h8) {
    using namespace f16x3;
    h8* out = (h8*)out_h8;
    for (int l = 0; l < n_mid; ++l)
        for (int wv = 0; wv < 4; ++wv)
            for (int half = 0; half < 2; ++half)
                for (int tap = 0; tap < 9; ++tap)
                    for (int lane = 0; lane < 64; ++lane) {
                        h8 hi, lo;
                        for (int j = 0; j < 8; ++j) {
                            const int cout = 16 * wv + (lane & 15), cin = 32 * half + 8 * (lane >> 4) + j;
                            const float v = w_mid[(((size_t)l * C + cout) * C + cin) * 9 + tap];
                            const _Float16 h = (_Float16)v;
                            hi[j] = h;
                            lo[j] = (_Float16)((v - (float)h) * LO_SCALE);
                        }
                        const size_t base = (((size_t)l * 4 + wv) * NW + (half * 9 + tap) * 2) * 64 + lane;
                        out[base] = hi;
                        out[base + 64] = lo;
                    }
}

size_t f16x3_weight_bytes(int n_mid) { return (size_t)n_mid * 4 * f16x3::NW * 64 * sizeof(h8); }

// One 64->64 layer.  in_a16 / out: A16 buffers (out fp32 NCHW when out_f32).  zeros: >= 16 zero bytes.
int f16x3_layer(const void* in_a16, void* out, const void* wpack_layer, const float* bias, const void* zeros,
                int H, int W, int batch, int num_cu, int out_f32, float slope, hipStream_t s, unsigned long long* stamps) {
    using namespace f16x3;
    const int ntiles = batch * (H / TR) * (W / TC);
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    const size_t ldsb = 2 * (size_t)HALF_BYTES;
    static bool attr = false;
    if (!attr) {
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_mid_f16x3<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_mid_f16x3<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_mid_f16x3<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_mid_f16x3<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_mid_f16x3<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        attr = true;
    }
    const h8* in = (const h8*)in_a16;
    const h8* wp = (const h8*)wpack_layer;
    const h8* z = (const h8*)zeros;
    if (slope != 0.f) {
        if (out_f32) k_mid_f16x3<true, true><<<grid, 256, ldsb, s>>>(in, out, wp, bias, z, H, W, ntiles, slope, stamps);
        else k_mid_f16x3<false, true><<<grid, 256, ldsb, s>>>(in, out, wp, bias, z, H, W, ntiles, slope, stamps);
    } else {
        if (out_f32) k_mid_f16x3<true, false><<<grid, 256, ldsb, s>>>(in, out, wp, bias, z, H, W, ntiles, 0.f, stamps);
        else if (stamps != nullptr) k_mid_f16x3<false, false, true><<<grid, 256, ldsb, s>>>(in, out, wp, bias, z, H, W, ntiles, 0.f, stamps);
        else k_mid_f16x3<false, false><<<grid, 256, ldsb, s>>>(in, out, wp, bias, z, H, W, ntiles, 0.f, stamps);
    }
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

}  // namespace pnp
