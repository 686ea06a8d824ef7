// dncnn.hip -- DnCNN-17 prox (reference denoisers/RealSN_DnCNN.py:16-42 around the network of
// denoisers/DeepDenoisers/model/models.py:5-22 / realSN_models.py:4-21) on gfx950.
//
//   k_first   1 -> 64 channels, 3x3, ReLU; fused with the min-max normalisation and the
//             "1 + sigma/255/2" range scaling of the wrapper (RealSN_DnCNN.py:19-29)
//   k_mid     64 -> 64 channels, 3x3, BatchNorm folded into weights + bias, ReLU:
//             implicit GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, the
//             reference network's own precision).  x15 per forward = 99.8 % of the FLOPs.
//   k_last    64 -> 1 channel, 3x3; fused with x = xtilde - r, the inverse scaling and the squared
//             error of Problem.PSNR (RealSN_DnCNN.py:36-40, problems/problem.py:33-35)
//
// k_mid design (one workgroup = 4 waves = one wave per SIMD, persistent over output tiles):
//   * output tile: 8 rows x 32 columns x 64 channels.  D[cout][pixel] = sum_k W[cout][k] X[k][pixel]
//     with K = 9 taps x 64 channels = 576: A operand = weights (rows = cout), B operand =
//     activations (columns = 32 adjacent pixels of one row) so that every accumulator register
//     holds 32 adjacent pixels of one channel -> 128-byte coalesced stores into NCHW.
//   * WEIGHT-STATIONARY: wave (nh, mg) owns couts [32nh, 32nh+32) and rows [4mg, 4mg+4).  Its
//     576 x 32 weight slice sits in 288 VGPRs for the life of the kernel (the unified 512-entry
//     register file of gfx950 makes that possible at one wave per SIMD) -- weights are fetched from
//     L2 once per launch, never per tile, and never go through LDS.
//   * activations: the (8+2) x (32+2) halo tile of 32 input channels (one K-half) lives in LDS as
//     channel planes [cin][10][34] -> the B operand of an MFMA is ONE conflict-free ds_read_b32
//     (32 adjacent pixels per half-wave).  Two K-halves = two LDS buffers = a natural double buffer:
//     while the MFMAs chew on one half, the next half (or the next tile's first half) is in flight
//     global -> VGPR, and is written to the other buffer at the end of the phase.
//   * per wave and phase: 576 MFMAs (64 cycles each) vs 576 ds_read_b32 + 43 global loads: the
//     matrix pipe is the only busy resource by a wide margin.
#include "common.h"
#include "wino44.h"
#include "wino44b.h"
#include "tilewalk.h"
#include "reduce.h"
#include <vector>
#include <algorithm>
#include <cstdlib>

namespace pnp {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int C = 64;             // feature channels
constexpr int TR = 8, TC = 32;    // output tile rows / cols
constexpr int PR = TR + 2;        // halo rows
constexpr int PC = 40;            // halo row stride in LDS: columns [tx0-4, tx0+36) = ten 16-byte chunks,
                                  // so every global->LDS piece is an aligned dwordx4 (the 3x3 halo needs
                                  // only tx0-1 .. tx0+32; the 6 extra floats buy 4x fewer DMA instructions)
constexpr int XOFF = 3;           // LDS column of image column tx0-1
constexpr int PLANE = PR * PC;    // 400 floats per channel; 400 % 32 == 16 -> the 4 k-rows of a B operand
                                  // (lanes 0-15 / 16-31 of a half-wave) fall on disjoint LDS banks
constexpr int HALF_C = 32;        // channels per K-half
constexpr int HALF_PAYLOAD = HALF_C * PLANE;          // 12800 floats = 50 KB of halo tile per K-half
constexpr int HALF_LDS = 52 * 256;                    // LDS buffer rounded up to 13 DMA pieces per wave (52 KB):
                                                      // every wave issues the same, branch-free piece sequence
constexpr int CHUNKS = HALF_PAYLOAD / 4;              // 3200 16-byte chunks = 50 wave-pieces per half
constexpr int KSTEPS_HALF = 9 * (HALF_C / 4);         // 72 MFMA K-steps (K=4 each) per half
constexpr int MT = 16;                                // 16-pixel M-tiles per output tile (8 rows x 2)

// Stage one K-half of an input halo tile global (NCHW) -> LDS by LDS-DMA (global_load_lds_dwordx4):
// no staging registers, no ds_write.  The LDS image [cin][10][40] is linear in the chunk index
// q = cin*100 + row*10 + chunk, so one wave-instruction (64 lanes x 16 B) fills 64 consecutive
// chunks; the SOURCE address is per lane; chunks outside the image read a zero line instead.
__device__ __forceinline__ void dma_piece(int pc, const float* __restrict__ in, const float* __restrict__ zeros,
                                          float* ldsbuf, int H, int W, int b, int ty0, int tx0, int half, int lane,
                                          bool valid_tile) {
    const int q = pc * 64 + lane;
    const int cin = q / 100, r = q - cin * 100;
    const int ry = r / 10, cx = r - ry * 10;
    const int y = ty0 - 1 + ry, x = tx0 - 4 + 4 * cx;
    const float* src = zeros;
    if (valid_tile && y >= 0 && y < H && x >= 0 && x < W)
        src = in + (((size_t)b * C + half * HALF_C + cin) * H + y) * W + x;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(ldsbuf + pc * 256), 16, 0, 0);
}

constexpr int PIECES = CHUNKS / 64;                   // 50 wave-pieces per half
constexpr int PIECES_PER_WAVE = (PIECES + 3) / 4;     // 13 (the last one only for waves 0,1)

__device__ __forceinline__ void dma_half(const float* __restrict__ in, const float* __restrict__ zeros,
                                         float* ldsbuf, int H, int W, int b, int ty0, int tx0, int half, int tid,
                                         bool valid_tile) {
    const int wv = tid >> 6, lane = tid & 63;
#pragma unroll 1
    for (int pc = wv; pc < PIECES; pc += 4) dma_piece(pc, in, zeros, ldsbuf, H, W, b, ty0, tx0, half, lane, valid_tile);
}

// STAMP / ABL: diagnostic builds only (pnp_dncnn_debug_clock): s_memtime / s_memrealtime around the tile
// loop and its phases; ABL bit 0 replaces the LDS reads by register values, bit 1 drops the DMA.
// LEAKY: the activation is LeakyReLU(slope) instead of ReLU (the MMO network, reference denoisers/MMODenoise.py:84).
template <bool RELU, bool STAMP = false, int ABL = 0, bool LEAKY = false>
__global__ __launch_bounds__(256, 1) void k_mid(const float* __restrict__ in, float* __restrict__ out,
                                                const float* __restrict__ wpack, const float* __restrict__ bias,
                                                const float* __restrict__ zeros, int H, int W, int ntiles,
                                                unsigned long long* __restrict__ stamps = nullptr, float slope = 0.f) {
    __shared__ float lds[2 * HALF_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = W / TC, tiles_per_img = tiles_x * (H / TR);

    // weight slice of this wave (16 couts x 576): wreg[s] = W'[16wv + (lane&15)][k = 4s + (lane>>4)]
    float wreg[2 * KSTEPS_HALF];
#pragma unroll
    for (int s = 0; s < 2 * KSTEPS_HALF; ++s) wreg[s] = wpack[((size_t)wv * 2 * KSTEPS_HALF + s) * 64 + lane];
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[16 * wv + 4 * (lane >> 4) + r];

    const int lbase = (lane >> 4) * PLANE + (lane & 15) + XOFF;     // lane part of the B-operand LDS address
    int loff[4];                                               // lane part of the output offsets (per register)
#pragma unroll
    for (int r = 0; r < 4; ++r) loff[r] = (16 * wv + 4 * (lane >> 4) + r) * H * W + (lane & 15);

    // Per-lane descriptors of this wave's DMA pieces (tile independent): element offset of the chunk relative
    // to the tile origin ((b*64 + half*32)*H + ty0 - 1)*W + tx0 - 4, and its (halo row, 4*chunk column).
    int poff[PIECES_PER_WAVE], pry[PIECES_PER_WAVE], pcx4[PIECES_PER_WAVE];
#pragma unroll
    for (int i = 0; i < PIECES_PER_WAVE; ++i) {
        const int q = (wv + 4 * i) * 64 + lane;
        const int cin = q / 100, r = q - cin * 100;
        pry[i] = r / 10;
        pcx4[i] = 4 * (r - pry[i] * 10);
        poff[i] = (cin * H + pry[i]) * W + pcx4[i];
    }

    const TileWalk tw_ = tile_walk(ntiles);
    int tile = tw_.first;
    {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        dma_half(in, zeros, lds, H, W, b, (t2 / tiles_x) * TR, (t2 % tiles_x) * TC, 0, tid, tile < tw_.limit);
    }
    __syncthreads();                                        // (drains the DMA: vmcnt(0) + barrier)
    unsigned long long t0 = 0, r0 = 0, acc_compute = 0, acc_barrier = 0, acc_epi = 0, tp = 0;
    if (STAMP) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    for (; tile < tw_.limit; tile += tw_.step) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int ty0 = (t2 / tiles_x) * TR, tx0 = (t2 % tiles_x) * TC;
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // prefetch target: the other K-half of this tile, or the first K-half of the next tile
            float* nbuf = lds + (half ^ 1) * HALF_LDS;
            const int nt = tile + tw_.step;
            const int nb = half == 0 ? b : nt / tiles_per_img;
            const int n2 = nt - nb * tiles_per_img;
            const int nty0 = half == 0 ? ty0 : (n2 / tiles_x) * TR, ntx0 = half == 0 ? tx0 : (n2 % tiles_x) * TC;
            const bool nvalid = half == 0 ? true : nt < tw_.limit;
            const float* nsrc0 = in + (((size_t)nb * C + (half ^ 1) * HALF_C) * H + nty0 - 1) * (size_t)W + ntx0 - 4;

            // keep the buffer base in a register of its own: every B-operand address is then base +
            // a 16-bit immediate (< 48 KB) instead of one v_add per LDS read
            int xb_off = half * HALF_LDS + lbase;
            asm volatile("" : "+v"(xb_off));
            const float* xb = lds + xb_off;
            if (STAMP) tp = __builtin_amdgcn_s_memtime();
            // B operands of one (channel quad c4, dx) group: the 10 halo rows x 2 column halves; each
            // value feeds up to three taps (dy): 20 LDS reads per 48 MFMAs.  Software pipeline: the reads
            // of group g+1 and one DMA piece of the next buffer are issued underneath the MFMAs of group g.
            float xr[2][PR][2];
            constexpr int NG = (HALF_C / 4) * 3;           // 24 groups per half
#pragma unroll
            for (int ry = 0; ry < PR; ++ry)
#pragma unroll
                for (int h = 0; h < 2; ++h) xr[0][ry][h] = (ABL & 1) ? (float)(lane + ry + h) : xb[ry * PC + 16 * h];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int c4 = g / 3, dx = g % 3;
                if (g + 1 < NG) {
                    const int c4n = (g + 1) / 3, dxn = (g + 1) % 3;
#pragma unroll
                    for (int ry = 0; ry < PR; ++ry)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            xr[(g + 1) & 1][ry][h] = (ABL & 1) ? xr[g & 1][ry][h] : xb[(4 * c4n) * PLANE + ry * PC + 16 * h + dxn];
                }
                if (g < PIECES_PER_WAVE && !(ABL & 2)) {
                    // branch-free and identical in every wave (a branch here would split the MFMA scheduling
                    // region): pieces 50/51 and the pieces of a non-existent next tile just move zeros
                    const int pc = wv + 4 * g;
                    const int y = nty0 - 1 + pry[g], x = ntx0 - 4 + pcx4[g];
                    const bool ok = nvalid & (pc < PIECES) & ((unsigned)y < (unsigned)H) & ((unsigned)x < (unsigned)W);
                    const float* src = ok ? nsrc0 + poff[g] : zeros;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(nbuf + pc * 256), 16, 0, 0);
                }
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int s = half * KSTEPS_HALF + (dy * 3 + dx) * (HALF_C / 4) + c4;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s], xr[g & 1][(m >> 1) + dy][m & 1], acc[m], 0, 0, 0);
                }
                // interleave: 2 MFMA, then 1 LDS read (ds_read2 pairs count once), ... rest MFMA
#pragma unroll
                for (int i = 0; i < 20; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_compute += t - tp; tp = t; }
            __syncthreads();                                // next buffer landed (vmcnt(0)) + everyone done reading
            if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_barrier += t - tp; tp = t; }
        }

        // epilogue: bias (+ReLU); each accumulator register = 16 adjacent pixels of one channel.
        // 32-bit in-image offsets off a per-tile scalar base: one v_add per store, no 64-bit multiplies.
        // (The transposed operand order -- 4 adjacent pixels per lane, one float4 store per M-tile -- was
        // measured SLOWER: each store instruction then touches 16 channel planes instead of 4.)
        float* ob = out + (size_t)b * C * H * W + ty0 * W + tx0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int so = (m >> 1) * W + 16 * (m & 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[m][r] + bv[r];
                if (RELU) v = v > 0.f ? v : (LEAKY ? slope * v : 0.f);
                ob[loff[r] + so] = v;
            }
        }
        if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_epi += t - tp; }
    }
    if (STAMP && tid == 0) {
        stamps[5 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        stamps[5 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
        stamps[5 * blockIdx.x + 2] = acc_compute;
        stamps[5 * blockIdx.x + 3] = acc_barrier;
        stamps[5 * blockIdx.x + 4] = acc_epi;
    }
}

// ------------------------------------------------------------------------------- Winograd F(2,3) variant
// Same tile / LDS image / DMA pipeline as k_mid, but the 3 horizontal taps go through the 1-D Winograd
// minimal-filtering transform F(2,3): per output PAIR (x, x+1) and input row,
//     V = B^T d  (d = inputs x-1 .. x+2):  V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3
//     U = G g    (g = the 3 horizontal weights): U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2
//     m_xi = sum_{dy,cin} U_xi V_xi ;   y(x) = m0 + m1 + m2 ,  y(x+1) = m1 - m2 - m3
// i.e. 4 x 3 x 64 multiply-adds per output pair instead of 2 x 9 x 64: 2/3 of the matrix-core work of the
// direct form for the same (exact-arithmetic) result; fp32 throughout, rounding differs from the fmaf chain
// at the 1e-7 level.  An M-tile is the 16 pixel pairs of one 32-pixel output row; a wave keeps
// 8 rows x 4 xi accumulators (128 regs) and its 4 x 3 x 16 = 192 transformed weights in registers.
// The B-operand transform (4 LDS values -> 4 V values, 4 VALU ops) feeds up to 12 MFMAs.
constexpr int WINO_U = 2 * (HALF_C / 4) * 3 * 4;      // 192 transformed-weight registers per wave
struct __attribute__((packed, aligned(4))) f2u { float a, b; };   // 4-byte-aligned float pair
typedef float f32x2v __attribute__((ext_vector_type(2)));

// The raw Winograd inputs of one group -- 5 halo rows x (d0,d1),(d2,d3) -- as ten hand-issued ds_read2_b32 off ONE
// base register with immediate offsets (row stride PC = 40 dwords).  hipcc pairs these loads as (d0,d3),(d1,d2)
// with a second base register and an extra v_add per row, and every VALU instruction here costs matrix-pipe issue.
// The results are asynchronous: the caller waits lgkmcnt(0) (wino_lds_wait) before the first use.
__device__ __forceinline__ void wino_lds_load(f32x2v (&dd)[5][2], unsigned lds_byte_addr) {
    f32x2v r0, r1, r2, r3, r4, r5, r6, r7, r8, r9;
    asm volatile(
        "ds_read2_b32 %0, %10 offset0:0 offset1:1\n"
        "ds_read2_b32 %1, %10 offset0:2 offset1:3\n"
        "ds_read2_b32 %2, %10 offset0:40 offset1:41\n"
        "ds_read2_b32 %3, %10 offset0:42 offset1:43\n"
        "ds_read2_b32 %4, %10 offset0:80 offset1:81\n"
        "ds_read2_b32 %5, %10 offset0:82 offset1:83\n"
        "ds_read2_b32 %6, %10 offset0:120 offset1:121\n"
        "ds_read2_b32 %7, %10 offset0:122 offset1:123\n"
        "ds_read2_b32 %8, %10 offset0:160 offset1:161\n"
        "ds_read2_b32 %9, %10 offset0:162 offset1:163\n"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9)
        : "v"(lds_byte_addr)
        : "memory");
    dd[0][0] = r0; dd[0][1] = r1; dd[1][0] = r2; dd[1][1] = r3; dd[2][0] = r4; dd[2][1] = r5;
    dd[3][0] = r6; dd[3][1] = r7; dd[4][0] = r8; dd[4][1] = r9;
}
// The wait takes the ten register pairs as in/out operands, so every consumer is data-dependent on it and no pass
// can move a use of the (still in flight) load results above the s_waitcnt.
// B^T d of one halo row as two packed-f32 adds: (d0-d2, -d1-d2) and (d1-d2, d1-d3).  The middle two components are
// the NEGATED textbook ones (d1+d2, d2-d1); the inverse transform in the epilogue flips their signs back.  hipcc
// lowers the first shuffle+negate to movs + two adds, so both are spelled out.  NO wait states inside: the caller
// must put >= 2 instructions between this and the first MFMA that reads the results.
__device__ __forceinline__ void wino_bt(f32x2v& v01, f32x2v& v23, f32x2v A, f32x2v Bq) {
    asm volatile("v_pk_add_f32 %0, %2, %3 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]\n\t"
                 "v_pk_add_f32 %1, %2, %3 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                 : "=&v"(v01), "=&v"(v23) : "v"(A), "v"(Bq));
}
// The Winograd kernel's MFMAs are hand-issued so that the REGISTER FILES are split the other way round from what
// hipcc picks: accumulators in VGPRs (the epilogue's VALU reads them in place), the 192 tile-invariant transformed
// weights in AGPRs (gfx90a+ MFMAs read SrcA from either file).  hipcc keeps accumulators in AGPRs, copies ~60
// weights through v_accvgpr_read every tile and reads all 128 accumulators back for the epilogue.  The first MFMA
// of an accumulator in a tile uses the constant-zero SrcC form, so accumulators are never cleared.
// Hazards are the caller's: >= 2 instructions between a VALU write of `v` and the MFMA, and wait states between the
// last MFMA and a VALU read of an accumulator (the hazard recognizer does not look inside inline asm).
__device__ __forceinline__ void mfma_wa(f32x4& acc, float w_agpr, float v) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w_agpr), "v"(v));
}
__device__ __forceinline__ void mfma_wa_first(f32x4& acc, float w_agpr, float v) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc) : "a"(w_agpr), "v"(v));
}
__device__ __forceinline__ void wino_lds_wait(f32x2v (&dd)[5][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(dd[0][0]), "+v"(dd[0][1]), "+v"(dd[1][0]), "+v"(dd[1][1]), "+v"(dd[2][0]), "+v"(dd[2][1]),
                   "+v"(dd[3][0]), "+v"(dd[3][1]), "+v"(dd[4][0]), "+v"(dd[4][1])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
static_assert(PC == 40, "wino_lds_load hard-codes the 40-dword LDS row stride");

template <bool RELU, bool STAMP = false, bool LEAKY = false>
__global__ __launch_bounds__(256, 1) void k_mid_wino(const float* __restrict__ in, float* __restrict__ out,
                                                     const float* __restrict__ upack, const float* __restrict__ bias,
                                                     const float* __restrict__ zeros, int H, int W, int ntiles,
                                                     unsigned long long* __restrict__ stamps = nullptr, float slope = 0.f) {
    __shared__ float lds[2 * HALF_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = W / TC, tiles_per_img = tiles_x * (H / TR);

    // ureg[((half*8 + c4)*3 + dy)*4 + xi] = U_xi[cout = 16wv + (lane&15)][cin = 32half + 4c4 + (lane>>4)][dy]
    float ureg[WINO_U];
#pragma unroll
    for (int s = 0; s < WINO_U; ++s) ureg[s] = upack[((size_t)wv * WINO_U + s) * 64 + lane];
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[16 * wv + 4 * (lane >> 4) + r];

    // lane (k-row kq = lane>>4, pair j = lane&15) reads d0..d3 at LDS columns XOFF + 2j + {0,1,2,3}
    const int lbase = (lane >> 4) * PLANE + 2 * (lane & 15) + XOFF;
    int loff[4];                                               // output offsets: pixel pair 2j of channel ...
#pragma unroll
    for (int r = 0; r < 4; ++r) loff[r] = (16 * wv + 4 * (lane >> 4) + r) * H * W + 2 * (lane & 15);

    // DMA piece descriptors, ONE register each: bits 0..27 = element offset of the lane's 16-byte chunk inside the
    // half's 32 channel planes, bits 28..31 = which image edge would put the chunk outside (top row of the halo,
    // bottom row, left chunk, right chunk).  A tile's own edge mask (scalar) then decides validity with one AND.
    unsigned pdesc[PIECES_PER_WAVE];
#pragma unroll
    for (int i = 0; i < PIECES_PER_WAVE; ++i) {
        const int q = (wv + 4 * i) * 64 + lane;
        const int cin = q / 100, r = q - cin * 100;
        const int ry = r / 10, cx4 = 4 * (r - ry * 10);
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
    unsigned long long t0 = 0, r0 = 0, acc_compute = 0, acc_barrier = 0, acc_epi = 0, tp = 0;
    if (STAMP) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    for (; tile < tw_.limit; tile += tw_.step) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int ty0 = (t2 / tiles_x) * TR, tx0 = (t2 % tiles_x) * TC;
        f32x4 acc[TR][4];                                       // written first by mfma_wa_first (half 0, channel quad 0, dy 0)

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
            if (STAMP) tp = __builtin_amdgcn_s_memtime();

            // group = (channel quad c4, block of 5 halo rows): 10 ds_read2 + 20 transform ops + 48 MFMAs
            constexpr int NG = (HALF_C / 4) * 2;               // 16 groups per half
            f32x2v d[2][5][2];
            const unsigned xb_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + 4u * (unsigned)xb_off;
            wino_lds_load(d[0], xb_addr);
            // All non-MFMA work of a group -- retire the group's LDS reads, issue the next group's, one DMA piece, the ten
            // packed adds of B^T d for the group's five halo rows -- is issued as ONE block in front of the group's 48
            // MFMAs.  On gfx950 every excursion from the MFMA stream to the vector ALU and back costs ~10 cycles on top
            // of ~4.3 per instruction (tools/microbench/mfma_f32_valu.hip: one v_pk_add_f32 between two fp32 MFMAs takes
            // them from 32.3 to 46.8 cycles), so the number of excursions counts, not only the instruction count.  It
            // also puts >= 2 instructions between every transform and the MFMA that reads it (VALU -> MFMA wait
            // states; the hazard recognizer does not see inside the asm).
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int c4 = g / 2, rb = g % 2;
                wino_lds_wait(d[g & 1]);                        // issued a whole group ago (or just above for g = 0)
                if (g + 1 < NG) {
                    const int c4n = (g + 1) / 2, rbn = (g + 1) % 2;
                    wino_lds_load(d[(g + 1) & 1], xb_addr + 4u * ((4 * c4n) * PLANE + (5 * rbn) * PC));
                }
                if (g < PIECES_PER_WAVE) {
                    const int pc = wv + 4 * g;
                    const bool ok = (nvalid & (pc < PIECES)) & ((pdesc[g] & nedge) == 0u);
                    const float* src = ok ? nsrc0 + (pdesc[g] & 0x0FFFFFFFu) : zeros;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(nbuf + pc * 256), 16, 0, 0);
                }
                f32x2v v01[5], v23[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) wino_bt(v01[i], v23[i], d[g & 1][i][0], d[g & 1][i][1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int ry = 5 * rb + i;
                    const float V[4] = {v01[i].x, v01[i].y, v23[i].x, v23[i].y};
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int r = ry - dy;
                        if (r >= 0 && r < TR) {
#pragma unroll
                            for (int xi = 0; xi < 4; ++xi) {
                                const float wgt = ureg[((half * (HALF_C / 4) + c4) * 3 + dy) * 4 + xi];
                                if (half == 0 && c4 == 0 && dy == 0) mfma_wa_first(acc[r][xi], wgt, V[xi]);
                                else mfma_wa(acc[r][xi], wgt, V[xi]);
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_compute += t - tp; tp = t; }
            __syncthreads();
            if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_barrier += t - tp; tp = t; }
        }

        // epilogue: inverse transform, bias (+ReLU); a lane holds pixel pair (2j, 2j+1) of 4 channels per row
        // (the barrier above sits between the last MFMA and these VALU reads; the explicit wait states make the
        // MFMA-write -> VALU-read distance independent of what the barrier costs)
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        float* ob = out + (size_t)b * C * H * W + ty0 * W + tx0;
#pragma unroll
        for (int r = 0; r < TR; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float m0 = acc[r][0][q], m1 = acc[r][1][q], m2 = acc[r][2][q], m3 = acc[r][3][q];
                float2 v;
                v.x = (m0 - m1 - m2) + bv[q];                   // m1, m2 carry the flipped signs of the packed transform
                v.y = (m2 - m1 - m3) + bv[q];
                if (RELU) { v.x = v.x > 0.f ? v.x : (LEAKY ? slope * v.x : 0.f); v.y = v.y > 0.f ? v.y : (LEAKY ? slope * v.y : 0.f); }
                *reinterpret_cast<float2*>(ob + loff[q] + r * W) = v;
            }
        }
        if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_epi += t - tp; }
    }
    if (STAMP && tid == 0) {
        stamps[5 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        stamps[5 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
        stamps[5 * blockIdx.x + 2] = acc_compute;
        stamps[5 * blockIdx.x + 3] = acc_barrier;
        stamps[5 * blockIdx.x + 4] = acc_epi;
    }
}

// ------------------------------------------------------------------------------- first layer
// xt = ((z - lo) / (hi - lo)) * srange + sshift ; act[c] = relu(sum_t w[c][t] * xt[tap t])
// MMO form (clamp01): xt = clamp(z, 0, 1); act[c] = leaky_relu(b[c] + sum_t ..., slope)   (MMODenoise.py:30,90-91)
template <typename T>
__global__ __launch_bounds__(256) void k_first(const T* __restrict__ z, const T* __restrict__ mm,
                                               const float* __restrict__ w, const float* __restrict__ bfirst,
                                               float* __restrict__ out, int H, int W, double srange, double sshift,
                                               int clamp01, float slope) {
    __shared__ float ws[C * 9];
    __shared__ float bs[C];
    for (int i = threadIdx.x; i < C * 9; i += 256) ws[i] = w[i];
    if (threadIdx.x < C) bs[threadIdx.x] = bfirst[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const T lo = mm ? mm[2 * b] : (T)0, hi = mm ? mm[2 * b + 1] : (T)1;
    const T* zi = z + (size_t)b * H * W;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        float q = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            T u = zi[yy * W + xx];
            if (clamp01) {
                u = u < (T)0 ? (T)0 : (u > (T)1 ? (T)1 : u);
            } else {
                u = (u - lo) / (hi - lo);
                u = u * (T)srange + (T)sshift;
            }
            q = (float)u;
        }
        v[t] = q;
    }
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) a = fmaf(ws[c * 9 + t], v[t], a);
        a += bs[c];                                             // 0 for the DnCNN family (bias-free first layer)
        out[((size_t)b * C + c) * H * W + p] = a > 0.f ? a : slope * a + 0.f;
    }
}

// ------------------------------------------------------------------------------- last layer
// r = sum_{c,t} w[c][t] * act[c][tap t];  x = xt - r;  x = (x - sshift)/srange;  z = x*(hi-lo)+lo
// HBM/L2-bound (reads the 64-channel activation once): 16 x 64 output tile per workgroup, 4 channels
// at a time staged in LDS as [4][18][72] (columns tx0-4 .. tx0+67, 16-byte aligned rows; 21 KB, so a whole
// B = 16 grid of 1024 workgroups is resident at once), the next slab's loads in flight during the current one, every
// thread owns a 1 x 4 pixel strip: one ds_read_b128 + two ds_read_b32 per row and channel.
constexpr int LT_R = 16, LT_C = 64, LT_CK = 4, LT_PR = LT_R + 2, LT_PC = LT_C + 8;

template <typename T>
__global__ __launch_bounds__(256) void k_last(const float* __restrict__ act, const float* __restrict__ w,
                                              const T* zin, const T* __restrict__ mm,      // zin may alias zout (in-place prox): no __restrict__
                                              T* zout, float* __restrict__ r_out,
                                              const T* __restrict__ xrec, double* __restrict__ sse_part, int H, int W,
                                              double srange, double sshift, int skip01, float blast) {
    __shared__ float ws[C * 9];
    __shared__ __attribute__((aligned(16))) float tile[LT_CK * LT_PR * LT_PC];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    for (int i = tid; i < C * 9; i += 256) ws[i] = w[i];
    const int b = blockIdx.z;
    const int ty0 = blockIdx.y * LT_R, tx0 = blockIdx.x * LT_C;
    const int py = tid >> 4, px = (tid & 15) * 4;               // this thread's strip inside the tile
    const float* ab = act + (size_t)b * C * H * W;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    // software pipeline over the 8-channel slabs: the global loads of slab k+1 are in flight (in registers) while
    // slab k is consumed out of LDS, so a workgroup never sits on an exposed HBM round trip per slab
    constexpr int SLAB = LT_CK * LT_PR * (LT_PC / 4);          // float4 elements of one slab (2592)
    constexpr int PER_T = (SLAB + 255) / 256;                  // 11 per thread
    float4 pre[PER_T];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < PER_T; ++k) {
            const int i = tid + 256 * k;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < SLAB) {
                const int ch = i / (LT_PR * (LT_PC / 4)), rem = i - ch * (LT_PR * (LT_PC / 4));
                const int ry = rem / (LT_PC / 4), cx = rem - ry * (LT_PC / 4);
                const int y = ty0 - 1 + ry, x = tx0 - 4 + 4 * cx;
                if (y >= 0 && y < H && x >= 0 && x < W)
                    v = *reinterpret_cast<const float4*>(ab + ((size_t)(c0 + ch) * H + y) * W + x);
            }
            pre[k] = v;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < C; c0 += LT_CK) {
        __syncthreads();                                        // the previous slab has been consumed
#pragma unroll
        for (int k = 0; k < PER_T; ++k) {
            const int i = tid + 256 * k;
            if (i < SLAB) reinterpret_cast<float4*>(tile)[i] = pre[k];   // slab layout == linear float4 index
        }
        __syncthreads();
        if (c0 + LT_CK < C) fetch(c0 + LT_CK);
#pragma unroll
        for (int ch = 0; ch < LT_CK; ++ch) {
            const float* wc = ws + (c0 + ch) * 9;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float* row = tile + (ch * LT_PR + py + dy) * LT_PC + px + 4;     // LDS col of pixel px
                const float4 m = *reinterpret_cast<const float4*>(row);
                const float l = row[-1], rr = row[4];
                const float w0 = wc[dy * 3], w1 = wc[dy * 3 + 1], w2 = wc[dy * 3 + 2];
                r[0] = fmaf(w0, l, fmaf(w1, m.x, fmaf(w2, m.y, r[0])));
                r[1] = fmaf(w0, m.x, fmaf(w1, m.y, fmaf(w2, m.z, r[1])));
                r[2] = fmaf(w0, m.y, fmaf(w1, m.z, fmaf(w2, m.w, r[2])));
                r[3] = fmaf(w0, m.z, fmaf(w1, m.w, fmaf(w2, rr, r[3])));
            }
        }
    }
    double err = 0.0;
    const int y = ty0 + py;
    if (y < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = tx0 + px + j;
            if (x >= W) continue;
            const size_t p = (size_t)b * H * W + (size_t)y * W + x;
            r[j] += blast;                                      // 0 for the DnCNN family (bias-free last layer)
            if (r_out != nullptr) r_out[p] = r[j];
            if (zout != nullptr && skip01) {
                // MMO form: out = clip(clamp(x,0,1) + net(clamp(x,0,1)), 0, 1), fp32 like the reference's tensors
                // (MMODenoise.py:30-32,98, then the float64 np.clip of :128 which is a no-op after the clamp)
                const float xin = (float)zin[p];
                float v = (xin < 0.f ? 0.f : (xin > 1.f ? 1.f : xin)) + r[j];
                v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
                zout[p] = (T)v;
                if (xrec != nullptr) {
                    const double d = (double)xrec[p] - (double)(T)v;
                    err += d * d;
                }
            } else if (zout != nullptr) {
                const T lo = mm[2 * b], hi = mm[2 * b + 1];
                T v = (zin[p] - lo) / (hi - lo);
                v = v * (T)srange + (T)sshift;                  // xtilde, kept in T (f64 in the reference wrapper)
                v = v - (T)r[j];                                // f64 - f32 in the reference wrapper
                v = (v - (T)sshift) / (T)srange;
                v = v * (hi - lo) + lo;
                zout[p] = v;
                if (xrec != nullptr) {
                    const double d = (double)xrec[p] - (double)v;
                    err += d * d;
                }
            }
        }
    }
    if (sse_part != nullptr) {
        err = wave_sum(err);
        if ((tid & 63) == 0) red[tid >> 6] = err;
        __syncthreads();
        if (tid == 0)
            sse_part[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

__global__ void k_sum_parts(const double* __restrict__ part, int nparts, double* __restrict__ out) {
    double s = 0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[(size_t)blockIdx.x * nparts + i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

}  // namespace pnp

using namespace pnp;

struct pnp_dncnn_plan {
    int n_mid, H, W, batch, num_cu;
    float *w_first, *w_last, *wpack, *upack, *bias;   // device (upack: Winograd F(2,3)-transformed weights)
    float* upack44;                              // Winograd F(4x4,3x3)-transformed weights (dncnn_wino44.hip)
    uint16_t* upack44b;                          // the same, split into three bf16 terms (dncnn_wino44b.hip; conv mode 6)
    float* b_first;                              // [64] device, zeros unless pnp_dncnn_set_affine
    float b_last, slope;                         // last-layer bias, LeakyReLU slope (0 = ReLU)
    int use_wino;
    float *act0, *act1, *zeros;                  // [B][64][H][W] x2; a zero word for halo padding
    double* mm;                                  // [B][2] (as double or float depending on call)
    double* sse_part;                            // [B][H*W/256]
    // optional in-band timing of the MFMA layers (hipEvents on the caller's stream, no sync)
    bool profile;
    std::vector<hipEvent_t> ev;                  // pairs: [2i] before, [2i+1] after the n_mid launches
    size_t ev_used;
    double prof_ms;
    long prof_launches;
};

extern "C" int pnp_dncnn_plan_create(pnp_dncnn_plan** out, int n_mid, const float* w_first, const float* w_mid,
                                     const float* b_mid, const float* w_last, int H, int W, int batch) {
    PNP_CHECK_ARG(out && w_first && w_mid && b_mid && w_last, "null argument");
    PNP_CHECK_ARG(n_mid >= 1 && batch >= 1, "n_mid and batch must be >= 1");
    PNP_CHECK_ARG(H % TR == 0 && W % TC == 0 && (H * W) % 256 == 0, "H must be a multiple of 8, W of 32");
    auto* p = new pnp_dncnn_plan{};
    p->n_mid = n_mid; p->H = H; p->W = W; p->batch = batch;
    hipDeviceProp_t prop;
    int dev = 0;
    PNP_CHECK_HIP(hipGetDevice(&dev));
    PNP_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    p->num_cu = prop.multiProcessorCount;
    // pack the mid-layer weights into MFMA (16x16x4) register order:
    //   wpack[l][wv][s][lane] = W[l][cout = 16wv + (lane&15)][cin][tap], k-step s = half*72 + tap*8 + c4,
    //   cin = 32*half + 4*c4 + (lane>>4)
    std::vector<float> pack((size_t)n_mid * 4 * 2 * KSTEPS_HALF * 64);
    for (int l = 0; l < n_mid; ++l)
        for (int wv = 0; wv < 4; ++wv)
            for (int s = 0; s < 2 * KSTEPS_HALF; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int half = s / KSTEPS_HALF, rs = s % KSTEPS_HALF, tap = rs / (HALF_C / 4), c4 = rs % (HALF_C / 4);
                    const int cout = 16 * wv + (lane & 15), cin = HALF_C * half + 4 * c4 + (lane >> 4);
                    pack[(((size_t)l * 4 + wv) * 2 * KSTEPS_HALF + s) * 64 + lane] =
                        w_mid[(((size_t)l * C + cout) * C + cin) * 9 + tap];
                }
    // Winograd F(2,3)-transformed weights (along dx): upack[l][wv][s][lane], s = ((half*8 + c4)*3 + dy)*4 + xi
    std::vector<float> upk((size_t)n_mid * 4 * WINO_U * 64);
    for (int l = 0; l < n_mid; ++l)
        for (int wv = 0; wv < 4; ++wv)
            for (int s = 0; s < WINO_U; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int xi = s % 4, dy = (s / 4) % 3, c4 = (s / 12) % (HALF_C / 4), half = s / (12 * (HALF_C / 4));
                    const int cout = 16 * wv + (lane & 15), cin = HALF_C * half + 4 * c4 + (lane >> 4);
                    const float* g = w_mid + (((size_t)l * C + cout) * C + cin) * 9 + dy * 3;
                    const double g0 = g[0], g1 = g[1], g2 = g[2];
                    const double u = xi == 0 ? g0 : xi == 1 ? 0.5 * (g0 + g1 + g2) : xi == 2 ? 0.5 * (g0 - g1 + g2) : g2;
                    upk[(((size_t)l * 4 + wv) * WINO_U + s) * 64 + lane] = (float)u;
                }
    {
        // conv kernel of the middle layers: 5 = Winograd F(4x4,3x3) (default where the image size allows it), 4 = F(4,3)
        // along x, 1 = F(2,3) along x, 0 = direct, 3 = opt-in split-fp16
        const char* ev = getenv("PNP_DNCNN_WINOGRAD");
        p->use_wino = ev ? atoi(ev) : 5;
        if (p->use_wino != 6 && p->use_wino != 5 && p->use_wino != 1 && p->use_wino != 0) p->use_wino = 5;      // (unknown value: the default)
        if (p->use_wino >= 5 && !wino44_supports(H, W)) p->use_wino = 1;
    }
    const size_t act_bytes = (size_t)batch * C * H * W * sizeof(float);
    hipError_t e = hipMalloc(&p->wpack, pack.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->wpack, pack.data(), pack.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&p->upack, upk.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->upack, upk.data(), upk.size() * sizeof(float), hipMemcpyHostToDevice);
    {
        std::vector<float> u44(wino44_weight_floats(n_mid));
        wino44_pack_weights(w_mid, n_mid, u44.data());
        if (e == hipSuccess) e = hipMalloc(&p->upack44, u44.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(p->upack44, u44.data(), u44.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    if (wino44b_supports(H, W)) {
        std::vector<uint16_t> u44b(wino44b_weight_halfwords(n_mid));
        wino44b_pack_weights(w_mid, n_mid, u44b.data());
        if (e == hipSuccess) e = hipMalloc(&p->upack44b, u44b.size() * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMemcpy(p->upack44b, u44b.data(), u44b.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(&p->bias, (size_t)n_mid * C * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->bias, b_mid, (size_t)n_mid * C * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&p->w_first, C * 9 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->w_first, w_first, C * 9 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&p->w_last, C * 9 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->w_last, w_last, C * 9 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&p->b_first, C * sizeof(float));
    if (e == hipSuccess) e = hipMemset(p->b_first, 0, C * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&p->act0, act_bytes);
    if (e == hipSuccess) e = hipMalloc(&p->act1, act_bytes);
    if (e == hipSuccess) e = hipMalloc(&p->zeros, 256);
    if (e == hipSuccess) e = hipMemset(p->zeros, 0, 256);
    if (e == hipSuccess) e = hipMalloc(&p->mm, (size_t)batch * 2 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&p->sse_part, (size_t)batch * (H * W / 256) * sizeof(double));
    if (e != hipSuccess) {
        set_error(std::string("pnp_dncnn_plan_create: ") + hipGetErrorString(e));
        for (void* q : {(void*)p->wpack, (void*)p->upack, (void*)p->bias, (void*)p->w_first, (void*)p->w_last, (void*)p->act0,
                        (void*)p->act1, (void*)p->zeros, (void*)p->mm, (void*)p->sse_part, (void*)p->b_first, (void*)p->upack44,
                        (void*)p->upack44b})
            if (q) (void)hipFree(q);
        delete p;
        return PNP_ERR_HIP;
    }
    *out = p;
    return PNP_OK;
}

extern "C" int pnp_dncnn_plan_destroy(pnp_dncnn_plan* p) {
    if (!p) return PNP_OK;
    for (void* q : {(void*)p->wpack, (void*)p->upack, (void*)p->bias, (void*)p->w_first, (void*)p->w_last, (void*)p->act0, (void*)p->act1,
                    (void*)p->zeros, (void*)p->mm, (void*)p->sse_part, (void*)p->b_first, (void*)p->upack44, (void*)p->upack44b})
        (void)hipFree(q);
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    delete p;
    return PNP_OK;
}

namespace {
template <typename T>
int run_dncnn(pnp_dncnn_plan* p, const T* z_in, bool normalise, double sigma_net, T* z_out, float* r_out,
              const T* xrec, double* sse_out, hipStream_t s, bool mmo = false) {
    const int H = p->H, W = p->W, B = p->batch, HW = H * W;
    T* mm = normalise ? (T*)p->mm : nullptr;             // raw network (no wrapper scaling): lo = 0, hi = 1
    double srange = 1.0, sshift = 0.0;
    if (normalise) {
        k_minmax<T><<<B, 256, 0, s>>>(z_in, HW, mm);
        PNP_CHECK_LAUNCH();
        srange = 1.0 + sigma_net / 255.0 / 2.0;
        sshift = (1.0 - srange) / 2.0;
    }
    dim3 pg(HW / 256, B);
    k_first<T><<<pg, 256, 0, s>>>(z_in, mm, p->w_first, p->b_first, p->act0, H, W, srange, sshift, mmo ? 1 : 0, p->slope);
    PNP_CHECK_LAUNCH();
    const int ntiles = B * (H / TR) * (W / TC);
    const int grid = ntiles < p->num_cu ? ntiles : p->num_cu;
    float *src = p->act0, *dst = p->act1;
    const bool prof = p->profile && p->ev_used + 2 <= p->ev.size();
    if (prof) PNP_CHECK_HIP(hipEventRecord(p->ev[p->ev_used], s));
    for (int l = 0; l < p->n_mid; ++l) {
        if (p->use_wino == 6) {
            const int rc = wino44b_layer(src, dst, p->upack44b + (size_t)l * wino44b_weight_halfwords(1), p->bias + (size_t)l * C,
                                         H, W, B, p->num_cu, p->slope, s);
            if (rc != PNP_OK) return rc;
        } else if (p->use_wino == 5) {
            const int rc = wino44_layer(src, dst, p->upack44 + (size_t)l * wino44_weight_floats(1), p->bias + (size_t)l * C, p->zeros,
                                        H, W, B, p->num_cu, p->slope, s);
            if (rc != PNP_OK) return rc;
        } else if (p->slope != 0.f) {                             // LeakyReLU builds exist for the two production kernels
            if (p->use_wino)
                k_mid_wino<true, false, true><<<grid, 256, 0, s>>>(src, dst, p->upack + (size_t)l * 4 * WINO_U * 64,
                                                                  p->bias + (size_t)l * C, p->zeros, H, W, ntiles, nullptr, p->slope);
            else
                k_mid<true, false, 0, true><<<grid, 256, 0, s>>>(src, dst, p->wpack + (size_t)l * 4 * 2 * KSTEPS_HALF * 64,
                                                                p->bias + (size_t)l * C, p->zeros, H, W, ntiles, nullptr, p->slope);
        } else if (p->use_wino)
            k_mid_wino<true><<<grid, 256, 0, s>>>(src, dst, p->upack + (size_t)l * 4 * WINO_U * 64, p->bias + (size_t)l * C,
                                                  p->zeros, H, W, ntiles);
        else
            k_mid<true><<<grid, 256, 0, s>>>(src, dst, p->wpack + (size_t)l * 4 * 2 * KSTEPS_HALF * 64, p->bias + (size_t)l * C,
                                             p->zeros, H, W, ntiles);
        PNP_CHECK_LAUNCH();
        float* t = src; src = dst; dst = t;
    }
    if (prof) { PNP_CHECK_HIP(hipEventRecord(p->ev[p->ev_used + 1], s)); p->ev_used += 2; }
    dim3 lg((W + LT_C - 1) / LT_C, (H + LT_R - 1) / LT_R, B);
    k_last<T><<<lg, 256, 0, s>>>(src, p->w_last, z_in, mm, z_out, r_out, xrec, sse_out ? p->sse_part : nullptr, H, W,
                                 srange, sshift, mmo ? 1 : 0, p->b_last);
    PNP_CHECK_LAUNCH();
    if (sse_out) {
        k_sum_parts<<<B, 64, 0, s>>>(p->sse_part, (int)(lg.x * lg.y), sse_out);
        PNP_CHECK_LAUNCH();
    }
    return PNP_OK;
}
}  // namespace

// ---- test hooks: one middle layer on caller-provided buffers (guard-band tests)
extern "C" size_t pnp_dncnn_debug_w44_floats(void) { return wino44_weight_floats(1); }

extern "C" int pnp_dncnn_debug_w44_weights(pnp_dncnn_plan* p, int layer, float* dst, void* stream) {
    PNP_CHECK_ARG(p && dst && layer >= 0 && layer < p->n_mid, "bad argument");
    PNP_CHECK_HIP(hipMemcpyAsync(dst, p->upack44 + (size_t)layer * wino44_weight_floats(1), wino44_weight_floats(1) * sizeof(float),
                                 hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PNP_OK;
}

extern "C" int pnp_dncnn_debug_mid_layer(pnp_dncnn_plan* p, int layer, const float* in, float* out, const float* w44_override,
                                         int w44_rows, void* stream) {
    PNP_CHECK_ARG(p && in && out && layer >= 0 && layer < p->n_mid && w44_rows >= 0 && w44_rows <= 2, "bad argument");
    PNP_CHECK_ARG(w44_override == nullptr || p->use_wino == 5, "w44_override needs the F(4x4,3x3) kernel (mode 5)");
    hipStream_t s = (hipStream_t)stream;
    const int H = p->H, W = p->W, B = p->batch, l = layer;
    const int ntiles = B * (H / TR) * (W / TC);
    const int grid = ntiles < p->num_cu ? ntiles : p->num_cu;
    if (p->use_wino == 6)
        return wino44b_layer(in, out, p->upack44b + (size_t)l * wino44b_weight_halfwords(1), p->bias + (size_t)l * C, H, W, B, p->num_cu,
                             p->slope, s);
    if (p->use_wino == 5)
        return wino44_layer(in, out, w44_override ? w44_override : p->upack44 + (size_t)l * wino44_weight_floats(1),
                            p->bias + (size_t)l * C, p->zeros, H, W, B, p->num_cu, p->slope, s, w44_rows);
    PNP_CHECK_ARG(p->slope == 0.f, "the hook runs the ReLU builds of the direct / F(2,3) kernels");
    if (p->use_wino)
        k_mid_wino<true><<<grid, 256, 0, s>>>(in, out, p->upack + (size_t)l * 4 * WINO_U * 64, p->bias + (size_t)l * C, p->zeros, H, W, ntiles);
    else
        k_mid<true><<<grid, 256, 0, s>>>(in, out, p->wpack + (size_t)l * 4 * 2 * KSTEPS_HALF * 64, p->bias + (size_t)l * C, p->zeros, H, W, ntiles);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

extern "C" int pnp_dncnn_set_affine(pnp_dncnn_plan* p, const float* b_first, float b_last, float negative_slope) {
    PNP_CHECK_ARG(p != nullptr, "null plan");
    PNP_CHECK_ARG(negative_slope >= 0.f && negative_slope < 1.f, "negative_slope must be in [0, 1)");
    if (b_first) PNP_CHECK_HIP(hipMemcpy(p->b_first, b_first, C * sizeof(float), hipMemcpyHostToDevice));
    else PNP_CHECK_HIP(hipMemset(p->b_first, 0, C * sizeof(float)));
    p->b_last = b_last;
    p->slope = negative_slope;
    return PNP_OK;
}

extern "C" int pnp_dncnn_set_winograd(pnp_dncnn_plan* p, int enable) {
    PNP_CHECK_ARG(p != nullptr, "null plan");
    PNP_CHECK_ARG(enable == 0 || enable == 1 || enable == 5 || enable == 6,
                  "mode must be 0 (direct), 1 (Winograd F(2,3) along x), 5 (Winograd F(4x4,3x3)) or 6 (the same on 3 x bf16 splits)");
    PNP_CHECK_ARG(!(enable >= 5 && !wino44_supports(p->H, p->W)), "Winograd F(4x4,3x3) needs H % 8 == 0 and W % 64 == 0");
    p->use_wino = enable;
    return PNP_OK;
}

extern "C" int pnp_dncnn_profile_begin(pnp_dncnn_plan* p, int max_calls) {
    PNP_CHECK_ARG(p && max_calls > 0, "bad argument");
    while (p->ev.size() < (size_t)2 * max_calls) {
        hipEvent_t e;
        PNP_CHECK_HIP(hipEventCreate(&e));
        p->ev.push_back(e);
    }
    p->profile = true; p->ev_used = 0; p->prof_ms = 0; p->prof_launches = 0;
    return PNP_OK;
}

extern "C" int pnp_dncnn_profile_end(pnp_dncnn_plan* p, double* avg_ms_per_launch, long* launches) {
    PNP_CHECK_ARG(p && avg_ms_per_launch && launches, "null argument");
    p->profile = false;
    for (size_t i = 0; i + 1 < p->ev_used; i += 2) {
        PNP_CHECK_HIP(hipEventSynchronize(p->ev[i + 1]));
        float ms = 0;
        PNP_CHECK_HIP(hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]));
        p->prof_ms += ms;
        p->prof_launches += p->n_mid;
    }
    *launches = p->prof_launches;
    *avg_ms_per_launch = p->prof_launches ? p->prof_ms / (double)p->prof_launches : 0.0;
    return PNP_OK;
}

// Diagnostic: in-kernel shader clock and phase shares of the conv kernel under load.  Runs `reps` back-to-back
// conv launches on the plan's activation buffers (contents arbitrary), the last one stamped; returns the median
// over workgroups of (shader cycles, 100 MHz reference ticks) spent in the tile loop.  PNP_DEBUG_STAMPS=1 prints
// the compute / barrier / epilogue split, PNP_DEBUG_ABL selects an ablation build.
extern "C" int pnp_dncnn_debug_clock(pnp_dncnn_plan* p, int reps, double* cycles, double* ref_ticks, void* stream) {
    PNP_CHECK_ARG(p && cycles && ref_ticks && reps >= 1, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = p->batch * (p->H / TR) * (p->W / TC);
    const int grid = ntiles < p->num_cu ? ntiles : p->num_cu;
    unsigned long long* d = nullptr;
    PNP_CHECK_HIP(hipMalloc(&d, (size_t)grid * 5 * sizeof(unsigned long long)));
    if (p->use_wino >= 5) {                                     // F(4x4,3x3): 8 x 64 regions, 4 stamps per workgroup
        const int nt5 = p->batch * (p->H / 8) * (p->W / 64), g5 = nt5 < p->num_cu ? nt5 : p->num_cu;
        const int rc = p->use_wino == 6 ? wino44b_debug_clock(p->act0, p->act1, p->upack44b, p->bias, p->H, p->W, p->batch, p->num_cu, reps, d, s)
                                        : wino44_debug_clock(p->act0, p->act1, p->upack44, p->bias, p->H, p->W, p->batch, p->num_cu, reps, d, s);
        if (rc != PNP_OK) { (void)hipFree(d); return rc; }
        std::vector<unsigned long long> h5((size_t)g5 * 4);
        hipError_t e5 = hipMemcpyAsync(h5.data(), d, h5.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
        if (e5 == hipSuccess) e5 = hipStreamSynchronize(s);
        (void)hipFree(d);
        PNP_CHECK_HIP(e5);
        std::vector<double> c5(g5), r5(g5);
        double wsum = 0, esum = 0;
        double rsum = 0;
        for (int i = 0; i < g5; ++i) {
            c5[i] = (double)h5[4 * i]; r5[i] = (double)h5[4 * i + 1]; esum += h5[4 * i + 3];
            wsum += (double)(h5[4 * i + 2] & 0xFFFFFFFFull); rsum += (double)(h5[4 * i + 2] >> 32);       // (mode 6 packs a second counter above bit 32)
        }
        if (getenv("PNP_DEBUG_STAMPS"))
            fprintf(stderr, "[k_mid_wino44 stamps] mean cycles per WG: chunk-end wait + barrier %.0f  epilogue %.0f  (mode 6: steps 0..5 of the chunks %.0f)\n",
                    wsum / g5, esum / g5, rsum / g5);
        std::sort(c5.begin(), c5.end());
        std::sort(r5.begin(), r5.end());
        *cycles = c5[g5 / 2];
        *ref_ticks = r5[g5 / 2];
        return PNP_OK;
    }
    for (int i = 0; i < reps - 1; ++i) {
        if (p->use_wino) k_mid_wino<true><<<grid, 256, 0, s>>>(p->act0, p->act1, p->upack, p->bias, p->zeros, p->H, p->W, ntiles);
        else k_mid<true><<<grid, 256, 0, s>>>(p->act0, p->act1, p->wpack, p->bias, p->zeros, p->H, p->W, ntiles);
    }
    const char* abl = getenv("PNP_DEBUG_ABL");
    const int ab = abl ? atoi(abl) : 0;
    if (p->use_wino) k_mid_wino<true, true><<<grid, 256, 0, s>>>(p->act0, p->act1, p->upack, p->bias, p->zeros, p->H, p->W, ntiles, d);
    else if (ab == 2) k_mid<true, true, 2><<<grid, 256, 0, s>>>(p->act0, p->act1, p->wpack, p->bias, p->zeros, p->H, p->W, ntiles, d);
    else k_mid<true, true><<<grid, 256, 0, s>>>(p->act0, p->act1, p->wpack, p->bias, p->zeros, p->H, p->W, ntiles, d);
    std::vector<unsigned long long> h((size_t)grid * 5);
    hipError_t e = hipMemcpyAsync(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    PNP_CHECK_HIP(e);
    std::vector<double> c(grid), r(grid);
    for (int i = 0; i < grid; ++i) { c[i] = (double)h[5 * i]; r[i] = (double)h[5 * i + 1]; }
    if (getenv("PNP_DEBUG_STAMPS")) {
        double a = 0, b = 0, ep = 0;
        for (int i = 0; i < grid; ++i) { a += h[5 * i + 2]; b += h[5 * i + 3]; ep += h[5 * i + 4]; }
        fprintf(stderr, "[k_mid stamps] mean cycles per WG: compute %.0f  barrier %.0f  epilogue %.0f\n", a / grid, b / grid, ep / grid);
    }
    std::sort(c.begin(), c.end());
    std::sort(r.begin(), r.end());
    *cycles = c[grid / 2];
    *ref_ticks = r[grid / 2];
    return PNP_OK;
}

extern "C" int pnp_dncnn_forward(pnp_dncnn_plan* p, const float* x, float* r, void* stream) {
    PNP_CHECK_ARG(p && x && r, "null argument");
    return run_dncnn<float>(p, x, false, 0.0, nullptr, r, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int pnp_mmo_denoise(pnp_dncnn_plan* p, const void* z_in, void* z_out, int dtype, const void* xrec,
                               double* sse_out, void* stream) {
    PNP_CHECK_ARG(p && z_in && z_out, "null argument");
    PNP_CHECK_ARG(!(sse_out && !xrec), "sse_out needs xrec");
    if (dtype == PNP_F32)
        return run_dncnn<float>(p, (const float*)z_in, false, 0.0, (float*)z_out, nullptr, (const float*)xrec, sse_out,
                                (hipStream_t)stream, true);
    if (dtype == PNP_F64)
        return run_dncnn<double>(p, (const double*)z_in, false, 0.0, (double*)z_out, nullptr, (const double*)xrec, sse_out,
                                 (hipStream_t)stream, true);
    PNP_CHECK_ARG(false, "bad dtype");
}

extern "C" int pnp_dncnn_denoise(pnp_dncnn_plan* p, const void* z_in, void* z_out, int dtype, double sigma_net,
                                 const void* xrec, double* sse_out, void* stream) {
    PNP_CHECK_ARG(p && z_in && z_out, "null argument");
    PNP_CHECK_ARG(!(sse_out && !xrec), "sse_out needs xrec");
    if (dtype == PNP_F32)
        return run_dncnn<float>(p, (const float*)z_in, true, sigma_net, (float*)z_out, nullptr, (const float*)xrec, sse_out,
                                (hipStream_t)stream);
    if (dtype == PNP_F64)
        return run_dncnn<double>(p, (const double*)z_in, true, sigma_net, (double*)z_out, nullptr, (const double*)xrec,
                                 sse_out, (hipStream_t)stream);
    PNP_CHECK_ARG(false, "bad dtype");
}
