// dncnn_wino44b.hip -- the 64 -> 64 channel 3x3 layer of the DnCNN prox as the two-dimensional Winograd algorithm F(4x4, 3x3) of
// dncnn_wino44.hip, with the 36 transformed-domain GEMMs on the BF16 matrix cores at fp32-class accuracy: every fp32 factor is
// split EXACTLY into three bf16 terms (a = a1 + a2 + a3, 8 + 8 + 8 significant bits, by truncation) and the six products that carry
// more than 2^-24 of a*b are summed in the fp32 accumulator:
//     a*b ~ a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)
// Opt-in (conv mode 6): fp32-class, but not the reference's arithmetic operation for operation (reference
// denoisers/DeepDenoisers/model/models.py:13-17 runs torch's fp32 conv).
//
// Organisation -- what differs from the fp32 kernel and why:
//   * v_mfma_f32_32x32x16_bf16: K = 16 = two half-waves x 8 input channels, so a chunk of 8 channels and its six product terms
//     are exactly three MFMAs per transform point:  (a1 | a2) x (b1 | b1),  (a1 | a2) x (b2 | b2),  (a3 | a1) x (b1 | b3)
//     -- the weight operands (a1 | a2), (a3 | a1) come pre-split from L2, two 16-byte loads per lane, point and chunk; the
//     activation operands are 16-byte reads of the planes b1, b2, b3 of the V image;
//   * a 32 x 32 accumulator tile is 16 registers, 36 points would be 576: the four waves are 2 halves of the output channels x
//     2 halves of the transform points (rows y < 3 / y >= 3 of the 6 x 6 points), 18 x 16 = 288 accumulators per lane as before
//     (256 AGPRs + 32 VGPRs).  The output transform Y = A^T M A sums over all 36 points: each wave runs the pass along x on its
//     three rows, the two waves of a pair exchange HALF of the results through LDS (the V buffer the last chunk has released)
//     and each finishes the pass along y, bias, ReLU and the stores for half of the pair's 32 channels;
//   * the V image holds bf16 triples: [point][plane][block][8 channels] = 6 bytes per value, 2 x 54 KB, which with the two
//     24 KB input buffers of the LDS-DMA is 156 KB of the CU's 160 KB.  A thread transforms one (channel, block) patch as
//     before; the terms are the high halves of (v, v - v1, v - v1 - v2), paired with the neighbour channel's by v_permlane16_swap
//     and stored four bytes at a time (put_pair).
// Everything else (persistent workgroups in the XCD-aware order, LDS-DMA of the halo planes with range-checked padding, the
// patch transform on the packed-f32 ALU, the chunk pipeline MFMA(k) | transform(k + 1) | DMA(k + 2)) is as in dncnn_wino44.hip.
#include "common.h"
#include "wino44b.h"
#include "tilewalk.h"
#include <vector>
#include <utility>
#include <cstdlib>
#include <cstring>

namespace pnp {
namespace w44b {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int C = 64;
constexpr int TC = 64;                                // region: 8 rows x 64 columns = 2 x 16 blocks of 4 x 4
constexpr int TR = 8, PR = TR + 2;
constexpr int PC = 72;                                // LDS row: image columns [tx0 - 4, tx0 + 68)
constexpr int KC = 8;                                 // input channels per chunk
constexpr int NCH = C / KC;
constexpr int PLANE = 768;                            // PR x 72 payload + pad (0 mod 64 dwords)
constexpr int DBUF = KC * PLANE;                      // floats per input buffer: 24 DMA pieces of 1 KiB
constexpr int PPW = DBUF / 256 / 4;                   // 6 pieces per wave
constexpr int VPLB = 512;                             // bytes per plane of a point: [32 blocks][8 channels] bf16
constexpr int VXI = 3 * VPLB;                         // bytes per transform point
constexpr int VBUF = 36 * VXI;                        // 55 296 bytes
constexpr int D_BYTES = 2 * DBUF * 4;                 // 49 152
constexpr int LDS_BYTES = D_BYTES + 2 * VBUF + 256;   // + the layer's 64 biases: 160 000 of 163 840
constexpr int URING = 20;                             // weight loads in flight per lane
constexpr int NLOADS = NCH * 18 * 2;                  // weight loads per region and lane
constexpr unsigned DUMMY = 1u << 27;                  // descriptor flag: padding chunk of a plane

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) f32x2 lds_f2;
typedef __attribute__((address_space(3))) f32x4 lds_f4;
typedef __attribute__((address_space(3))) char lds_c;
typedef __attribute__((address_space(3))) unsigned short lds_u16;

// B^T of F(4,3) applied to six values
__device__ __forceinline__ void bt6(float d0, float d1, float d2, float d3, float d4, float d5, float (&v)[6]) {
    const float t1 = __builtin_fmaf(-4.f, d2, d4), t2 = __builtin_fmaf(-4.f, d1, d3);
    const float t3 = d4 - d2, sd = d3 - d1;
    v[0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
    v[1] = t1 + t2;
    v[2] = t1 - t2;
    v[3] = __builtin_fmaf(2.f, sd, t3);
    v[4] = __builtin_fmaf(-2.f, sd, t3);
    v[5] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
}
// A^T of F(4,3) applied to six values
template <typename T> __device__ __forceinline__ void at6(T m0, T m1, T m2, T m3, T m4, T m5, T (&y)[4]) {
    const T s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y[0] = (m0 + s12) + s34;
    y[1] = 2.f * d34 + d12;
    y[2] = 4.f * s34 + s12;
    y[3] = (8.f * d34 + d12) + m5;
}

// hand-issued MFMAs (see dncnn_wino44.hip: left to the compiler the 288 accumulators do not stay put); the three MFMAs of a
// point are one accumulation chain on the same 16 registers, which the matrix pipe runs back to back
template <bool AG> __device__ __forceinline__ void mfma(f32x16& acc, f32x4 a, f32x4 b) {
    if (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <bool AG> __device__ __forceinline__ void mfma_first(f32x16& acc, f32x4 a, f32x4 b) {
    if (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
constexpr bool in_agpr(int t) { return t < 16; }

// one 1-KiB piece global -> LDS: lane's 16 bytes from rsrc.base + voff (an offset beyond num_records reads zeros)
__device__ __forceinline__ void dma_piece_asm(unsigned voff, i32x4 rsrc, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(lds_byte_addr) : "memory");
}

struct Patch {
    f32x2 a; f32x4 m; f32x2 e;                                 // one patch row in flight: LDS columns 4tc + 2..3, 4..7, 8..9
    f32x2 t[6][3];                                             // row transforms as register pairs: (V0, V5), (V1, V2), (V3, V4) of row r
    f32x2 v[6];                                                // one transformed pair of columns on its way to the V image
};
struct Ctx {
    f32x16 acc[18];
    f32x4 ur[URING];
    Patch P;
    f32x4 b[2][3];                                             // B operands of the current / next point: [parity][(b1|b1), (b2|b2), (b1|b3)]
    const lds_f* dsrc[2];                                      // this lane's patch in the two d buffers
    lds_c* vdst[2];                                            // where its 4-byte stores go in the two V buffers (put_pair)
    const lds_c* vsrcA[2];                                     // its B operands: plane b1 of the wave's first point, block lane & 31 ...
    const lds_c* vsrcB[2];                                     //   ... and plane b1 (lower half-wave) / b3 (upper) for the third MFMA
    const __attribute__((address_space(1))) char* ucur;        // weight stream of this wave: scalar cursor (1 KiB per load) ...
    unsigned ulane;                                            //   ... + the lane's 16 bytes
    __device__ __forceinline__ f32x4 uload_next() {
        const f32x4 u = *(const __attribute__((address_space(1))) f32x4*)(ucur + ulane);
        ucur += 1024;
        asm volatile("" : "+s"(ucur));
        return u;
    }
};
#define PNP_SLOT() __builtin_amdgcn_sched_barrier(0)

template <int DPAR, int R, int HALF> __device__ __forceinline__ void patch_load(Ctx& c) {
    const lds_f* row = c.dsrc[DPAR] + R * PC;                    // 16-byte aligned
    if (HALF == 0) { c.P.a = *(const lds_f2*)(row + 2); c.P.e = *(const lds_f2*)(row + 8); }
    else c.P.m = *(const lds_f4*)(row + 4);
}
__device__ __forceinline__ f32x2 pk_sum_diff(f32x2 a) {             // (a.lo + a.hi, a.hi - a.lo)
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ f32x2 pk_hi_pm_2lo(f32x2 a) {            // (a.hi + 2 a.lo, a.hi - 2 a.lo)
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, 2.0, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ void bt6_pk(f32x2 q0, f32x2 q1, f32x2 q2, f32x2 q3, f32x2 q4, f32x2 q5, f32x2 (&v)[6]) {
    const f32x2 t1 = q4 - 4.f * q2, t2 = q3 - 4.f * q1;
    const f32x2 t3 = q4 - q2, sd = q3 - q1;
    v[0] = 4.f * q0 + (q4 - 5.f * q2);
    v[1] = t1 + t2;
    v[2] = t1 - t2;
    v[3] = 2.f * sd + t3;
    v[4] = t3 - 2.f * sd;
    v[5] = 4.f * q1 + (q5 - 5.f * q3);
}
// The exact three-way split of a transformed value: the high halves of v, r1 = v - hi(v) and r2 = r1 - hi(r1) (both differences are
// exact; r2 has at most 8 significant bits left) are the bf16 terms.
// A lane transforms ONE channel; 2-byte stores of single terms put four lanes on every LDS bank (the loop was bound by the LDS
// write port).  So two values of the lane, P at point XI and Q at point XI + 18, first go through v_permlane16_swap with the lane
// 16 further (the same block, the next channel): the even row ends up with both channels of point XI, the odd row with both
// channels of point XI + 18, and every term is ONE 4-byte store per lane.  `vdst` carries the row's part of the address: the odd
// row's 18 points further and two dwords rotated inside the 16-byte item -- points 18..35 hold their channels in the order
// 4..7, 0..3 (wino44b_pack_weights packs the weights of the waves that own those points the same way), which puts the two rows of
// a store on different banks.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
template <int XI> __device__ __forceinline__ void put_pair(lds_c* vdst, float P, float Q) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(P), __float_as_uint(Q), false, false);
    const u32x2 u0 = {(unsigned)sw[0], (unsigned)sw[1]};                        // (channel 2 wv, channel 2 wv + 1)
    const f32x2 v0 = __builtin_bit_cast(f32x2, u0);
    const f32x2 r1 = v0 - __builtin_bit_cast(f32x2, u0 & 0xFFFF0000u);
    const u32x2 u1 = __builtin_bit_cast(u32x2, r1);
    const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, u1 & 0xFFFF0000u);
    const u32x2 u2 = __builtin_bit_cast(u32x2, r2);
    *(__attribute__((address_space(3))) unsigned*)(vdst + XI * VXI) = __builtin_amdgcn_perm(u0.y, u0.x, 0x07060302u);
    *(__attribute__((address_space(3))) unsigned*)(vdst + XI * VXI + VPLB) = __builtin_amdgcn_perm(u1.y, u1.x, 0x07060302u);
    *(__attribute__((address_space(3))) unsigned*)(vdst + XI * VXI + 2 * VPLB) = __builtin_amdgcn_perm(u2.y, u2.x, 0x07060302u);
}
// the same split for ONE value and 2-byte stores (the first chunk of a workgroup's first region, outside the pipeline)
template <int XI> __device__ __forceinline__ void put_v(lds_c* vlo, lds_c* vhi, float v) {
    lds_c* const vdst = XI < 18 ? vlo : vhi;
    const unsigned u0 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u0 & 0xFFFF0000u);
    const unsigned u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
    const unsigned u2 = __float_as_uint(r2);
    *(lds_u16*)(vdst + XI * VXI) = (unsigned short)(u0 >> 16);
    *(lds_u16*)(vdst + XI * VXI + VPLB) = (unsigned short)(u1 >> 16);
    *(lds_u16*)(vdst + XI * VXI + 2 * VPLB) = (unsigned short)(u2 >> 16);
}
// Transform schedule of a chunk over the 18 steps of the MFMA loop: steps 0..5 = row transform of patch row T (and the reads of
// row T + 1), steps 6 + 4 cp + q (cp = 0..2, q = 0..3) = column pair cp: (0, 5) / (1, 2) / (3, 4); its column transform at
// q = 0; its twelve values leave as six (y, y + 3) pairs: two pairs in each of the steps q = 0, 1, one in q = 2, 3
constexpr int col_pair(int t) { return (t - 6) / 4; }
constexpr int col_q(int t) { return (t - 6) % 4; }
template <int T> __device__ __forceinline__ void slice_valu(Ctx& c) {
    Patch& P = c.P;
    if constexpr (T < 6) {
        asm volatile("" :: "v"(P.a.x), "v"(P.e.y));
        const f32x2 p12 = {P.m.x, P.m.y}, p34 = {P.m.z, P.m.w};
        const f32x2 A = p34 - 4.f * p12;                          // (t2, t1)
        const f32x2 Bv = p34 - p12;                               // (sd, t3)
        P.t[T][1] = pk_sum_diff(A);
        P.t[T][2] = pk_hi_pm_2lo(Bv);
        float v0, v5;
        asm("v_fma_f32 %0, 4.0, %1, %2" : "=v"(v0) : "v"(P.a.y), "v"(__builtin_fmaf(-5.f, P.m.y, P.m.w)));
        asm("v_fma_f32 %0, 4.0, %1, %2" : "=v"(v5) : "v"(P.m.x), "v"(__builtin_fmaf(-5.f, P.m.z, P.e.x)));
        P.t[T][0] = f32x2{v0, v5};
    } else if constexpr (col_q(T) == 0) {
        constexpr int cp = col_pair(T);
        bt6_pk(P.t[0][cp], P.t[1][cp], P.t[2][cp], P.t[3][cp], P.t[4][cp], P.t[5][cp], P.v);
    }
}
template <int DPAR, int T> __device__ __forceinline__ void slice_lds(Ctx& c) {
    if constexpr (T < 5) {
        patch_load<DPAR, T + 1, 0>(c);
        patch_load<DPAR, T + 1, 1>(c);
    } else if constexpr (T >= 6) {
        constexpr int cp = col_pair(T), q = col_q(T), xa = cp == 0 ? 0 : cp == 1 ? 1 : 3, xb = cp == 0 ? 5 : cp == 1 ? 2 : 4;
        constexpr int u0 = q < 2 ? 2 * q : q + 2, nu = q < 2 ? 2 : 1;         // pair index u = 2 y + (0: column xa, 1: column xb), y = 0..2
        {
            constexpr int y = u0 >> 1, x = (u0 & 1) ? xb : xa;
            put_pair<y * 6 + x>(c.vdst[DPAR], (u0 & 1) ? c.P.v[y].y : c.P.v[y].x, (u0 & 1) ? c.P.v[y + 3].y : c.P.v[y + 3].x);
        }
        if constexpr (nu == 2) {
            constexpr int u1 = u0 + 1, y = u1 >> 1, x = (u1 & 1) ? xb : xa;
            put_pair<y * 6 + x>(c.vdst[DPAR], (u1 & 1) ? c.P.v[y].y : c.P.v[y].x, (u1 & 1) ? c.P.v[y + 3].y : c.P.v[y + 3].x);
        }
    }
}
// B operand R (0: (b1|b1), 1: (b2|b2), 2: (b1|b3)) of the wave's point T from V buffer VPAR
template <int VPAR, int T, int R> __device__ __forceinline__ void b_load(Ctx& c) {
    if (R == 0) c.b[T & 1][0] = *(const lds_f4*)(c.vsrcA[VPAR] + T * VXI);
    else if (R == 1) c.b[T & 1][1] = *(const lds_f4*)(c.vsrcA[VPAR] + T * VXI + VPLB);
    else c.b[T & 1][2] = *(const lds_f4*)(c.vsrcB[VPAR] + T * VXI);
}

// step (K, T) of the main loop: the wave's point T of chunk K; dma(piece) issues DMA piece `piece` of chunk K + 2
// VAR (ablation builds, timing only): 10 = no transform arithmetic, 11 = no DMA, 12 = no weight reloads, 13 = no B reads,
// 14 = no V writes (nor their split arithmetic), 15 = bare MFMAs, 16 = no patch reads
template <int K, int T, int VAR, typename DMA> __device__ __forceinline__ void step(Ctx& c, DMA&& dma) {
    constexpr int SQ = K * 18 + T, VPAR = K & 1, DPAR = (K + 1) & 1;
    const f32x4 ua = c.ur[(2 * SQ) % URING], ub = c.ur[(2 * SQ + 1) % URING];
    const f32x4 b0 = c.b[T & 1][0], b1 = c.b[T & 1][1], b2 = c.b[T & 1][2];
    if constexpr (K == 0) mfma_first<in_agpr(T)>(c.acc[T], ua, b0);
    else mfma<in_agpr(T)>(c.acc[T], ua, b0);
    PNP_SLOT();
    if constexpr (T + 1 < 18 && VAR != 13 && VAR != 15) b_load<VPAR, T + 1, 0>(c);
    if constexpr (VAR != 10 && VAR != 15) slice_valu<T>(c);
    PNP_SLOT();
    mfma<in_agpr(T)>(c.acc[T], ua, b1);
    PNP_SLOT();
    if constexpr (T + 1 < 18 && VAR != 13 && VAR != 15) b_load<VPAR, T + 1, 1>(c);
    if constexpr (T == 0 && VAR != 11 && VAR != 15) {
        // all of the chunk's DMA pieces at once, in front of every weight load the chunk issues: a weight load cannot retire before
        // the pieces ahead of it in the wave's queue have come back, so what counts is when the LAST piece goes out
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma(i);
    }
    if constexpr (VAR != 15 && !(VAR == 14 && T >= 6) && !(VAR == 16 && T < 5)) slice_lds<DPAR, T>(c);
    PNP_SLOT();
    mfma<in_agpr(T)>(c.acc[T], ub, b2);
    PNP_SLOT();
    if constexpr (T + 1 < 18 && VAR != 13 && VAR != 15) b_load<VPAR, T + 1, 2>(c);
    if constexpr (2 * SQ + URING + 1 < NLOADS && VAR != 12 && VAR != 15) {
        c.ur[(2 * SQ) % URING] = c.uload_next();
        c.ur[(2 * SQ + 1) % URING] = c.uload_next();
    }
    PNP_SLOT();
}
template <int K, int VAR, int T0, typename DMA, int... T> __device__ __forceinline__ void chunk_steps(Ctx& c, DMA&& dma, std::integer_sequence<int, T...>) {
    (step<K, T0 + T, VAR>(c, dma), ...);
}
// weight loads issued behind the chunk's DMA pieces (step 0: the pieces go out before that step's two reloads)
template <int K> constexpr int reloads_behind_dma() {
    int n = 0;
    for (int p = 0; p < 18; ++p) n += (2 * (K * 18 + p) + URING + 1 < NLOADS) ? 2 : 0;
    return n;
}
// chunk K of a tile: MFMAs on V buffer K & 1, transform of chunk K + 1, DMA of chunk K + 2
template <int K, bool STAMP, int VAR, typename DMA> __device__ __forceinline__ void chunk(Ctx& c, DMA&& dma, unsigned long long& t_wait, unsigned long long& t_rows) {
    patch_load<(K + 1) & 1, 0, 0>(c);
    patch_load<(K + 1) & 1, 0, 1>(c);
    b_load<K & 1, 0, 0>(c);
    b_load<K & 1, 0, 1>(c);
    b_load<K & 1, 0, 2>(c);
    PNP_SLOT();
    unsigned long long tb = 0;
    if (STAMP) tb = __builtin_amdgcn_s_memtime();
    chunk_steps<K, VAR, 0>(c, dma, std::make_integer_sequence<int, 6>{});
    if (STAMP) t_rows += __builtin_amdgcn_s_memtime() - tb;
    chunk_steps<K, VAR, 6>(c, dma, std::make_integer_sequence<int, 12>{});
    unsigned long long ta = 0;
    if (STAMP) ta = __builtin_amdgcn_s_memtime();
    // this chunk's DMA pieces have landed: vector-memory operations leave the queue in issue order
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VAR == 12 || VAR == 15 ? 0 : reloads_behind_dma<K>()) : "memory");
    __syncthreads();
    if (STAMP) t_wait += __builtin_amdgcn_s_memtime() - ta;
}
template <bool STAMP, int VAR, typename MK, int... K> __device__ __forceinline__ void all_chunks(Ctx& c, MK&& mk, unsigned long long& t_wait, unsigned long long& t_rows, std::integer_sequence<int, K...>) {
    (chunk<K, STAMP, VAR>(c, mk(std::integral_constant<int, K>{}), t_wait, t_rows), ...);
}

// the first chunk of a workgroup's first tile, outside the pipeline
template <int X> __device__ __forceinline__ void transform0_col(const float (&t)[6][6], lds_c* vlo, lds_c* vhi) {
    float v[6];
    bt6(t[0][X], t[1][X], t[2][X], t[3][X], t[4][X], t[5][X], v);
    put_v<X>(vlo, vhi, v[0]); put_v<6 + X>(vlo, vhi, v[1]); put_v<12 + X>(vlo, vhi, v[2]);
    put_v<18 + X>(vlo, vhi, v[3]); put_v<24 + X>(vlo, vhi, v[4]); put_v<30 + X>(vlo, vhi, v[5]);
}
// (vlo / vhi: the lane's 2-byte slot in a plane of a point < 18 / >= 18 -- see put_pair for the channel order)
__device__ __forceinline__ void transform0(const float* dsrc, lds_c* vlo, lds_c* vhi) {
    float t[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const float* row = dsrc + r * PC;                         // LDS column 4 tc of patch row r
        bt6(row[3], row[4], row[5], row[6], row[7], row[8], t[r]);
    }
    transform0_col<0>(t, vlo, vhi); transform0_col<1>(t, vlo, vhi); transform0_col<2>(t, vlo, vhi);
    transform0_col<3>(t, vlo, vhi); transform0_col<4>(t, vlo, vhi); transform0_col<5>(t, vlo, vhi);
}

// ---- epilogue of a region, for the wave's half XH of the transform points (rows y = 3 XH + y') --------------------------------------
// Lane l holds block n = l & 31 and, in accumulator register r, output channel 32 ch + 8 (r >> 2) + (r & 3) + 4 (l >> 5).
// Register pairs rp = (2 rp, 2 rp + 1); the wave FINISHES pairs 4 XH .. 4 XH + 3 and sends its x-pass results of the other four to
// its partner (wave ^ 2), two pairs per round through `xbuf` (the V buffer the last chunk released; 12 KiB per wave and round:
// [y'][output column i][lane][pair k] as 8-byte items, so that the partner reads 16 bytes per (y', i)).
// Register pressure is what shapes it: one A^T pass at a time, results written to LDS or stored at once, scheduling barriers
// between the groups (left alone, hipcc reads all 288 accumulators first and spills: 49 k instead of 8 k cycles per region).
struct AccQ { f32x4 q[18][4]; };                                // the accumulators as quads: q[t][r >> 2]
template <int XH, int YY, int RP> __device__ __forceinline__ void xpass(AccQ& c, const lds_f* bias_l, f32x2 (&t4)[4]) {
    f32x2 m[6];
#pragma unroll
    for (int x = 0; x < 6; ++x) {
        const f32x4 q = c.q[6 * YY + x][RP >> 1];
        m[x] = (RP & 1) ? f32x2{q.z, q.w} : f32x2{q.x, q.y};
    }
    if (XH == 0 && YY == 1) m[1] += *(const lds_f2*)(bias_l + 8 * (RP >> 1) + 2 * (RP & 1));   // the bias through M[1][1]
    at6(m[0], m[1], m[2], m[3], m[4], m[5], t4);
}
template <int XH, int RD, int K, int YY> __device__ __forceinline__ void send_one(AccQ& c, const lds_f* bias_l, lds_c* xw) {
    f32x2 t4[4];
    xpass<XH, YY, 4 * (1 - XH) + 2 * RD + K>(c, bias_l, t4);
#pragma unroll
    for (int i = 0; i < 4; ++i) *(lds_f2*)(xw + (YY * 4 + i) * 1024 + K * 8) = t4[i];
    __builtin_amdgcn_sched_barrier(0);
}
template <int XH, int RD, int K, bool LEAKY>
__device__ __forceinline__ void finish_one(AccQ& c, const lds_f* bias_l, const lds_c* xr, __attribute__((address_space(1))) char* ob, unsigned so,
                                           ptrdiff_t hw4, ptrdiff_t w4, float slope) {
    constexpr int RP = 4 * XH + 2 * RD + K, R0 = 2 * RP;                // accumulator registers R0, R0 + 1
    f32x2 Tm[3][4], Tp[3][4];
#pragma unroll
    for (int yy = 0; yy < 3; ++yy)
#pragma unroll
        for (int i = 0; i < 4; ++i) Tp[yy][i] = *(const lds_f2*)(xr + (yy * 4 + i) * 1024 + K * 8);
    xpass<XH, 2, RP>(c, bias_l, Tm[2]);
    __builtin_amdgcn_sched_barrier(0);
    xpass<XH, 1, RP>(c, bias_l, Tm[1]);
    __builtin_amdgcn_sched_barrier(0);
    xpass<XH, 0, RP>(c, bias_l, Tm[0]);
    __builtin_amdgcn_sched_barrier(0);
    f32x2 y[4][4];                                                       // [output column i][output row o]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (XH == 0) at6(Tm[0][i], Tm[1][i], Tm[2][i], Tp[0][i], Tp[1][i], Tp[2][i], y[i]);
        else at6(Tp[0][i], Tp[1][i], Tp[2][i], Tm[0][i], Tm[1][i], Tm[2][i], y[i]);
    }
    // channel of register R0: 8 (R0 >> 2) + (R0 & 3) (+ the lane's 4 h, in `so`); R0 + 1: the next channel
    __attribute__((address_space(1))) char* oc = ob + (ptrdiff_t)(8 * (R0 >> 2) + (R0 & 3)) * hw4;
    asm volatile("" : "+s"(oc));
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        f32x4 v0, v1;
        if (LEAKY) {
            v0.x = fmaxf(y[0][o].x, slope * y[0][o].x); v0.y = fmaxf(y[1][o].x, slope * y[1][o].x);
            v0.z = fmaxf(y[2][o].x, slope * y[2][o].x); v0.w = fmaxf(y[3][o].x, slope * y[3][o].x);
            v1.x = fmaxf(y[0][o].y, slope * y[0][o].y); v1.y = fmaxf(y[1][o].y, slope * y[1][o].y);
            v1.z = fmaxf(y[2][o].y, slope * y[2][o].y); v1.w = fmaxf(y[3][o].y, slope * y[3][o].y);
        } else {
            v0.x = fmaxf(y[0][o].x, 0.f); v0.y = fmaxf(y[1][o].x, 0.f); v0.z = fmaxf(y[2][o].x, 0.f); v0.w = fmaxf(y[3][o].x, 0.f);
            v1.x = fmaxf(y[0][o].y, 0.f); v1.y = fmaxf(y[1][o].y, 0.f); v1.z = fmaxf(y[2][o].y, 0.f); v1.w = fmaxf(y[3][o].y, 0.f);
        }
        *(__attribute__((address_space(1))) f32x4*)(oc + so) = v0;
        oc += hw4;
        asm volatile("" : "+s"(oc));
        *(__attribute__((address_space(1))) f32x4*)(oc + so) = v1;
        oc += w4 - hw4;
        asm volatile("" : "+s"(oc));
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int XH, int RD, bool LEAKY>
__device__ __forceinline__ void epilogue_round(Ctx& cx, AccQ& c, lds_c* xw, const lds_c* xr, const lds_f* bias_l, __attribute__((address_space(1))) char* ob,
                                               unsigned so, ptrdiff_t hw4, ptrdiff_t w4, float slope,
                                               const __attribute__((address_space(1))) char* ubase) {
    send_one<XH, RD, 0, 2>(c, bias_l, xw); send_one<XH, RD, 1, 2>(c, bias_l, xw);          // row 2 first: it holds the two VGPR accumulators
    send_one<XH, RD, 0, 1>(c, bias_l, xw); send_one<XH, RD, 1, 1>(c, bias_l, xw);
    send_one<XH, RD, 0, 0>(c, bias_l, xw); send_one<XH, RD, 1, 0>(c, bias_l, xw);
    __syncthreads();
    finish_one<XH, RD, 0, LEAKY>(c, bias_l, xr, ob, so, hw4, w4, slope);
    if (RD == 1) {
        // the ring's first entries of the next tile go out before the last block (no weight load is in flight while most of the
        // accumulators are read: a spilled in-flight load costs its whole latency)
        cx.ucur = ubase;
#pragma unroll
        for (int i = 0; i < URING; ++i) cx.ur[i] = cx.uload_next();
        __builtin_amdgcn_sched_barrier(0);
    }
    finish_one<XH, RD, 1, LEAKY>(c, bias_l, xr, ob, so, hw4, w4, slope);
    __syncthreads();                                                     // the exchange buffer is free again
}
template <int XH, bool LEAKY>
__device__ __forceinline__ void epilogue(Ctx& cx, AccQ& c, lds_c* xbuf, const lds_f* bias_l, int wv, int lane,
                                         __attribute__((address_space(1))) char* ob, unsigned so, ptrdiff_t hw4, ptrdiff_t w4, float slope,
                                         const __attribute__((address_space(1))) char* ubase) {
    lds_c* const xw = xbuf + ((wv * 12) * 64 + lane) * 16;
    const lds_c* const xr = xbuf + (((wv ^ 2) * 12) * 64 + lane) * 16;
    epilogue_round<XH, 0, LEAKY>(cx, c, xw, xr, bias_l, ob, so, hw4, w4, slope, ubase);
    epilogue_round<XH, 1, LEAKY>(cx, c, xw, xr, bias_l, ob, so, hw4, w4, slope, ubase);
}

// STAMP: diagnostic build only (wino44b_debug_clock)
template <bool LEAKY, bool STAMP = false, int VAR = 0>
__global__ __launch_bounds__(256, 1) void k_mid_wino44b(const float* __restrict__ in, float* __restrict__ out,
                                                        const uint4* __restrict__ upack, const float* __restrict__ bias,
                                                        int H, int W, int ntiles, float slope,
                                                        unsigned long long* __restrict__ stamps = nullptr) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wv & 1, xh = wv >> 1;                           // the wave's half of the output channels / of the transform points
    const int tiles_x = W / TC, units_per_img = tiles_x * (H / 8);
    auto region = [&](int t, int& b, int& ty0, int& tx0) {
        b = t / units_per_img;
        const int t2 = t - b * units_per_img;
        ty0 = (t2 / tiles_x) * 8;
        tx0 = (t2 % tiles_x) * TC;
    };
    lds_c* const ldsp = (lds_c*)lds;
    lds_f* const bias_lds = (lds_f*)(ldsp + D_BYTES + 2 * VBUF);
    if (tid < C) bias_lds[tid] = bias[tid];

    // transform item of this thread: block (g, tc), chunk channel 2 wv + j
    const int tc = lane & 15, j = (lane >> 4) & 1, g = lane >> 5;
    const int d_off = 4 * ((2 * wv + j) * (PLANE / 4) + g * PC + tc);               // floats; 16-byte aligned
    const ptrdiff_t hw4 = (ptrdiff_t)4 * H * W, w4 = (ptrdiff_t)4 * W;
    // V stores (put_pair): the lane's row of 16 (2 g + j) holds, after the swap, channels 2 wv, 2 wv + 1 of block (g, tc) for a
    // point < 18 (j = 0) or 18 further (j = 1: two dwords rotated inside the item)
    const int v_off = (g * 16 + tc) * 16 + ((wv + 2 * j) & 3) * 4 + j * 18 * VXI;
    // MFMA / epilogue role of the lane: block n = lane & 31 (block row n >> 4, column n & 15), half-wave h = lane >> 5
    const int n = lane & 31, h = lane >> 5;
    const unsigned st_off = 4u * (unsigned)((4 * h) * H * W + (4 * (n >> 4)) * W + 4 * (n & 15));

    unsigned pdesc[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = (wv + 4 * i) * 64 + lane;
        const int c = q / (PLANE / 4), r = q - c * (PLANE / 4);
        const int ry = r / 18, cx4 = 4 * (r - ry * 18);
        const unsigned edge = (ry == 0 ? 1u : 0u) | (ry == PR - 1 ? 2u : 0u) | (cx4 == 0 ? 4u : 0u) | (cx4 == TC + 4 ? 8u : 0u);
        pdesc[i] = r < PR * 18 ? ((unsigned)((c * H + ry) * W + cx4) | (edge << 28)) : DUMMY;
    }
    const unsigned lds0 = (unsigned)(size_t)ldsp;
    struct TileDma { size_t base; unsigned voff[PPW]; };
    const size_t chunk_bytes = (size_t)KC * H * W * 4;
    auto tile_dma = [&](int t) {
        TileDma td;
        int b, ty0, tx0;
        region(t, b, ty0, tx0);
        td.base = (size_t)in + 4 * ((((size_t)b * C) * H + ty0 - 1) * (size_t)W + tx0 - 4);
        const unsigned bad = t < ntiles ? ((((ty0 == 0 ? 1u : 0u) | (ty0 + TR == H ? 2u : 0u) | (tx0 == 0 ? 4u : 0u) | (tx0 + TC == W ? 8u : 0u)) << 28) | DUMMY)
                                        : 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < PPW; ++i) td.voff[i] = (pdesc[i] & bad) == 0u ? 4u * (pdesc[i] & 0x07FFFFFFu) : 0x80000000u;
        return td;
    };
    auto chunk_rsrc = [&](const TileDma& td, int k) {
        const size_t base = td.base + (size_t)k * chunk_bytes;
        i32x4 rs;
        rs.x = (int)(unsigned)base; rs.y = (int)(unsigned)(base >> 32) & 0xFFFF; rs.z = (int)0x80000000u; rs.w = 0x00020000;
        return rs;
    };
    auto dma_piece = [&](const TileDma& td, i32x4 rs, int buf, int i) {
        dma_piece_asm(td.voff[i], rs, lds0 + 4u * (unsigned)(buf * DBUF + (wv + 4 * i) * 256));
    };

    Ctx c;
    c.dsrc[0] = (const lds_f*)ldsp + d_off;              c.dsrc[1] = (const lds_f*)ldsp + DBUF + d_off;
    c.vdst[0] = ldsp + D_BYTES + v_off;                  c.vdst[1] = ldsp + D_BYTES + VBUF + v_off;
    c.vsrcA[0] = ldsp + D_BYTES + xh * 18 * VXI + n * 16; c.vsrcA[1] = c.vsrcA[0] + VBUF;
    c.vsrcB[0] = c.vsrcA[0] + h * 2 * VPLB;              c.vsrcB[1] = c.vsrcB[0] + VBUF;
    asm volatile("" : "+v"(c.vsrcB[0]), "+v"(c.vsrcB[1]));
    const __attribute__((address_space(1))) char* const ubase = (const __attribute__((address_space(1))) char*)upack + (size_t)wv * NLOADS * 1024;
    c.ucur = ubase;
    c.ulane = 16u * lane;

    const TileWalk tw_ = tile_walk(ntiles);
    int tile = tw_.first;
    const int limit = tw_.limit < ntiles ? tw_.limit : ntiles;
    {
        const TileDma td0 = tile_dma(tile < limit ? tile : ntiles);
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_piece(td0, chunk_rsrc(td0, 0), 0, i);
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_piece(td0, chunk_rsrc(td0, 1), 1, i);
#pragma unroll
        for (int i = 0; i < URING; ++i) c.ur[i] = c.uload_next();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        lds_c* const v0 = ldsp + D_BYTES + (g * 16 + tc) * 16;
        transform0((const float*)lds + d_off, v0 + (2 * wv + j) * 2, v0 + ((2 * wv + j + 4) & 7) * 2);
        __syncthreads();
    }

    unsigned long long t0 = 0, r0 = 0, t_epi = 0, t_wait = 0, t_rows = 0;
    if (STAMP) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (; tile < limit; tile += tw_.step) {
        asm volatile("" : "+v"(c.ulane));                                        // the weight loads stay inside the tile loop
        int b, ty0, tx0;
        region(tile, b, ty0, tx0);
        const int ntile = tile + tw_.step < limit ? tile + tw_.step : ntiles;      // ntiles = "none": zeros

        const TileDma cur = tile_dma(tile), nxt = tile_dma(ntile);
        all_chunks<STAMP, VAR>(c, [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            const TileDma& td = K + 2 < NCH ? cur : nxt;
            const i32x4 rs = chunk_rsrc(td, (K + 2) % NCH);
            return [&, rs](int piece) { dma_piece(td, rs, K & 1, piece); };
        }, t_wait, t_rows, std::make_integer_sequence<int, NCH>{});

        unsigned long long te = 0;
        if (STAMP) te = __builtin_amdgcn_s_memtime();
        asm volatile("; W44B_EPILOGUE_BEGIN\n\ts_nop 15\n\ts_nop 7" ::: "memory");      // MFMA write -> VALU read distance
        // every accumulator is re-defined here by an empty asm, ordered behind the wait states (dncnn_wino44.hip)
        // ... and taken apart into quads of their own (a 16-register accumulator read element by element makes hipcc copy and
        // spill whole tuples)
        AccQ aq;
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            if (in_agpr(t)) asm volatile("" : "+a"(c.acc[t]));
            else asm volatile("" : "+v"(c.acc[t]));
            aq.q[t][0] = __builtin_shufflevector(c.acc[t], c.acc[t], 0, 1, 2, 3);
            aq.q[t][1] = __builtin_shufflevector(c.acc[t], c.acc[t], 4, 5, 6, 7);
            aq.q[t][2] = __builtin_shufflevector(c.acc[t], c.acc[t], 8, 9, 10, 11);
            aq.q[t][3] = __builtin_shufflevector(c.acc[t], c.acc[t], 12, 13, 14, 15);
            if (in_agpr(t)) asm volatile("" : "+a"(aq.q[t][0]), "+a"(aq.q[t][1]), "+a"(aq.q[t][2]), "+a"(aq.q[t][3]));
            else asm volatile("" : "+v"(aq.q[t][0]), "+v"(aq.q[t][1]), "+v"(aq.q[t][2]), "+v"(aq.q[t][3]));
        }
        unsigned so = st_off;
        asm volatile("" : "+v"(so));
        __attribute__((address_space(1))) char* const ob = (__attribute__((address_space(1))) char*)out + 4 * ((((size_t)b * C + 32 * ch) * H + ty0) * (size_t)W + tx0);
        lds_c* const xbuf = ldsp + D_BYTES + VBUF;                               // V buffer 1: chunk 7's, released by its barrier
        const lds_f* const bias_l = bias_lds + 32 * ch + 4 * h;
        if (xh == 0) epilogue<0, LEAKY>(c, aq, xbuf, bias_l, wv, lane, ob, so, hw4, w4, slope, ubase);
        else epilogue<1, LEAKY>(c, aq, xbuf, bias_l, wv, lane, ob, so, hw4, w4, slope, ubase);
        asm volatile("; W44B_EPILOGUE_END" ::: "memory");
        if (STAMP) t_epi += __builtin_amdgcn_s_memtime() - te;
    }
    if (STAMP && tid == 0) {
        stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
        stamps[4 * blockIdx.x + 2] = t_wait + (t_rows << 32);          // (two counters in one word: steps 0..5 of the chunks above bit 32)
        stamps[4 * blockIdx.x + 3] = t_epi;
    }
}

}  // namespace w44b

bool wino44b_supports(int H, int W) { return H % 8 == 0 && W % w44b::TC == 0; }

size_t wino44b_weight_halfwords(int n_mid) { return (size_t)n_mid * 4 * w44b::NLOADS * 64 * 8; }

// w_mid [n_mid][64][64][3][3] (BN folded) -> upack[l][wave][chunk k][point t][operand ab][lane][8 channels] (bf16 bit patterns):
// wave = 2 xh + ch, point xi = 18 xh + t, cout = 32 ch + (lane & 31), cin = 8 k + ((e + 4 xh) & 7); operand 0 = (a1 | a2), 1 = (a3 | a1) by
// half-wave (lane >> 5), with U = G g G^T evaluated in float64, rounded to fp32 (the fp32 kernel's value) and split by truncation
void wino44b_pack_weights(const float* w_mid, int n_mid, uint16_t* out) {
    static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    auto flt = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    for (int l = 0; l < n_mid; ++l)
        for (int cout = 0; cout < w44b::C; ++cout)
            for (int cin = 0; cin < w44b::C; ++cin) {
                const float* g = w_mid + (((size_t)l * w44b::C + cout) * w44b::C + cin) * 9;
                const int ch = cout >> 5, r = cout & 31, k = cin / w44b::KC, e = cin % w44b::KC;
                for (int xy = 0; xy < 6; ++xy)
                    for (int xx = 0; xx < 6; ++xx) {
                        double u = 0;
                        for (int dy = 0; dy < 3; ++dy)
                            for (int dx = 0; dx < 3; ++dx) u += G[xy][dy] * G[xx][dx] * (double)g[dy * 3 + dx];
                        const float a = (float)u;
                        const uint32_t u1 = bits(a) & 0xFFFF0000u;
                        const float r1 = a - flt(u1);
                        const uint32_t u2 = bits(r1) & 0xFFFF0000u;
                        const float r2 = r1 - flt(u2);
                        const uint16_t a1 = (uint16_t)(u1 >> 16), a2 = (uint16_t)(u2 >> 16), a3 = (uint16_t)(bits(r2) >> 16);
                        const int xi = 6 * xy + xx, xh = xi / 18, t = xi % 18, wave = 2 * xh + ch;
                        const int es = (e + 4 * xh) & 7;                          // points 18..35 keep their channels in the order 4..7, 0..3 (put_pair)
                        const size_t base = ((((size_t)(l * 4 + wave) * w44b::NCH + k) * 18 + t) * 2) * 512;
                        out[base + (size_t)r * 8 + es] = a1;                      // operand 0, lower half-wave
                        out[base + (size_t)(32 + r) * 8 + es] = a2;               // operand 0, upper half-wave
                        out[base + 512 + (size_t)r * 8 + es] = a3;                // operand 1, lower
                        out[base + 512 + (size_t)(32 + r) * 8 + es] = a1;         // operand 1, upper
                    }
            }
}

int wino44b_layer(const float* in, float* out, const uint16_t* upack_layer, const float* bias, int H, int W, int batch, int num_cu,
                  float slope, hipStream_t s) {
    const int units = batch * (H / 8) * (W / w44b::TC);
    const int grid = units < num_cu ? units : num_cu;
    const uint4* up = (const uint4*)upack_layer;
    if (slope != 0.f) w44b::k_mid_wino44b<true><<<grid, 256, 0, s>>>(in, out, up, bias, H, W, units, slope, nullptr);
    else w44b::k_mid_wino44b<false><<<grid, 256, 0, s>>>(in, out, up, bias, H, W, units, 0.f, nullptr);
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

// diagnostic: `reps` back-to-back launches, the last one stamped; per workgroup {shader cycles, 100 MHz ticks, cycles in the
// chunk-end waits + barriers, cycles in the epilogue}
int wino44b_debug_clock(const float* in, float* out, const uint16_t* upack_layer, const float* bias, int H, int W, int batch,
                        int num_cu, int reps, unsigned long long* stamps_dev, hipStream_t s) {
    const int ntiles = batch * (H / 8) * (W / w44b::TC);
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    for (int i = 0; i < reps - 1; ++i)
        w44b::k_mid_wino44b<false><<<grid, 256, 0, s>>>(in, out, (const uint4*)upack_layer, bias, H, W, ntiles, 0.f, nullptr);
    const int var = getenv("PNP_W44_VAR") ? atoi(getenv("PNP_W44_VAR")) : 0;
    if (var == 0) w44b::k_mid_wino44b<false, true><<<grid, 256, 0, s>>>(in, out, (const uint4*)upack_layer, bias, H, W, ntiles, 0.f, stamps_dev);
#ifdef PNP_W44_ABLATIONS   // timing-only builds (wrong results)
#define PNP_W44_ABL(V) else if (var == V) w44b::k_mid_wino44b<false, true, V><<<grid, 256, 0, s>>>(in, out, (const uint4*)upack_layer, bias, H, W, ntiles, 0.f, stamps_dev);
    PNP_W44_ABL(10) PNP_W44_ABL(11) PNP_W44_ABL(12) PNP_W44_ABL(13) PNP_W44_ABL(14) PNP_W44_ABL(15) PNP_W44_ABL(16)
#undef PNP_W44_ABL
#endif
    else PNP_CHECK_ARG(false, "PNP_W44_VAR: this library was built without -DPNP_W44_ABLATIONS");
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

}  // namespace pnp
