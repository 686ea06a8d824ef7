// dncnn_wino44.hip -- the 64 -> 64 channel 3x3 layer of the DnCNN prox (reference denoisers/DeepDenoisers/model/models.py:
// 13-17: conv + BatchNorm + ReLU, BN folded by the caller) as the TWO-dimensional Winograd minimal-filtering algorithm
// F(4x4, 3x3): per 4x4 output block and (cout, cin)
//     V = B^T d B  (d = the 6x6 input patch)      U = G g G^T  (6x6 per (cout, cin), pre-transformed at plan creation)
//     M_xi = sum_cin U_xi V_xi  (36 multiply-adds per cin for 16 outputs)          Y = A^T M A
// i.e. ONE QUARTER of the direct form's matrix-core work (F(4,3) along x alone: one half); fp32 throughout.
//
// Organisation (one workgroup = 4 waves = one per SIMD, persistent over 8 x 64 output regions in the XCD-aware order; 4 x 64
// regions with half the accumulators for launches that would leave CUs idle -- Geo<NG> below):
//   * the 36 transformed-domain products are 36 independent [64 cout] x [64 cin] x [32 blocks] GEMMs; wave wv owns output
//     channels [16 wv, 16 wv + 16) and keeps ALL 36 x 2 accumulator quads (288 registers) for the region's 2 x 16 blocks;
//   * input channels go by in chunks of 8: the chunk's (8+2) x 72 halo planes arrive by LDS-DMA, every thread transforms
//     ONE (channel, block) patch -- the transform is shared by the four waves, i.e. by all 64 output channels -- and
//     writes its 36 values to the V image [xi][row of blocks][k-row][block][k-step], from which a wave's B operands of
//     one xi are a single conflict-free ds_read_b64 per block row;
//   * transformed weights do not fit registers (36 x 64 x 64): they stream from L2 in MFMA operand order, one 16-byte
//     load per lane and xi pair and chunk through a ring of 18 loads in flight, each value used by two MFMAs (the two
//     block rows);
//   * software pipeline over chunks: while the MFMAs of chunk k run, the same wave transforms chunk k + 1 and the DMA of
//     chunk k + 2 is in flight; one barrier per chunk.
#include "common.h"
#include "wino44.h"
#include "tilewalk.h"
#include <vector>
#include <utility>
#include <cstdlib>

namespace pnp {
namespace w44 {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int C = 64;
constexpr int TC = 64;                                // region width: 16 blocks of 4 x 4
constexpr int PC = 72;                                // LDS row: image columns [tx0 - 4, tx0 + 68) = eighteen 16-byte chunks
constexpr int KC = 8;                                 // input channels per chunk
constexpr int NCH = C / KC;
// NG = block rows per region: 2 (8 x 64 outputs, 72 accumulator quads per wave; the throughput form) or 1 (4 x 64, 36 quads:
// twice the regions for launches that would otherwise leave CUs idle -- a single 256 x 256 image is 128 regions of 8 x 64)
template <int NG_> struct Geo {
    static constexpr int NG = NG_;
    static constexpr int TR = 4 * NG, PR = TR + 2;                 // output rows, halo rows
    static constexpr int PLANE = NG == 2 ? 768 : 512;              // PR x 72 payload + pad: 0 mod 64 dwords (ds_read_b128 lane groups mix two planes)
    static constexpr int DBUF = KC * PLANE;                        // 24 / 16 DMA pieces of 1 KiB
    static constexpr int PPW = DBUF / 256 / 4;                     // 6 / 4 pieces per wave
    static constexpr int VPL = 128 * NG;                           // floats per xi plane of V: [NG block rows][4 k-rows][16 blocks][2 k-steps]
    static constexpr int VBUF = 36 * VPL;
    static constexpr int LDS_FLOATS = 2 * DBUF + 2 * VBUF;         // 120 KiB / 68 KiB
    static constexpr int NQ_AGPR = NG == 2 ? 64 : 36;              // accumulator quads kept in AGPRs (of 72: block row 0 and xi < 28 of row 1; of 36: all)
    static constexpr bool in_agpr(int g, int xi) { return g * 36 + xi < NQ_AGPR; }
};
constexpr int URING = 18;                              // weight loads in flight per lane
constexpr unsigned DUMMY = 1u << 27;                  // descriptor flag: padding chunk of a plane

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
    y[1] = 2.f * d34 + d12;                                 // contracted to (packed) fma
    y[2] = 4.f * s34 + s12;
    y[3] = (8.f * d34 + d12) + m5;
}

// hand-issued MFMAs: the accumulator quad lives in AGPRs (AG) or VGPRs; the first product of a tile takes the constant-zero
// SrcC form.  (Left to the compiler, all 72 quads are sent to the 256 AGPRs and the overflow is shuffled around.)
// hipcc does not look inside inline asm, so its hazard recognizer protects nothing here: if register pressure makes it
// split an accumulator's live range, the copy it puts next to the MFMA reads the 4th register (written by the last pass)
// stale.  Wait states inside every statement cost 30 cycles per MFMA, so instead the kernel keeps the pressure low enough
// that no accumulator is ever moved (all 256 AGPRs are accumulators, the epilogue reads them in small batches) and
// tests/test_cpu_host.py::test_w44_accumulators_untouched checks the generated code for it.
#define PNP_MFMA_PRE ""
#define PNP_MFMA_POST ""
template <bool AG> __device__ __forceinline__ void mfma(f32x4& acc, float w, float v) {
    if (AG) asm volatile(PNP_MFMA_PRE "v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" PNP_MFMA_POST : "+a"(acc) : "v"(w), "v"(v));
    else asm volatile(PNP_MFMA_PRE "v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" PNP_MFMA_POST : "+v"(acc) : "v"(w), "v"(v));
}
template <bool AG> __device__ __forceinline__ void mfma_first(f32x4& acc, float w, float v) {
    if (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" PNP_MFMA_POST : "=&a"(acc) : "v"(w), "v"(v));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" PNP_MFMA_POST : "=&v"(acc) : "v"(w), "v"(v));
}

// ---- memory streams of the main loop ------------------------------------------------------------------------------------------
// * LDS-DMA (activations): inline asm, invisible to the compiler's s_waitcnt insertion -- visible, it makes every LDS read
//   after a DMA wait for vmcnt(0), which drains the weight ring at every step.  One hand-counted vmcnt wait per chunk
//   (vector-memory operations leave the queue in issue order; sched_barriers pin the order of everything else around it).
// * weights: plain loads, a ring of URING 16-byte values per lane; the compiler waits for them itself (its counts do not
//   include the DMA pieces, so its waits are stricter than needed while pieces are in flight, never weaker).
// * LDS: plain reads / writes, software-pipelined in the source (B operands one xi ahead, patch rows one slice ahead).
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) f32x2 lds_f2;
typedef __attribute__((address_space(3))) f32x4 lds_f4;

// one 1-KiB piece global -> LDS: lane's 16 bytes from rsrc.base + voff (an offset beyond num_records reads zeros)
__device__ __forceinline__ void dma_piece_asm(unsigned voff, i32x4 rsrc, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(lds_byte_addr) : "memory");
}

struct Patch {
    f32x2 a; f32x4 m; f32x2 e;                                 // one patch row in flight: LDS columns 4tc + 2..3, 4..7, 8..9
    f32x2 t[6][3];                                             // row transforms as register pairs: (V0, V5), (V1, V2), (V3, V4) of row r
    f32x2 v[6];                                                // one transformed pair of columns on its way to the V image
};
template <int NG_> struct Ctx {
    static constexpr int NG = NG_;
    using G = Geo<NG_>;
    f32x4 acc[NG_][36];
    f32x4 ur[URING];
    Patch P;
    f32x2 b[2][NG_];                                           // B operands of the current / next xi: [parity][block row]
    const lds_f* dsrc[2];                                      // this lane's patch in the two d buffers
    lds_f* vdst[2];                                            // its V item in the two V buffers
    const lds_f* vsrc0[2];                                     // its B operands (block row 0 / 1) in the two V buffers;
    const lds_f* vsrc1[2];                                     //   separate, laundered bases: no ds_read2st64_b64 merging
    // weight stream of this wave: a scalar cursor (1 KiB per load; advanced on the scalar ALU, re-defined through an empty asm so
    // that it stays ONE register pair instead of 144 hoisted addresses) + the lane's 16 bytes as a 32-bit vector offset
    const __attribute__((address_space(1))) char* ucur;
    unsigned ulane;
    __device__ __forceinline__ f32x4 uload_next() {
        const f32x4 u = *(const __attribute__((address_space(1))) f32x4*)(ucur + ulane);
        ucur += 1024;
        asm volatile("" : "+s"(ucur));
        return u;
    }
};
// One LDS read beside an f32 MFMA is free, two in the same gap cost about an MFMA (tools/microbench/mfma_f32_fillers.hip),
// and vector-ALU work is cheapest in blocks; so a step (two xi = 8 MFMAs) has fixed slots, pinned by sched_barriers:
//     M1 | B(xi1)[0] | M2 | B(xi1)[1] | M3 | VALU block of transform slice p | M4 | slice LDS op 1 |
//     M5 | B(xi0')[0] | M6 | B(xi0')[1] | M7 | DMA piece p, slice LDS op 2 | M8 | slice LDS op 3, ring reload
#define PNP_SLOT() __builtin_amdgcn_sched_barrier(0)

// patch row R of d buffer DPAR, in two halves; all ten floats are "used" (slice_valu) so that the reads stay one
// conflict-free ds_read_b128 and two ds_read_b64 (narrowed to the six needed values they become three 4-way
// bank-conflicting ds_read2_b32)
template <int DPAR, int R, int HALF, typename CT> __device__ __forceinline__ void patch_load(CT& c) {
    const lds_f* row = c.dsrc[DPAR] + R * PC;                    // 16-byte aligned
    if (HALF == 0) { c.P.a = *(const lds_f2*)(row + 2); c.P.e = *(const lds_f2*)(row + 8); }
    else c.P.m = *(const lds_f4*)(row + 4);
}
// The transform on the packed-f32 ALU (a v_pk_* beside f32 MFMAs costs what one plain instruction does).  Row pass of
// B^T d B: the loaded row holds (d1, d2) and (d3, d4) as aligned register pairs, so
//     (t2, t1) = (d3, d4) - 4 (d1, d2)      (sd, t3) = (d3, d4) - (d1, d2)
//     (V1, V2) = (t1 + t2, t1 - t2)         (V3, V4) = (t3 + 2 sd, t3 - 2 sd)        [half-selects: op_sel, hand-written]
// and V0, V5 (their inputs straddle the pairs) stay scalar: 8 instructions instead of 12.  The results are kept as the
// pairs (V0, V5), (V1, V2), (V3, V4), so the column pass runs on whole pairs: 12 packed instructions for two columns.
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
// B^T of F(4,3) on six pairs
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
// transform slice SL of the patch (d buffer DPAR -> V buffer DPAR): 0..5 = row transforms, 6 / 8 / 10 = column transforms of
// the column pairs (0, 5) / (1, 2) / (3, 4); the six value pairs wait in P.v for their write slots
constexpr int col_pair(int sl) { return (sl - 6) / 2; }
constexpr bool is_col_slice(int sl) { return sl == 6 || sl == 8 || sl == 10; }
template <int SL, typename CT> __device__ __forceinline__ void slice_valu(CT& c) {
    Patch& P = c.P;
    if constexpr (SL < 6) {
        asm volatile("" :: "v"(P.a.x), "v"(P.e.y));
        const f32x2 p12 = {P.m.x, P.m.y}, p34 = {P.m.z, P.m.w};
        const f32x2 A = p34 - 4.f * p12;                          // (t2, t1)
        const f32x2 Bv = p34 - p12;                               // (sd, t3)
        P.t[SL][1] = pk_sum_diff(A);
        P.t[SL][2] = pk_hi_pm_2lo(Bv);
        // (the last FMA of each as asm with its own destination: hipcc picks the two-address v_fmac_f32 and then moves the
        //  result into the pair)
        float v0, v5;
        asm("v_fma_f32 %0, 4.0, %1, %2" : "=v"(v0) : "v"(P.a.y), "v"(__builtin_fmaf(-5.f, P.m.y, P.m.w)));
        asm("v_fma_f32 %0, 4.0, %1, %2" : "=v"(v5) : "v"(P.m.x), "v"(__builtin_fmaf(-5.f, P.m.z, P.e.x)));
        P.t[SL][0] = f32x2{v0, v5};
    } else if constexpr (is_col_slice(SL)) {
        constexpr int cp = col_pair(SL);
        bt6_pk(P.t[0][cp], P.t[1][cp], P.t[2][cp], P.t[3][cp], P.t[4][cp], P.t[5][cp], P.v);
    }
}
// LDS operation N (0..2) of slice SL: the next patch row's reads (SL < 5) or two of the six pairs of V writes (column slices)
template <int DPAR, int SL, int N, typename CT> __device__ __forceinline__ void slice_lds(CT& c) {
    constexpr int VPL = CT::G::VPL;
    if constexpr (SL < 5) {
        if constexpr (N < 2) patch_load<DPAR, SL + 1, N>(c);
    } else if constexpr (is_col_slice(SL)) {
        constexpr int cp = col_pair(SL), xa = cp == 0 ? 0 : cp == 1 ? 1 : 3, xb = cp == 0 ? 5 : cp == 1 ? 2 : 4;
#pragma unroll
        for (int y = 2 * N; y < 2 * N + 2; ++y) {
            c.vdst[DPAR][(y * 6 + xa) * VPL] = c.P.v[y].x;
            c.vdst[DPAR][(y * 6 + xb) * VPL] = c.P.v[y].y;
        }
    }
}
template <int VPAR, int XI, int GR, typename CT> __device__ __forceinline__ void b_load(CT& c) {
    if constexpr (GR < CT::NG) {
        if (GR == 0) c.b[XI & 1][0] = *(const lds_f2*)(c.vsrc0[VPAR] + XI * CT::G::VPL);
        else c.b[XI & 1][GR] = *(const lds_f2*)(c.vsrc1[VPAR] + XI * CT::G::VPL);
    }
}
// MFMA of block row GR (nothing for a block row the region does not have): first k-step of a xi (constant-zero SrcC in
// chunk 0) / second k-step
template <int K, int XI, int GR, typename CT> __device__ __forceinline__ void mfma_j0(CT& c, float u, f32x2 (&b)[CT::NG]) {
    if constexpr (GR < CT::NG) {
        if constexpr (K == 0) mfma_first<CT::G::in_agpr(GR, XI)>(c.acc[GR][XI], u, b[GR].x);
        else mfma<CT::G::in_agpr(GR, XI)>(c.acc[GR][XI], u, b[GR].x);
    }
}
template <int XI, int GR, typename CT> __device__ __forceinline__ void mfma_j1(CT& c, float u, f32x2 (&b)[CT::NG]) {
    if constexpr (GR < CT::NG) mfma<CT::G::in_agpr(GR, XI)>(c.acc[GR][XI], u, b[GR].y);
}

// step (K, P) of the main loop: xi = 2P, 2P + 1; dma(piece) issues DMA piece `piece` of chunk K + 2.  With one block row
// (NG = 1) the slots of M2, M4, M6, M8 and of their B reads are empty.
// VAR (ablation builds, timing only): 10 = no transform arithmetic, 11 = no DMA, 12 = no weight reloads, 13 = no B reads,
// 14 = no transform LDS traffic, 15 = bare MFMAs
template <int K, int P, int VAR, typename CT, typename DMA> __device__ __forceinline__ void step(CT& c, DMA&& dma) {
    constexpr int X0 = 2 * P, X1 = 2 * P + 1, SQ = K * 18 + P;
    constexpr int VPAR = K & 1, DPAR = (K + 1) & 1, NG = CT::NG;
    const f32x4 u = c.ur[SQ % URING];
    f32x2 b0[NG], b1[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) b0[g] = c.b[0][g];
    mfma_j0<K, X0, 0>(c, u.x, b0);                   PNP_SLOT();          // M1
    if constexpr (VAR != 13 && VAR != 15) b_load<VPAR, X1, 0>(c);
    PNP_SLOT();
    mfma_j0<K, X0, 1>(c, u.x, b0);                   PNP_SLOT();          // M2
    if constexpr (VAR != 13 && VAR != 15) b_load<VPAR, X1, 1>(c);
    PNP_SLOT();
    mfma_j1<X0, 0>(c, u.y, b0);                      PNP_SLOT();          // M3
    if constexpr (VAR != 10 && VAR != 15) slice_valu<P>(c);
    PNP_SLOT();
    mfma_j1<X0, 1>(c, u.y, b0);                      PNP_SLOT();          // M4
    if constexpr (VAR != 14 && VAR != 15) slice_lds<DPAR, P, 0>(c);
    PNP_SLOT();
#pragma unroll
    for (int g = 0; g < NG; ++g) b1[g] = c.b[1][g];
    mfma_j0<K, X1, 0>(c, u.z, b1);                   PNP_SLOT();          // M5
    if constexpr (X1 + 1 < 36 && VAR != 13 && VAR != 15) b_load<VPAR, X1 + 1, 0>(c);
    PNP_SLOT();
    mfma_j0<K, X1, 1>(c, u.z, b1);                   PNP_SLOT();          // M6
    if constexpr (X1 + 1 < 36 && VAR != 13 && VAR != 15) b_load<VPAR, X1 + 1, 1>(c);
    PNP_SLOT();
    mfma_j1<X1, 0>(c, u.w, b1);                      PNP_SLOT();          // M7
    if constexpr (P < CT::G::PPW && VAR != 11 && VAR != 15) dma(P);
    if constexpr (VAR != 14 && VAR != 15) slice_lds<DPAR, P, 1>(c);
    PNP_SLOT();
    mfma_j1<X1, 1>(c, u.w, b1);                      PNP_SLOT();          // M8
    if constexpr (VAR != 14 && VAR != 15) slice_lds<DPAR, P, 2>(c);
    if constexpr (VAR != 12 && VAR != 15 && SQ + URING < NCH * 18) c.ur[SQ % URING] = c.uload_next();
    PNP_SLOT();
}
template <int K, int VAR, typename CT, typename DMA, int... P> __device__ __forceinline__ void chunk_steps(CT& c, DMA&& dma, std::integer_sequence<int, P...>) {
    (step<K, P, VAR>(c, dma), ...);
}
// weight reloads issued in steps p0 .. 17 of chunk K
template <int K> constexpr int reloads_from(int p0) {
    int n = 0;
    for (int p = p0; p < 18; ++p) n += (K * 18 + p + URING < NCH * 18) ? 1 : 0;
    return n;
}
// chunk K of a tile: MFMAs on V buffer K & 1, transform of chunk K + 1, DMA of chunk K + 2
template <int K, bool STAMP, int VAR, typename CT, typename DMA> __device__ __forceinline__ void chunk(CT& c, DMA&& dma, unsigned long long& t_wait) {
    patch_load<(K + 1) & 1, 0, 0>(c);
    patch_load<(K + 1) & 1, 0, 1>(c);
    b_load<K & 1, 0, 0>(c);
    b_load<K & 1, 0, 1>(c);
    PNP_SLOT();
    chunk_steps<K, VAR>(c, dma, std::make_integer_sequence<int, 18>{});
    unsigned long long ta = 0;
    if (STAMP) ta = __builtin_amdgcn_s_memtime();
    // this chunk's DMA pieces have landed: they were issued in steps 0 .. PPW - 1, each before its step's weight reload, so
    // at most the reloads of steps PPW .. 17 may still be in flight (the last chunks of a tile reload nothing: see step())
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(reloads_from<K>(CT::G::PPW)) : "memory");
    __syncthreads();
    if (STAMP) t_wait += __builtin_amdgcn_s_memtime() - ta;
}
template <bool STAMP, int VAR, typename CT, typename MK, int... K> __device__ __forceinline__ void all_chunks(CT& c, MK&& mk, unsigned long long& t_wait, std::integer_sequence<int, K...>) {
    (chunk<K, STAMP, VAR>(c, mk(std::integral_constant<int, K>{}), t_wait), ...);
}

// the first chunk of a workgroup's first tile, outside the pipeline
template <int VPL> __device__ __forceinline__ void transform0(const float* dsrc, float* vdst) {
    float t[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const float* row = dsrc + r * PC;                         // LDS column 4 tc of patch row r
        bt6(row[3], row[4], row[5], row[6], row[7], row[8], t[r]);
    }
#pragma unroll
    for (int x = 0; x < 6; ++x) {
        float v[6];
        bt6(t[0][x], t[1][x], t[2][x], t[3][x], t[4][x], t[5][x], v);
#pragma unroll
        for (int y = 0; y < 6; ++y) vdst[(y * 6 + x) * VPL] = v[y];
    }
}

// STAMP: diagnostic build only (wino44_debug_clock): s_memtime / s_memrealtime around the tile loop, the chunk-end waits and
// the epilogue; the stamps go to their own buffer
template <bool LEAKY, int NG = 2, bool STAMP = false, int VAR = 0>
__global__ __launch_bounds__(256, 1) void k_mid_wino44(const float* __restrict__ in, float* __restrict__ out,
                                                       const float4* __restrict__ upack, const float* __restrict__ bias,
                                                       int H, int W, int ntiles, float slope,
                                                       unsigned long long* __restrict__ stamps = nullptr, int tile0 = 0) {
    using G = Geo<NG>;
    constexpr int TR = G::TR, PR = G::PR, PLANE = G::PLANE, DBUF = G::DBUF, PPW = G::PPW, VPL = G::VPL, VBUF = G::VBUF;
    __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Regions are numbered in units of 8 x 64 pixels (`tile0` + ...) whatever the form: a 4 x 64 region u is sub-row u & 1 of
    // unit tile0 + (u >> 1), so that a launch of the one-row form can take over the units a two-row launch left (the last,
    // partly filled wave of workgroups: wino44_layer)
    const int tiles_x = W / TC, units_per_img = tiles_x * (H / 8);
    auto region = [&](int u, int& b, int& ty0, int& tx0) {
        const int t = tile0 + (NG == 2 ? u : u >> 1);
        b = t / units_per_img;
        const int t2 = t - b * units_per_img;
        ty0 = (t2 / tiles_x) * 8 + (NG == 2 ? 0 : 4 * (u & 1));
        tx0 = (t2 % tiles_x) * TC;
    };

    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[16 * wv + 4 * (lane >> 4) + r];

    // transform item of this thread: block (g, tc), chunk channel 2 wv + j  (k-row wv, k-step j of the MFMA B operand)
    // (with one block row the upper half-wave repeats the lower half's items: same values to the same addresses)
    const int tc = lane & 15, j = (lane >> 4) & 1, g = NG == 2 ? lane >> 5 : 0;
    const int d_off = 4 * ((2 * wv + j) * (PLANE / 4) + g * PC + tc);               // 16-byte aligned
    const ptrdiff_t hw4 = (ptrdiff_t)4 * H * W, w4 = (ptrdiff_t)4 * W;
    const unsigned st_off = 4u * (unsigned)(4 * (lane >> 4) * H * W + 4 * tc);      // output: channel 4 (lane >> 4) of the wave's 16, column 4 tc
    const int v_off = g * 128 + wv * 32 + tc * 2 + j;

    // DMA piece descriptors: bits 0..26 = element offset of the lane's 16-byte chunk inside the chunk's 8 channel planes,
    // bit 27 = padding, bits 28..31 = which image edge would put the chunk outside
    unsigned pdesc[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = (wv + 4 * i) * 64 + lane;
        const int c = q / (PLANE / 4), r = q - c * (PLANE / 4);
        const int ry = r / 18, cx4 = 4 * (r - ry * 18);
        const unsigned edge = (ry == 0 ? 1u : 0u) | (ry == PR - 1 ? 2u : 0u) | (cx4 == 0 ? 4u : 0u) | (cx4 == TC + 4 ? 8u : 0u);
        pdesc[i] = r < PR * 18 ? ((unsigned)((c * H + ry) * W + cx4) | (edge << 28)) : DUMMY;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    // DMA state of a tile t (t == ntiles: "none", every lane reads zeros): the buffer base of its chunk 0 and, per piece, the
    // lane's byte offset (or an offset beyond num_records) -- the same for all chunks of the tile, whose bases are
    // chunk_bytes apart
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
    // piece i of a chunk -> d buffer `buf`
    auto dma_piece = [&](const TileDma& td, i32x4 rs, int buf, int i) {
        dma_piece_asm(td.voff[i], rs, lds0 + 4u * (unsigned)(buf * DBUF + (wv + 4 * i) * 256));
    };

    float* const dbuf = lds;
    float* const vbuf = lds + 2 * DBUF;

    Ctx<NG> c;
    lds_f* const ldsp = (lds_f*)lds;
    c.dsrc[0] = ldsp + d_off;                     c.dsrc[1] = ldsp + DBUF + d_off;
    c.vdst[0] = ldsp + 2 * DBUF + v_off;          c.vdst[1] = ldsp + 2 * DBUF + VBUF + v_off;
    c.vsrc0[0] = ldsp + 2 * DBUF + 2 * lane;      c.vsrc0[1] = ldsp + 2 * DBUF + VBUF + 2 * lane;
    c.vsrc1[0] = c.vsrc0[0] + 128;                c.vsrc1[1] = c.vsrc0[1] + 128;             // (block row 1; unused when NG = 1)
    asm volatile("" : "+v"(c.vsrc1[0]), "+v"(c.vsrc1[1]));
    // transformed weights: the same stream of NCH x 18 16-byte loads per lane for every tile, kept URING loads ahead
    const __attribute__((address_space(1))) char* const ubase = (const __attribute__((address_space(1))) char*)upack + (size_t)(wv * NCH * 18) * 1024;
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
        transform0<VPL>(dbuf + d_off, vbuf + v_off);
        __syncthreads();
    }

    unsigned long long t0 = 0, r0 = 0, t_wait = 0, t_epi = 0;
    if (STAMP) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (; tile < limit; tile += tw_.step) {
        asm volatile("" : "+v"(c.ulane));                                        // the weight loads stay inside the tile loop
        int b, ty0, tx0;
        region(tile, b, ty0, tx0);
        const int ntile = tile + tw_.step < limit ? tile + tw_.step : ntiles;      // ntiles = "none": zeros

        const TileDma cur = tile_dma(tile), nxt = tile_dma(ntile);
        all_chunks<STAMP, VAR>(c, [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            // chunk K + 2 (of this tile, or chunk 0 / 1 of the next) -> the d buffer chunk K was transformed from
            const TileDma& td = K + 2 < NCH ? cur : nxt;
            const i32x4 rs = chunk_rsrc(td, (K + 2) % NCH);
            return [&, rs](int piece) { dma_piece(td, rs, K & 1, piece); };
        }, t_wait, std::make_integer_sequence<int, NCH>{});

        unsigned long long te = 0;
        if (STAMP) te = __builtin_amdgcn_s_memtime();
        asm volatile("; W44_EPILOGUE_BEGIN\n\ts_nop 15\n\ts_nop 7" ::: "memory");      // MFMA write -> VALU read distance
        // ... which only holds if no read of an accumulator is scheduled above it: to hipcc the MFMA results are ready where
        // the asm statements stand, and it hoists the epilogue's v_accvgpr_reads right behind them.  Every accumulator is
        // therefore re-defined here by an empty asm (ordered behind the wait states; no code).
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2)
#pragma unroll
            for (int xi = 0; xi < 36; xi += 4) {
                if (G::in_agpr(g2, xi)) asm volatile("" : "+a"(c.acc[g2][xi]), "+a"(c.acc[g2][xi + 1]), "+a"(c.acc[g2][xi + 2]), "+a"(c.acc[g2][xi + 3]));
                else asm volatile("" : "+v"(c.acc[g2][xi]), "+v"(c.acc[g2][xi + 1]), "+v"(c.acc[g2][xi + 2]), "+v"(c.acc[g2][xi + 3]));
            }
        // epilogue: Y = A^T M A, bias, ReLU; a lane holds block (g2, tc) of channels 16 wv + 4 (lane >> 4) + i.  Two
        // channels at a time on the packed-f32 ALU (the halves of an accumulator quad are register pairs), two columns /
        // rows per scheduling region so that at most two dozen accumulator copies are in flight; the bias enters through
        // M[1][1], whose weight is 1 in all sixteen outputs.
        // stores: scalar base (tile, wave, channel pair, row: scalar ALU) + the lane's 32-bit byte offset
        unsigned so = st_off;                 // (re-defined inside the loop: its zero-extension must sit next to the stores for
        asm volatile("" : "+v"(so));          //  instruction selection to fold it into the scalar-base addressing mode)
        __attribute__((address_space(1))) char* const ob = (__attribute__((address_space(1))) char*)out + 4 * ((((size_t)b * C + 16 * wv) * H + ty0) * (size_t)W + tx0);
#pragma unroll
        for (int blk = 0; blk < 2 * NG; ++blk) {
            {
                const int g2 = NG == 2 ? 1 - (blk >> 1) : 0, pi = blk & 1;   // block row 1 first: it frees the eight VGPR accumulator quads
                // no weight load is in flight across the epilogue (a spilled in-flight load costs its whole latency); the
                // ring's first entries of the next tile go out before the last block
                if (blk == 2 * NG - 1) {
                    c.ucur = ubase;
#pragma unroll
                    for (int i = 0; i < URING; ++i) c.ur[i] = c.uload_next();
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x2 bias2 = {bv[2 * pi], bv[2 * pi + 1]};
                // store cursor: one scalar register pair walked over the block's eight rows (re-defined through empty asms: left
                // alone, hipcc keeps all 32 store bases in scalar registers and spills them)
                __attribute__((address_space(1))) char* oc = ob + (ptrdiff_t)(2 * pi) * hw4 + (ptrdiff_t)(4 * g2) * w4;
                asm volatile("" : "+s"(oc));
                f32x2 s[6][4];                                  // s[x][r]: A^T along y of column x
#pragma unroll
                for (int x = 0; x < 6; ++x) {
                    f32x2 m[6];
#pragma unroll
                    for (int y = 0; y < 6; ++y) {
                        const f32x4 q = c.acc[g2][6 * y + x];
                        m[y] = pi ? f32x2{q.z, q.w} : f32x2{q.x, q.y};
                    }
                    if (x == 1) m[1] += bias2;
                    at6(m[0], m[1], m[2], m[3], m[4], m[5], s[x]);
                    if (x & 1) __builtin_amdgcn_sched_barrier(0);          // (two columns / rows per scheduling region: -0.35 % against one)
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f32x2 y[4];
                    at6(s[0][r], s[1][r], s[2][r], s[3][r], s[4][r], s[5][r], y);
                    f32x4 v0, v1;
                    if (LEAKY) {
                        v0.x = fmaxf(y[0].x, slope * y[0].x); v0.y = fmaxf(y[1].x, slope * y[1].x);
                        v0.z = fmaxf(y[2].x, slope * y[2].x); v0.w = fmaxf(y[3].x, slope * y[3].x);
                        v1.x = fmaxf(y[0].y, slope * y[0].y); v1.y = fmaxf(y[1].y, slope * y[1].y);
                        v1.z = fmaxf(y[2].y, slope * y[2].y); v1.w = fmaxf(y[3].y, slope * y[3].y);
                    } else {
                        v0.x = fmaxf(y[0].x, 0.f); v0.y = fmaxf(y[1].x, 0.f); v0.z = fmaxf(y[2].x, 0.f); v0.w = fmaxf(y[3].x, 0.f);
                        v1.x = fmaxf(y[0].y, 0.f); v1.y = fmaxf(y[1].y, 0.f); v1.z = fmaxf(y[2].y, 0.f); v1.w = fmaxf(y[3].y, 0.f);
                    }
                    *(__attribute__((address_space(1))) f32x4*)(oc + so) = v0;          // channel 2 pi, row 4 g2 + r
                    oc += hw4;
                    asm volatile("" : "+s"(oc));
                    *(__attribute__((address_space(1))) f32x4*)(oc + so) = v1;          // channel 2 pi + 1
                    oc += w4 - hw4;
                    asm volatile("" : "+s"(oc));
                    if (r & 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("; W44_EPILOGUE_END" ::: "memory");
        if (STAMP) t_epi += __builtin_amdgcn_s_memtime() - te;
    }
    if (STAMP && tid == 0) {
        stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
        stamps[4 * blockIdx.x + 2] = t_wait;
        stamps[4 * blockIdx.x + 3] = t_epi;
    }
}

}  // namespace w44

bool wino44_supports(int H, int W) { return H % 8 == 0 && W % w44::TC == 0; }

size_t wino44_weight_floats(int n_mid) { return (size_t)n_mid * 4 * w44::NCH * 18 * 64 * 4; }

// w_mid [n_mid][64][64][3][3] (BN folded) -> upack[l][wv][chunk k][xi pair p][lane][e]:  xi = 2 p + (e >> 1), k-step j = e & 1,
// U_xi[cout = 16 wv + (lane & 15)][cin = 8 k + 2 (lane >> 4) + j],  U = G g G^T,  xi = 6 xi_y + xi_x
void wino44_pack_weights(const float* w_mid, int n_mid, float* out) {
    static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (int l = 0; l < n_mid; ++l)
        for (int wv = 0; wv < 4; ++wv)
            for (int k = 0; k < w44::NCH; ++k)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 2; ++j) {
                        const int cout = 16 * wv + (lane & 15), cin = w44::KC * k + 2 * (lane >> 4) + j;
                        const float* g = w_mid + (((size_t)l * w44::C + cout) * w44::C + cin) * 9;
                        for (int xy = 0; xy < 6; ++xy)
                            for (int xx = 0; xx < 6; ++xx) {
                                double u = 0;
                                for (int dy = 0; dy < 3; ++dy)
                                    for (int dx = 0; dx < 3; ++dx) u += G[xy][dy] * G[xx][dx] * (double)g[dy * 3 + dx];
                                const int xi = 6 * xy + xx, p = xi >> 1, e = 2 * (xi & 1) + j;
                                out[((((size_t)(l * 4 + wv) * w44::NCH + k) * 18 + p) * 64 + lane) * 4 + e] = (float)u;
                            }
                    }
}

int wino44_layer(const float* in, float* out, const float* upack_layer, const float* bias, const float* zeros, int H, int W,
                 int batch, int num_cu, float slope, hipStream_t s, int force) {
    // 8 x 64 regions (72 accumulator quads per wave) in full waves of one region per CU; what is left -- a launch smaller than
    // the chip, or the last, partly filled wave -- goes through the 4 x 64 form, twice as many regions of half the work (one
    // 256 x 256 image: 256 regions instead of 128; three images: 256 + 256 instead of 384 in two waves).  Same bits either way.
    // force = 1 / 2 (test hook pnp_dncnn_debug_mid_layer only) takes one form for the whole layer.
    const int units = batch * (H / 8) * (W / w44::TC);
    int full = force == 1 ? 0 : force == 2 ? units : (units / num_cu) * num_cu;              // units done as 8 x 64 regions
    if (force == 0 && 2 * (units - full) > num_cu) full = units;      // (more than half a wave left: one more 8 x 64 wave is cheaper than two 4 x 64 waves)
    const float4* up = (const float4*)upack_layer;
    if (full > 0) {
        const int grid = full < num_cu ? full : num_cu;
        if (slope != 0.f) w44::k_mid_wino44<true, 2><<<grid, 256, 0, s>>>(in, out, up, bias, H, W, full, slope, nullptr, 0);
        else w44::k_mid_wino44<false, 2><<<grid, 256, 0, s>>>(in, out, up, bias, H, W, full, 0.f, nullptr, 0);
        PNP_CHECK_LAUNCH();
    }
    if (full < units) {
        const int n1 = 2 * (units - full), grid = n1 < num_cu ? n1 : num_cu;
        if (slope != 0.f) w44::k_mid_wino44<true, 1><<<grid, 256, 0, s>>>(in, out, up, bias, H, W, n1, slope, nullptr, full);
        else w44::k_mid_wino44<false, 1><<<grid, 256, 0, s>>>(in, out, up, bias, H, W, n1, 0.f, nullptr, full);
        PNP_CHECK_LAUNCH();
    }
    return PNP_OK;
}

// diagnostic: `reps` back-to-back launches, the last one stamped; per workgroup {shader cycles, 100 MHz ticks, cycles in the
// chunk-end waits + barriers, cycles in the epilogue} of the tile loop
int wino44_debug_clock(const float* in, float* out, const float* upack_layer, const float* bias, int H, int W, int batch,
                       int num_cu, int reps, unsigned long long* stamps_dev, hipStream_t s) {
    const int ntiles = batch * (H / 8) * (W / w44::TC);
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    for (int i = 0; i < reps - 1; ++i)
        w44::k_mid_wino44<false, 2><<<grid, 256, 0, s>>>(in, out, (const float4*)upack_layer, bias, H, W, ntiles, 0.f);
    const int var = getenv("PNP_W44_VAR") ? atoi(getenv("PNP_W44_VAR")) : 0;
    if (var == 0) w44::k_mid_wino44<false, 2, true><<<grid, 256, 0, s>>>(in, out, (const float4*)upack_layer, bias, H, W, ntiles, 0.f, stamps_dev);
#ifdef PNP_W44_ABLATIONS   // timing-only builds (wrong results): 10 = no transform
                           // arithmetic, 11 = no DMA, 12 = no weight reloads, 13 = no B reads, 14 = no transform LDS traffic, 15 = bare MFMAs
#define PNP_W44_ABL(V) else if (var == V) w44::k_mid_wino44<false, 2, true, V><<<grid, 256, 0, s>>>(in, out, (const float4*)upack_layer, bias, H, W, ntiles, 0.f, stamps_dev);
    PNP_W44_ABL(10) PNP_W44_ABL(11) PNP_W44_ABL(12) PNP_W44_ABL(13) PNP_W44_ABL(14) PNP_W44_ABL(15)
#undef PNP_W44_ABL
#endif
    else PNP_CHECK_ARG(false, "PNP_W44_VAR: this library was built without -DPNP_W44_ABLATIONS");
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

}  // namespace pnp
