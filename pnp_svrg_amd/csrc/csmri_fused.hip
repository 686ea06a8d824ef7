// csmri_fused.hip -- one whole inner iteration of pnp_svrg on CSMRI in ONE kernel (f32, 256 x 256):
//
//     z <- prox_TV( alpha * Re ifft2( sel o fft2(a - b) ) + beta * c1 + gamma * c2 )        [+ noise estimate, PSNR error]
//
// i.e. reference algorithms/pnp_svrg.py:52-80 (minibatch SVRG direction via problems/CSMRI.py:83-89, step,
// estimate_sigma, TVDenoiser.denoise, Problem.PSNR) with a = z, b = w, c1 = z, c2 = mu.  The four streaming kernels of
// csmri.hip / prox.hip move the half spectrum through HBM three times and the stepped image once (3.3 MB per problem-
// iteration against 2.4 MB algorithmic); here ONE workgroup owns one image and keeps it in registers from the first load
// to the last store, so HBM sees only the operands: a, b, c1, c2, xrec in, z out (1.5 MB).
//
// 512 threads = 8 wavefronts = two per SIMD, i.e. 256 VGPRs per lane: 128 of them hold the image (a 256 x 256 f32 image
// is 128 registers x 512 lanes), the rest is working space -- at 1024 threads (128 VGPRs) the same kernel spilled ~300
// registers per lane to scratch and the spill traffic alone exceeded the HBM traffic it was meant to save (measured).
//
// Register layouts of the image (all hand-overs between them go through the CU's LDS, half an image = 128 KB at a time):
//   R  "rows"    : every lane holds 16-byte pieces of rows -- row pair rp = 16 pass + 2 wave + (lane >> 5), columns
//                  4 cb .. 4 cb + 3 and 128 + 4 cb .., cb = lane & 31, for the 8 passes.  EVERY global access of the kernel
//                  is made in this layout: one dwordx4 per lane, a half-wave covering 512 contiguous bytes.  (Round 2 loaded
//                  and stored in the F and C layouts, four bytes per lane in 64-byte segments: in-kernel clock stamps showed
//                  53 % of a workgroup's time in those phases at ~10 B/clk/CU -- the quad rate of dword accesses, not HBM.)
//   F  "FFT"     : fft.h's lane + 16 r layout: lane group g (16 lanes) holds the complex row-pair signal
//                  Z[p][r] = (row 2 rp, row 2 rp + 1) at column l + 16 r, rp = 32 p + g
//   C  "columns" : the prox's layout: wave wv owns image columns [16 wv, 16 wv + 16) and 128 + the same; lane = column + 16 q
//                  keeps rows [64 q, 64 q + 64) of both
// Phases:
//   1  operands     : a, b in R, a - b, R -> F
//   1' rows forward : a group packs two real rows into one complex FFT-256
//   2  columns      : the 256 KiB raw spectrum does not fit LDS, so it crosses in two halves of 128 k-space columns
//                     chosen so that kx and W - kx travel together (the split of the packed transforms needs both):
//                     row side writes [kx][row pair]; every group takes two column pairs, splits each into the true
//                     half-spectrum column (256 points), FFT -> selector weights (bit-packed mask o minibatch) -> inverse
//                     FFT in registers, re-packs and writes back; row side reads its entries back
//   3  rows inverse : one complex inverse FFT per row pair = two real rows; F -> R; epilogue alpha*g + beta*c1 + gamma*c2 in R
//                     (gradient-only mode stores here)
//   4  re-layout    : R -> C
//   5  prox         : prox_tv.h -- per-column MAD noise estimate, Haar BayesShrink; C -> R; squared error, store
// The FFT scratch of a lane group is private to it and the group lies inside one wavefront, so the in-FFT exchanges need
// no workgroup barrier (a wavefront's LDS operations complete in order); barriers separate only the phases that hand data
// between wavefronts.  DENOISE = false stops after the noise estimate and stores the stepped image (the DnCNN prox takes
// over from there).
#include "fft.h"
#include <type_traits>
#include <cstdlib>

// Diagnostic build (-DPNP_FUSED_CLOCK, tools/fused_clock.py): thread 0 of every workgroup stamps the shader clock at the phase
// boundaries, so the phases can be timed in steady state (workgroups of a full launch are out of step with each other, unlike
// the PNP_FUSED_STOP builds where every CU is in the same phase and the loads of all of them saturate HBM together).
#ifdef PNP_FUSED_CLOCK
#define PNP_STAMP_MAXB 4096
__device__ unsigned long long g_fused_stamps[PNP_STAMP_MAXB * 16];
// the stamps stay in scalar registers until the end of the kernel (stamping through memory raised the vector-register pressure
// enough for hipcc to spill a ground-truth load that was still in flight -- tools/check_fused_isa.py -DPNP_FUSED_CLOCK)
#define PNP_STAMP_DECL unsigned long long stamp_[16] = {0}
#define PNP_STAMP_PARAM , unsigned long long (&stamp_)[16]
#define PNP_STAMP_ARG , stamp_
#define PNP_STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp_[k] = __builtin_readcyclecounter(); asm volatile("" ::: "memory"); } while (0)
#define PNP_STAMP_NW(k) do { asm volatile("" ::: "memory"); stamp_[k] = __builtin_readcyclecounter(); asm volatile("" ::: "memory"); } while (0)
#define PNP_STAMP_FLUSH do { if (threadIdx.x == 0 && blockIdx.x < PNP_STAMP_MAXB) { _Pragma("unroll") for (int k_ = 0; k_ < 16; ++k_) g_fused_stamps[blockIdx.x * 16 + k_] = stamp_[k_]; } } while (0)
extern "C" int pnp_debug_fused_stamps(unsigned long long* host_out, int nblocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fused_stamps), (size_t)nblocks * 16 * sizeof(unsigned long long));
}
#else
#define PNP_STAMP_DECL
#define PNP_STAMP_PARAM
#define PNP_STAMP_ARG
#define PNP_STAMP(k) do { } while (0)
#define PNP_STAMP_NW(k) do { } while (0)
#define PNP_STAMP_FLUSH do { } while (0)
#endif

namespace pnp {

constexpr int FN = 256;                                   // image side
constexpr int FT = 512;                                   // threads per workgroup
constexpr int FG = FT / 16;                               // 32 lane groups
constexpr int FP = (FN / 2) / FG;                         // 4 row-pair passes per group
constexpr int F_SCR = 16 * 17;                            // complex elements of one group's FFT scratch
constexpr int F_RS = 129;                                 // row stride (complex) of the transposition buffer [128 kx][129]
constexpr size_t F_LDS_BYTES = (size_t)128 * F_RS * sizeof(cx<float>);   // 132 096 B >= 32 scratches and 128*257*4

// FFT-256 of one lane group (16 lanes x 16 registers, element lane + 16 r, natural order in and out); twiddles from an
// LDS table; group-private scratch; no workgroup barrier (see header).
template <bool INV>
__device__ __forceinline__ void fft256(cx<float> (&v)[16], const cx<float>* twl, cx<float>* scr, int lane) {
    dft_reg<float, 16, INV>(v);
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        const cx<float> tw = twl[(lane * r) & 255];
        v[r] = cmul(v[r], INV ? cconj(tw) : tw);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[r * 17 + lane] = v[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = scr[lane * 17 + r];
    __builtin_amdgcn_wave_barrier();
    dft_reg<float, 16, INV>(v);
}

__device__ __forceinline__ float dpp_xor1(float v) {       // value of lane ^ 1 (quad_perm [1,0,3,2]), VALU only
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// which half a k-space column travels in, and its row in the transposition buffer
__device__ __forceinline__ bool in_half(int kx, int half) {
    const bool a = kx <= 63 || kx == 128 || kx >= 193;
    return half == 0 ? a : !a;
}
// the same for kx = l + 16 r with the row index r a compile-time constant: fourteen of the sixteen r belong to one half
// for every lane (only kx = 128 and kx = 192, lane 0 of r = 8 and r = 12, sit on the other side), and said so the
// compiler sees that those registers of Z are dead between the hand-over and the reload -- with a per-lane predicate
// on every r it kept all 128 alive through both halves and spilled 80 registers around the column transforms
__device__ __forceinline__ bool in_half_r(int r, int l, int half) {
    if (r == 8) return (l == 0) == (half == 0);
    if (r == 12) return (l == 0) == (half == 1);
    return (r < 4 || r > 12) == (half == 0);
}
__device__ __forceinline__ int kx_local(int kx, int half) {
    if (half == 0) return kx <= 63 ? kx : (kx == 128 ? 64 : kx - 128);           // 0..63, 64, 65..127
    return kx <= 127 ? kx - 64 : kx - 65;                                        // 64..127 -> 0..63, 129..192 -> 64..127
}

// One column pair (ca, cb = W - ca; packed: ca = 0, cb = W/2) of the raw spectrum in the transposition buffer: split
// into the true half-spectrum column, FFT along h, selector weights, inverse FFT, Hermitian re-expansion.  `col_load` /
// `col_store` touch the buffer (the caller puts workgroup barriers around the part between them).
struct ColPair { int la, lb; bool packed; };

__device__ __forceinline__ void col_load(cx<float> (&v)[16], const cx<float>* ldc, const ColPair& c, int l) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // h = l + 16 r: row pair rp = h >> 1; even lanes fetch column ca, odd lanes column cb, then trade
        const int rp = (l + 16 * r) >> 1;
        const cx<float> own = ldc[((l & 1) ? c.lb : c.la) * F_RS + rp];
        const cx<float> oth = {dpp_xor1(own.x), dpp_xor1(own.y)};
        const cx<float> zk = (l & 1) ? oth : own, zm = (l & 1) ? own : oth;
        if (c.packed) {
            // packed column: (kx = 0, kx = 128) of row 2rp as (re, im) on even lanes, of row 2rp+1 on odd lanes
            v[r] = (l & 1) ? cx<float>{zk.y, zm.y} : cx<float>{zk.x, zm.x};
        } else {
            // split of the two packed real rows: A (row 2rp) on even lanes, B (row 2rp+1) on odd lanes
            v[r] = (l & 1) ? cx<float>{0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x)}
                           : cx<float>{0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};
        }
    }
}

__device__ __forceinline__ void col_transform(cx<float> (&v)[16], const cx<float>* twl, cx<float>* scr, const uint32_t* sb,
                                              const ColPair& c, int l, const cx<float>* __restrict__ yhc) {
    fft256<false>(v, twl, scr, l);                          // along h: element ky = l + 16 r
    auto bit = [&](int slot, int ky) -> float { return (float)((sb[slot * 8 + (ky >> 5)] >> (ky & 31)) & 1u); };
    if (c.packed) {
        // the packed column holds two real-input transforms: separate, weight, re-pack (k_cols, blockIdx.x == 0)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[l + 16 * r] = v[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ky = l + 16 * r, km = (FN - ky) & (FN - 1);
            const cx<float> pk = v[r], pm = scr[km];
            const cx<float> A = {0.5f * (pk.x + pm.x), 0.5f * (pk.y - pm.y)};
            const cx<float> B = {0.5f * (pk.y + pm.y), -0.5f * (pk.x - pm.x)};
            const float wA = 0.5f * (bit(0, ky) + bit(0, km)), wB = 0.5f * (bit(1, ky) + bit(1, km));
            v[r] = {fma_(wA, A.x, -(wB * B.y)), fma_(wA, A.y, wB * B.x)};
        }
        __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ky = l + 16 * r, km = (FN - ky) & (FN - 1);
            const float wgt = 0.5f * (bit(0, ky) + bit(1, km));
            v[r] = {wgt * v[r].x, wgt * v[r].y};
        }
    }
    if (yhc != nullptr) {                                   // packed data term of this half-spectrum column (pnp_csmri_pack_y)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = csub(v[r], yhc[l + 16 * r]);
    }
    fft256<true>(v, twl, scr, l);                           // back to h = l + 16 r
}

__device__ __forceinline__ void col_store(const cx<float> (&v)[16], cx<float>* ldc, const ColPair& c, int l) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // Hermitian re-expansion (k_rows_inv): even lane has q.a,q.b (row 2rp), odd lane q.c,q.d (row 2rp+1)
        const int rp = (l + 16 * r) >> 1;
        const cx<float> own = v[r], oth = {dpp_xor1(v[r].x), dpp_xor1(v[r].y)};
        cx<float> o;
        if (c.packed) o = (l & 1) ? cx<float>{oth.y, own.y} : cx<float>{own.x, oth.x};       // zp[128] | zp[0]
        else o = (l & 1) ? cx<float>{oth.x + own.y, own.x - oth.y}                            // zp[W - kx] = {a + d, c - b}
                         : cx<float>{own.x - oth.y, own.y + oth.x};                           // zp[kx]     = {a - d, b + c}
        ldc[((l & 1) ? c.lb : c.la) * F_RS + rp] = o;
    }
}

// ---------------------------------------------------------------------------------------------- layouts and crossings
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }

// Global accesses in the R layout are HAND-ISSUED (inline asm, as the LDS-DMA of the conv kernel): left to hipcc, the 16-byte
// loads of an operand batch were serialised against the next batch (it re-used their destination registers and waited), and
// under register pressure every ground-truth load was followed by `s_waitcnt vmcnt(0)` and a spill -- the phases ran at the
// old 4-byte rate.  Issued by hand, a batch goes out back to back, the next one is requested before the current one is
// consumed, and ONE counted wait stands in front of each use:
//   * address = wave-uniform base (scalar register pair) + the lane's 32-bit byte offset + immediate;
//   * hipcc knows nothing of the loads in flight: gwait() is `s_waitcnt vmcnt(N)` followed by an empty asm that re-defines the
//     batch's registers, so no use can be scheduled above the wait.  N = the number of hand-issued vector-memory operations
//     issued AFTER the batch (operations leave the queue in issue order; whatever else hipcc has in the queue -- scratch
//     traffic -- only makes the wait longer, never shorter);
//   * a destination register must not be read (or spilled) between its load and its wait: tools/check_fused_isa.py verifies
//     that on the generated code (CPU test test_fused_loads_untouched).
template <int IMM> __device__ __forceinline__ void gld(f4& dst, const float* sbase, unsigned voff) {
    // (s_nop 4: hipcc may have moved the uniform base into its scalar registers with a v_readfirstlane right in front of this
    //  statement -- a VALU write of an SGPR needs 5 wait states before a VMEM instruction reads it, and the hazard recognizer
    //  does not pad in front of inline asm; a diagnostic build died of exactly that with an address beyond the aperture)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3 ; PNP_GLD" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
// (a store of more than 64 bits must not be followed directly by a write of its data registers: hipcc's hazard recognizer pads
//  its own stores, but does not look inside inline asm -- without the s_nop the next piece's v_movs, which re-use the same
//  four registers, corrupted the data in flight)
template <int IMM> __device__ __forceinline__ void gst(float* sbase, unsigned voff, f4 v) {
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}
// the four pieces (column half h2, row of the pair) of NP passes from pass p0 on: [k][h2][row01]
template <int NP> __device__ __forceinline__ void gld_passes(f4 (&d)[NP][2][2], const float* img, int p0, unsigned voff) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const float* sb = img + (p0 + k) * 8192;
        gld<0>(d[k][0][0], sb, voff);
        gld<1024>(d[k][0][1], sb, voff);
        gld<512>(d[k][1][0], sb, voff);
        gld<1536>(d[k][1][1], sb, voff);
    }
}
template <int NP> __device__ __forceinline__ void gst_passes(float* img, int p0, unsigned voff, const f4 (&d)[NP][2][2]) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        float* sb = img + (p0 + k) * 8192;
        gst<0>(sb, voff, d[k][0][0]);
        gst<1024>(sb, voff, d[k][0][1]);
        gst<512>(sb, voff, d[k][1][0]);
        gst<1536>(sb, voff, d[k][1][1]);
    }
}
template <int N, int NP> __device__ __forceinline__ void gwait(f4 (&d)[NP][2][2]) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#pragma unroll
    for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(d[k][0][0]), "+v"(d[k][0][1]), "+v"(d[k][1][0]), "+v"(d[k][1][1]));
}

// R layout of one lane: RImg[pass][h2][row01] = row 2 rp + row01, columns 4 (cb + 32 h2) .. + 3, rp = 16 pass + 2 wv + u
struct RLane {
    int wv, u, cb;
    unsigned gbase;                                            // element offset of (row 2 (2 wv + u), column 4 cb) in the image
    unsigned voff;                                             // the same in bytes (the lane part of a hand-issued access)
    __device__ __forceinline__ RLane(int t) : wv(t >> 6), u((t >> 5) & 1), cb(t & 31) { gbase = (unsigned)(2 * wv + u) * 512u + 4u * cb; voff = 4u * gbase; }
    __device__ __forceinline__ unsigned goff(int pass, int h2, int row01) const { return gbase + (unsigned)(pass * 8192 + h2 * 128 + row01 * 256); }
};
typedef f4 RImg[8][2][2];

// F-side hand-over buffer (half an image = 64 row pairs x 256 complex = 128 KB): element (row pair rp_local, column c) at
// rp_local * 256 + (c ^ 16 (rp_local & 1)) -- the XOR puts the two lane groups of a half-wave (consecutive row pairs, the same
// 16 columns) on different halves of the bank row.  In float units the R side addresses 16-byte pieces {x_c, y_c, x_c+1, y_c+1}.
__device__ __forceinline__ unsigned fbuf_r_addr(const RLane& L, int pass_l, int h2, int s) {      // float index of piece s (columns +2s, +2s+1)
    const int rp_local = pass_l * 16 + 2 * L.wv + L.u;                                             // parity = u
    return (unsigned)(rp_local * 512 + 8 * (L.cb ^ (4 * L.u)) + 256 * h2 + 4 * s);
}

// R -> F for the row pairs [64 H, 64 H + 64): d = the R registers of passes 4 H .. 4 H + 3
__device__ __forceinline__ void r_to_f_write(const f4 (&d)[4][2][2], float* ldf, const RLane& L) {
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const f4 X = d[pl][h2][0], Y = d[pl][h2][1];
            st4(ldf + fbuf_r_addr(L, pl, h2, 0), f4{X.x, Y.x, X.y, Y.y});
            st4(ldf + fbuf_r_addr(L, pl, h2, 1), f4{X.z, Y.z, X.w, Y.w});
        }
}
__device__ __forceinline__ void f_read(cx<float> (&Z)[FP][16], const cx<float>* ldc, int H, int g, int l) {
    const int sw = 16 * (g & 1);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const cx<float>* rowp = ldc + (pl * 32 + g) * 256 + l;
        const cx<float>* rE = rowp + sw;
        const cx<float>* rO = rowp - sw;
#pragma unroll
        for (int r = 0; r < 16; ++r) Z[2 * H + pl][r] = (r & 1) ? rO[16 * r] : rE[16 * r];
    }
}
__device__ __forceinline__ void f_write(const cx<float> (&Z)[FP][16], cx<float>* ldc, int H, int g, int l) {
    const int sw = 16 * (g & 1);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        cx<float>* rowp = ldc + (pl * 32 + g) * 256 + l;
        cx<float>* rE = rowp + sw;
        cx<float>* rO = rowp - sw;
#pragma unroll
        for (int r = 0; r < 16; ++r) ((r & 1) ? rO : rE)[16 * r] = Z[2 * H + pl][r];
    }
}
__device__ __forceinline__ void f_to_r_read(f4 (&d)[4][2][2], const float* ldf, const RLane& L) {
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const f4 q0 = ld4(ldf + fbuf_r_addr(L, pl, h2, 0)), q1 = ld4(ldf + fbuf_r_addr(L, pl, h2, 1));
            d[pl][h2][0] = f4{q0.x, q0.z, q1.x, q1.z};
            d[pl][h2][1] = f4{q0.y, q0.w, q1.y, q1.w};
        }
}

// C-side hand-over buffer (half the image's columns = 256 rows x 128 floats = 128 KB): element (row, local column c) at
// row * 128 + (c ^ 16 ((row >> 6) & 1)) -- lanes q and q + 1 of a column (rows 64 apart) land 16 banks apart.
__device__ __forceinline__ unsigned cbuf_r_addr(const RLane& L, int pass, int row01) {
    const int row = pass * 32 + 4 * L.wv + 2 * L.u + row01;                                         // (row >> 6) & 1 == (pass >> 1) & 1
    return (unsigned)(row * 128 + 4 * (((pass >> 1) & 1) ? (L.cb ^ 4) : L.cb));
}
__device__ __forceinline__ unsigned cbuf_c_base(int wv, int cl, int q) { return (unsigned)(64 * q * 128 + ((16 * wv + cl) ^ (16 * (q & 1)))); }

// phases 1-3: the gradient step; leaves alpha * g + beta * c1 + gamma * c2 in R (the R layout).
// a, b, c1, c2: THIS image's arrays (wave-uniform pointers -> scalar base + 32-bit lane offset addressing).  No __restrict__
// on them: out may alias a and c1, and the batches below are ordered by memory clobbers, which the compiler may ignore for
// loads it knows to be invariant.
// OUTER (the outer-loop refresh folded into the first inner iteration, algorithms/pnp_svrg.py:32-57 at j = 0): the
// scaled transform IS mu = grad_full(z); the epilogue stores it, stores w = z (the operand c1 it has to load anyway) and
// leaves z + gamma * mu -- what the plain form computes from an all-zero difference z - w plus mu, bit for bit.
// NOPS: epilogue operand arrays actually present (0, 1: c1 with coefficient beta, 2: c1 and c2) -- known at launch, so the
// epilogue is straight-line code (with run-time tests hipcc kept both prefetch buffers alive everywhere and spilled them while
// their loads were in flight)
template <bool OUTER = false, int NOPS = 2>
__device__ __forceinline__ void fused_gradient(RImg& R, const float* a, const float* b,
                                               const uint32_t* __restrict__ bits, const cx<float>* __restrict__ twtab, cx<float>* twl, cx<float>* ldc,
                                               uint32_t (*sbits)[FG][2][16], const cx<float>* __restrict__ yh, float scale, float beta,
                                               const float* c1, float gamma, const float* c2, int t, int g, int l,
                                               float* w_out, float* mu_out PNP_STAMP_PARAM) {
    cx<float>* scr = ldc + g * F_SCR;
    float* ldf = reinterpret_cast<float*>(ldc);
    const RLane L(t);
    // selector bits of this group's four column pairs (two per half): fetched now, under the operand loads of phase 1 --
    // inside phase 2 their round trip to L2 sat exposed between two workgroup barriers, once per half
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = k * FG + g;
            const int ca = half == 0 ? idx : 64 + idx;
            const int cb = (half == 0 && idx == 0) ? 128 : FN - ca;
            sbits[half][g][k][l] = bits[(size_t)(l < 8 ? ca : cb) * 8 + (l & 7)];
        }
    cx<float> twv = {0.f, 0.f};
    if (t < FN) twv = twtab[t];
    cx<float> Z[FP][16];
    // ------------------------------------------------------------------ 1: operands in R, a - b, R -> F
    // Register budget (256 per lane): half an image of each operand is 64 registers.  Both operands of a half are requested
    // back to back; the second half's `a` goes out before the first half's hand-over, its `b` after it.
    {
        f4 A[4][2][2], Bv[4][2][2];
        gld_passes<4>(A, a, 0, L.voff);
        if (b != nullptr) gld_passes<4>(Bv, b, 0, L.voff);
#pragma unroll
        for (int H = 0; H < 2; ++H) {
            gwait<0, 4>(A);
            if (b != nullptr) {
                gwait<0, 4>(Bv);
#pragma unroll
                for (int pl = 0; pl < 4; ++pl)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                        for (int q = 0; q < 2; ++q) A[pl][h2][q] -= Bv[pl][h2][q];
            }
            if (H == 1) __syncthreads();                        // the first half's readers are done with the buffer
            r_to_f_write(A, ldf, L);
            if (H == 0) {
                if (t < FN) twl[t] = twv;   // twiddles: requested before the operands, landed with them
                gld_passes<4>(A, a, 4, L.voff);
                if (b != nullptr) gld_passes<4>(Bv, b, 4, L.voff);
            }
            __syncthreads();
            f_read(Z, ldc, H, g, l);
            asm volatile("" ::: "memory");
        }
    }
    __syncthreads();                                            // the buffer becomes FFT scratch; also orders twl and the selector bits
    PNP_STAMP(1);
#pragma unroll
    for (int p = 0; p < FP; ++p) {
        fft256<false>(Z[p], twl, scr, l);
        asm volatile("" ::: "memory");
    }
    PNP_STAMP(2);

    // ------------------------------------------------------------------ 2: columns, two halves
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                    // FFT scratch / previous half's reads are done
#pragma unroll
        for (int p = 0; p < FP; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kx = l + 16 * r;
                if (in_half_r(r, l, half)) ldc[kx_local(kx, half) * F_RS + p * FG + g] = Z[p][r];
            }
        // this group's two column pairs and their selector bits (rows ca and W - ca of the transposed bit mask)
        ColPair cp[2];
        const cx<float>* yhc[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = k * FG + g;                     // 0..63
            const int ca = half == 0 ? idx : 64 + idx;
            yhc[k] = yh != nullptr ? yh + (size_t)ca * FN : nullptr;
            const bool packed = half == 0 && idx == 0;      // columns 0 and 128
            const int cb = packed ? 128 : FN - ca;
            cp[k] = {kx_local(ca, half), kx_local(cb, half), packed};
        }
        __syncthreads();
        cx<float> v0[16], v1[16];
        col_load(v0, ldc, cp[0], l);
        col_load(v1, ldc, cp[1], l);
        __syncthreads();                                    // every group has its columns: the buffer becomes FFT scratch
        col_transform(v0, twl, scr, sbits[half][g][0], cp[0], l, yhc[0]);
        asm volatile("" ::: "memory");                      // one transform's working registers at a time
        col_transform(v1, twl, scr, sbits[half][g][1], cp[1], l, yhc[1]);
        __syncthreads();                                    // all FFT scratch use is over: the buffer carries data again
        col_store(v0, ldc, cp[0], l);
        col_store(v1, ldc, cp[1], l);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < FP; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kx = l + 16 * r;
                if (in_half_r(r, l, half)) Z[p][r] = ldc[kx_local(kx, half) * F_RS + p * FG + g];
            }
    }
    __syncthreads();
    PNP_STAMP(3);

    // ------------------------------------------------------------------ 3: rows inverse, F -> R, epilogue in R
#pragma unroll
    for (int p = 0; p < FP; ++p) {
        fft256<true>(Z[p], twl, scr, l);
        asm volatile("" ::: "memory");
    }
    PNP_STAMP(4);
#pragma unroll
    for (int p = 0; p < FP; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) Z[p][r] = {scale * Z[p][r].x, scale * Z[p][r].y};
    // epilogue operands: batches of two passes (8 pieces = 32 registers), two batches in flight.  With the refresh folded in
    // the first two are requested here, so that they arrive under the F -> R hand-over (the plain form has no registers to
    // spare across the hand-over: it requests them behind it)
    f4 U[2][2][2][2];                                           // [buffer][pass of the batch][h2][row01]
    if (OUTER) {
        gld_passes<2>(U[0], c1, 0, L.voff);
        gld_passes<2>(U[1], c1, 2, L.voff);
    }
#pragma unroll
    for (int H = 0; H < 2; ++H) {
        __syncthreads();                                        // FFT scratch / the first half's readers are done
        f_write(Z, ldc, H, g, l);
        __syncthreads();
        f4 d[4][2][2];
        f_to_r_read(d, ldf, L);
#pragma unroll
        for (int pl = 0; pl < 4; ++pl)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int q = 0; q < 2; ++q) R[4 * H + pl][h2][q] = d[pl][h2][q];
        asm volatile("" ::: "memory");
    }
    if (OUTER) {
        // batch k: passes 2k, 2k + 1.  mu = R (stored), w = c1 (stored), R <- c1 + gamma * mu
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // operations issued after batch k's loads: batch k + 1's 8 loads; and for k >= 1 the 16 stores of step k - 1, which
            // went out between the loads of batch k + 1 ... (see the order below: loads of k + 2 are issued at the end of step k)
            if (k == 0) gwait<8, 2>(U[0]);
            else if (k < 3) gwait<8, 2>(U[k & 1]);               // queue behind batch k: [16 stores of step k - 1 are OLDER] batch k + 1
            else gwait<0, 2>(U[k & 1]);
            f4 mu2[2][2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int q = 0; q < 2; ++q) mu2[j][h2][q] = R[2 * k + j][h2][q];
            gst_passes<2>(mu_out, 2 * k, L.voff, mu2);
            gst_passes<2>(w_out, 2 * k, L.voff, U[k & 1]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f4 uv = U[k & 1][j][h2][q];
                        f4& rv = R[2 * k + j][h2][q];
                        rv = f4{fma_(gamma, rv.x, uv.x), fma_(gamma, rv.y, uv.y), fma_(gamma, rv.z, uv.z), fma_(gamma, rv.w, uv.w)};
                    }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                asm volatile("" : "+v"(R[2 * k + j][0][0]), "+v"(R[2 * k + j][0][1]), "+v"(R[2 * k + j][1][0]), "+v"(R[2 * k + j][1][1]));
            if (k + 2 < 4) gld_passes<2>(U[k & 1], c1, 2 * (k + 2), L.voff);
        }
        return;
    }
    if constexpr (NOPS > 0) {
        gld_passes<2>(U[0], c1, 0, L.voff);
        gld_passes<2>(U[1], c1, 2, L.voff);
        // one operand array = four batches; on entry its batches 0 and 1 are in flight; NEXT: another array follows (its first
        // two batches are requested from here, so that they are in flight when its turn comes)
        auto stage = [&](const float* src, float cf, auto NEXT, const float* nxt) {
            constexpr bool HAS_NEXT = decltype(NEXT)::value;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // behind batch k in the queue: batch k + 1 (8 loads) when there is one
                if (k < 3 || HAS_NEXT) gwait<8, 2>(U[k & 1]);
                else gwait<0, 2>(U[k & 1]);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const f4 uv = U[k & 1][j][h2][q];
                            f4& rv = R[2 * k + j][h2][q];
                            rv = f4{fma_(cf, uv.x, rv.x), fma_(cf, uv.y, rv.y), fma_(cf, uv.z, rv.z), fma_(cf, uv.w, rv.w)};
                        }
                // the batch is consumed before its buffer is requested again: an empty asm takes the results, and the requests
                // (volatile asm as well) cannot pass it -- left alone hipcc defers the arithmetic, keeps every batch in
                // registers of its own and spills
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    asm volatile("" : "+v"(R[2 * k + j][0][0]), "+v"(R[2 * k + j][0][1]), "+v"(R[2 * k + j][1][0]), "+v"(R[2 * k + j][1][1]));
                if (k + 2 < 4) gld_passes<2>(U[k & 1], src, 2 * (k + 2), L.voff);
                else if (HAS_NEXT) gld_passes<2>(U[k & 1], nxt, 2 * (k - 2), L.voff);
            }
        };
        if constexpr (NOPS == 2) {
            stage(c1, beta, std::true_type{}, c2);
            stage(c2, gamma, std::false_type{}, nullptr);
        } else {
            stage(c1, beta, std::false_type{}, nullptr);
        }
    }
}

}  // namespace pnp

// everything from here on follows pywt / skimage product for product: no FMA contraction (exact zeros in the wavelet
// coefficients are semantically significant); the FFT code above keeps the compiler's default (contraction on)
#pragma clang fp contract(off)
#include "prox_tv.h"

namespace pnp {

// MODE: 0 = the whole iteration; 1 = stop after the noise estimate and store the stepped image (another prox follows);
//       2 = the gradient only (phases 1-3: grad_full with its data term, or any other use of pnp_csmri_grad_sel that fits
//           this kernel).
enum { FUSED_FULL = 0, FUSED_NO_DENOISE = 1, FUSED_GRAD = 2 };
// the LDS a workgroup needs beside the 132 KB hand-over buffer
struct FusedShared {
    cx<float> twl[FN];
    uint32_t sbits[2][FG][2][16];
    double red[8];
    float sig_sh;
};

// De-synchronisation of the CUs.  Every workgroup does the same work, so the 256 CUs of a launch march through the phases in
// lock step: during the three memory phases ALL of them pull on HBM (saturated, ~10 B/clk per CU), during the compute phases
// none does (in-kernel clock stamps: the memory phases take the same time whether loads are 4 or 16 bytes wide).  The first
// workgroup of every CU (the first `stagger_n` of the grid) therefore starts (blockIdx % groups) * units * 1024 cycles
// late; the offsets persist through the later workgroups of the launch, and one group's memory phases meet the others'
// transforms.
__device__ __forceinline__ void startup_stagger(int stagger_n, int stagger_groups, int stagger_units) {
    if (stagger_units > 0 && (int)blockIdx.x < stagger_n) {
        const int n = ((int)blockIdx.x % stagger_groups) * stagger_units;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
    }
}

// One whole iteration of one image (`a` ... `out`, `xrec`, `w_out`, `mu_out`: THIS image's arrays; sse_out / sigma_out: this
// image's slots).  Whole-workgroup collective.
template <int MODE, bool OUTER, int NOPS>
__device__ __forceinline__ void svrg_iter_body(unsigned char* lds_raw, FusedShared& sh, const float* a, const float* b,
                                               const uint32_t* __restrict__ bits, const cx<float>* __restrict__ yh,
                                               const cx<float>* __restrict__ twtab, float scale, float beta, const float* c1,
                                               float gamma, const float* c2, float* oi, float sigma_modifier, float fallback_sigma,
                                               const float* xri, double* __restrict__ sse_out, float* __restrict__ sigma_out,
                                               float* w_out, float* mu_out) {
    constexpr bool DENOISE = MODE == FUSED_FULL;
    cx<float>* ldc = reinterpret_cast<cx<float>*>(lds_raw);
    float* ldf = reinterpret_cast<float*>(lds_raw);
    cx<float>* twl = sh.twl;
    uint32_t (*sbits)[FG][2][16] = sh.sbits;
    double* red = sh.red;
    float& sig_sh = sh.sig_sh;
    // (laundered: called from a loop -- k_svrg_outer -- hipcc would otherwise hoist every lane-dependent address and every
    //  per-pass base out of the loop and keep them alive across the whole body: 86 scalar and 265 vector registers spilled)
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    asm volatile("" : "+s"(a), "+s"(b), "+s"(bits), "+s"(yh), "+s"(c1), "+s"(c2), "+s"(oi), "+s"(xri), "+s"(w_out), "+s"(mu_out));
    const int g = t >> 4, l = t & 15, wv = t >> 6, lane64 = t & 63;

    PNP_STAMP_DECL;
    PNP_STAMP(0);
    const RLane L(t);
    RImg R;
    fused_gradient<OUTER, NOPS>(R, a, b, bits, twtab, twl, ldc, sbits, yh, scale, beta, c1, gamma, c2, t, g, l, w_out, mu_out PNP_STAMP_ARG);
    if (MODE == FUSED_GRAD) {
        gst_passes<8>(oi, 0, L.voff, R);
        return;
    }
    PNP_STAMP(5);

    // ------------------------------------------------------------------ 4: R -> C
    // wave wv owns image columns [16 wv, 16 wv + 16) and [128 + 16 wv, ...); lane = column + 16 * chunk keeps rows
    // [64 chunk, 64 chunk + 64) of both
    float x[2][64];
    const int cl = lane64 & 15, q = lane64 >> 4;
    const unsigned cbase = cbuf_c_base(wv, cl, q);
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 8; ++pass)
#pragma unroll
            for (int r01 = 0; r01 < 2; ++r01) st4(ldf + cbuf_r_addr(L, pass, r01), R[pass][h2][r01]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 64; ++i) x[h2][i] = ldf[cbase + 128 * i];
    }
    PNP_STAMP(6);

    // ------------------------------------------------------------------ 5: noise estimate, prox, C -> R, error, store
    const bool want_err = DENOISE && xri != nullptr;
    // sigma_est = mean over the 256 columns of the per-column MAD estimate
    {
        const float s0 = column_sigma<float, 64>(x[0], q), s1 = column_sigma<float, 64>(x[1], q);
        double part = q == 0 ? (double)s0 + (double)s1 : 0.0;
        part = wave_sum(part);
        if (lane64 == 0) red[wv] = part;
        __syncthreads();
        if (t == 0) {
            double s = 0;
            for (int i = 0; i < FT / 64; ++i) s += red[i];
            sig_sh = (float)(s / (double)FN);
        }
        __syncthreads();
    }
    const float sigma_est = sig_sh;
    PNP_STAMP(7);
    if (sigma_out != nullptr && t == 0) *sigma_out = sigma_est;
    if (DENOISE) {
        const float sigma = sigma_est > 0.f ? sigma_est * sigma_modifier : fallback_sigma;
        haar_bayes_shrink<float, FN>(x[0], sigma * sigma);
        haar_bayes_shrink<float, FN>(x[1], sigma * sigma);
    }
    PNP_STAMP(8);
    // C -> R per column half, then error + store piece by piece.  The ground truth of the first half is requested once the
    // registers of x[0] are free (written to the hand-over buffer), that of the second half right after the first has arrived:
    // it is in flight during the whole first half.
    double err = 0.0;
    auto finish = [&](auto WANT) {
        constexpr bool ERR = decltype(WANT)::value;
        f4 xa[8][2][1][1], xb[8][2][1][1];                      // ground truth of the two column halves: [pass][row01]
        auto gt_load = [&](f4 (&dst)[8][2][1][1], int h2) {
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                if (h2 == 0) { gld<0>(dst[pass][0][0][0], xri + pass * 8192, L.voff); gld<1024>(dst[pass][1][0][0], xri + pass * 8192, L.voff); }
                else { gld<512>(dst[pass][0][0][0], xri + pass * 8192, L.voff); gld<1536>(dst[pass][1][0][0], xri + pass * 8192, L.voff); }
            }
        };
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            __syncthreads();                                    // (h2 == 1: the first half's readers are done)
#pragma unroll
            for (int i = 0; i < 64; ++i) ldf[cbase + 128 * i] = x[h2][i];
            if (ERR && h2 == 0) gt_load(xa, 0);
            __syncthreads();
            if (ERR) {
                if (h2 == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int pass = 0; pass < 8; ++pass) asm volatile("" : "+v"(xa[pass][0][0][0]), "+v"(xa[pass][1][0][0]));
                    gt_load(xb, 1);
                } else {
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // behind xb in the queue: the first half's 16 stores
#pragma unroll
                    for (int pass = 0; pass < 8; ++pass) asm volatile("" : "+v"(xb[pass][0][0][0]), "+v"(xb[pass][1][0][0]));
                }
            }
            PNP_STAMP_NW(10 + 2 * h2);
            float e = 0.f;
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                f4 v[2];
#pragma unroll
                for (int r01 = 0; r01 < 2; ++r01) {
                    v[r01] = ld4(ldf + cbuf_r_addr(L, pass, r01));
                    if (ERR) {
                        const f4 df = (h2 == 0 ? xa[pass][r01][0][0] : xb[pass][r01][0][0]) - v[r01];
                        e += df.x * df.x;
                        e += df.y * df.y;
                        e += df.z * df.z;
                        e += df.w * df.w;
                    }
                }
                if (h2 == 0) { gst<0>(oi + pass * 8192, L.voff, v[0]); gst<1024>(oi + pass * 8192, L.voff, v[1]); }
                else { gst<512>(oi + pass * 8192, L.voff, v[0]); gst<1536>(oi + pass * 8192, L.voff, v[1]); }
            }
            err += (double)e;
            PNP_STAMP_NW(11 + 2 * h2);
        }
    };
    if (want_err) finish(std::true_type{});
    else finish(std::false_type{});
    if (sse_out != nullptr && want_err) {
        err = wave_sum(err);
        __syncthreads();
        if (lane64 == 0) red[wv] = err;
        __syncthreads();
        if (t == 0) {
            double s = 0;
            for (int i = 0; i < FT / 64; ++i) s += red[i];
            *sse_out = s;
        }
    }
#ifdef PNP_FUSED_CLOCK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PNP_STAMP(9);
    PNP_STAMP_FLUSH;
}

// MODE: 0 = the whole iteration; 1 = stop after the noise estimate and store the stepped image (another prox follows);
//       2 = the gradient only.  OUTER: the outer refresh folded in.  NOPS: epilogue operand arrays present.
template <int MODE, bool OUTER = false, int NOPS = 2>
__global__ __launch_bounds__(FT) void k_svrg_iter(const float* a, const float* b,
                                                  const uint32_t* __restrict__ bitsT, const cx<float>* __restrict__ yh,
                                                  const cx<float>* __restrict__ twtab,
                                                  float scale, const float* __restrict__ alpha_vec, float beta, const float* c1,
                                                  float gamma, const float* c2, float* out,
                                                  float sigma_modifier, float fallback_sigma, const float* __restrict__ xrec,
                                                  double* __restrict__ sse_out, float* __restrict__ sigma_out,
                                                  float* w_out, float* mu_out, int stagger_n, int stagger_groups, int stagger_units) {
    startup_stagger(stagger_n, stagger_groups, stagger_units);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ FusedShared sh;
    const int prob = blockIdx.x;
    const size_t img = (size_t)prob * FN * FN;
    if (alpha_vec != nullptr) scale *= alpha_vec[prob];
    svrg_iter_body<MODE, OUTER, NOPS>(lds_raw, sh, a + img, b != nullptr ? b + img : nullptr, bitsT + (size_t)prob * FN * 8,
                                      yh != nullptr ? yh + (size_t)prob * (FN / 2) * FN : nullptr, twtab, scale, beta,
                                      c1 != nullptr ? c1 + img : nullptr, gamma, c2 != nullptr ? c2 + img : nullptr, out + img,
                                      sigma_modifier, fallback_sigma, xrec != nullptr ? xrec + img : nullptr,
                                      sse_out != nullptr ? sse_out + prob : nullptr, sigma_out != nullptr ? sigma_out + prob : nullptr,
                                      OUTER ? w_out + img : nullptr, OUTER ? mu_out + img : nullptr);
}

// A whole OUTER iteration of pnp_svrg with the TV prox in one launch (algorithms/pnp_svrg.py:32-95 for T2 inner iterations): the
// workgroup that owns an image runs the folded refresh + first inner iteration and then the T2 - 1 plain inner iterations of
// THAT image back to back.  Inner iterations of different images never meet, so there is nothing to synchronise across
// workgroups; between two iterations of one image the data goes through memory (z, w, mu are written and read by the very
// same lanes, in the R layout) -- one s_waitcnt and an L1 invalidate.  What it buys over T2 launches: no launch tails, the
// start-up stagger paid once per T2 rounds, and the operands an image re-reads every iteration (w, mu, the ground truth, z)
// come back while only the 256 images in flight compete for the caches, not the whole batch.
// selbits: [T2][batch][W][H/32] (slot j = inner iteration j; slot 0 is not read), sse_log: [n_log][batch], row (log_row0 + j) % n_log.
__global__ __launch_bounds__(FT) void k_svrg_outer(float* z, float* w, float* mu, const uint32_t* __restrict__ mask_bits,
                                                   const cx<float>* __restrict__ yh, const float* __restrict__ alpha_vec,
                                                   const uint32_t* __restrict__ selbits, int T2, float scale_inner, float gamma,
                                                   const cx<float>* __restrict__ twtab, float sigma_modifier, float fallback_sigma,
                                                   const float* __restrict__ xrec, double* __restrict__ sse_log, int log_row0, int n_log,
                                                   float* __restrict__ sigma_out, int stagger_n, int stagger_groups, int stagger_units) {
    startup_stagger(stagger_n, stagger_groups, stagger_units);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ FusedShared sh;
    const int prob = blockIdx.x, batch = gridDim.x;
    const size_t img = (size_t)prob * FN * FN;
    const float inv_n = 1.0f / ((float)FN * (float)FN);
    float* zi = z + img;
    // outer refresh + inner iteration 0 (pnp_csmri_svrg_outer_step)
    svrg_iter_body<FUSED_FULL, true, 0>(lds_raw, sh, zi, nullptr, mask_bits + (size_t)prob * FN * 8, yh + (size_t)prob * (FN / 2) * FN, twtab,
                                        inv_n * alpha_vec[prob], 1.0f, zi, gamma, nullptr, zi, sigma_modifier, fallback_sigma, xrec + img,
                                        sse_log + (size_t)(log_row0 % n_log) * batch + prob, sigma_out + prob, w + img, mu + img);
#pragma unroll 1
    for (int j = 1; j < T2; ++j) {
        // this iteration reads what the last one wrote (same lanes, same addresses): stores done, no stale line in the L1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        svrg_iter_body<FUSED_FULL, false, 2>(lds_raw, sh, zi, w + img, selbits + ((size_t)j * batch + prob) * FN * 8, nullptr, twtab,
                                             scale_inner, 1.0f, zi, gamma, mu + img, zi, sigma_modifier,
                                             fallback_sigma, xrec + img, sse_log + (size_t)((log_row0 + j) % n_log) * batch + prob,
                                             sigma_out + prob, nullptr, nullptr);
    }
}

// stagger (see startup_stagger): only launches of more than one workgroup per CU pay for it and profit from it
static int stagger_config(int* num_cu_out, int* groups, int* units) {
    static int num_cu = 0, st_groups = 2, st_units = 40;    // same-box sweep (tools/dev/stagger_sweep.py): 0.588 ms per config-2 step without, 0.575 with (2, 40), slower from (8, 20) on
    if (num_cu == 0) {
        int d0 = 0;
        hipDeviceProp_t prop;
        PNP_CHECK_HIP(hipGetDevice(&d0));
        PNP_CHECK_HIP(hipGetDeviceProperties(&prop, d0));
        if (const char* ev = getenv("PNP_FUSED_STAGGER")) {   // "groups,units" (A/B timing); "0" switches it off
            int g = 0, u = 0;
            if (sscanf(ev, "%d,%d", &g, &u) == 2 && g >= 1 && u >= 0) { st_groups = g; st_units = u; }
            else st_units = 0;
        }
        num_cu = prop.multiProcessorCount;
    }
    *num_cu_out = num_cu; *groups = st_groups; *units = st_units;
    return PNP_OK;
}

static int fused_lds_optin() {
    // > 64 KiB of dynamic LDS needs the opt-in, once per device (the attribute is per device)
    static unsigned long long attr_done = 0;
    int dev = 0;
    PNP_CHECK_HIP(hipGetDevice(&dev));
    if (!((attr_done >> (dev & 63)) & 1ull)) {
#define PNP_FUSED_ATTR(...) PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_iter<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES))
        PNP_FUSED_ATTR(0, false, 0); PNP_FUSED_ATTR(0, false, 1); PNP_FUSED_ATTR(0, false, 2);
        PNP_FUSED_ATTR(1, false, 0); PNP_FUSED_ATTR(1, false, 1); PNP_FUSED_ATTR(1, false, 2);
        PNP_FUSED_ATTR(2, false, 0); PNP_FUSED_ATTR(2, false, 1); PNP_FUSED_ATTR(2, false, 2);
        PNP_FUSED_ATTR(0, true, 0); PNP_FUSED_ATTR(1, true, 0);
#undef PNP_FUSED_ATTR
        PNP_CHECK_HIP(hipFuncSetAttribute((const void*)k_svrg_outer, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES));
        attr_done |= 1ull << (dev & 63);
    }
    return PNP_OK;
}

// one launch = one outer iteration (k_svrg_outer); scales computed exactly as the per-iteration entry points compute them
int csmri_fused_outer_launch(int batch, const void* twtab, void* z, void* w, void* mu, const uint32_t* mask_bits, const void* yh,
                             const void* alpha_vec, const uint32_t* selbits, int T2, double lr, int mini_batch_size,
                             double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_log, int log_row0, int n_log,
                             void* sigma_out, void* stream) {
    int num_cu = 0, st_groups = 0, st_units = 0;
    { const int rc = stagger_config(&num_cu, &st_groups, &st_units); if (rc != PNP_OK) return rc; }
    { const int rc = fused_lds_optin(); if (rc != PNP_OK) return rc; }
    const double alpha = -lr / (double)mini_batch_size;
    const float scale_inner = (float)(alpha / ((double)FN * (double)FN));
    k_svrg_outer<<<batch, FT, F_LDS_BYTES, (hipStream_t)stream>>>((float*)z, (float*)w, (float*)mu, mask_bits, (const cx<float>*)yh,
                                                                 (const float*)alpha_vec, selbits, T2, scale_inner, (float)(-lr),
                                                                 (const cx<float>*)twtab, (float)sigma_modifier, (float)fallback_sigma,
                                                                 (const float*)xrec, sse_log, log_row0, n_log, (float*)sigma_out, num_cu,
                                                                 st_groups, 0);     // (no stagger: same-box A/B 0.539 ms per step without, 0.546 with (2, 40))
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

// plan internals live in csmri.hip (pnp_csmri_svrg_step / pnp_csmri_grad_sel); the kernel only needs the plan's twiddle table.
// mode: FUSED_FULL / FUSED_NO_DENOISE / FUSED_GRAD; w_out / mu_out != NULL: the outer refresh folded in (pnp_csmri_svrg_outer_step)
int csmri_fused_launch(int batch, const void* twtab, const void* a, const void* b, const uint32_t* bitsT, const void* yh,
                       double alpha, const void* alpha_vec, double beta, const void* c1, double gamma, const void* c2, void* out,
                       int mode, double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_out, void* sigma_out,
                       void* stream, void* w_out, void* mu_out) {
    const float scale = (float)(alpha / ((double)FN * (double)FN));
    hipStream_t s = (hipStream_t)stream;
    int num_cu = 0, st_groups = 0, st_units = 0;
    { const int rc = stagger_config(&num_cu, &st_groups, &st_units); if (rc != PNP_OK) return rc; }
    const int stagger_units = batch > num_cu ? st_units : 0;
    { const int rc = fused_lds_optin(); if (rc != PNP_OK) return rc; }
    // epilogue operands as the kernel takes them: (c1, beta) first, then (c2, gamma); a lone c2 moves to the first slot
    int nops = (c1 != nullptr ? 1 : 0) + (c2 != nullptr ? 1 : 0);
    if (c1 == nullptr && c2 != nullptr) { c1 = c2; beta = gamma; c2 = nullptr; }
    const bool outer = w_out != nullptr;
#define PNP_FUSED_LAUNCH(...)                                                                                             \
    k_svrg_iter<__VA_ARGS__><<<batch, FT, F_LDS_BYTES, s>>>((const float*)a, (const float*)b, bitsT, (const cx<float>*)yh,    \
                                                       (const cx<float>*)twtab, scale, (const float*)alpha_vec, (float)beta,    \
                                                       (const float*)c1, (float)gamma, (const float*)c2, (float*)out,         \
                                                       (float)sigma_modifier, (float)fallback_sigma, (const float*)xrec,      \
                                                       sse_out, (float*)sigma_out, (float*)w_out, (float*)mu_out, num_cu,      \
                                                       st_groups, stagger_units)
#define PNP_FUSED_BY_NOPS(MD)                                                     \
    do { if (nops == 0) PNP_FUSED_LAUNCH(MD, false, 0); else if (nops == 1) PNP_FUSED_LAUNCH(MD, false, 1); else PNP_FUSED_LAUNCH(MD, false, 2); } while (0)
    if (outer) {                                                // the outer refresh folded into the first inner iteration
        if (mode == FUSED_FULL) PNP_FUSED_LAUNCH(0, true, 0);
        else PNP_FUSED_LAUNCH(1, true, 0);
    } else if (mode == FUSED_GRAD) PNP_FUSED_BY_NOPS(2);
    else if (mode == FUSED_FULL) PNP_FUSED_BY_NOPS(0);
    else PNP_FUSED_BY_NOPS(1);
#undef PNP_FUSED_BY_NOPS
#undef PNP_FUSED_LAUNCH
    PNP_CHECK_LAUNCH();
    return PNP_OK;
}

}  // namespace pnp
