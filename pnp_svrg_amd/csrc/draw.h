// draw.h -- device-side minibatch draws (problems/problem.py:110-117, problems/CSMRI.py:66-74:
// `np.random.choice(candidates, size, replace=False)`) as counter-based keys + a threshold.
//
// Every candidate position i gets the 32-bit key  mb_key(state, i),  state = mix64(mix64(mix64(seed) + step) +
// problem); the `mb` smallest (key, i) pairs are the minibatch (uniform without replacement; the pairs are distinct,
// so a draw is deterministic).  Only the THRESHOLD pair (T, P) of each (problem, step) is computed (k_draw_thr);
// consumers re-derive membership (mb_member) where they need it, so a minibatch never exists as an array unless a
// caller asks for one.  The fields are absorbed one at a time, so streams of different seeds / steps / problems are
// unrelated (no XOR-aliasing between the fields).  NOT the NumPy legacy stream: reference-identical draws come from
// the host.
#pragma once
#include "common.h"
#include <cstdlib>

namespace pnp {

inline int draw_fast_path() { return getenv("PNP_DRAW_NO_FAST") == nullptr ? 1 : 0; }

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {      // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
struct MbDesc { uint64_t state; uint32_t T, P; };
// Per-position key: a 32-bit mixer (two 32-bit multiplies; 64-bit multiplies are several quarter-rate instructions each
// on gfx950 and this runs once per k-space position) of the position XOR the stream's low word, XOR the stream's high
// word.  For a fixed stream it is a BIJECTION of the position, so keys never tie inside one draw.
__host__ __device__ __forceinline__ uint32_t mb_key(uint64_t state, uint32_t i) {
    uint32_t x = (uint32_t)state ^ i;
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x ^ (uint32_t)(state >> 32);
}
__device__ __forceinline__ bool mb_member(const MbDesc& d, uint32_t i) {
    const uint32_t key = mb_key(d.state, i);
    return key < d.T || (key == d.T && i <= d.P);
}

// One workgroup per (problem, step): a whole outer iteration's draws are one launch.
// Radix select, 12 + 12 + 8 bits: histogram of the current digit in LDS, pick the bucket that holds rank `mb`; as
// soon as that bucket has at most DRAW_LIST keys they are collected and ranked by brute force.
// MASKED: candidates are the set bits of a transposed bit-packed H x W mask (word [kx][ky >> 5], bit ky & 31),
//         position i = ky*W + kx (the flat row-major index np.flatnonzero counts);
// else  : candidates are all of 0 .. H*W-1 (pass H = 1, W = M for a length-M measurement vector).
constexpr int DRAW_BINS = 4096, DRAW_LIST = 1024;

template <bool MASKED>
__global__ __launch_bounds__(256) void k_draw_thr(const uint32_t* __restrict__ bitsT, int H, int W, int mb, uint64_t seed,
                                                  uint32_t step0, const uint32_t* __restrict__ step_dev,
                                                  MbDesc* __restrict__ mbd, uint32_t* __restrict__ selbits, int fast) {
    __shared__ int hist[DRAW_BINS];
    __shared__ unsigned long long cand[DRAW_LIST];
    __shared__ int wtot[4];
    __shared__ int s_bucket, s_before, s_count, s_n;
    __shared__ unsigned long long s_thr;
    const int prob = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t step = step0 + blockIdx.y;
    if (step_dev != nullptr) step += *step_dev;                  // device-resident counter (hipGraph replays)
    const uint64_t state = mix64(mix64(mix64(seed) + step) + (uint64_t)prob);
    const int wpr = MASKED ? H / 32 : 1, nwords = MASKED ? W * wpr : (H * W + 31) / 32, total_pos = H * W;
    const uint32_t* bits = MASKED ? bitsT + (size_t)prob * nwords : nullptr;
    // candidates of "word" wd: (bit mask, kx, first ky) for a mask; 32 consecutive positions otherwise
    auto word_bits = [&](int wd) -> uint32_t {
        if (MASKED) return bits[wd];
        const int rem = total_pos - 32 * wd;
        return rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);
    };
    auto pos_of = [&](int wd, int bt) -> uint32_t {
        if (MASKED) { const int kx = wd / wpr, kyb = (wd - kx * wpr) * 32; return (uint32_t)((kyb + bt) * W + kx); }
        return (uint32_t)(32 * wd + bt);
    };
    if (tid == 0) s_thr = ~0ull;                                 // default: every candidate (mb >= their number)
    // selbits (optional): the minibatch itself as a bit array in the layout of the candidates ([W][H/32] words of the
    // transposed mask, or ceil(M/32) words), one row per (step, problem) -- 8 KiB per 256 x 256 problem-step
    uint32_t* sb = selbits != nullptr ? selbits + ((size_t)blockIdx.y * gridDim.x + prob) * nwords : nullptr;
    bool emitted = false;

    // ---- fast path: ONE sweep over the candidates.  The rank-mb key of M0 i.i.d. uniform keys sits near mb/M0 * 2^32
    // with a standard deviation of ~sqrt(mb) key spacings; keys below a +-3.5 sigma window (in units of the top 12 bits) are
    // certain members, keys inside it are collected and ranked exactly (the ranking is quadratic in their number, hence 3.5 sigma: one draw in ~2000 misses).  If the window misses (or overflows the list)
    // the general radix select below takes over, so the result is the same either way (`fast` = 0, set by the environment
    // variable PNP_DRAW_NO_FAST, forces the general select: tests compare the two).
    if (fast) {
        int m0 = 0;
        for (int wd = tid; wd < nwords; wd += 256) m0 += __builtin_popcount(word_bits(wd));
        m0 = wave_sum(m0);
        if (lane == 0) wtot[wv] = m0;
        if (tid == 0) s_n = 0;
        __syncthreads();
        m0 = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
        if (m0 > mb) {
            const float per_bucket = (float)m0 / (float)DRAW_BINS;
            const float bg = (float)mb / per_bucket, wdt = 3.5f * sqrtf((float)mb) / per_bucket + 1.f;
            const int lo = (int)fmaxf(0.f, floorf(bg - wdt)), hi = (int)fminf((float)(DRAW_BINS - 1), ceilf(bg + wdt));
            int below = 0;
            for (int wd = tid; wd < nwords; wd += 256) {
                uint32_t m = word_bits(wd), bl = 0;
                while (m) {
                    const int bt = __builtin_ctz(m);
                    m &= m - 1;
                    const uint32_t i = pos_of(wd, bt);
                    const uint32_t key = mb_key(state, i);
                    const int top = (int)(key >> 20);
                    if (top < lo) {
                        bl |= 1u << bt;
                    } else if (top <= hi) {
                        const int pos = atomicAdd(&s_n, 1);
                        if (pos < DRAW_LIST) cand[pos] = ((unsigned long long)key << 32) | i;
                    }
                }
                below += __builtin_popcount(bl);
                if (sb != nullptr) sb[wd] = bl;
            }
            below = wave_sum(below);
            if (lane == 0) wtot[wv] = below;
            __syncthreads();
            const int before = wtot[0] + wtot[1] + wtot[2] + wtot[3], ncand = s_n, kk = mb - before;
            if (ncand <= DRAW_LIST && kk >= 1 && kk <= ncand) {             // uniform across the workgroup
                for (int a = tid; a < ncand; a += 256) {
                    const unsigned long long mine = cand[a];
                    int rank = 0;
                    for (int q = 0; q < ncand; ++q) rank += cand[q] < mine ? 1 : 0;
                    if (rank == kk - 1) {
                        const MbDesc d = {state, (uint32_t)(mine >> 32), (uint32_t)mine};
                        mbd[(size_t)blockIdx.y * gridDim.x + prob] = d;
                    }
                    if (sb != nullptr && rank <= kk - 1) {
                        const uint32_t i = (uint32_t)mine;
                        if (MASKED) { const int ky = i / W, kx = i - ky * W; atomicOr(&sb[kx * wpr + (ky >> 5)], 1u << (ky & 31)); }
                        else atomicOr(&sb[i >> 5], 1u << (i & 31));
                    }
                }
                return;
            }
            __syncthreads();                                     // fall back to the general select
        }
    }

    uint32_t prefix = 0;                                         // the digits fixed so far (high bits of the key)
    int k = mb, fixed_bits = 0;                                  // 1-based rank still to locate inside the prefix bucket
    bool done = false;
    for (int level = 0; level < 3 && !done; ++level) {
        const int dbits = level < 2 ? 12 : 8, shift = 32 - fixed_bits - dbits;
        for (int i = tid; i < DRAW_BINS; i += 256) hist[i] = 0;
        __syncthreads();
        for (int wd = tid; wd < nwords; wd += 256) {
            uint32_t m = word_bits(wd);
            while (m) {
                const int bt = __builtin_ctz(m);
                m &= m - 1;
                const uint32_t key = mb_key(state, pos_of(wd, bt));
                if (fixed_bits == 0 || (key >> (32 - fixed_bits)) == prefix) atomicAdd(&hist[(key >> shift) & ((1u << dbits) - 1)], 1);
            }
        }
        __syncthreads();
        // exclusive scan over the bins: 16 consecutive bins per thread, wave scan, wave totals
        int c[16], tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { c[j] = hist[16 * tid + j]; tot += c[j]; }
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int base = 0;
        for (int q = 0; q < wv; ++q) base += wtot[q];
        const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        if (level == 0 && total <= mb) break;                    // uniform: every candidate is in the minibatch (emitted below)
        const int excl = base + incl - tot;
        if (excl < k && k <= excl + tot) {                       // exactly one thread
            int before = excl, j = 0;
            while (k > before + c[j]) { before += c[j]; ++j; }
            s_bucket = 16 * tid + j;
            s_before = before;
            s_count = c[j];
        }
        __syncthreads();
        prefix = (prefix << dbits) | (uint32_t)s_bucket;
        fixed_bits += dbits;
        k -= s_before;
        const int count = s_count;
        __syncthreads();
        if (count <= DRAW_LIST || level == 2) {
            // collect the bucket's (key, i) pairs and rank them; in the same sweep, emit the bits of every key BELOW
            // the bucket (certain members) -- the bucket's own members are OR-ed in once the threshold is known
            if (tid == 0) s_n = 0;
            __syncthreads();
            const bool emit_here = sb != nullptr && count <= DRAW_LIST;
            for (int wd = tid; wd < nwords; wd += 256) {
                uint32_t m = word_bits(wd), below = 0;
                while (m) {
                    const int bt = __builtin_ctz(m);
                    m &= m - 1;
                    const uint32_t i = pos_of(wd, bt);
                    const uint32_t key = mb_key(state, i);
                    const uint32_t top = key >> (32 - fixed_bits);
                    if (top == prefix) {
                        const int pos = atomicAdd(&s_n, 1);
                        if (pos < DRAW_LIST) cand[pos] = ((unsigned long long)key << 32) | i;
                    } else if (top < prefix) {
                        below |= 1u << bt;
                    }
                }
                if (emit_here) sb[wd] = below;
            }
            __syncthreads();
            if (count <= DRAW_LIST) {
                for (int a = tid; a < count; a += 256) {
                    const unsigned long long mine = cand[a];
                    int rank = 0;
                    for (int q = 0; q < count; ++q) rank += cand[q] < mine ? 1 : 0;
                    if (rank == k - 1) s_thr = mine;
                    if (emit_here && rank <= k - 1) {               // a member inside the threshold bucket
                        const uint32_t i = (uint32_t)mine;
                        if (MASKED) { const int ky = i / W, kx = i - ky * W; atomicOr(&sb[kx * wpr + (ky >> 5)], 1u << (ky & 31)); }
                        else atomicOr(&sb[i >> 5], 1u << (i & 31));
                    }
                }
                emitted = emit_here;
            } else if (tid == 0) {
                // more than DRAW_LIST candidates share one 32-bit key (never happens with a sane hash; kept exact):
                // walk the positions in increasing order and stop at the k-th tie
                int seen = 0;
                for (uint32_t i = 0; i < (uint32_t)total_pos && seen < k; ++i) {
                    bool cnd = true;
                    if (MASKED) { const int ky = i / W, kx = i - ky * W; cnd = (bits[kx * wpr + (ky >> 5)] >> (ky & 31)) & 1u; }
                    if (cnd && mb_key(state, i) == prefix && ++seen == k) s_thr = ((unsigned long long)prefix << 32) | i;
                }
            }
            done = true;
        }
    }
    __syncthreads();
    const MbDesc d = {state, (uint32_t)(s_thr >> 32), (uint32_t)s_thr};
    if (tid == 0) mbd[(size_t)blockIdx.y * gridDim.x + prob] = d;
    if (sb != nullptr && !emitted) {                             // everything selected, or the (never taken) slow paths
        for (int wd = tid; wd < nwords; wd += 256) {
            uint32_t m = word_bits(wd), sel = 0;
            while (m) {
                const int bt = __builtin_ctz(m);
                m &= m - 1;
                if (mb_member(d, pos_of(wd, bt))) sel |= 1u << bt;
            }
            sb[wd] = sel;
        }
    }
}

}  // namespace pnp
