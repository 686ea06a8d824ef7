// Host-side: NumPy's legacy `np.random.choice(a, size, replace=False)` on the global MT19937 stream, restated in C.
//
// The drop-in loops must consume the legacy stream exactly as the reference does (problems/CSMRI.py:72,
// problems/problem.py:114: one `np.random.choice(..., replace=False)` per inner iteration), and at B = 1 that draw -- a full
// Fisher-Yates shuffle of every sampled location -- is what bounds the loop (170-380 us per draw inside NumPy against
// ~65 us of device work per inner iteration).  Same algorithm, same stream, fewer nanoseconds per element:
//   choice(a, size, replace=False)  =  a[permutation(len(a))[:size]]
//   permutation(n)                  =  shuffle(arange(n)):  for i = n-1 .. 1:  j = interval(i);  swap(x[i], x[j])
//   interval(max)                   =  rejection sampling of (next_uint32() & mask) <= max, mask = smallest 2^k - 1 >= max
//   next_uint32()                   =  MT19937 (Matsumoto & Nishimura), state = 624 words + position, as
//                                      np.random.get_state() / set_state() exchange it
// No device work, no allocation beyond the caller's buffers.  tests/test_cpu_host.py pins it against NumPy itself.
#include <cstdint>
#include "../../include/pnp_hip.h"

namespace {

struct Mt {
    uint32_t* key;
    int pos;
    void refill() {
        constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
        int kk = 0;
        for (; kk < 624 - 397; ++kk) {
            const uint32_t y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        }
        for (; kk < 623; ++kk) {
            const uint32_t y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        }
        const uint32_t y = (key[623] & UPPER) | (key[0] & LOWER);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        pos = 0;
    }
    // the next 624 - pos outputs (tempered) in one pass: the shuffle below consumes them from `buf`
    uint32_t buf[624];
    int bpos = 624;
    inline void fill() {
        if (pos == 624) refill();
        const int n = 624 - pos;
        for (int k = 0; k < n; ++k) {
            uint32_t y = key[pos + k];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            buf[624 - n + k] = y;
        }
        bpos = 624 - n;
        pos = 624;
    }
    inline uint32_t next() {
        if (bpos == 624) fill();
        return buf[bpos++];
    }
    // the stream's position: buffered outputs that were not consumed are still the state's next ones
    inline int position() const { return bpos == 624 ? pos : bpos; }
};

}  // namespace

extern "C" int pnp_legacy_choice(uint32_t* mt_key, int* mt_pos, const int64_t* pool, int pop, int size, int32_t* work,
                                 int64_t* out) {
    if (mt_key == nullptr || mt_pos == nullptr || work == nullptr || out == nullptr || pop < 1 || size < 0 || size > pop ||
        *mt_pos < 0 || *mt_pos > 624)
        return PNP_ERR_ARG;
    Mt mt;
    mt.key = mt_key;
    mt.pos = *mt_pos;
    for (int i = 0; i < pop; ++i) work[i] = i;
    // Rejection sampling without an unpredictable branch: every candidate runs the same sequence -- a rejected one swaps
    // x[i] with itself and leaves i where it is (acceptance is 50-100 % likely, which a predictor cannot learn).
    int i = pop - 1;
    while (i >= 1) {
        const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz((uint32_t)i);        // smallest 2^k - 1 >= i: constant while i > mask / 2
        const int lo = (int)(mask >> 1);
        while (i > lo) {
            const uint32_t j = mt.next() & mask;
            const bool acc = j <= (uint32_t)i;
            const uint32_t jj = acc ? j : (uint32_t)i;
            const int32_t ti = work[i], tj = work[jj];
            work[i] = tj;
            work[jj] = ti;
            i -= acc ? 1 : 0;
        }
    }
    if (pool != nullptr) for (int k = 0; k < size; ++k) out[k] = pool[work[k]];
    else for (int k = 0; k < size; ++k) out[k] = work[k];
    *mt_pos = mt.position();
    return PNP_OK;
}
