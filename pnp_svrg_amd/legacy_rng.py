"""`np.random.choice(pool, size, replace=False)` on NumPy's global legacy stream, through the C restatement
`pnp_legacy_choice` (`csrc/legacy_rng.cpp`): the same values, the same state of the stream afterwards, a third of the time.

The reference draws one minibatch per inner iteration this way (`problems/CSMRI.py:72`, `problems/problem.py:114`); at B = 1
that full Fisher-Yates shuffle is what bounds the drop-in loops (170-380 us inside NumPy against ~65 us of device work per inner
iteration).  The C function works on the MT19937 state in place (`bit_generator.ctypes.state_address`: 624 key words + the
position) after one self-check against `np.random.get_state()`; if the layout is not the expected one it goes through
`get_state()` / `set_state()` copies instead."""
import ctypes

import numpy as np

from . import _native as N

_work = {}
_inplace = None


def _state_in_place():
    """(address, ok): the global RandomState's MT19937 state struct, checked once against get_state()."""
    global _inplace
    try:                                                        # (private NumPy attributes: any surprise -> the get_state path)
        bg = np.random.mtrand._rand._bit_generator
        if type(bg).__name__ != 'MT19937':                      # np.random.set_bit_generator(...): not the legacy stream's layout
            return 0, False
        addr = bg.ctypes.state_address
    except AttributeError:
        _inplace = False
        return 0, False
    if _inplace is None:
        st = np.random.get_state()
        key = np.frombuffer((ctypes.c_uint32 * 624).from_address(addr), dtype=np.uint32)
        pos = ctypes.c_int.from_address(addr + 624 * 4).value
        _inplace = bool(st[0] == 'MT19937' and np.array_equal(key, st[1]) and pos == st[2])
    return addr, _inplace


def choice(pool, size):
    """np.random.choice(pool, size, replace=False) for a 1-D integer array `pool` or an int (= arange(pool)); int64 result.
    Anything else -- float / object pools (NumPy returns the pool's dtype), 0-d or n-d arrays, sizes that are not plain
    non-negative ints -- goes to np.random.choice itself, with NumPy's own results and errors."""
    if isinstance(pool, (int, np.integer)) and not isinstance(pool, (bool, np.bool_)):
        pop, pool_arr, pool_ptr = int(pool), None, None
    else:
        pa = np.asarray(pool)
        if pa.ndim != 1 or pa.dtype.kind not in 'iu':
            return np.random.choice(pool, size, replace=False)
        pool_arr = np.ascontiguousarray(pa, dtype=np.int64)
        pop, pool_ptr = pool_arr.shape[0], pool_arr.ctypes.data
    if not isinstance(size, (int, np.integer)) or isinstance(size, (bool, np.bool_)):
        return np.random.choice(pool, size, replace=False)
    size = int(size)
    if size < 0 or size > pop or pop < 1:
        return np.random.choice(pool, size, replace=False)          # NumPy's own errors / corner cases
    work = _work.get(pop)
    if work is None:
        work = _work[pop] = np.empty(pop, np.int32)
    out = np.empty(size, np.int64)
    try:
        lock = np.random.mtrand._rand._bit_generator.lock
    except AttributeError:
        lock = None
    if lock is not None:
        with lock:
            addr, ok = _state_in_place()
            if ok:
                N.call('pnp_legacy_choice', addr, ctypes.cast(addr + 624 * 4, ctypes.POINTER(ctypes.c_int)), pool_ptr, pop, size,
                       work.ctypes.data, out.ctypes.data)
                return out
    st = np.random.get_state()
    if st[0] != 'MT19937':
        return np.random.choice(pool, size, replace=False)          # another bit generator: NumPy's own draw
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = ctypes.c_int(int(st[2]))
    N.call('pnp_legacy_choice', key.ctypes.data, ctypes.byref(pos), pool_ptr, pop, size, work.ctypes.data, out.ctypes.data)
    np.random.set_state((st[0], key, pos.value, st[3], st[4]))
    return out
