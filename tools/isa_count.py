"""Static instruction mix of one kernel in a hipcc -S listing:  tools/isa_count.py file.s <substring of the mangled name> ..."""
import re, sys
from collections import Counter
f = sys.argv[1]
lines = open(f).read().split('\n')
for want in sys.argv[2:]:
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and want in l.split(':')[0] and l.rstrip().split(';')[0].rstrip().endswith(':'))
    c = Counter()
    n = 0
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith('s_endpgm'):
            break
        if not l.startswith('\t') or t.startswith(('.', ';')) or not t:
            continue
        op = t.split()[0]
        n += 1
        kind = ('valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'salu' if op.startswith('s_') else
                'scratch' if op.startswith('scratch_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else 'other')
        c[kind] += 1
        if op in ('s_waitcnt', 's_barrier'):
            c[op] += 1
    print(want, n, dict(c))
