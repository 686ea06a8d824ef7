"""Static check of the hand-issued global loads of the one-kernel iteration (csrc/csmri_fused.hip) on the generated code.

hipcc does not know that the inline-asm `global_load_dwordx4 ... ; PNP_GLD` results arrive later: it may read, overwrite or
spill a destination register right behind the load.  The kernel guards every batch with a hand-counted `s_waitcnt vmcnt(N)`;
this script verifies on the ISA of every k_svrg_iter instantiation that

  * no instruction names a destination register of a tagged load before a wait that covers it has been passed, on ANY path:
    a wait `vmcnt(K)` covers a load when at least K vector-memory operations were issued after it on every path (operations
    leave the queue in issue order);
  * branches are forward, or loops without a hand-issued load in flight (the analysis is one pass over the listing, states merged
    at labels).

    python tools/check_fused_isa.py [listing.s]        (without argument: compiles csmri_fused.hip with hipcc -S first)

Also prints, per kernel, the spill traffic between workgroup barriers (where the register pressure bites)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VMEM = ('global_', 'buffer_', 'scratch_', 'flat_')


def listing(path=None, defines=()):
    if path:
        return open(path).read()
    src = os.path.join(ROOT, 'pnp_svrg_amd', 'csrc', 'csmri_fused.hip')
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'f.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-function', *defines, '-x', 'hip',
                        '--cuda-device-only', '-S', src, '-o', out], check=True, cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
        return open(out).read()


def regs_of(text):
    """VGPR numbers named in an operand string."""
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def kernels(txt):
    lines = txt.split('\n')
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith(('_ZN3pnp11k_svrg_iter', '_ZN3pnp12k_svrg_outer')) and l.split(';')[0].rstrip().endswith(':'):
            name = l.split(':')[0]
            body = []
            i += 1
            while i < len(lines) and not lines[i].strip().startswith('s_endpgm'):
                body.append(lines[i])
                i += 1
            yield name, body
        i += 1


def check(name, body):
    # state: {load id: (dst regs, min younger ops)}; per-label merged states
    pending = {}
    state = {}
    alive = True                                       # fallthrough reachable
    seen_labels = set()
    errors, n_loads, n_waits = [], 0, 0
    last_code = ''
    seg, segs = {'sst': 0, 'sld': 0}, []

    def merge(a, b):
        out = dict(a)
        for k, (regs, y) in b.items():
            out[k] = (regs, min(y, out[k][1])) if k in out else (regs, y)
        return out

    for ln, raw in enumerate(body):
        t = raw.strip()
        if not t or t.startswith(';'):
            continue
        if re.match(r'^\.LBB\d+_\d+:', t):
            lab = t.split(':')[0]
            seen_labels.add(lab)
            inc = pending.pop(lab, None)
            if inc is not None:
                state = merge(state, inc) if alive else inc
                alive = True
            continue
        if t.startswith('.'):
            continue
        prev_code = last_code
        code = t.split(';')[0].strip()
        last_code = code
        op = code.split()[0]
        operands = code[len(op):]
        if not alive:
            continue
        if op == 's_barrier':
            segs.append(seg)
            seg = {'sst': 0, 'sld': 0}
        if op.startswith('scratch_store'):
            seg['sst'] += 1
        if op.startswith('scratch_load'):
            seg['sld'] += 1
        # hazard: any in-flight destination named here
        used = regs_of(operands)
        for k, (regs, y) in list(state.items()):
            if used & regs and not (op.startswith('global_load') and 'PNP_GLD' in t and k == ln):
                errors.append(f'{name}: line {ln}: `{code}` names v{sorted(used & regs)} of the load issued at line {k} (in flight, {y} younger operations)')
                del state[k]
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', code)
            if m:
                n_waits += 1
                K = int(m.group(1))
                state = {k: v for k, v in state.items() if v[1] < K}
        if (op.startswith('global_load') and 'PNP_GLD' in t) or (op == 'global_store_dwordx4' and prev_code != 's_nop 4' and False):
            pass
        if op.startswith('global_') and ('PNP_GLD' in t) and prev_code != 's_nop 4':
            errors.append(f'{name}: line {ln}: hand-issued load without its s_nop 4 (VALU-written scalar base)')
        if op.startswith(VMEM):
            state = {k: (r, y + 1) for k, (r, y) in state.items()}
            if 'PNP_GLD' in t:
                n_loads += 1
                dst = regs_of(operands.split(',')[0])
                state[ln] = (dst, 0)
        if op.startswith('s_cbranch') or op == 's_branch':
            lab = operands.strip()
            if lab in seen_labels:
                # a loop (the start-up delay) is fine as long as no hand-issued load is in flight around it
                if state:
                    errors.append(f'{name}: backward branch to {lab} with {len(state)} hand-issued loads in flight')
                continue
            pending[lab] = merge(pending[lab], state) if lab in pending else dict(state)
            if op == 's_branch':
                alive = False
                state = {}
    segs.append(seg)
    if state:
        errors.append(f'{name}: {len(state)} tagged loads never waited for')
    return errors, n_loads, n_waits, segs


def main():
    args = sys.argv[1:]
    defines = [a for a in args if a.startswith('-D')]          # e.g. -DPNP_FUSED_CLOCK: the diagnostic build
    paths = [a for a in args if not a.startswith('-D')]
    txt = listing(paths[0] if paths else None, defines)
    bad = []
    nk = 0
    for name, body in kernels(txt):
        nk += 1
        errors, n_loads, n_waits, segs = check(name, body)
        spill = ' '.join(f'{i}:{s["sst"]}/{s["sld"]}' for i, s in enumerate(segs) if s['sst'] or s['sld'])
        print(f'{name[:40]}...: {n_loads} hand-issued loads, {n_waits} vmcnt waits, {len(errors)} violations; scratch stores/loads per barrier segment: {spill}')
        bad += errors
    for e in bad[:40]:
        print('VIOLATION', e)
    if nk == 0:
        print('no k_svrg_iter kernel found')
        return 2
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
