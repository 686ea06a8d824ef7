"""Static check of the generated code of k_mid_wino44 (csrc/dncnn_wino44.hip).

The kernel's MFMAs are inline asm, so hipcc's hazard recognizer does not protect their results: a compiler-inserted copy
of an accumulator register next to an MFMA (live-range split, spill) reads the registers of the late passes stale.  This
script compiles the file to assembly and verifies, for every instantiation of the kernel, that between the first MFMA and
the end of the tile loop NO instruction other than an MFMA reads or writes an accumulator register outside the epilogue
(marked W44_EPILOGUE_BEGIN / _END, which starts with the required wait states).  Exit code 0 = clean."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'pnp_svrg_amd', 'csrc', 'dncnn_wino44.hip')
REG = re.compile(r'\b([av])(\d+)\b|\b([av])\[(\d+):(\d+)\]')


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def check(asm_text):
    """Per kernel instantiation (also (c): m0 is used by the LDS-DMA statements only): (a) exactly 72 (36 for the one-block-row form) accumulator quads, each the destination of exactly 16 MFMAs (an accumulator
    that was moved shows up as extra quads with fewer); (b) outside the epilogue no instruction other than an MFMA names an
    accumulator AGPR (the 8 VGPR quads of the two-row form are legitimately reused between tiles, (a) covers them)."""
    problems, kernels = [], 0
    blocks = re.split(r'\n(?=_ZN3pnp3w4412k_mid_wino44[^\n]*:\s)', asm_text)
    for blk in blocks[1:]:
        name = blk.split(':', 1)[0]
        m = re.search(r'ILb[01]ELi([12])ELb0ELi0EEEv', name)       # <LEAKY, NG, STAMP = false, VAR = 0>: stamped / ablation builds are not checked
        if not m:
            continue
        nq = 36 * int(m.group(1))
        lines = blk.split('\n')
        end = next((i for i, l in enumerate(lines) if l.startswith('.Lfunc_end')), len(lines))
        lines = lines[:end]
        kernels += 1
        quads = {}
        for l in lines:
            if 'v_mfma' in l:
                m = re.search(r'v_mfma_f32_16x16x4_f32 ([av])\[(\d+):(\d+)\]', l)
                quads[(m.group(1), int(m.group(2)))] = quads.get((m.group(1), int(m.group(2))), 0) + 1
        bad = {k: n for k, n in quads.items() if n != 16}
        if len(quads) != nq or bad:
            problems.append(f'{name}: {len(quads)} accumulator quads ({nq} expected); MFMA count != 16 for {sorted(bad.items())[:8]}')
        agpr = {('a', r) for (cls, r0) in quads if cls == 'a' for r in range(r0, r0 + 4)}
        mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
        in_epi, n_epi = False, 0
        for i in range(mf[0] if mf else 0, len(lines)):
            l = lines[i].strip()
            if 'W44_EPILOGUE_BEGIN' in l:
                in_epi, n_epi = True, n_epi + 1
            elif 'W44_EPILOGUE_END' in l:
                in_epi = False
            if in_epi or not l or l.startswith(';') or l.startswith('.') or 'v_mfma' in l:
                continue
            code = l.split(';')[0]
            hit = regs(code) & agpr
            if hit:
                problems.append(f'{name}: line {i}: `{code.strip()}` touches accumulator register(s) {sorted(hit)[:4]}')
        if n_epi == 0:
            problems.append(f'{name}: no epilogue marker found')
        # (c) M0 belongs to the hand-written LDS-DMA statements (it is not on their clobber list: hipcc reserves it):
        # every mention of m0 must be one of their `s_mov_b32 m0, ...`, one per DMA instruction
        m0 = [l for l in lines if re.search(r'\bm0\b', l.split(';')[0])]
        dma = [l for l in lines if 'buffer_load_dwordx4' in l and ' lds' in l]
        if len(m0) != len(dma) or any('s_mov_b32 m0' not in l for l in m0):
            problems.append(f'{name}: {len(m0)} uses of m0 for {len(dma)} LDS-DMA instructions (the compiler touches m0?)')
    return kernels, problems


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, 'w44.s')
            subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm',
                            '-pragma-unroll-threshold=200000', '-fno-slp-vectorize', '-x', 'hip', '--cuda-device-only', '-S', SRC, '-o', out],
                           check=True, stderr=subprocess.DEVNULL)
            text = open(out).read()
    kernels, problems = check(text)
    print(f'{kernels} kernel instantiation(s) checked, {len(problems)} problem(s)')
    for p in problems[:40]:
        print('  ' + p)
    return 1 if problems or kernels == 0 else 0


if __name__ == '__main__':
    sys.exit(main())
