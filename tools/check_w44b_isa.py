"""Static check of the generated code of k_mid_wino44b (csrc/dncnn_wino44b.hip), the 3 x bf16 split F(4x4,3x3) conv kernel.

Its MFMAs are inline asm, so hipcc's hazard recognizer does not protect their results (tools/check_w44_isa.py has the story).
For every production instantiation this script verifies on the assembly that (a) there are exactly 18 accumulator tuples of 16
registers (16 in AGPRs, 2 in VGPRs), each the destination of exactly 24 MFMAs (8 chunks x 3 products) -- an accumulator that was
moved shows up as extra tuples with fewer; (b) from a tuple's first MFMA to the epilogue marker (behind which the wait states
stand) no instruction other than an MFMA names one of its registers -- BEFORE its first MFMA of a region hipcc may, and does,
park spilled values in a not-yet-live accumulator AGPR; (c) m0 is used by the LDS-DMA statements only.  Exit code 0 = clean."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'pnp_svrg_amd', 'csrc', 'dncnn_wino44b.hip')
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
    problems, kernels = [], 0
    blocks = re.split(r'\n(?=_ZN3pnp4w44b13k_mid_wino44b[^\n]*:\s)', asm_text)
    for blk in blocks[1:]:
        name = blk.split(':', 1)[0]
        if not re.search(r'ILb[01]ELb0ELi0EEEv', name):             # <LEAKY, STAMP = false, VAR = 0>: stamped / ablation builds are not checked
            continue
        lines = blk.split('\n')
        end = next((i for i, l in enumerate(lines) if l.startswith('.Lfunc_end')), len(lines))
        lines = lines[:end]
        kernels += 1
        tuples = {}
        for l in lines:
            if 'v_mfma' in l:
                m = re.search(r'v_mfma_f32_32x32x16_bf16 ([av])\[(\d+):(\d+)\]', l)
                if not m or int(m.group(3)) - int(m.group(2)) != 15:
                    problems.append(f'{name}: unexpected MFMA `{l.strip()}`')
                    continue
                tuples[(m.group(1), int(m.group(2)))] = tuples.get((m.group(1), int(m.group(2))), 0) + 1
        bad = {k: n for k, n in tuples.items() if n != 24}
        na = sum(1 for (c, _) in tuples if c == 'a')
        if len(tuples) != 18 or na != 16 or bad:
            problems.append(f'{name}: {len(tuples)} accumulator tuples ({na} in AGPRs; 18 / 16 expected); MFMA count != 24 for {sorted(bad.items())[:8]}')
        mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
        begin = [i for i, l in enumerate(lines) if 'W44B_EPILOGUE_BEGIN' in l]
        if len(begin) != 1 or not mf or begin[0] < mf[-1]:
            problems.append(f'{name}: epilogue marker missing or not behind the last MFMA')
            continue
        live = set()                                             # registers of the tuples whose first MFMA has been seen
        for i in range(mf[0], begin[0]):
            l = lines[i].strip()
            if not l or l.startswith(';') or l.startswith('.'):
                continue
            code = l.split(';')[0]
            if 'v_mfma' in code:
                m = re.search(r'v_mfma_f32_32x32x16_bf16 ([av])\[(\d+):(\d+)\]', code)
                live.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
                continue
            hit = regs(code) & live
            if hit:
                problems.append(f'{name}: line {i}: `{code.strip()}` touches live accumulator register(s) {sorted(hit)[:4]}')
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
            out = os.path.join(td, 'w44b.s')
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
