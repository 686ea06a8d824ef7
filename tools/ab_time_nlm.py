"""Same-box A/B timing of library builds on the NLM prox (256 x 256, B = 64, f32)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:]
res = {n: [] for n in names}
for rnd in range(3):
    for n in names:
        env = dict(os.environ, PNP_HIP_LIB=os.path.join(ROOT, 'pnp_svrg_amd', 'lib', 'ab', n + '.so'))
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'time_nlm.py')], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if 'float32 B=64' in l][0]
        res[n].append(float(line.split(':')[1].split('ms')[0]))
for n in names:
    print(n, ' '.join('%.4f' % v for v in res[n]), 'ms per NLM launch (B = 64)', flush=True)
