"""Same-box A/B timing of library builds on the config-2 bench (one-kernel PnP-SVRG + TV iteration, B = 1024)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# names: library builds under pnp_svrg_amd/lib/ab/ ("tree" = the in-tree library); NAME+flag+flag appends bench.py flags
# (e.g. tree+--no-fold)
names = sys.argv[1:]
res = {n: [] for n in names}
for rnd in range(3):
    for n in names:
        lib, *flags = n.split('+')
        env = dict(os.environ)
        if lib != 'tree':
            env['PNP_HIP_LIB'] = os.path.join(ROOT, 'pnp_svrg_amd', 'lib', 'ab', lib + '.so')
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'tv', '--no-cpu-baseline', '--no-secondary'] + flags,
                             env=env, capture_output=True, text=True, timeout=300)
        res[n].append(json.loads(out.stdout.strip().splitlines()[-1])['ms_per_step'])
for n in names:
    print(n, ' '.join('%.4f' % v for v in res[n]), 'ms/step  min %.4f' % min(res[n]), flush=True)
