"""development helper: config-2 bench under different stagger settings (PNP_FUSED_STAGGER=groups,units of 1024 cycles)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for st in sys.argv[1:]:
    env = dict(os.environ, PNP_FUSED_STAGGER=st)
    vals = []
    for _ in range(2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'tv', '--no-cpu-baseline', '--no-secondary', '--steps', '40'],
                             env=env, capture_output=True, text=True, timeout=300)
        vals.append(json.loads(out.stdout.strip().splitlines()[-1])['ms_per_step'])
    print(f'stagger {st:8s}: ' + ' '.join('%.4f' % v for v in vals) + ' ms/step', flush=True)
