"""development helper: config-2 bench, one launch per outer iteration against one per inner iteration, under stagger settings"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for st in sys.argv[1:]:
    for flag in ([], ['--no-outer-kernel']):
        env = dict(os.environ, PNP_FUSED_STAGGER=st)
        vals = []
        for _ in range(2):
            out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'tv', '--no-cpu-baseline', '--no-secondary', '--steps', '40'] + flag,
                                 env=env, capture_output=True, text=True, timeout=300)
            try:
                vals.append(json.loads(out.stdout.strip().splitlines()[-1])['ms_per_step'])
            except Exception:
                print(out.stderr[-1500:]); raise
        print(f'stagger {st:8s} {"per-inner launches" if flag else "one launch per outer"}: ' + ' '.join('%.4f' % v for v in vals) + ' ms/step', flush=True)
