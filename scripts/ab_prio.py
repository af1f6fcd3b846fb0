"""Priority of the weight gradients' stream: same-session A/B (the class attribute must be set before the model is built)."""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else 'n/a')
code = '''
import sys, runpy
sys.path.insert(0, %r)
from sacv_amd import engine
engine.DetectorEngine.SIDE_PRIORITY = int(sys.argv.pop(1))
sys.argv[0] = 'bench.py'
runpy.run_path(%r, run_name='__main__')
''' % (root, os.path.join(root, 'bench.py'))
for rep in range(3):
    for prio in (0, 1, 2, -1):
        r = subprocess.run([sys.executable, '-c', code, str(prio), '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-kernel-timer'],
                           capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print('side priority', prio, d['ms_per_step'], d['ms_per_step_median'], flush=True)
        except Exception as e:
            print('side priority', prio, 'failed', r.stderr[-300:], flush=True)
