"""C3 fused steps under several library builds (each in its own subprocess), stage medians.  usage: pad_ab.py lib1.so lib2.so ..."""
import os, subprocess, sys, json, statistics
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(HERE))
from stein_amd import _lib
if LIBPATH: _lib.LIB_PATH = LIBPATH
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
n, d = N, D
torch.manual_seed(0)
theta = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda"); gd = AdagradGradientDescent(learning_rate=1e-3)
for _ in range(6):
    phi = eng.compute_phi(theta, G); gd.apply_(theta, phi, eng.sqnorm)
steps = 30
_lib.timing_reserve(steps)
for _ in range(steps):
    phi = eng.compute_phi(theta, G, timing=True); gd.apply_(theta, phi, eng.sqnorm)
torch.cuda.synchronize()
per = _lib.timing_read(steps)
print(json.dumps({k: sum(c[k] for c in per) / len(per) for k in _lib.T_STAGES}))
'''
def run(lib, n, d):
    path = "" if lib == "shipped" else os.path.join(HERE, lib)
    code = CHILD.replace("HERE", repr(HERE)).replace("LIBPATH", repr(path)).replace("N, D", "%d, %d" % (n, d))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if out.returncode: raise SystemExit(out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])
if __name__ == "__main__":
    n, d = int(sys.argv[1]), int(sys.argv[2]); libs = sys.argv[3:]
    acc = {l: [] for l in libs}
    for r in range(3):
        for l in libs: acc[l].append(run(l, n, d))
    for l in libs:
        print(l, {k: round(statistics.median(x[k] for x in acc[l]), 4) for k in acc[l][0]}, flush=True)
