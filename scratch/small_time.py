import os, sys, subprocess, json
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(HERE))
from stein_amd import _lib
_lib.LIB_PATH = LIBPATH
from stein_amd.engine import SvgdEngine
n, d = N, D
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda")
for _ in range(20): eng.compute_phi(T, G)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(200): eng.compute_phi(T, G)
e1.record(); torch.cuda.synchronize()
print(e0.elapsed_time(e1) / 200 * 1000)
'''
n, d = int(sys.argv[1]), int(sys.argv[2])
for lib in sys.argv[3:]:
    code = CHILD.replace("HERE", repr(HERE)).replace("LIBPATH", repr(os.path.join(HERE, lib))).replace("N, D", "%d, %d" % (n, d))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(lib, out.stdout.strip().splitlines()[-1] if out.returncode == 0 else out.stderr[-500:], "us per compute_phi")
