"""Score kernels (GLM at the C3 shape, BNN at the C4 shape) per library build: python scratch/score_ab.py libA.so libB.so ..."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(HERE))
from stein_amd import _lib
_lib.LIB_PATH = LIBPATH
from stein_amd.scores import GlmScore, BnnScore
dev = "cuda"
def timeit(f, reps=100):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
n, nf, batch = 16384, 255, 50
X = torch.randn(batch, nf, device=dev); y = (torch.rand(batch, device=dev) < 0.5).float()
theta = torch.randn(n, nf + 1, device=dev) * 0.1
glm = GlmScore("logistic", n_feats=nf, w_col=1, alpha_col=0, n_train=16000)
out = torch.empty_like(theta)
print("glm us", round(timeit(lambda: glm(theta, {"X": X, "y": y}, out=out)), 2))
'''
for lib in sys.argv[1:]:
    code = CHILD.replace("HERE", repr(HERE)).replace("LIBPATH", repr(os.path.join(HERE, lib)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(lib, out.stdout.strip() if out.returncode == 0 else out.stderr[-800:])
