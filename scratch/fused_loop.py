"""A plain loop of fused steps under a given library build, for rocprofv3: python3 scratch/fused_loop.py <lib|shipped> n d steps"""
import os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from stein_amd import _lib
if sys.argv[1] != "shipped":
    _lib.LIB_PATH = os.path.join(HERE, sys.argv[1])
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
n, d, steps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(0)
theta = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda"); gd = AdagradGradientDescent(learning_rate=1e-3)
for _ in range(steps):
    phi = eng.compute_phi(theta, G); gd.apply_(theta, phi, eng.sqnorm)
torch.cuda.synchronize()
