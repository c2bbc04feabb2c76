"""Bisect helper: the panel kernel under a given library build (STAMPLIB), stage by stage with progress lines."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
if os.environ.get("STAMPLIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["STAMPLIB"])
from stein_amd.engine import SvgdEngine
dev = "cuda"
def staged(n, d, sym):
    T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
    eng = SvgdEngine(n, d, device=dev, x3=True, small=False); st = eng.stages
    st.rownorms(T, n, d, eng.rownorm); st.x3_prepare(T, G, n, d, eng.planes)
    for k in range(3):
        st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, symmetric=sym, planes=eng.planes, kernel=_lib.STAGE_PANEL)
        torch.cuda.synchronize()
    print("staged ok", n, d, sym, flush=True)
def fused(n, d, steps):
    T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
    eng = SvgdEngine(n, d, device=dev)
    for k in range(steps):
        eng.compute_phi(T, G); torch.cuda.synchronize(); T = T + 1e-4 * eng.phi
        print("  fused step", k, "h2", eng.h2.item(), flush=True)
    print("fused ok", n, d, eng.window_stats(), flush=True)
for job in sys.argv[1:]:
    kind, n, d = job.split(":")
    if kind == "s": staged(int(n), int(d), True)
    elif kind == "f": staged(int(n), int(d), False)
    else: fused(int(n), int(d), 6)
