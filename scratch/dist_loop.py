"""A plain loop of staged distance passes (panel kernel, symmetric C3 block, no window, no histogram) under a given library
build, for rocprofv3: python3 scratch/dist_loop.py <lib|shipped> [reps] [waves]"""
import os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from stein_amd import _lib
if sys.argv[1] != "shipped":
    _lib.LIB_PATH = os.path.join(HERE, sys.argv[1])
from stein_amd.engine import SvgdEngine
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
if len(sys.argv) > 3: _lib.call("stein_debug_dpanel_waves", int(sys.argv[3]))
n, d = 16384, 256
torch.manual_seed(0)
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda", small=False)
st = eng.stages
st.rownorms(T, n, d, eng.rownorm); st.x3_prepare(T, G, n, d, eng.planes)
ev = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, symmetric=True, planes=eng.planes, kernel=_lib.STAGE_PANEL)
    b.record(); ev.append((a, b))
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev[2:])
print("distance pass alone (%s): median %.4f ms, min %.4f" % (sys.argv[1], ts[len(ts) // 2], ts[0]))
