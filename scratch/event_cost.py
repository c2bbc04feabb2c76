"""What do the stage-boundary HIP events of STEIN_FLAG_TIMING cost the step they time?  usage: event_cost.py <c2|c3>"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, d, dt = (4096, 128, torch.bfloat16) if which == "c2" else (16384, 256, torch.float32)
torch.manual_seed(0)
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda").to(dt)
eng = SvgdEngine(n, d, device="cuda", dtype=dt); gd = AdagradGradientDescent(learning_rate=1e-3)
def run(steps, timing, apply_events):
    if timing: _lib.timing_reserve(steps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        phi = eng.compute_phi(T.to(dt), G, timing=timing)
        if apply_events:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
        gd.apply_(T, phi, eng.sqnorm)
        if apply_events: b.record()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    if timing: _lib.timing_read(steps)
    return ms
run(10, False, False)
for rep in range(3):
    print("%s: plain %.4f | stage events %.4f | stage + apply events %.4f ms/step" % (which, run(40, False, False), run(40, True, False), run(40, True, True)))
