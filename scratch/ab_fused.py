"""Same-box A/B of library builds on the FUSED step (what bench.py times): stage events of stein_svgd_phi + the apply kernel.
usage: python scratch/ab_fused.py <c3|c2|c4|n,d[,bf16]> libA.so libB.so ...   (paths relative to scratch/; "shipped" = stein_amd/libsteinhip.so)
Each build runs in its own subprocess, interleaved over three rounds; prints the median stage times (ms)."""
import os, subprocess, sys, json, statistics
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, torch, json, statistics
sys.path.insert(0, os.path.dirname(HERE))
from stein_amd import _lib
if LIBPATH: _lib.LIB_PATH = LIBPATH
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
n, d, bf = N, D, BF
dt = torch.bfloat16 if bf else torch.float32
torch.manual_seed(0)
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda").to(dt)
eng = SvgdEngine(n, d, device="cuda", dtype=dt, window=WINDOW); gd = AdagradGradientDescent(learning_rate=1e-3)
import time as _t
_t0 = _t.perf_counter()
while (_t.perf_counter() - _t0) < 0.1:        # settle: the clock governor needs ~50 ms of this load (profiles/r04_step_trend.txt)
    for _ in range(10):
        phi = eng.compute_phi(T.to(dt), G); gd.apply_(T, phi, eng.sqnorm)
    torch.cuda.synchronize()
steps = 30
_lib.timing_reserve(steps)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps):
    phi = eng.compute_phi(T.to(dt), G, timing=True); gd.apply_(T, phi, eng.sqnorm)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
per = _lib.timing_read(steps)
out = {k: statistics.median(c[k] for c in per) for k in _lib.T_STAGES}
out["wall"] = wall
out["h2"] = float(eng.h2.item())
print(json.dumps(out))
'''
SHAPES = {"c3": (16384, 256, False), "c2": (4096, 128, True), "c4": (8192, 2001, False)}
def run(lib, n, d, bf, window):
    path = "" if lib == "shipped" else os.path.join(HERE, lib)
    code = (CHILD.replace("HERE", repr(HERE)).replace("LIBPATH", repr(path)).replace("N, D, BF", "%d, %d, %s" % (n, d, bf))
            .replace("WINDOW", str(window)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if out.returncode: raise SystemExit(out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])
if __name__ == "__main__":
    what = sys.argv[1]
    window = True
    if what.endswith(":miss"): what, window = what[:-5], False
    if what in SHAPES: n, d, bf = SHAPES[what]
    else:
        p = what.split(","); n, d, bf = int(p[0]), int(p[1]), len(p) > 2
    libs = sys.argv[2:]
    acc = {l: [] for l in libs}
    for r in range(3):
        for l in libs: acc[l].append(run(l, n, d, bf, window))
    for l in libs:
        keys = acc[l][0].keys()
        print(what, "window" if window else "no window", l, {k: round(statistics.median(x[k] for x in acc[l]), 4) for k in keys}, flush=True)
