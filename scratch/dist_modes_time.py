"""Kernel time of the distance pass per epilogue flavour (sym|full x window|hist|plain), shipped library."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
if os.environ.get("LIB"): _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["LIB"])
from stein_amd.engine import SvgdEngine, _ptr, _dt, _stream
n, d = int(sys.argv[1]), int(sys.argv[2])
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda", x3=True, small=False)
st = eng.stages
for _ in range(4): eng.compute_phi(T, G)
sel, spec, hist = eng.select_state, eng.spec_section, eng.hist
state = sel.clone()
out = []
for sym in (True, False):
    for mode in ("window", "hist", "plain"):
        ts = []
        for rep in range(12):
            sel.copy_(state)
            if mode == "window": st.spec_begin(hist, sel, spec, n * n)
            else: st.median_begin(hist, sel, n * n)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            fl = _lib.STAGE_SYMMETRIC if sym else 0
            e0.record()
            if mode == "window":
                _lib.call("stein_distance_block_spec", _ptr(T), _ptr(eng.rownorm), n, d, 0, n, _dt(T), _ptr(eng.dist), eng.ld_dist,
                          _ptr(hist[0]), _ptr(eng.planes), fl, _ptr(sel), _ptr(spec), _stream(T))
            else:
                st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, hist0=hist[0] if mode == "hist" else None,
                                  symmetric=sym, planes=eng.planes)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        out.append("%s/%s %.4f" % ("sym" if sym else "full", mode, float(np.median(ts[2:]))))
print("n=%d d=%d  ms: %s" % (n, d, "  ".join(out)))
