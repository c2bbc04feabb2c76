"""Phase cycles of k_distance_x3 per epilogue flavour: sym|full x window|hist|plain; -DSTEIN_STAMPS build."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPLIB", "libsteinhip_stamps.so"))
from stein_amd.engine import SvgdEngine, _ptr, _dt, _stream
lib = _lib.load()
n, d = int(sys.argv[1]), int(sys.argv[2])
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda", x3=True, small=False)
st = eng.stages
for _ in range(4): eng.compute_phi(T, G)
lib.stein_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
buf = (ctypes.c_uint64 * 8)()
sel, spec, hist = eng.select_state, eng.spec_section, eng.hist
state = sel.clone()
for sym in (True, False):
    for mode in ("window", "hist", "plain"):
        for rep in range(2):
            sel.copy_(state)
            if mode == "window": st.spec_begin(hist, sel, spec, n * n)
            else: st.median_begin(hist, sel, n * n)
            torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 1)
            fl = _lib.STAGE_SYMMETRIC if sym else 0
            if mode == "window":
                _lib.call("stein_distance_block_spec", _ptr(T), _ptr(eng.rownorm), n, d, 0, n, _dt(T), _ptr(eng.dist), eng.ld_dist,
                          _ptr(hist[0]), _ptr(eng.planes), fl, _ptr(sel), _ptr(spec), _stream(T))
            else:
                st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, hist0=hist[0] if mode == "hist" else None,
                                  symmetric=sym, planes=eng.planes)
            torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 0)
        v = np.array(list(buf), dtype=np.float64); nb = v[7]
        print("%-5s %-7s wgs %5d  main loop %6.0f | epilogue: setup %6.0f bodies %6.0f staged stores %6.0f tail %6.0f" %
              ("sym" if sym else "full", mode, nb, v[2] / nb, v[3] / nb, v[4] / nb, v[5] / nb, v[6] / nb), flush=True)
