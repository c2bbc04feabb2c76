"""Phase cycles of k_distance_x3 (one wave per workgroup): -DSTEIN_STAMPS build.
STAMPLIB=lib_stamps.so python scratch/stamps_dist.py n d"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPLIB", "lib_stamps.so"))
from stein_amd.engine import SvgdEngine, _ptr, _dt, _stream
lib = _lib.load()
n, d = int(sys.argv[1]), int(sys.argv[2])
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda", x3=True, small=False)
st = eng.stages
for _ in range(30): eng.compute_phi(T, G)
lib.stein_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
lib.stein_debug_xcd.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
xb = (ctypes.c_uint64 * 16)()
buf = (ctypes.c_uint64 * 8)()
sel, spec, hist = eng.select_state, eng.spec_section, eng.hist
state = sel.clone()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for sym in (True, False):
    for mode in ("window", "plain"):
        acc = np.zeros(8); ms = 0.0; reps = 5
        for rep in range(reps):
            for _ in range(10): eng.compute_phi(T, G)
            sel.copy_(state)
            if mode == "window": st.spec_begin(hist, sel, spec, n * n)
            else: st.median_begin(hist, sel, n * n)
            torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 1); lib.stein_debug_xcd(xb, 1)
            fl = _lib.STAGE_SYMMETRIC if sym else 0
            e0.record()
            if mode == "window":
                _lib.call("stein_distance_block_spec", _ptr(T), _ptr(eng.rownorm), n, d, 0, n, _dt(T), _ptr(eng.dist), eng.ld_dist,
                          _ptr(hist[0]), _ptr(eng.planes), fl, _ptr(sel), _ptr(spec), _stream(T))
            else:
                st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, hist0=None, symmetric=sym, planes=eng.planes)
            e1.record(); torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 0); lib.stein_debug_xcd(xb, 0)
            acc += np.array(list(buf), dtype=np.float64); ms += e0.elapsed_time(e1)
        xv = np.array(list(xb), dtype=np.float64)
        t0 = xv[8:].min()
        print("      last launch, per XCD: first workgroup end %s us, last workgroup end %s us" % (((xv[8:] - t0) * 1e-2).round(1).tolist(), ((xv[:8] - t0) * 1e-2).round(1).tolist()), flush=True)
        nb = acc[7]
        tot = acc[:4].sum() / nb
        print("%-5s %-7s %.3f ms  wgs %6d | per workgroup (cycles): wait loads+LDS store %7.0f  barriers %7.0f  issue+MFMA %7.0f  epilogue %7.0f  total %7.0f"
              % ("sym" if sym else "full", mode, ms / reps, nb / reps, acc[0] / nb, acc[1] / nb, acc[2] / nb, acc[3] / nb, tot), flush=True)
