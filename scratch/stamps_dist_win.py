"""Phase cycles of k_distance_x3 in WINDOW mode (steady state of the fused call); -DSTEIN_STAMPS build."""
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
for _ in range(4): eng.compute_phi(T, G)          # predictor history: the window exists from here on
lib.stein_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
buf = (ctypes.c_uint64 * 8)()
sel, spec, hist = eng.select_state, eng.spec_section, eng.hist
for rep in range(2):
    st.spec_begin(hist, sel, spec, n * n)
    torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 1)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    _lib.call("stein_distance_block_spec", _ptr(T), _ptr(eng.rownorm), n, d, 0, n, _dt(T), _ptr(eng.dist), eng.ld_dist,
              _ptr(hist[0]), _ptr(eng.planes), _lib.STAGE_SYMMETRIC, _ptr(sel), _ptr(spec), _stream(T))
    e1.record(); torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 0)
u = sel[64:128].cpu().numpy().view(np.uint32)
print("window width", int(u[4]), "entries", int(u[5]))
v = np.array(list(buf), dtype=np.float64); nb = v[7]
print("kernel ms", e0.elapsed_time(e1), "workgroups", nb)
for k, nm in enumerate(["wait loads + LDS stores", "barriers", "issue + frags + MFMA", "epilogue"]):
    print("%-26s %9.0f cycles per workgroup" % (nm, v[k] / nb))
