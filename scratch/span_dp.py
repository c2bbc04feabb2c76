"""Begin / end of every wave of the (otherwise shipped) k_distance_panel: -DSTEIN_WGEND build, one plain store per wave.
usage: STAMPLIB=lib_wgend.so python scratch/span_dp.py n d [world]"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPLIB", "lib_wgend.so"))
from stein_amd.engine import SvgdEngine, HipStages
lib = _lib.load()
lib.stein_debug_dp_span.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
n, d = int(sys.argv[1]), int(sys.argv[2]); world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = "cuda"
def report(tag):
    buf = (ctypes.c_uint64 * 4096)()
    lib.stein_debug_dp_span(buf)
    a = np.array(list(buf), dtype=np.float64).reshape(2048, 2)
    t0 = a[:, 0].min()
    beg, end = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01
    print("%s: wave begin %.1f .. %.1f us; wave end: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us" %
          (tag, beg.min(), beg.max(), end.min(), np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max()))
    e = end.reshape(256, 8)
    print("   older waves (0-3) end: median %.1f max %.1f | younger (4-7): median %.1f max %.1f" % (np.median(e[:, :4]), e[:, :4].max(), np.median(e[:, 4:]), e[:, 4:].max()))
    print("   per XCD (32 logical workgroups each) last wave end: %s" % [round(float(e[32 * x:32 * x + 32].max()), 1) for x in range(8)])
    print("   per XCD median wave end: %s" % [round(float(np.median(e[32 * x:32 * x + 32])), 1) for x in range(8)])
if world == 1:
    T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
    eng = SvgdEngine(n, d, device=dev)
    for _ in range(8):
        eng.compute_phi(T, G); T = T + 1e-4 * eng.phi
    torch.cuda.synchronize()
    report("fused n=%d d=%d (window on), last launch" % (n, d))
else:
    nl = n // world
    st = HipStages()
    total, offs, extra = st.workspace_layout(nl, n, d, _lib.FLAG_X3)
    ws = torch.empty(total, dtype=torch.uint8, device=dev)
    ld = extra[_lib.WSX_LD_DIST]
    r = ws[offs[_lib.WS_ROWNORM]:offs[_lib.WS_ROWNORM] + n * 4].view(torch.float32)
    D = ws[offs[_lib.WS_DIST]:offs[_lib.WS_DIST] + nl * ld * 4].view(torch.float32).view(nl, ld)
    planes = ws[offs[_lib.WS_PLANES]:total]
    T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
    st.rownorms(T, n, d, r); st.x3_prepare(T, G, n, d, planes)
    for _ in range(3):
        st.distance_block(T, r, n, d, 0, nl, D, ld, planes=planes, kernel=_lib.STAGE_PANEL)
    torch.cuda.synchronize()
    report("row block %d x %d d=%d (plain epilogue), last launch" % (nl, n, d))
