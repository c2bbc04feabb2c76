"""Strip-by-strip timeline of k_distance_panel (-DSTEIN_STAMPS build): duration of every wave's strips, aligned at the end of
the wave, so that a slow LAST strip (the tail of the launch) shows.
usage: STAMPLIB=lib_stamps.so python scratch/trace_dp.py n d [world]"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPLIB", "lib_stamps.so"))
from stein_amd.engine import SvgdEngine, HipStages
lib = _lib.load()
lib.stein_debug_dp_trace.argtypes = [ctypes.POINTER(ctypes.c_uint32)]
n, d = int(sys.argv[1]), int(sys.argv[2]); world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = "cuda"
def report(tag):
    buf = (ctypes.c_uint32 * (2048 * 64))()
    lib.stein_debug_dp_trace(buf)
    a = np.array(list(buf), dtype=np.int64).reshape(2048, 64)
    cnt = a[:, 0]
    print("%s: strips per wave min %d max %d" % (tag, cnt.min(), cnt.max()))
    K = 12
    durs = np.full((2048, K), np.nan)       # durs[w, k]: duration (us) of the strip k places before the wave's last one
    ends = np.zeros(2048)
    for wv in range(2048):
        c = int(cnt[wv])
        if c < 2: continue
        m = min(c, 62)
        t = np.array([a[wv, 1 + ((c - m + i) % 62)] for i in range(m)], dtype=np.float64) * 0.01   # us since the wave began
        dd = np.diff(t)
        ends[wv] = t[-1]
        for k in range(min(K, len(dd))):
            durs[wv, k] = dd[len(dd) - 1 - k]
    print("   strip duration (us) by position counted from the wave's LAST strip (0 = last): median / mean / p90 over the 2048 waves")
    for k in range(K):
        col = durs[:, k][~np.isnan(durs[:, k])]
        if len(col): print("     last-%2d: %7.1f %7.1f %7.1f" % (k, np.median(col), col.mean(), np.percentile(col, 90)))
    print("   wave end times (us since the wave began): min %.0f median %.0f max %.0f" % (ends[ends > 0].min(), np.median(ends[ends > 0]), ends.max()))
    for wv in (0, 4, 8 * 100 + 1, 8 * 100 + 5, 8 * 200 + 3, 8 * 200 + 7):
        c = int(cnt[wv]); m = min(c, 62)
        t = np.array([a[wv, 1 + ((c - m + i) % 62)] for i in range(m)], dtype=np.float64) * 0.01
        print("   wg %3d wave %d: strip ends (us) %s" % (wv // 8, wv % 8, " ".join("%.0f" % x for x in t[-24:])))
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
