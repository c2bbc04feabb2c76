"""Where a window miss spends its time: (1) the staged histogram passes (k_hist<level>, one launch each) on the image a fused
step left behind, per level; (2) the fused call's median stage with the window disabled (k_spec_select + k_hist_all) for
several counts of virtual workgroups.  One GPU.  usage: python scratch/select_time.py [c3|c2]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, d, dtype = {"c3": (16384, 256, torch.float32), "c2": (4096, 128, torch.bfloat16)}[which]
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
T = torch.randn(n, d, generator=g).to(dev).to(dtype)
G = torch.randn(n, d, generator=g).to(dev).to(dtype)

def ev():
    return torch.cuda.Event(enable_timing=True)

# (1) staged passes on the fused call's image
eng = SvgdEngine(n, d, device=dev, dtype=dtype, window=False, small=False)
eng.compute_phi(T, G); torch.cuda.synchronize()
st, D, ld, hist, sel = eng.stages, eng.dist, eng.ld_dist, eng.hist, eng.select_state
for rep in range(2):
    st.median_begin(hist, sel, n * n)
    for level in range(3):
        a, b = ev(), ev()
        a.record(); st.median_hist_pass(D, ld, n, n, level, sel, hist, symmetric=True); b.record()
        h2, med = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        st.median_resolve(hist, level, n, sel, h2, med)
        torch.cuda.synchronize()
        if rep: print("%s staged k_hist<%d> symmetric: %.1f us" % (which, level, a.elapsed_time(b) * 1e3), flush=True)
print("%s staged h2 %.6f vs fused %.6f" % (which, h2.item(), eng.h2.item()))
# (2) the fused miss path's median stage for several virtual-workgroup counts
for nvb in (0, 512, 1024, 1536, 2048):
    _lib.call("stein_debug_hist_all_vblocks", nvb)
    e2 = SvgdEngine(n, d, device=dev, dtype=dtype, window=False, small=False)
    for _ in range(3): e2.compute_phi(T, G)
    _lib.timing_reserve(10)
    for _ in range(10): e2.compute_phi(T, G, timing=True)
    calls = _lib.timing_read(10)
    med = sorted(c["median"] for c in calls)[5]
    dist = sorted(c["distance"] for c in calls)[5]
    print("%s fused, window off, nvb %4d: median stage %.1f us, distance %.1f us, h2 %.6f" % (which, nvb, med * 1e3, dist * 1e3, e2.h2.item()), flush=True)
_lib.call("stein_debug_hist_all_vblocks", 0)
# (3) with the window: the first two steps miss and take level 0 inside the launch too
e3 = SvgdEngine(n, d, device=dev, dtype=dtype, small=False)
_lib.timing_reserve(6)
for _ in range(6): e3.compute_phi(T, G, timing=True)
print("%s fused with window, steps 0..5 median stage us:" % which, ["%.1f" % (c["median"] * 1e3) for c in _lib.timing_read(6)])
