"""Long determinism soak of the panel distance kernels at full size (C3 symmetric, the C5 rank block, C4's deep K) with a
second stream hammering memory: every repetition must reproduce the first image bit for bit.  usage: soak_dp.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import HipStages
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda"
st = HipStages()
side = torch.cuda.Stream(device=dev)
a = torch.empty(128 << 20, dtype=torch.float32, device=dev); b = torch.empty_like(a)
for name, n, d, nl, sym in (("C3 symmetric", 16384, 256, 16384, True), ("C4 symmetric, K in chunks", 8192, 2001, 8192, True),
                            ("C5 rank block", 131072, 256, 16384, False)):
    total, offs, extra = st.workspace_layout(nl, n, d, _lib.FLAG_X3)
    ws = torch.empty(total, dtype=torch.uint8, device=dev)
    ld = extra[_lib.WSX_LD_DIST]
    r = ws[offs[_lib.WS_ROWNORM]:offs[_lib.WS_ROWNORM] + n * 4].view(torch.float32)
    D = ws[offs[_lib.WS_DIST]:offs[_lib.WS_DIST] + nl * ld * 4].view(torch.float32).view(nl, ld)
    planes = ws[offs[_lib.WS_PLANES]:total]
    T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
    st.rownorms(T, n, d, r); st.x3_prepare(T, G, n, d, planes)
    def run():
        st.distance_block(T, r, n, d, 0, nl, D, ld, symmetric=sym, planes=planes, kernel=_lib.STAGE_PANEL)
    D.fill_(float("nan")); run(); torch.cuda.synchronize()
    first = D.clone()
    bad = 0
    for rep in range(reps):
        with torch.cuda.stream(side):
            for _ in range(rep % 5):
                b.copy_(a)
        run(); torch.cuda.synchronize()
        same = (D == first) | (D.isnan() & first.isnan())
        if not bool(same.all()):
            bad += 1
            print("   %s: repetition %d differs in %d entries" % (name, rep, int((~same).sum())), flush=True)
    print("%s (n=%d d=%d rows=%d): %d repetitions, %d differed" % (name, n, d, nl, reps, bad), flush=True)
    del ws, D, first, T, G
    torch.cuda.empty_cache()
