"""Same-box A/B of the two distance kernels (per-tile k_distance_x3 vs panel-resident k_distance_panel).
usage: python scratch/dist_ab.py [c3] [c5]
  c3: fused C3 steps (window path), library stage events, engines interleaved
  c5: rank 0's 16384 x 131072 row block of C5 (window form of the staged calls), torch events"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
if os.environ.get("STAMPLIB"): _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["STAMPLIB"])   # a scratch/build_variant.py library
from stein_amd.engine import SvgdEngine, HipStages
from stein_amd.optimizers import AdagradGradientDescent
what = sys.argv[1:] or ["c3", "c5"]
dev = "cuda"
if "c3" in what:
    n, d = 16384, 256
    torch.manual_seed(0)
    res = {}
    engs = {"tiles": SvgdEngine(n, d, device=dev, tile_distance=True), "panel": SvgdEngine(n, d, device=dev)}
    th = {k: torch.randn(n, d, device=dev, generator=torch.Generator(dev).manual_seed(1)) for k in engs}
    G = torch.randn(n, d, device=dev)
    gd = {k: AdagradGradientDescent(learning_rate=1e-3) for k in engs}
    for k, e in engs.items():
        for _ in range(6):
            phi = e.compute_phi(th[k], G); gd[k].apply_(th[k], phi, e.sqnorm)
    for rnd in range(3):
        for k, e in engs.items():
            steps = 20
            _lib.timing_reserve(steps)
            for _ in range(steps):
                phi = e.compute_phi(th[k], G, timing=True); gd[k].apply_(th[k], phi, e.sqnorm)
            torch.cuda.synchronize()
            per = _lib.timing_read(steps)
            res.setdefault(k, []).append({s: sum(c[s] for c in per) / len(per) for s in _lib.T_STAGES})
    for k in engs:
        print("C3 fused", k, {s: round(statistics.median(r[s] for r in res[k]), 4) for s in _lib.T_STAGES},
              "window", engs[k].window_stats(), flush=True)
    del engs
    torch.cuda.empty_cache()
if "c5" in what:
    n, d, world = 131072, 256, 8
    nl = n // world
    st = HipStages()
    total, offs, extra = st.workspace_layout(nl, n, d, _lib.FLAG_X3)
    ws = torch.empty(total, dtype=torch.uint8, device=dev)
    ld = extra[_lib.WSX_LD_DIST]
    r = ws[offs[_lib.WS_ROWNORM]:offs[_lib.WS_ROWNORM] + n * 4].view(torch.float32)
    D = ws[offs[_lib.WS_DIST]:offs[_lib.WS_DIST] + nl * ld * 4].view(torch.float32).view(nl, ld)
    hist = ws[offs[_lib.WS_HIST]:offs[_lib.WS_HIST] + 3 * 2 * 2048 * 8].view(torch.int64).view(3, 2, 2048)
    sel = ws[offs[_lib.WS_SELECT]:offs[_lib.WS_SELECT] + 128]
    sel.zero_()
    spec = ws[offs[_lib.WS_SPEC]:offs[_lib.WS_PLANES]]
    planes = ws[offs[_lib.WS_PLANES]:total]
    T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
    st.rownorms(T, n, d, r); st.x3_prepare(T, G, n, d, planes)
    out = {"tiles": [], "panel": []}
    for rep in range(4):
        for name, kernel in (("tiles", _lib.STAGE_TILES), ("panel", _lib.STAGE_PANEL)):
            st.spec_begin(hist, sel, spec, n * n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st.distance_block_spec(T, r, n, d, 0, nl, D, ld, hist[0], sel, spec, planes=planes, kernel=kernel)
            e1.record(); torch.cuda.synchronize()
            if rep: out[name].append(e0.elapsed_time(e1))
    for k, v in out.items():
        print("C5 rank block 16384 x 131072 (no window yet: plain epilogue)", k, [round(x, 3) for x in v], flush=True)
