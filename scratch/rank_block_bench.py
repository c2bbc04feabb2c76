"""Per-rank stage times of a sharded run, measured on ONE GPU: the stages of rank 0 of `world` ranks (row block
[0, n/world) of an n-particle problem), without the collectives.  usage: rank_block_bench.py n d world"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import HipStages
n, d, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nl = n // world
dev = "cuda"
st = HipStages()
flags = _lib.FLAG_X3
total, offs, extra = st.workspace_layout(nl, n, d, flags)
ws = torch.empty(total, dtype=torch.uint8, device=dev)
sec = lambda k, nb: ws[offs[k]:offs[k] + nb]
ld = extra[_lib.WSX_LD_DIST]
rows = (nl + 127) // 128 * 128
r = sec(_lib.WS_ROWNORM, n * 4).view(torch.float32)
D = sec(_lib.WS_DIST, rows * ld * 4).view(torch.float32)
hist = sec(_lib.WS_HIST, 3 * 2 * 2048 * 8).view(torch.int64).view(3, 2, 2048)
sel = sec(_lib.WS_SELECT, 128)
spec = ws[offs[_lib.WS_SPEC]:offs[_lib.WS_PLANES]]
planes = ws[offs[_lib.WS_PLANES]:total]
T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
h2 = torch.zeros(1, device=dev); med = torch.zeros(1, device=dev)
phi = torch.empty(nl, d, device=dev); sq = torch.zeros(1, dtype=torch.float64, device=dev)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
res = {}
for rep in range(4):
    T.add_(1e-4 * torch.randn_like(T))
    marks = [("start", ev())]
    st.rownorms(T, n, d, r); st.x3_prepare(T, G, n, d, planes); marks.append(("prepare", ev()))
    st.spec_begin(hist, sel, spec, n * n)
    st.distance_block_spec(T, r, n, d, 0, nl, D, ld, hist[0], sel, spec, planes=planes); marks.append(("distance", ev()))
    st.spec_tally(sel, spec); st.spec_pick(sel, spec, n, h2, med); marks.append(("window tally+pick (no all-reduce)", ev()))
    flagsv = sel[_lib.SPEC_HIT_OFFSET:_lib.SPEC_SKIP_L0_OFFSET + 4].cpu()
    hit = bool(flagsv[:4].view(torch.int32).item()); skip0 = bool(flagsv[-4:].view(torch.int32).item())
    marks.append(("flag read-back", ev()))
    if not hit:     # a single rank's table only holds ITS rows: with world > 1 the pick misses here, so the radix passes run
        for lv in range(3):
            if lv > 0 or not skip0:
                st.median_hist_pass(D, ld, nl, n, lv, sel, hist)
            st.median_resolve(hist, lv, n, sel, h2, med)
    marks.append(("radix passes", ev()))
    st.spec_update(sel)
    st.contract_partial(D, ld, T, G, n, d, 0, nl, h2, ws, planes); marks.append(("contract", ev()))
    st.contract_finish(T, n, d, 0, nl, h2, phi, sq, None, ws, flags); marks.append(("finish", ev()))
    torch.cuda.synchronize()
    if rep:
        for (a, ea), (b, eb) in zip(marks[:-1], marks[1:]):
            res.setdefault(b, []).append(ea.elapsed_time(eb))
print("n=%d d=%d world=%d rows/rank=%d  (h2 here is the median of THIS block's rows only)" % (n, d, world, nl))
for k, v in res.items():
    print("  %-36s %8.3f ms" % (k, sum(v) / len(v)))
