import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine, untile_distances
n, d = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda"
T = torch.tensor(np.random.default_rng(7).normal(size=(n, d)), dtype=torch.float32, device=dev)
G = torch.tensor(np.random.default_rng(8).normal(size=(n, d)), dtype=torch.float32, device=dev)
eng = SvgdEngine(n, d, device=dev, x3=True)
ref = SvgdEngine(n, d, device=dev, x3=True, tile_distance=True)
for step in range(5):
    eng.dist.fill_(float("nan")); ref.dist.fill_(float("nan"))
    phi = eng.compute_phi(T, G).clone(); phir = ref.compute_phi(T, G).clone()
    torch.cuda.synchronize()
    v = eng.dist.view(n // 128, eng.ld_dist // 32, 128, 32); vr = ref.dist.view(n // 128, eng.ld_dist // 32, 128, 32)
    miss = 0; bad = []
    for I in range(n // 128):
        up = v[I, 4 * I:n // 32]; upr = vr[I, 4 * I:n // 32]
        nanm = torch.isnan(up)
        miss += int(nanm.sum())
        diff = (up - upr).abs()
        diff[nanm] = 0
        if diff.max().item() > 1e-3: bad.append((I, diff.max().item(), int((diff > 1e-3).sum())))
        if I and not bool(torch.isnan(v[I, :4 * I]).all()): print("  wrote below diagonal in row tile", I)
    st = eng.select_state.view(torch.int32).cpu()
    print("step", step, "missing", miss, "bad tiles", bad[:6], "h2", eng.h2.item(), ref.h2.item(), "hit", int(st[16 + 7]), "skip_l0", int(st[16 + 13]),
          "count", int(st[16 + 5]), "overflow", int(st[16 + 6]), "stats", eng.window_stats(), ref.window_stats(), flush=True)
    T = T + 1e-3 * phir
