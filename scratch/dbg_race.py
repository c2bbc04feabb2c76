import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.engine import SvgdEngine
def ref(T, G, h2):
    T, G = T.double(), G.double(); n = T.shape[0]
    r = (T * T).sum(1); D = r[:, None] + r[None, :] - 2.0 * (T @ T.T)
    K = torch.exp(-D / h2 / 2.0)
    return (K @ G + (K.sum(1)[:, None] * T - K @ T) / h2) / n
for n, d, dt, reps in [(1024, 128, torch.bfloat16, 40), (4096, 128, torch.bfloat16, 20), (1024, 128, torch.float32, 40), (4096, 256, torch.float32, 20), (2048, 2001, torch.float32, 6)]:
    g = torch.Generator().manual_seed(n + d)
    T = torch.randn(n, d, generator=g).cuda(); G = torch.randn(n, d, generator=g).cuda()
    if dt == torch.bfloat16: T, G = T.bfloat16().float(), G.bfloat16().float()
    eng = SvgdEngine(n, d, device="cuda", small=False, dtype=dt)
    tol = 1e-5 if dt == torch.float32 else 4e-3
    bad = {}
    first = None
    for rep in range(reps):
        phi = eng.compute_phi(T.to(dt), G.to(dt)).double().clone(); torch.cuda.synchronize()
        if first is None:
            r = ref(T, G, eng.h2.item()); first = phi
        e = (phi - r).norm(dim=1) / r.norm(dim=1)
        blocks = [i // 128 for i in (e > 10 * tol).nonzero().flatten().tolist()]
        for b in set(blocks): bad[b] = bad.get(b, 0) + 1
    print((n, d), str(dt).split(".")[-1], "split", eng.split, "calls", reps, "bad row blocks {block: calls}:", dict(sorted(bad.items())), flush=True)
