import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.engine import SvgdEngine
n, d = 1024, 128
g = torch.Generator().manual_seed(n + d)
T = torch.randn(n, d, generator=g).cuda().bfloat16(); G = torch.randn(n, d, generator=g).cuda().bfloat16()
eng = SvgdEngine(n, d, device="cuda", small=False, dtype=torch.bfloat16)
dK = torch.empty(n, d, device="cuda")
phi = eng.compute_phi(T, G, dK_out=dK).double(); torch.cuda.synchronize()
h2 = eng.h2.item()
Td, Gd = T.double(), G.double()
r = (Td * Td).sum(1); D = r[:, None] + r[None, :] - 2.0 * (Td @ Td.T)
K = torch.exp(-D / h2 / 2.0)
dKr = (K.sum(1)[:, None] * Td - K @ Td) / h2
KG = K @ Gd
kg_gpu = phi * n - dK.double()
for name, a, b in (("K.G", kg_gpu, KG), ("dK", dK.double(), dKr)):
    e = (a - b).norm(dim=1) / b.norm(dim=1)
    print(name, "rel err per 128-row block:", [round(float(e[i:i + 128].max()), 4) for i in range(0, n, 128)])
    ec = (a - b).norm(dim=0) / b.norm(dim=0)
    print(name, "rel err per 32-col block:", [round(float(ec[i:i + 32].max()), 4) for i in range(0, d, 32)])
# row sums: dK = (rs * T - K.T)/h2 -> compare K.T via phi path impossible; print a few rows
i = 200
print("row", i, "dK gpu", dK[i, :4].tolist(), "ref", dKr[i, :4].tolist())
e = ((kg_gpu - KG).abs() / KG.abs().max())
blk = e[128:256]
print("block 1: rows with err>1e-2:", (blk.max(1).values > 1e-2).nonzero().flatten().tolist()[:40])
print("block 1: cols with err>1e-2:", (blk.max(0).values > 1e-2).nonzero().flatten().tolist()[:40])
# which j range is responsible?  recompute the partial sums over j < 128, 128..255, >= 256 and see which one is off
Kb = K[128:256]
for lo, hi in ((0, 128), (128, 256), (256, 1024)):
    part = Kb[:, lo:hi] @ Gd[lo:hi]
    print("j in [%d,%d): norm of the partial %.4e" % (lo, hi, part.norm().item()))
diff = (kg_gpu - KG)[128:256]
for lo, hi in ((0, 128), (128, 256), (256, 1024)):
    part = Kb[:, lo:hi] @ Gd[lo:hi]
    # if the GPU dropped / doubled / misplaced this partial, diff would correlate with it
    c = (diff * part).sum() / (part * part).sum()
    print("projection of the error on the partial over [%d,%d): %.4f" % (lo, hi, c.item()))
# the same projection with the G rows of the tile permuted by the rotation candidates
D2 = eng.dist_matrix().double()
print("D mirrored vs fp64: max rel err block(1,0) %.3e, block(0,1) %.3e" % (((D2 - D)[128:256, :128].abs().max() / D.abs().max()).item(), ((D2 - D)[:128, 128:256].abs().max() / D.abs().max()).item()))
