"""Error of phi against the fp64 oracle for the three GEMM paths (run once per STEIN_SPLIT_KIND: the kind is read once
per process).  usage: python scratch/accuracy.py            -> spawns itself for h2 / b3"""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import torch
    from oracle import svgd_oracle as orc
    from stein_amd.engine import SvgdEngine
    cases = [(1536, 256, 1.0, 1.0), (4096, 64, 1.0, 1.0), (2048, 128, 1e-3, 1e4), (1000, 130, 50.0, 1e-6)]
    for n, d, st, sg in cases:
        rng = np.random.default_rng(n + d)
        T64 = rng.normal(size=(n, d)) * st * np.exp(rng.normal(size=(1, d)) * 2.0)    # columns of very different scale
        G64 = rng.normal(size=(n, d)) * sg * np.exp(rng.normal(size=(1, d)) * 2.0)
        T = torch.tensor(T64, dtype=torch.float32, device="cuda"); G = torch.tensor(G64, dtype=torch.float32, device="cuda")
        ref = orc.compute_phi(T.double().cpu().numpy(), G.double().cpu().numpy(), exact64=True) if "exact64" in orc.compute_phi.__code__.co_varnames else None
        if ref is None:
            # fp64 everything, same median rule
            Td, Gd = T.double().cpu().numpy(), G.double().cpu().numpy()
            r = (Td * Td).sum(1); D = r[:, None] + r[None, :] - 2 * Td @ Td.T
            med = np.median(D); h2 = med / np.log(n); K = np.exp(-D / h2 / 2)
            ref = (K @ Gd + (K.sum(1)[:, None] * Td - K @ Td) / h2) / n
        out = {}
        for name, x3 in (("split", True), ("fp32mfma", False)):
            eng = SvgdEngine(n, d, device="cuda", x3=x3)
            phi = eng.compute_phi(T, G).double().cpu().numpy()
            colscale = np.abs(ref).max(0) + 1e-300
            out[name] = (np.abs(phi - ref).max(0) / colscale).max(), np.linalg.norm(phi - ref) / np.linalg.norm(ref)
        print(os.environ.get("STEIN_SPLIT_KIND", "h2"), (n, d), {k: ("%.2e" % a, "%.2e" % b) for k, (a, b) in out.items()})

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        main()
    else:
        for kind in ("h2", "b3"):
            env = dict(os.environ, STEIN_SPLIT_KIND=kind)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
