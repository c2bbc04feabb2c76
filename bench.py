#!/usr/bin/env python3
"""bench.py -- SVGD particle-updates/s on MI355X, with the MFMA roofline of the dominant kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full particle update given the score matrix G (resident in HBM): the fused C call
stein_svgd_phi (row norms, operand scales and split planes -> distance pass on the 16-bit matrix cores with the
speculative median window -> exact median -> fused exp + split-precision MFMA contraction K.[G|theta] -> phi,
|phi|^2) followed by clip + Adagrad apply (lr 1e-3, alpha 0.9); particles move every step.  Workload:
BASELINE.json's roofline config "n=16384, d=256, fp32" (C3, the configuration the north star's contraction
target is quoted on); with N ranks the same n is sharded by rows (strong scaling): all-gathers of the theta and G
rows, the median collectives (stein_amd/engine.py) and one scalar all-reduce per step over RCCL.  The secondary
entry times C5 (n=131072) the same way; a one-GPU run also times the other BASELINE configs (C1, C2 bf16, C4) for 20
steps each and reports them under `other_configs` (they are parity-test cases, not the headline value).

value = n * K / (max-over-ranks wall time of the K timed steps), bracketed by barrier + synchronize.
roofline: the K.[G|theta] contraction kernel (k_phi_x3fs, or k_phi_partial with STEIN_X3=0), algorithmic flops
4*n_local*n*d per launch (2*n^2*d for K.G plus 2*n^2*d for K.theta), duration from HIP events
recorded around each of its launches on the launching stream inside the timed region (by the library at one rank:
STEIN_FLAG_TIMING); peak = 157.3 TFLOP/s dense fp32 MFMA (MI355X_MICROARCH.md).
cpu_baseline: the NumPy oracle (a port: the reference's TF-1.12 kernel graph cannot run anywhere here) timed
on the host on a bounded row block of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA = 157.3e12      # MI355X_MICROARCH.md, chip-level parameters (dense fp32-input MFMA)
PEAK_BF16_MFMA = 2.5e15        # same table: dense bf16 MFMA
WORKLOADS = {
    "c3": dict(n=16384, d=256, name="C3 bayesian-logreg-shaped synthetic block n=16384 d=256 fp32 (roofline config)"),
    "c5": dict(n=131072, d=256, name="C5 n=131072 d=256 fp32, rows sharded over the ranks"),
    "c2": dict(n=4096, d=128, bf16=True, name="C2 n=4096 d=128, theta and score fed to the kernels in bf16 (fp32 master copy)"),
    "c2f32": dict(n=4096, d=128, name="C2-shaped n=4096 d=128 run in fp32"),
    "c4": dict(n=8192, d=2001, name="C4 BNN-shaped n=8192 d=2001 (H=666) fp32"),
    "c1": dict(n=100, d=10, name="C1 n=100 d=10"),
}


def make_inputs(n, d, row0, n_local, device, torch):
    # T ~ N(0,1) seed 0, G ~ N(0,1) seed 1, generated in fp64 then cast
    T = np.random.default_rng(0).normal(size=(n, d))
    G = np.random.default_rng(1).normal(size=(n, d))
    Tl = torch.tensor(T[row0:row0 + n_local], dtype=torch.float32, device=device).contiguous()
    Gl = torch.tensor(G[row0:row0 + n_local], dtype=torch.float32, device=device).contiguous()
    return T, G, Tl, Gl


class StageClock:
    """Records a HIP event (torch.cuda.Event on the launching = current stream) at every stage mark."""

    def __init__(self, torch):
        self.torch = torch
        self.steps = []
        self.cur = None

    def begin_step(self):
        self.cur = []
        self.steps.append(self.cur)

    def mark(self, label):
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record()
        self.cur.append((label, ev))

    def summary(self):
        """-> {label: mean ms between this mark and the next one}"""
        acc = {}
        for marks in self.steps:
            for (la, ea), (_, eb) in zip(marks[:-1], marks[1:]):
                acc.setdefault(la, []).append(ea.elapsed_time(eb))
        return {k: float(np.mean(v)) for k, v in acc.items()}


def run_workload(torch, dist, args, wl, device, rank, world, group, steps, warmup, clock_stages=True):
    from stein_amd import _lib
    from stein_amd.engine import SvgdEngine
    from stein_amd.optimizers import AdagradGradientDescent
    n, d = wl["n"], wl["d"]
    n_local = n // world
    row0 = rank * n_local
    T64, G64, theta, G = make_inputs(n, d, row0, n_local, device, torch)
    bf16 = bool(wl.get("bf16"))      # BASELINE config 2: the kernels see bf16 theta / score; theta's master copy stays fp32
    eng = SvgdEngine(n, d, device=device, group=group, dtype=torch.bfloat16 if bf16 else torch.float32)
    gd = AdagradGradientDescent(learning_rate=1e-3, alpha=0.9)
    if bf16:
        G = G.to(torch.bfloat16)
    feed = (lambda: theta.to(torch.bfloat16)) if bf16 else (lambda: theta)
    clock = StageClock(torch)

    # Single rank: the fused C call (what SteinSampler.update_particles runs), with the library recording HIP events
    # at its stage boundaries on the launching stream (STEIN_FLAG_TIMING).  Several ranks: the staged calls with a
    # HIP event at every mark.
    fused = world == 1
    apply_events = []
    if fused and clock_stages:
        _lib.timing_reserve(steps)

    def step(timed):
        if fused:
            phi = eng.compute_phi(feed(), G, timing=timed and clock_stages)
            if timed and clock_stages:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gd.apply_(theta, phi, eng.sqnorm)
                e1.record()
                apply_events.append((e0, e1))
            else:
                gd.apply_(theta, phi, eng.sqnorm)
        elif timed and clock_stages:
            clock.begin_step()
            phi = eng.compute_phi(feed(), G, mark=clock.mark)
            gd.apply_(theta, phi, eng.sqnorm)
            clock.mark("apply_end")
        else:
            phi = eng.compute_phi(feed(), G, mark=(lambda s: None) if clock_stages else None)
            gd.apply_(theta, phi, eng.sqnorm)

    for _ in range(warmup):
        step(False)
    if world > 1:
        dist.barrier(group=group)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier(group=group)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(te.item())
    if fused and clock_stages:
        per_call = _lib.timing_read(steps)
        stages = {k: round(sum(c[k] for c in per_call) / max(1, len(per_call)), 4) for k in _lib.T_STAGES}
        stages["apply"] = round(sum(a.elapsed_time(b) for a, b in apply_events) / max(1, len(apply_events)), 4)
    else:
        stages = clock.summary() if clock_stages else {}
        if "end" in stages:            # interval end -> apply_end is the optimizer apply kernel
            stages["apply"] = stages.pop("end")
    finite = bool(torch.isfinite(theta).all().item())
    return dict(n=n, d=d, n_local=n_local, elapsed=elapsed, stages=stages, finite=finite, split=eng.split,
                ws_bytes=eng.ws_bytes, T64=T64, G64=G64, x3=eng.x3)


def cpu_baseline(wl, T64, G64, rows):
    from oracle import svgd_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        info = threadpool_info()
        cores = max([i.get("num_threads", 1) for i in info] or [1])
        blas = ",".join(sorted({"%s %s" % (i.get("internal_api"), i.get("version")) for i in info}))
    except Exception:
        cores, blas = os.cpu_count() or 1, "unknown"
    n, d = wl["n"], wl["d"]
    rows = min(rows, n)
    gd = orc.AdagradState(learning_rate=1e-3, alpha=0.9)
    t0 = time.perf_counter()
    orc.svgd_step_rows(T64, G64, 0, rows, gd, np.float32)
    dt = time.perf_counter() - t0
    return dict(value=rows / dt, unit="particle-updates/s", cores=int(cores), kind="port",
                sample="one step of the NumPy oracle (fp32 kernel, fp64 contraction, %s) on rows [0,%d) of the same "
                       "n=%d d=%d inputs: %.1f s of CPU work; the reference's TF-1.12 graph cannot run here" %
                       (blas, rows, n, d, dt),
                seconds=dt)


# products per fp32 operand pair in the split path (stein_x3.hip: KIND 2 -> 3 fp16 products, KIND 3 -> 6 bf16 products)
NPROD = 3


def pmc_traffic(workload_key, x3):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (profiles/pmc_traffic.json), if any."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(workload_key, {}).get("k_phi_x3fs_hbm_bytes" if x3 else "k_phi_partial_hbm_bytes")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=0, help="override the workload's particle count (experiments)")
    ap.add_argument("--d", type=int, default=0, help="override the workload's parameter count (experiments)")
    ap.add_argument("--cpu-rows", type=int, default=16384, help="rows of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--secondary", default="c5", help="also time this workload briefly (extra key); 'none' to skip")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the brief C1 / C2 / C4 timings (one GPU)")
    args = ap.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path in stein_amd)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("STEIN_REHEARSAL_ONE_GPU"):   # rehearse the multi-rank path with all ranks on one card
        local_rank = 0
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist, group = None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("STEIN_DIST_BACKEND", "nccl")   # RCCL; "gloo" only for one-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
        group = dist.group.WORLD

    import __graft_entry__ as ge
    ge.build()

    wl = dict(WORKLOADS[args.workload])
    if args.n or args.d:
        wl["n"], wl["d"] = args.n or wl["n"], args.d or wl["d"]
        wl["name"] = "custom n=%d d=%d fp32" % (wl["n"], wl["d"])
    res = run_workload(torch, dist, args, wl, device, rank, world, group, args.steps, args.warmup)
    n, d, nl = res["n"], res["d"], res["n_local"]
    ms_per_step = res["elapsed"] / args.steps * 1e3
    value = n * args.steps / res["elapsed"]
    global NPROD
    if wl.get("bf16"):
        NPROD = 1               # bf16 inputs: one bf16 product per pair
    k_ms = res["stages"].get("contract")
    flops = 4.0 * nl * n * d
    achieved = flops / (k_ms * 1e-3) if k_ms else None
    out = {
        "metric": "SVGD particle-updates/sec",
        "value": value,
        "unit": "particle-updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "bf16" if wl.get("bf16") else "f32",
        "data": "synthetic",
        "config": {"workload": wl["name"], "n": n, "d": d, "rows_per_rank": nl,
                   "optimizer": "adagrad lr=1e-3 alpha=0.9",
                   "parallelism": "rows sharded x%d, all-gather(theta,G) + median (3 histogram all-reduces, or one window-table "
                                  "all-reduce when the local block has >= 2^27 entries) + 1 scalar all-reduce" % world
                   if world > 1 else "single GPU"},
        "element_updates_per_s": value * d,
        "pair_interactions_per_s": value * n,
        "roofline": {
            "kernel": ("k_phi_x3fs (exp + split-precision MFMA K.[G|theta] contraction, %d 16-bit products per fp32 pair)"
                       % NPROD if res["x3"] else "k_phi_partial (exp + fp32-input MFMA K.[G|theta] contraction)"),
            "bound": "mfma", "achieved": achieved / 1e12 if achieved else None, "peak": PEAK_FP32_MFMA / 1e12,
            "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA if achieved else None,
            "flops_per_launch": flops, "ms_per_launch": k_ms,
            "traffic": pmc_traffic(args.workload, res["x3"]) if world == 1 else None,
            "note": ("achieved = ALGORITHMIC fp32 flops (4 n_local n d) / kernel time against the dense fp32-input MFMA "
                     "peak, as SURVEY 7/8(d) prescribes for split-precision emulation; the kernel executes %dx that many "
                     "16-bit MFMA flops" % NPROD if res["x3"] else "fp32-input MFMA kernel"),
            "executed_mfma_tflops": (NPROD if res["x3"] else 1.0) * achieved / 1e12 if achieved else None,
            "frac_of_executed_dtype_peak": ((NPROD * achieved / PEAK_BF16_MFMA) if res["x3"] else achieved / PEAK_FP32_MFMA)
            if achieved else None,
        },
        "gemm_path": ({1: "bf16 inputs (1 product)", 3: "split fp16 x 2 (3 products)", 6: "split bf16 x 3 (6 products)"}[NPROD]) if res["x3"] else "fp32 mfma",
        "stage_ms": {k: round(v, 4) for k, v in res["stages"].items()},
        "full_step_tflops": 6.0 * nl * n * d / (ms_per_step * 1e-3) / 1e12,
        "finite": res["finite"],
    }
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, res["T64"], res["G64"], args.cpu_rows)
    del res

    if args.secondary != "none" and args.secondary != args.workload:
        wl2 = WORKLOADS[args.secondary]
        need = (wl2["n"] // world) * wl2["n"] * 4 * 1.15 + 4 * wl2["n"] * wl2["d"] * 8
        free = torch.cuda.mem_get_info(device)[0]
        if need < free * 0.9:
            torch.cuda.empty_cache()
            r2 = run_workload(torch, dist, args, wl2, device, rank, world, group, args.secondary_steps, 3)   # 3 warm-up steps: the median predictor needs two medians of history
            k2 = r2["stages"].get("contract")
            f2 = 4.0 * r2["n_local"] * r2["n"] * r2["d"]
            out["secondary"] = {
                "workload": wl2["name"], "n": r2["n"], "d": r2["d"], "steps": args.secondary_steps,
                "ms_per_step": r2["elapsed"] / args.secondary_steps * 1e3,
                "value": r2["n"] * args.secondary_steps / r2["elapsed"], "unit": "particle-updates/s",
                "scaling": "strong (fixed n=%d)" % r2["n"],
                "contract_tflops": f2 / (k2 * 1e-3) / 1e12 if k2 else None,
                "stage_ms": {k: round(v, 4) for k, v in r2["stages"].items()}, "finite": r2["finite"],
            }
    # one GPU: the remaining BASELINE configs (parity-test cases, timed briefly for the record; not the headline value)
    if world == 1 and args.workload == "c3" and not args.no_other_configs:
        others = {}
        for key in ("c1", "c2", "c4"):
            torch.cuda.empty_cache()
            ro = run_workload(torch, dist, args, dict(WORKLOADS[key]), device, rank, world, group, 20, 3, clock_stages=False)
            others[key] = {"workload": WORKLOADS[key]["name"], "n": ro["n"], "d": ro["d"], "steps": 20,
                           "ms_per_step": ro["elapsed"] / 20 * 1e3, "value": ro["n"] * 20 / ro["elapsed"],
                           "unit": "particle-updates/s", "finite": ro["finite"]}
            if not args.no_cpu_baseline:     # the NumPy oracle on the same inputs (C4: a bounded row block)
                others[key]["cpu_baseline"] = cpu_baseline(WORKLOADS[key], ro["T64"], ro["G64"],
                                                           2048 if key == "c4" else ro["n"])
            del ro
        out["other_configs"] = others
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier(group=group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
