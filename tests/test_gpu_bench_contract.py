"""GPU: bench.py prints ONE JSON line with the keys the driver's contract names (a short run of the C2 workload, which
takes a second; the default run is the same code on C3)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_line_has_the_contract_keys(cuda):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--steps", "4", "--warmup", "3",
           "--secondary", "none", "--no-other-configs", "--no-variants"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout                                   # exactly one line on stdout
    b = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    assert b["n_gpus"] == 1 and b["steps"] == 4 and b["warmup"] == 3 and b["higher_is_better"] is True
    assert b["vs_baseline"] is None and b["data"] == "synthetic" and "workload" in b["config"]
    assert abs(b["value"] - b["config"]["n"] * 1e3 / b["ms_per_step"]) <= 1e-6 * b["value"]     # particle-updates/s
    roof = b["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "mfma" and 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = b["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["value"] > 0
    # `achieved` / `frac` are the ALGORITHMIC 4 n_local n d flops per launch over the kernel's mean duration
    # (round 4; rounds 1-3 put the executed flops there).  The executed fraction, the counter reading and the fp32-peak view
    # stand beside it under their own names.
    for key in ("frac_executed", "frac_algorithmic", "mfma_busy_frac", "frac_vs_fp32_mfma_peak", "algorithmic_flops_per_launch",
                "ms_per_launch"):
        assert key in roof, key
    assert roof["frac_algorithmic"] == roof["frac"] and roof["frac"] <= roof["frac_executed"] <= 1.0
    n, d = b["config"]["n"], b["config"]["d"]
    assert roof["algorithmic_flops_per_launch"] == 4.0 * n * n * d
    alg = roof["algorithmic_flops_per_launch"] / (roof["ms_per_launch"] * 1e-3) / 1e12
    assert abs(roof["achieved"] - alg) <= 1e-9 * alg
    assert abs(roof["frac_vs_fp32_mfma_peak"] - alg / 157.3) <= 1e-9 * alg
    # every timed entry says how far the wall time per step is from the GPU time its HIP events bracket
    for key in ("events_ms", "wall_minus_events_ms", "host_stall_suspected", "instrumented_ms_per_step", "timing_note",
                "settle_steps", "first_block_ms_per_step"):
        assert key in b, key
    # (the headline's timed loop carries only the contraction's two events; the stage split and its gap describe the second,
    # fully instrumented loop, which cannot be faster than a loop with fewer events by more than noise)
    assert abs(b["wall_minus_events_ms"] - (b["instrumented_ms_per_step"] - b["events_ms"])) < 2e-4
    assert b["ms_per_step"] <= b["instrumented_ms_per_step"] * 1.05
    assert abs(roof["ms_per_launch"] - b["stage_ms"]["contract"]) <= 0.15 * b["stage_ms"]["contract"]
    # the host baseline says how many threads / cores it ran on and is timed on one thread as well, the distance pass reports
    # its HBM write rate
    for key in ("blas_threads", "logical_cpus", "physical_cores", "one_thread"):
        assert key in cpu, key
    assert cpu["one_thread"]["cores"] == 1 and cpu["one_thread"]["value"] > 0 and cpu["cores"] == cpu["blas_threads"]
    dp = b["distance_pass"]
    assert dp["bound"] == "hbm" and dp["achieved_write_GBps"] > 0 and 0.0 < dp["frac_of_hbm_peak"] < 1.0
    assert b["finite"] is True and b["parity_sample_relerr"] < 4e-3                            # bf16 inputs: K rounded to bf16


def test_train_on_batch_entry(cuda):
    """the C3-shaped SteinSampler.train_on_batch entry of the bench line: score recomputed every step, window statistics"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    ent = bench.train_on_batch_entry(torch, cuda, steps=6, warmup=3)
    assert ent["finite"] and ent["ms_per_step"] > 0 and ent["window"]["timed_steps"] == 6
    assert abs(ent["wall_minus_events_ms"] - (ent["ms_per_step"] - ent["events_ms"])) < 2e-4
    assert 0 <= ent["window"]["hits"] <= 6
