"""gpurun_out/<tag>/ (scratch/profile_round.sh) -> the tracked evidence under profiles/:
  profiles/<name>_bench.json          the bench line of that call
  profiles/<name>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same bench command (k_* kernels)
  profiles/<name>_pmc_summary.json    per-kernel FETCH_SIZE / WRITE_SIZE / MFMA-busy averages of the timed launches
  profiles/pmc_traffic.json           the dominant kernel's HBM bytes per launch, with its source and commit (bench.py)
usage: python scratch/summarize_profile.py <tag> <name> [commit]"""
import csv, collections, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, name + "_bench.json"))
rows = list(csv.DictReader(open(os.path.join(src, "stats", "c3_kernel_stats.csv"))))
with open(os.path.join(dst, name + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
    w.writeheader()
    for r in rows:
        if "k_" in r["Name"].split("(")[0]:
            w.writerow(r)
WARM = 4      # warm-up launches of the PMC command (--warmup 4), dropped: the median window has no history there

def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v[WARM:]) / len(v[WARM:]) if len(v) > WARM else sum(v) / len(v)) for c, v in cs.items()} for k, cs in acc.items()}

summ = collections.defaultdict(dict)
for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_mfma"):
    p = os.path.join(src, sub, "c3_counter_collection.csv")
    if os.path.exists(p):
        for k, cs in per_kernel(p).items():
            summ[k].update(cs)
for k, cs in summ.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        cs["hbm_bytes_per_launch_corrected"] = (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0
out = {"workload": "c3 n=16384 d=256 fp32 inputs, split fp16 x 2 GEMM path, fused call with the speculative median window in steady state, 1 GPU",
       "command": "scratch/profile_round.sh: rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python3 bench.py --steps 6 --warmup 4 "
                  "--no-cpu-baseline --secondary none --no-other-configs --no-variants, one run per counter set (FETCH_SIZE | WRITE_SIZE | "
                  "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES); per-launch averages over the 6 timed steps",
       "correction": "gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1 KiB",
       "commit": commit, "kernels": summ}
json.dump(out, open(os.path.join(dst, name + "_pmc_summary.json"), "w"), indent=1)
tr = {"source": "profiles/%s_pmc_summary.json" % name, "commit": commit, "c3": {}}
avg_ns = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) for r in rows}
for k, cs in summ.items():
    if k.startswith("k_phi_x3fs") and "hbm_bytes_per_launch_corrected" in cs:
        tr["c3"]["k_phi_x3fs_hbm_bytes"] = cs["hbm_bytes_per_launch_corrected"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and k in avg_ns:   # busy cycles summed over the 1024 SIMDs / (1024 x 2.4 GHz x time)
            tr["c3"]["k_phi_x3fs_mfma_busy_frac"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * 2.4 * avg_ns[k])
            tr["c3"]["k_phi_x3fs_avg_ns_rocprofv3"] = avg_ns[k]
    if k.startswith("k_distance") and "hbm_bytes_per_launch_corrected" in cs:
        tr["c3"]["k_distance_hbm_bytes"] = cs["hbm_bytes_per_launch_corrected"]
        tr["c3"]["k_distance_kernel"] = k
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and k in avg_ns:
            tr["c3"]["k_distance_mfma_busy_frac"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * 2.4 * avg_ns[k])
            tr["c3"]["k_distance_avg_ns_rocprofv3"] = avg_ns[k]
old = json.load(open(os.path.join(dst, "pmc_traffic.json")))
if "k_phi_partial_hbm_bytes" in old.get("c3", {}):
    tr["c3"]["k_phi_partial_hbm_bytes"] = old["c3"]["k_phi_partial_hbm_bytes"]
    tr["c3"]["k_phi_partial_source"] = "profiles/r01_c3_pmc_summary.json (round 1; that kernel is unchanged)"
json.dump(tr, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
for k in sorted(summ):
    print(k, {c: round(v, 1) for c, v in summ[k].items()})
