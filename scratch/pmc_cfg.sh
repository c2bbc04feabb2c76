#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / MFMA-busy counters of a bench workload's kernels: bash scratch/pmc_cfg.sh <tag> <workload>
set -e -o pipefail
TAG=$1; WL=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 6 --warmup 4 --no-cpu-baseline --secondary none --no-other-configs --no-variants --no-train-on-batch"
for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$N -o x -- $B > $OUT/$N.log 2>&1
done
python3 - <<PY
import csv, glob, os, collections, json
out = "$OUT"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(out + "/*/")):
    p = os.path.join(d, "x_counter_collection.csv")
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = os.path.join(d, "x_kernel_trace.csv")
    if "FETCH" in d:
        for r in csv.DictReader(open(t)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("k_"): dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for k, cs in acc.items():
    e = {c: sum(v[4:]) / max(1, len(v[4:])) for c, v in cs.items()}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e: e["hbm_bytes_per_launch_corrected"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    v = dur.get(k, [])
    if v: e["avg_us"] = sum(v[4:]) / max(1, len(v[4:]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and v: e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * 2.4e3 * e["avg_us"])
    res[k] = e
json.dump({"workload": "$WL", "kernels": res}, open(out + "/pmc_summary.json", "w"), indent=1)
for k, e in res.items(): print(k, {a: round(b, 3) for a, b in e.items()})
PY
