#!/bin/bash
# What clock does the chip hold inside each kernel of the C3 step?  rocprofv3 --pmc GRBM_COUNT GRBM_GUI_ACTIVE (cycles of the
# graphics clock domain elapsed / busy during a dispatch) joined with the dispatch's own begin / end timestamps:
# GRBM_COUNT / (end - begin) = the average shader clock over that launch.  The distance pass is run twice more with the
# window disabled (level-0 histogram epilogue) for comparison.  usage (gpurun, repo root): bash scratch/pmc_clock.sh <tag>
TAG=${1:-clock}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "GRBM_[A-Z_]*" $OUT/avail.txt | sort -u | tr '\n' ' ' > $OUT/grbm_counters.txt
for C in "GRBM_COUNT GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$N -o x -- python3 $GRAFT_REPO_ROOT/scratch/${SCRIPT:-fused_loop.py} ${LIB:-shipped} ${ARGS:-16384 256 12} > $OUT/$N.log 2>&1 || echo "counter set $N refused" >&2
done
python3 - <<PY
import csv, glob, os, collections, json
out = "$OUT"
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/*/")):
    pc = glob.glob(os.path.join(d, "**", "x_counter_collection.csv"), recursive=True)
    pk = glob.glob(os.path.join(d, "**", "x_kernel_trace.csv"), recursive=True)
    if not pc or not pk: continue
    dur = {}
    for r in csv.DictReader(open(pk[0])):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0].replace("void ", ""), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pc[0])):
        k, ns = dur.get(r["Dispatch_Id"], (None, None))
        if k and k.startswith("k_"):
            acc[k][r["Counter_Name"]].append((float(r["Counter_Value"]), ns))
    for k, cs in acc.items():
        for c, v in cs.items():
            v = v[4:] if len(v) > 6 else v          # drop the first steps (no window yet)
            res[k][c] = sum(x for x, _ in v) / len(v)
            res[k]["ns_in_" + c + "_run"] = sum(n for _, n in v) / len(v)
for k, cs in sorted(res.items()):
    line = {c: round(v, 1) for c, v in cs.items() if not c.startswith("ns_in_")}
    if "GRBM_COUNT" in cs:
        line["avg_clock_GHz"] = round(cs["GRBM_COUNT"] / cs["ns_in_GRBM_COUNT_run"], 3)
        line["ns"] = round(cs["ns_in_GRBM_COUNT_run"], 0)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
        line["mfma_busy_of_1024_simds_at_2p4GHz"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * 2.4 * cs["ns_in_SQ_VALU_MFMA_BUSY_CYCLES_run"]), 3)
    print(k, line)
json.dump(res, open(os.path.join(out, "clock_summary.json"), "w"), indent=1)
PY
