#!/bin/bash
# Clock and matrix-pipe occupancy of the timing-only ablations of k_distance_panel (scratch/dist_ablate.sh): is a part's added time
# a lower clock (power) or an emptier pipe (schedule)?  One rocprofv3 --pmc pass per library over scratch/dist_loop.py.
# clock = SQ_BUSY_CU_CYCLES / 256 / launch time; pipe busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024) / (SQ_BUSY_CU_CYCLES / 256).
# usage (gpurun, repo root): bash scratch/dist_ablate_pmc.sh <tag>
TAG=${1:-ablpmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in ${LIBS:-shipped lib_nostore.so lib_noepi.so lib_noepi_nostream.so lib_noepi_nopanel.so lib_mfmaonly.so}; do
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$L -o x -- python3 $GRAFT_REPO_ROOT/scratch/dist_loop.py $L 16 ${DL_ARGS:-} > $OUT/$L.log 2>&1 || echo "$L refused" >&2
done
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
for d in sorted(glob.glob(out + "/*/")):
    pc = glob.glob(os.path.join(d, "**", "x_counter_collection.csv"), recursive=True)
    pk = glob.glob(os.path.join(d, "**", "x_kernel_trace.csv"), recursive=True)
    if not pc or not pk: continue
    dur = {r["Dispatch_Id"]: float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(pk[0])) if "k_distance_panel" in r["Kernel_Name"]}
    acc = collections.defaultdict(dict)
    for r in csv.DictReader(open(pc[0])):
        if r["Dispatch_Id"] in dur: acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(acc, key=int)[3:]
    m = lambda f: sum(f(i) for i in ids) / len(ids)
    ns = m(lambda i: dur[i]); cu = m(lambda i: acc[i]["SQ_BUSY_CU_CYCLES"]); mf = m(lambda i: acc[i]["SQ_VALU_MFMA_BUSY_CYCLES"])
    print("%-24s %7.1f us  clock %.3f GHz  pipe busy %.3f  (MFMA busy/SIMD %.0f, LDS wait %.0f, LDS idx active %.0f)" % (
        os.path.basename(d.rstrip("/")), ns / 1e3, cu / 256 / ns, (mf / 1024) / (cu / 256), mf / 1024,
        m(lambda i: acc[i].get("SQ_WAIT_INST_LDS", 0)), m(lambda i: acc[i].get("SQ_LDS_IDX_ACTIVE", 0))))
PY
rm -rf $OUT/*/
