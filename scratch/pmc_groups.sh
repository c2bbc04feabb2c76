#!/bin/bash
# Sequencer-level counters of the two MFMA kernels of the C3 step (where do the waves wait?): one rocprofv3 --pmc pass per
# group over scratch/fused_loop.py.  usage (gpurun, repo root): bash scratch/pmc_groups.sh <tag>
TAG=${1:-groups}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/g$i -o x -- python3 $GRAFT_REPO_ROOT/scratch/fused_loop.py ${LIB:-shipped} 16384 256 10 > $OUT/g$i.log 2>&1 || echo "group $i refused: $C" >&2
done <<GROUPS
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_IFETCH_LEVEL
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA
GROUPS
python3 - <<PY
import csv, glob, os, collections, json
out = "$OUT"
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/g*/")):
    pc = glob.glob(os.path.join(d, "**", "x_counter_collection.csv"), recursive=True)
    if not pc: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pc[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_distance") or k.startswith("k_phi_x3fs"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            v = v[4:] if len(v) > 6 else v
            res[k][c] = sum(v) / len(v)
for k, cs in sorted(res.items()):
    print(k)
    for c in sorted(cs): print("   %-34s %16.0f" % (c, cs[c]))
json.dump(res, open(os.path.join(out, "groups_summary.json"), "w"), indent=1)
PY
rm -rf $OUT/g*/
