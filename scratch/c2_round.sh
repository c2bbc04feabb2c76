#!/bin/bash
# C2 (n=4096, d=2304 bf16) launch chain: bench line + rocprofv3 kernel stats of the same command.
# usage (via gpurun, from the repo root): bash scratch/c2_round.sh <tag>
set -e -o pipefail
TAG=${1:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
C2="--workload c2 --steps 200 --warmup 20 --no-cpu-baseline --secondary none --no-other-configs --no-variants --no-train-on-batch"
python3 bench.py $C2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py $C2 > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" >&2
