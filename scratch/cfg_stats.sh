#!/bin/bash
# rocprofv3 kernel stats of a bench workload other than the default: bash scratch/cfg_stats.sh <tag> <workload> <steps>
set -e -o pipefail
TAG=$1; WL=$2; STEPS=${3:-10}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $WL -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline --secondary none --no-other-configs --no-variants --no-train-on-batch > $OUT/bench_$WL.json 2> $OUT/stats_$WL.log
