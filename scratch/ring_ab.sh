#!/bin/bash
# same-box runs: four-slot ring against the five / six-slot ring (shipped), timings and phase stamps
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/ring_ab.txt
: > $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  STAMPLIB=lib_ring4.so timeout -k 10 300 python scratch/dist_ab.py c3 c5 2>&1 | grep panel | sed "s/^/ring4  /" >> $O || exit 1
  timeout -k 10 300 python scratch/dist_ab.py c3 c5 2>&1 | grep panel | sed "s/^/ring56 /" >> $O || exit 1
done
for L in stamps_ring4 stamps stamps_nostore; do
  STAMPLIB=lib_$L.so timeout -k 10 300 python scratch/stamps_dp.py 16384 256 2>&1 | grep -E "cycles per strip|median over waves|in-kernel clock" | sed "s/^/$L c3 /" >> $O || exit 1
  STAMPLIB=lib_$L.so timeout -k 10 300 python scratch/stamps_dp.py 131072 256 8 2>&1 | grep -E "cycles per strip|median over waves|in-kernel clock" | sed "s/^/$L c5 /" >> $O || exit 1
done
