#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/phimap_ab.txt
: > $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for shape in "8192 2001" "16384 1000" "4096 2001"; do
    STAMPLIB=lib_rowmajor.so timeout -k 10 300 python scratch/phimap_ab.py $shape 2>&1 | grep "n=" | sed "s/^/rowmajor /" >> $O || exit 1
    timeout -k 10 300 python scratch/phimap_ab.py $shape 2>&1 | grep "n=" | sed "s/^/blocked  /" >> $O || exit 1
  done
done
