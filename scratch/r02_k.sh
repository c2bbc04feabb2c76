#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python scratch/ab.py 16384 256 lib_rot0.so lib_rot1.so lib_rot2.so > $OUT/ab_rot.txt 2>&1; cat $OUT/ab_rot.txt
