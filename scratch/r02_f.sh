#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 100 ./scratch/probes/coissue > $OUT/coissue.txt 2>&1; cat $OUT/coissue.txt
