#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
STAMPLIB=lib_stamps.so timeout -k 10 120 python scratch/stamps.py 16384 256 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps.txt
timeout -k 10 500 python scratch/ab.py 16384 256 ../stein_amd/libsteinhip.so lib_p_prio1.so lib_p_prio3.so lib_rot0.so > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
