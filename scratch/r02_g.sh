#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python scratch/ab.py 16384 256 ../stein_amd/libsteinhip.so lib_c_nomfma.so lib_c_nolds.so lib_p_prio3.so > $OUT/ab_contract.txt 2>&1; cat $OUT/ab_contract.txt
for v in s_c_nomfma s_c_nolds; do
  echo "== $v" | tee -a $OUT/stamps_abl.txt
  STAMPLIB=lib_$v.so timeout -k 10 120 python scratch/stamps.py 16384 256 2>&1 | tee -a $OUT/stamps_abl.txt
done
