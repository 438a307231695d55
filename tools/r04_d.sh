#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/r04d; rm -rf $O; mkdir -p $O
timeout -k 10 120 ./tools/micro/mfma_shape > $O/mfma_shape.txt 2>&1; echo "mfma_shape rc=$?"; cat $O/mfma_shape.txt
export RVA_LIB_PATH=$ROOT/tools/_dbg/librva_exp.so
for spec in "128 128 3 1 40 32 56 90 52 91" "256 256 3 1 20 32 56 90 52 91" "128 128 3 1 80 32 56 90 52 91"; do
  timeout -k 10 120 python3 tools/sweep_run.py $spec >> $O/nosel.txt 2>&1; echo "sweep rc=$?"
done
cat $O/nosel.txt
