#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04e; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py -x -q -k "every_conv_variant" > $O/pytest_conv.log 2>&1; rc=$?; echo "pytest conv rc=$rc"; tail -5 $O/pytest_conv.log
[ $rc -eq 0 ] || exit 1
export RVA_LIB_PATH=$ROOT/tools/_dbg/librva_exp.so
for spec in "128 128 3 1 40 32 67 93 69 94 56 52" "256 256 3 1 20 32 69 94 67 93" "192 192 3 1 40 4 67 93 68 95" "288 288 3 1 20 4 68 95 67 93" "96 96 3 1 80 4 71 96 67 93" "128 128 3 1 80 32 69 94 31"; do
  timeout -k 10 120 python3 tools/sweep_run.py $spec >> $O/padl.txt 2>&1; echo "sweep rc=$?"
done
grep -v amdgpu.ids $O/padl.txt
