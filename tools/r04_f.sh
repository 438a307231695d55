#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04f; rm -rf $O; mkdir -p $O
export RVA_LIB_PATH=$ROOT/tools/_dbg/librva_exp.so
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py -x -q -k "every_conv_variant" > $O/pytest_conv_exp.log 2>&1; rc=$?; echo "pytest conv (experiment build, variants 92-99 included) rc=$rc"; tail -5 $O/pytest_conv_exp.log
[ $rc -eq 0 ] || exit 1
for spec in "128 128 3 1 40 32 52 92 91 56 97 90 58 99" "256 256 3 1 20 32 52 92 91 69" "128 128 3 1 80 32 52 92 91 31" "128 192 3 1 80 32 52 92 31" "256 192 3 1 40 32 60 98 52 92" "64 64 3 1 80 32 52 92 66"; do
  RVA_NOSEL=1 timeout -k 10 120 python3 tools/sweep_run.py $spec >> $O/pado.txt 2>&1; echo "sweep rc=$?"
done
grep -v amdgpu.ids $O/pado.txt
