#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04p; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py -x -q -k "every_conv_variant" > $O/pytest_conv.log 2>&1; rc=$?; echo "pytest conv rc=$rc"; tail -5 $O/pytest_conv.log
[ $rc -eq 0 ] || exit 1
for spec in "256 512 3 2 40 32 37 86 39 88" "128 256 3 2 80 32 64 89 39 88 37 86" "64 128 3 2 160 32 39 88 38 87" "256 256 3 2 40 32 37 86 39 88" "128 128 3 2 80 32 37 86 39 88" "128 128 3 1 40 32 37 86 82"; do
  timeout -k 10 120 python3 tools/sweep_run.py $spec >> $O/zl.txt 2>&1; echo "sweep rc=$?"
done
grep -v amdgpu.ids $O/zl.txt
