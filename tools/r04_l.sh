#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/r04l; rm -rf $O; mkdir -p $O
timeout -k 10 1150 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc=$rc"; tail -12 $O/pytest_gpu.log
ls $ROOT/gpurun_out/fp16_error_*.json 2>/dev/null
