#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/r04m; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_engine.py -x -q -k "fused_plan" -s > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v "fp16 error report" $O/pytest.log | tail -12
