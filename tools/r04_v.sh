#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/r04v; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py -x -q -k "jpeg or preview" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 25 $O/pytest.log
