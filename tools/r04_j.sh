#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04j; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 tools/_dbg/plan_identity.py > $O/identity.log 2>&1; rc=$?; echo "identity rc=$rc"; grep -v amdgpu.ids $O/identity.log | tail -8
[ $rc -eq 0 ] || exit 1
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc=$rc"; tail -8 $O/pytest_gpu.log
