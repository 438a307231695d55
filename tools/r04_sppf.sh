#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "pool_upsample" > gpurun_out/sppf_test.log 2>&1 || { tail -n 30 gpurun_out/sppf_test.log; exit 1; }
tail -n 2 gpurun_out/sppf_test.log
echo "== new"; timeout -k 10 120 python tools/sppf_time.py 2>&1 | grep batch || exit 1
echo "== RVA_SPPF_G1=1"; RVA_SPPF_G1=1 timeout -k 10 120 python tools/sppf_time.py 2>&1 | grep batch || exit 1
