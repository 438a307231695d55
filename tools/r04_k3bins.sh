#!/bin/bash
# K3 centre-bin filter: parity, then A/B timing of the tail under load
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/k3b_parity.log 2>&1 || { tail -n 40 gpurun_out/k3b_parity.log; exit 1; }
tail -n 3 gpurun_out/k3b_parity.log
for D in 16 64 256; do
  timeout -k 10 120 python tools/load_tail.py $D 40 2>&1 | grep "D=" | sed 's/^/bins   /' | tee -a gpurun_out/k3b_ab.log || exit 1
  RVA_K3_NOBINS=1 timeout -k 10 120 python tools/load_tail.py $D 40 2>&1 | grep "D=" | sed 's/^/nobins /' | tee -a gpurun_out/k3b_ab.log || exit 1
done
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/k3b_pipe.log 2>&1 || { tail -n 40 gpurun_out/k3b_pipe.log; exit 1; }
tail -n 3 gpurun_out/k3b_pipe.log
