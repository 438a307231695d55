#!/bin/bash
# stride-2 long-run kernels: parity, then interleaved A/B against the gather kernels on the five downsampling layers of YOLOv8s x 32
# and three of YOLOv8m x 4
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "every_conv_variant" > gpurun_out/s2_test.log 2>&1 || { tail -n 40 gpurun_out/s2_test.log; exit 1; }
tail -n 2 gpurun_out/s2_test.log
for spec in "64 128 3 2 160 32 39 37 74 64 86 87 88 89 90 91 92 93" "128 256 3 2 80 32 64 37 74 39 86 87 88 89 90 91 92 93" "256 512 3 2 40 32 37 74 64 86 87 88 89 90 91 92 93" \
            "128 128 3 2 80 32 37 74 86 87 88 89 90 91 92 93" "256 256 3 2 40 32 74 37 86 87 88 89 90 91 92 93" \
            "96 192 3 2 160 4 1 41 86 87 88 89 90 91 92 93" "192 384 3 2 80 4 74 37 86 87 88 89 90 91 92 93" "384 576 3 2 40 4 78 77 86 87 88 89 90 91 92 93"; do
  timeout -k 10 120 python tools/sweep_run.py $spec 2>&1 | grep "k3s2" | tee -a gpurun_out/s2_sweep.log || exit 1
done
