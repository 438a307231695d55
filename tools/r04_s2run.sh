#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export RVA_TUNE_CACHE_DIR=/tmp/rva_tune
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "every_conv_variant" > gpurun_out/s2_test.log 2>&1 || { tail -n 40 gpurun_out/s2_test.log; exit 1; }
tail -n 2 gpurun_out/s2_test.log
RVA_TUNE_CACHE=0 timeout -k 10 500 python tools/show_tuning.py 32 s > gpurun_out/s2_tuning_s32.txt 2>&1 || { tail -n 20 gpurun_out/s2_tuning_s32.txt; exit 1; }
grep "k3s2\|sum conv\|forward ms" gpurun_out/s2_tuning_s32.txt
RVA_TUNE_CACHE=0 timeout -k 10 500 python tools/show_tuning.py 4 m > gpurun_out/s2_tuning_m4.txt 2>&1 || { tail -n 20 gpurun_out/s2_tuning_m4.txt; exit 1; }
grep "k3s2\|sum conv\|forward ms" gpurun_out/s2_tuning_m4.txt
