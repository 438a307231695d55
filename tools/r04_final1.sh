#!/bin/bash
# snapshot writer + K7 timing + full GPU suite + driver-like bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "snapshot or skipped_and_missing or preview or jpeg" > gpurun_out/f1_snap.log 2>&1 || { tail -n 30 gpurun_out/f1_snap.log; exit 1; }
tail -n 3 gpurun_out/f1_snap.log
timeout -k 10 200 python tools/k7_time.py --out gpurun_out/k7_time.json > gpurun_out/f1_k7.log 2>&1 || { tail -n 20 gpurun_out/f1_k7.log; exit 1; }
cat gpurun_out/f1_k7.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/f1_gpu.log 2>&1 || { tail -n 40 gpurun_out/f1_gpu.log; exit 1; }
tail -n 3 gpurun_out/f1_gpu.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/f1_bench.json 2> gpurun_out/f1_bench.err || { tail -n 20 gpurun_out/f1_bench.err; exit 1; }
cat gpurun_out/f1_bench.json
