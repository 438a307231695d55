#!/bin/bash
# fused 32-channel bottleneck: parity, then forward alone and pipeline A/B against RVA_NO_PAIR32=1
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "bottleneck_pair" > gpurun_out/pair_test.log 2>&1 || { tail -n 40 gpurun_out/pair_test.log; exit 1; }
tail -n 2 gpurun_out/pair_test.log
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu > gpurun_out/pair_engine.log 2>&1 || { tail -n 40 gpurun_out/pair_engine.log; exit 1; }
tail -n 2 gpurun_out/pair_engine.log
rm -f gpurun_out/pair_ab.log
for i in 1 2 3; do
  for which in pair nopair; do
    if [ $which = nopair ]; then export RVA_NO_PAIR32=1 RVA_TUNE_CACHE_DIR=/tmp/rva_tune_b; else unset RVA_NO_PAIR32; export RVA_TUNE_CACHE_DIR=/tmp/rva_tune_a; fi
    timeout -k 10 300 python bench.py --gpus 1 --steps 300 --warmup 40 --no-cpu-baseline --no-extras > gpurun_out/pair_ab_$which.json 2> gpurun_out/pair_ab.err || { tail -n 20 gpurun_out/pair_ab.err; exit 1; }
    python - $which <<'PY' | tee -a gpurun_out/pair_ab.log
import json, sys
d = json.loads(open(f"gpurun_out/pair_ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["p99_latency_ms"], d["stages_ms"]["detector"])
PY
  done
done
