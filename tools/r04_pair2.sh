#!/bin/bash
# fused bottleneck, policy check: full GPU suite; depth-1 and paced A/B (RVA_PAIR32=1 / 0)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/pair2_gpu.log 2>&1 || { tail -n 40 gpurun_out/pair2_gpu.log; exit 1; }
tail -n 2 gpurun_out/pair2_gpu.log
rm -f gpurun_out/pair2_ab.log
for i in 1 2; do
  for which in 1 0; do
    export RVA_PAIR32=$which RVA_TUNE_CACHE_DIR=/tmp/rva_tune_p$which
    timeout -k 10 300 python bench.py --gpus 1 --steps 300 --warmup 40 --depth 1 --no-cpu-baseline --no-extras > gpurun_out/pair2_$which.json 2> gpurun_out/pair2.err || { tail -n 20 gpurun_out/pair2.err; exit 1; }
    python - $which <<'PY' | tee -a gpurun_out/pair2_ab.log
import json, sys
d = json.loads(open(f"gpurun_out/pair2_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("depth 1, RVA_PAIR32 =", sys.argv[1], d["value"], d["p99_latency_ms"], d["p50_latency_ms"], d["stages_ms"]["detector"])
PY
  done
done
