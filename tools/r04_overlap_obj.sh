#!/bin/bash
# kernel-selection objective under four chains: candidates timed n launches side by side, n = 2 / 3 / 4 (default = the depth, 4)
set -o pipefail
mkdir -p gpurun_out; rm -f gpurun_out/ovl.log
for rep in 1 2; do
for n in 4 3 2; do
  export RVA_TUNE_LAYER_OVERLAP=$n RVA_TUNE_CACHE_DIR=/tmp/rva_tune_ovl$n
  timeout -k 10 300 python bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --no-extras > gpurun_out/ovl_$n.json 2> gpurun_out/ovl.err || { tail -n 20 gpurun_out/ovl.err; exit 1; }
  python - $n <<'PY' | tee -a gpurun_out/ovl.log
import json, sys
d = json.loads(open(f"gpurun_out/ovl_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("objective", sys.argv[1], d["value"], d["p99_latency_ms"], d["stages_ms"]["detector"])
PY
done
done
