#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/k3f_gpu.log 2>&1 || { tail -n 40 gpurun_out/k3f_gpu.log; exit 1; }
tail -n 3 gpurun_out/k3f_gpu.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/k3f_bench.json 2> gpurun_out/k3f_bench.err || { tail -n 20 gpurun_out/k3f_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/k3f_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["p99_latency_ms"], d["stages_ms"], d["post_tracker_load_sweep"], {k: (v["frames_per_s"], v["vs_light_load"]) for k, v in d["load_sweep_end_to_end"].items() if isinstance(v, dict)})
PY
