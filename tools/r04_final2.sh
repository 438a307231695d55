#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/f2_gpu.log 2>&1 || { tail -n 40 gpurun_out/f2_gpu.log; exit 1; }
tail -n 3 gpurun_out/f2_gpu.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/f2_bench.json 2> gpurun_out/f2_bench.err || { tail -n 20 gpurun_out/f2_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/f2_bench.json").read().strip().splitlines()[-1])
print("s32", d["value"], d["p99_latency_ms"], d["stages_ms"], d["detector_frac_of_mfma_peak"], d["detector_frac_of_mfma_peak_in_pipeline"], d["long_run"]["frames_per_s"], d["paced_30fps"]["chains_1"]["p99_ms"])
PY
timeout -k 10 600 python bench.py --model m --streams 4 --steps 200 --warmup 40 --no-cpu-baseline --no-extras > gpurun_out/f2_bench_m4.json 2> gpurun_out/f2_bench_m4.err || { tail -n 20 gpurun_out/f2_bench_m4.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/f2_bench_m4.json").read().strip().splitlines()[-1])
print("m4", d["value"], d["p99_latency_ms"], d["stages_ms"], d.get("network_launch"))
PY
timeout -k 10 600 python bench.py --model m --streams 4 --steps 200 --warmup 40 --no-cpu-baseline --no-extras > gpurun_out/f2_bench_m4b.json 2> gpurun_out/f2_bench_m4b.err || exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/f2_bench_m4b.json").read().strip().splitlines()[-1])
print("m4 again", d["value"], d["p99_latency_ms"])
PY
