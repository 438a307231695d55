#!/bin/bash
# end-of-round rehearsal of what the driver runs: GPU tests, smoke(), the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/f3_gpu.log 2>&1 || { tail -n 40 gpurun_out/f3_gpu.log; exit 1; }
tail -n 3 gpurun_out/f3_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/f3_smoke.log 2>&1 || { tail -n 20 gpurun_out/f3_smoke.log; exit 1; }
tail -n 2 gpurun_out/f3_smoke.log
export TIMEFORMAT="bench wall %R s"
time (timeout -k 10 800 python bench.py > gpurun_out/f3_bench.json 2> gpurun_out/f3_bench.err) || { tail -n 20 gpurun_out/f3_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/f3_bench.json").read().strip().splitlines()[-1])
print(d["metric"], d["value"], d["unit"], d["steps"], d["warmup"], d["ms_per_step"], d["p99_latency_ms"])
print(json.dumps(d["roofline"])[:300])
print(json.dumps(d["cpu_baseline"])[:300])
print(json.dumps(d["other_configs"])[:1200])
print(d.get("extras_error"))
PY
