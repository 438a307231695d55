#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/r04t; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc=$rc"; tail -n 6 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err; echo "driver-like bench rc=$? in $(( $(date +%s) - t0 )) s"
python3 - <<PY
import json
d=json.loads(open('$O/bench_driver.log').read().strip().splitlines()[-1])
keep=('value','ms_per_step','p99_latency_ms','p50_latency_ms','latency_samples','host_submit_us_per_tick','network_launch','detector_frac_of_mfma_peak','detector_frac_of_mfma_peak_in_pipeline','detector_tflops','stages_ms','ticks_in_flight','long_run','paced_30fps','extras_error','post_tracker_load_sweep','load_sweep_end_to_end','roofline_4k','roofline_clip')
print(json.dumps({k:d.get(k) for k in keep}, indent=1))
print('roofline', json.dumps({k:v for k,v in d['roofline'].items() if k not in ('timing','kernel')}, indent=1))
print('cpu', json.dumps(d.get('cpu_baseline',{}))[:600])
PY
timeout -k 10 600 python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras > $O/bench_long.log 2>&1; python3 -c "
import json
d=json.loads(open('$O/bench_long.log').read().strip().splitlines()[-1]); print('1000 steps', {k:d.get(k) for k in ('value','ms_per_step','p99_latency_ms','p50_latency_ms','detector_frac_of_mfma_peak','detector_frac_of_mfma_peak_in_pipeline')})"
timeout -k 10 600 python3 bench.py --steps 500 --warmup 50 --depth 3 --no-cpu-baseline --no-extras > $O/bench_d3.log 2>&1; python3 -c "
import json
d=json.loads(open('$O/bench_d3.log').read().strip().splitlines()[-1]); print('depth 3', {k:d.get(k) for k in ('value','ms_per_step','p99_latency_ms','p50_latency_ms')})"
