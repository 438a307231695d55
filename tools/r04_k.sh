#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/r04k; rm -rf $O; mkdir -p $O
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err; echo "driver-like bench rc=$? in $(( $(date +%s) - t0 )) s"; tail -3 $O/bench_driver.err
python3 - <<PY
import json
d=json.loads(open('$O/bench_driver.log').read().strip().splitlines()[-1])
keep=('value','ms_per_step','p99_latency_ms','p50_latency_ms','latency_samples','latency_source','host_submit_us_per_tick','network_launch','detector_frac_of_mfma_peak','detector_frac_of_mfma_peak_in_pipeline','long_run','paced_30fps','extras_error','kernel_selection')
print(json.dumps({k:d.get(k) for k in keep}, indent=1))
print('roofline', json.dumps({k:v for k,v in d['roofline'].items() if k not in ('timing','kernel')}, indent=1))
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
for m in "n 4" "m 4"; do set -- $m
timeout -k 10 300 python3 bench.py --model $1 --streams $2 --steps 400 --warmup 40 --no-cpu-baseline --no-extras > $O/bench_$1$2.log 2>&1; echo "bench $m rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$O/bench_$1$2.log').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','p99_latency_ms','host_submit_us_per_tick','network_launch')})"
done
