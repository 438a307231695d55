#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04q; rm -rf $O; mkdir -p $O
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1'.split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','p99_latency_ms','ticks_in_flight')}, d['network_launch']['mode'], d['host_submit_us_per_tick']['mean'])"; }
for rep in 1 2; do
for d in 3 4 5 2; do
  timeout -k 10 300 python3 bench.py --model m --streams 4 --depth $d --steps 500 --warmup 40 --no-cpu-baseline --no-extras > $O/m4_d${d}_$rep.log 2>&1 || { echo FAIL; tail -3 $O/m4_d${d}_$rep.log; }
  show $O/m4_d${d}_$rep.log
done
done
timeout -k 10 300 python3 bench.py --model m --streams 32 --steps 200 --warmup 30 --no-cpu-baseline --no-extras > $O/m32.log 2>&1; show $O/m32.log
