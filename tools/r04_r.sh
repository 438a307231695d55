#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04r; rm -rf $O; mkdir -p $O
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1'.split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','p99_latency_ms','ticks_in_flight')}, d['network_launch']['mode'], d.get('chain_stream_probe'))"; }
for rep in 1 2 3; do
for cfg in "m 4 3" "m 4 4" "s 32 3" "s 32 4"; do set -- $cfg
  timeout -k 10 300 python3 bench.py --model $1 --streams $2 --depth $3 --steps 500 --warmup 40 --no-cpu-baseline --no-extras > $O/$1$2_d$3_$rep.log 2>&1 || { echo FAIL; tail -3 $O/$1$2_d$3_$rep.log; }
  show $O/$1$2_d$3_$rep.log
done
done
