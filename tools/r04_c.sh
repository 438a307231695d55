#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04c; rm -rf $O; mkdir -p $O
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1', {k:d.get(k) for k in ('value','ms_per_step','p99_latency_ms','host_submit_us_per_tick','hip_graph_scope')})"; }
for rep in 1 2; do
for m in "m 4" "s 32" "n 4"; do
  set -- $m
  timeout -k 10 300 python3 bench.py --model $1 --streams $2 --steps 400 --warmup 40 --no-cpu-baseline --no-extras > $O/b_$1$2_eager_$rep.log 2>&1 || { echo FAIL eager $m; tail -5 $O/b_$1$2_eager_$rep.log; exit 1; }
  show $O/b_$1$2_eager_$rep.log
  timeout -k 10 300 python3 bench.py --model $1 --streams $2 --steps 400 --warmup 40 --no-cpu-baseline --no-extras --net-graph > $O/b_$1$2_graph_$rep.log 2>&1 || { echo FAIL graph $m; tail -5 $O/b_$1$2_graph_$rep.log; exit 1; }
  show $O/b_$1$2_graph_$rep.log
done
done
