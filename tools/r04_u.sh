#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/r04u; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_engine.py -x -q -k "fused_plan or c_plan" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
export RVA_TUNE_CACHE_DIR=/tmp/rva_tune
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1'.split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','p99_latency_ms')}, d['network_launch']['mode'], d['stages_ms']['detector'])"; }
for rep in 1 2; do
for m in "m 4" "n 4"; do set -- $m
  timeout -k 10 300 python3 bench.py --model $1 --streams $2 --steps 500 --warmup 40 --no-cpu-baseline --no-extras > $O/pad_$1$2_$rep.log 2>&1 || { echo FAIL; tail -3 $O/pad_$1$2_$rep.log; }
  show $O/pad_$1$2_$rep.log
  RVA_NO_CIN_PAD=1 timeout -k 10 300 python3 bench.py --model $1 --streams $2 --steps 500 --warmup 40 --no-cpu-baseline --no-extras > $O/nopad_$1$2_$rep.log 2>&1 || { echo FAIL; tail -3 $O/nopad_$1$2_$rep.log; }
  show $O/nopad_$1$2_$rep.log
done
done
RVA_TUNE_CACHE=0 python3 tools/show_tuning.py 4 m 2>&1 | grep -v amdgpu | grep "48->\|sum conv\|forward"
