#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04b; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py -x -q -k "every_conv_variant" > $O/pytest_conv.log 2>&1; rc=$?; echo "pytest conv rc=$rc"; tail -5 $O/pytest_conv.log
[ $rc -eq 0 ] || exit 1
RVA_TUNE_CACHE=0 timeout -k 10 300 python3 tools/show_tuning.py 4 m > $O/m4_conv_tuning.txt 2>&1; echo "m4 tuning rc=$?"; tail -3 $O/m4_conv_tuning.txt
RVA_TUNE_CACHE=0 timeout -k 10 300 python3 tools/show_tuning.py 32 s > $O/s32_conv_tuning.txt 2>&1; echo "s32 tuning rc=$?"; tail -3 $O/s32_conv_tuning.txt
timeout -k 10 300 python3 bench.py --model m --streams 4 --steps 300 --warmup 30 --no-cpu-baseline --no-extras > $O/bench_m4.log 2>&1; echo "m4 bench rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$O/bench_m4.log').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','p99_latency_ms')})"
timeout -k 10 300 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras > $O/bench_s32.log 2>&1; echo "s32 bench rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$O/bench_s32.log').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','p99_latency_ms')}, d['roofline']['frac'], d['roofline']['avg_launch_us'])"
