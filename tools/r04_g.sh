#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04g; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_engine.py -x -q > $O/pytest_engine.log 2>&1; rc=$?; echo "pytest engine rc=$rc"; tail -5 $O/pytest_engine.log
[ $rc -eq 0 ] || exit 1
for spec in "128 128 3 1 40 32 52 80 56 81 58 82 84 85"; do
  RVA_NOSEL=1 RVA_LIB_PATH=$ROOT/tools/_dbg/librva_exp.so timeout -k 10 120 python3 tools/sweep_run.py $spec >> $O/nosel2.txt 2>&1; echo "sweep rc=$?"
done
grep -v amdgpu.ids $O/nosel2.txt
RVA_TUNE_CACHE=0 timeout -k 10 300 python3 tools/show_tuning.py 32 s > $O/s32_conv_tuning.txt 2>&1; echo "s32 tuning rc=$?"; tail -3 $O/s32_conv_tuning.txt
RVA_TUNE_CACHE=0 timeout -k 10 300 python3 tools/show_tuning.py 4 m > $O/m4_conv_tuning.txt 2>&1; echo "m4 tuning rc=$?"; tail -3 $O/m4_conv_tuning.txt
for m in "s 32" "m 4"; do set -- $m
timeout -k 10 300 python3 bench.py --model $1 --streams $2 --steps 400 --warmup 40 --no-cpu-baseline --no-extras > $O/bench_$1$2.log 2>&1; echo "bench $m rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$O/bench_$1$2.log').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','p99_latency_ms','detector_frac_of_mfma_peak','detector_frac_of_mfma_peak_in_pipeline')})"
done
