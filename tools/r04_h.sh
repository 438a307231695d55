#!/bin/bash
# same-box A/B of the round-4 kernel families in the pipeline: all variants vs the round-3 set (RVA_SKIP_VARIANTS)
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04h; rm -rf $O; mkdir -p $O
OLD="67 68 69 70 71 72 73 74 75 76 77 78 79 80 81 82 83 84 85"
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1'.split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','p99_latency_ms','detector_frac_of_mfma_peak','detector_frac_of_mfma_peak_in_pipeline')}, d['stages_ms'])"; }
for rep in 1 2 3; do
  for m in "s 32" "m 4"; do set -- $m
    timeout -k 10 300 python3 bench.py --model $1 --streams $2 --steps 500 --warmup 40 --no-cpu-baseline --no-extras > $O/new_$1$2_$rep.log 2>&1 || { echo FAIL; tail -3 $O/new_$1$2_$rep.log; exit 1; }
    show $O/new_$1$2_$rep.log
    RVA_SKIP_VARIANTS="$OLD" timeout -k 10 300 python3 bench.py --model $1 --streams $2 --steps 500 --warmup 40 --no-cpu-baseline --no-extras > $O/old_$1$2_$rep.log 2>&1 || { echo FAIL; tail -3 $O/old_$1$2_$rep.log; exit 1; }
    show $O/old_$1$2_$rep.log
  done
done
