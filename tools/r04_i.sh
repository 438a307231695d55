#!/bin/bash
# same-box A/B of the kernel selection objective: launches timed alone (1) vs three at a time on the chain streams (3)
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04i; rm -rf $O; mkdir -p $O
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1'.split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','p99_latency_ms','detector_frac_of_mfma_peak','detector_frac_of_mfma_peak_in_pipeline')}, d['stages_ms'])"; }
for rep in 1 2; do
  for m in "s 32" "m 4"; do set -- $m
    for ov in 3 1 2; do
    RVA_TUNE_CACHE=0 RVA_TUNE_LAYER_OVERLAP=$ov timeout -k 10 300 python3 bench.py --model $1 --streams $2 --steps 500 --warmup 40 --no-cpu-baseline --no-extras > $O/ov${ov}_$1$2_$rep.log 2>&1 || { echo FAIL; tail -5 $O/ov${ov}_$1$2_$rep.log; exit 1; }
    show $O/ov${ov}_$1$2_$rep.log
    done
  done
done
RVA_TUNE_CACHE=0 RVA_TUNE_LAYER_OVERLAP=3 timeout -k 10 300 python3 tools/show_tuning.py 32 s > $O/s32_conv_tuning_ov3.txt 2>&1; head -30 $O/s32_conv_tuning_ov3.txt
