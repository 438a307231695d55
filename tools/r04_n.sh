#!/bin/bash
# K1 experiments: non-temporal loads / stores and a high-priority K1 stream, same box, alternating; PMC passes for the 4K and clip K1
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04n; rm -rf $O; mkdir -p $O
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$1'.split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','p99_latency_ms')}, 'K1 us', r['avg_launch_us'], 'frac', r['frac'], 'cold', r.get('cold_launch_us'))"; }
for rep in 1 2; do
  for cfg in "base" "RVA_K1_NT=1" "RVA_K1_NT=2" "RVA_K1_NT=3" "RVA_K1_PRIO=1"; do
    env $( [ "$cfg" = base ] || echo $cfg ) timeout -k 10 300 python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline --net-graph off > $O/k1_${cfg}_$rep.log 2>&1 || { echo FAIL $cfg; tail -5 $O/k1_${cfg}_$rep.log; }
    show $O/k1_${cfg}_$rep.log
  done
done
cd /tmp
for mode in 4k clip; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/k1_${mode}_$c -- python3 $ROOT/tools/k1_only.py 20 $mode > $O/k1_${mode}_$c.log 2>&1; echo "k1 $mode $c rc=$?"
  done
  python3 $ROOT/tools/pmc_summary.py $O/k1_${mode}_FETCH_SIZE $O/k1_${mode}_WRITE_SIZE > $O/k1_${mode}_pmc.txt; cat $O/k1_${mode}_pmc.txt
done
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete
