#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04o; rm -rf $O; mkdir -p $O
python3 bench.py --steps 50 --warmup 30 --no-cpu-baseline --no-extras --net-graph off > $O/prime.log 2>&1   # fills the tuning cache
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --net-graph off > $O/bench.log 2>&1
echo "bench trace rc=$?"; tail -c 300 $O/bench.log
cd $ROOT
python3 tools/summarize_profile.py $O/bench $O/bench_kernel_stats_summary.csv "bench.py --steps 100 --warmup 20 --net-graph off, round 4 (mid-round)" > /dev/null
python3 tools/tick_breakdown.py $O/bench 90 > $O/tick_breakdown.csv; head -60 $O/tick_breakdown.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
