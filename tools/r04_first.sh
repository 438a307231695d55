#!/bin/bash
# Round-4 first evidence run: per-layer table of the batch-4 YOLOv8m plan (one GPU's share of BASELINE configs[3]), its bench line and kernel trace,
# and the headline line on the same box for reference.
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
O=$ROOT/gpurun_out/r04a; rm -rf $O; mkdir -p $O
RVA_TUNE_CACHE=0 python3 tools/show_tuning.py 4 m > $O/m4_conv_tuning.txt 2>&1; echo "m4 tuning rc=$?"
python3 bench.py --model m --streams 4 --steps 300 --warmup 30 --no-cpu-baseline --no-extras > $O/bench_m4.log 2>&1; echo "m4 bench rc=$?"; tail -c 600 $O/bench_m4.log
python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras > $O/bench_s32.log 2>&1; echo "s32 bench rc=$?"; tail -c 900 $O/bench_s32.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m4trace -- python3 $ROOT/bench.py --model m --streams 4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras > $O/m4trace.log 2>&1
echo "m4 trace rc=$?"
cd $ROOT
python3 tools/summarize_profile.py $O/m4trace $O/m4_kernel_stats_summary.csv "bench.py --model m --streams 4 --steps 100 --warmup 20, round 4" > /dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
du -sh $O
