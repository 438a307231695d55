#!/bin/bash
# Round-4 evidence run (GPU box): kernel traces of the bench command (default = four tick chains, and --depth 1), of the 4 x YOLOv8m
# share of BASELINE configs[3], of the tail under load; PMC passes on K1 and on the final convolution kernels; per-layer tuning tables.
# rocprofv3 is given the program itself after `--` (python3 ...), counters in their own passes (no trace domains with --pmc).
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT RVA_TUNE_CACHE_DIR=/tmp/rva_tune
R=${1:-r04}
OUT=$ROOT/gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 30 --warmup 30 --no-cpu-baseline --no-extras --net-graph off > $OUT/prime_s32.log 2>&1        # fills the kernel-selection cache (four chains)
python3 bench.py --steps 30 --warmup 30 --no-cpu-baseline --no-extras --net-graph off --depth 1 > $OUT/prime_s32_d1.log 2>&1
python3 bench.py --model m --streams 4 --steps 30 --warmup 30 --no-cpu-baseline --no-extras --net-graph off > $OUT/prime_m4.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --net-graph off > $OUT/bench.log 2>&1
echo "bench trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_d1 -- python3 $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --net-graph off --depth 1 > $OUT/bench_d1.log 2>&1
echo "bench depth-1 trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/m4 -- python3 $ROOT/bench.py --model m --streams 4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --net-graph off > $OUT/m4.log 2>&1
echo "m4 trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tail -- python3 $ROOT/tools/load_tail.py 256 40 > $OUT/tail.log 2>&1
echo "tail trace rc=$?"; tail -n 2 $OUT/tail.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/k1_$c -- python3 $ROOT/tools/k1_only.py 20 content > $OUT/k1_$c.log 2>&1
  echo "k1 $c rc=$?"
done
i=0
for spec in "128 128 3 1 40 32 82" "128 128 3 1 40 32 56" "128 128 3 1 40 32 81" "128 192 3 1 80 32 80" "256 256 3 1 20 32 69" "64 64 3 1 80 32 66" "768 512 1 1 20 32 37" "192 192 3 1 40 4 67"; do
  i=$((i+1))
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/conv${i}_a -- python3 $ROOT/tools/bench_one.py $spec > $OUT/conv${i}_a.log 2>&1
  echo "conv $spec pass a rc=$?"; tail -1 $OUT/conv${i}_a.log
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/conv${i}_b -- python3 $ROOT/tools/bench_one.py $spec > $OUT/conv${i}_b.log 2>&1
  echo "conv $spec pass b rc=$?"
done
cd $ROOT
python3 tools/summarize_profile.py $OUT/bench $OUT/bench_kernel_stats_summary.csv "bench.py --steps 100 --warmup 20 --net-graph off (four tick chains), round 4" > /dev/null
python3 tools/summarize_profile.py $OUT/bench_d1 $OUT/bench_depth1_kernel_stats_summary.csv "bench.py --steps 100 --warmup 20 --net-graph off --depth 1, round 4" > /dev/null
python3 tools/summarize_profile.py $OUT/m4 $OUT/m4_kernel_stats_summary.csv "bench.py --model m --streams 4 --steps 100 --warmup 20 --net-graph off (one GPU's share of BASELINE configs[3]), round 4" > /dev/null
python3 tools/summarize_profile.py $OUT/tail $OUT/tail_kernel_stats_summary.csv "tools/load_tail.py 256 40 (K2/K3/K4 alone at ~256 planted objects per frame x 32 streams), round 4" > /dev/null
python3 tools/tick_breakdown.py $OUT/bench 90 > $OUT/tick_breakdown.csv; head -12 $OUT/tick_breakdown.csv
python3 tools/tick_breakdown.py $OUT/bench_d1 90 > $OUT/tick_breakdown_d1.csv; head -4 $OUT/tick_breakdown_d1.csv
python3 tools/tick_breakdown.py $OUT/m4 90 > $OUT/tick_breakdown_m4.csv; head -4 $OUT/tick_breakdown_m4.csv
python3 tools/pmc_summary.py $OUT/k1_FETCH_SIZE $OUT/k1_WRITE_SIZE > $OUT/k1_pmc_summary.txt; cat $OUT/k1_pmc_summary.txt
for j in 1 2 3 4 5 6 7 8; do echo "== conv$j: $(tail -1 $OUT/conv${j}_a.log)"; python3 tools/pmc_summary.py $OUT/conv${j}_a $OUT/conv${j}_b; done > $OUT/conv_pmc_summary.txt
RVA_TUNE_CACHE=0 python3 tools/show_tuning.py 32 s > $OUT/s32_conv_tuning.txt 2>&1
RVA_TUNE_CACHE=0 RVA_TUNE_LAYER_OVERLAP=4 python3 tools/show_tuning.py 32 s > $OUT/s32_conv_tuning_overlap4.txt 2>&1
RVA_TUNE_CACHE=0 python3 tools/show_tuning.py 4 m > $OUT/m4_conv_tuning.txt 2>&1
RVA_TUNE_CACHE=0 RVA_TUNE_LAYER_OVERLAP=4 python3 tools/show_tuning.py 4 m > $OUT/m4_conv_tuning_overlap4.txt 2>&1
tail -n 3 $OUT/s32_conv_tuning.txt; tail -n 3 $OUT/m4_conv_tuning.txt
# keep the merge small: drop raw traces, keep summaries + stats
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
