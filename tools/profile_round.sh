#!/bin/bash
# Round-3 evidence run (GPU box): kernel trace of the bench command, of the tail under load, PMC passes on K1 and on the conv kernels.
# rocprofv3 is given the program itself after `--` (python3 ...), counters in their own passes (no trace domains with --pmc).
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
R=${1:-r03}
OUT=$ROOT/gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
echo "bench trace rc=$?"; tail -c 400 $OUT/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tail -- python3 $ROOT/tools/load_tail.py 256 40 > $OUT/tail.log 2>&1
echo "tail trace rc=$?"; tail -n 2 $OUT/tail.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/k1_$c -- python3 $ROOT/tools/k1_only.py 20 content > $OUT/k1_$c.log 2>&1
  echo "k1 $c rc=$?"
done
i=0
for spec in "128 128 3 1 40 32 56" "128 128 3 1 40 32 52" "64 64 3 1 80 32 66" "128 192 3 1 80 32 31" "256 256 3 1 20 32 52" "384 256 1 1 40 32 37" "768 512 1 1 20 32 37"; do
  i=$((i+1))
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/conv${i}_a -- python3 $ROOT/tools/bench_one.py $spec > $OUT/conv${i}_a.log 2>&1
  echo "conv $spec pass a rc=$?"; tail -1 $OUT/conv${i}_a.log
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/conv${i}_b -- python3 $ROOT/tools/bench_one.py $spec > $OUT/conv${i}_b.log 2>&1
  echo "conv $spec pass b rc=$?"
done
cd $ROOT
python3 tools/summarize_profile.py $OUT/bench $OUT/bench_kernel_stats_summary.csv "bench.py --steps 100 --warmup 20, round 3" > /dev/null
python3 tools/summarize_profile.py $OUT/tail $OUT/tail_kernel_stats_summary.csv "tools/load_tail.py 256 40 (K2/K3/K4 alone at ~256 planted objects per frame x 32 streams), round 3" > /dev/null
python3 tools/tick_breakdown.py $OUT/bench 90 > $OUT/tick_breakdown.csv; head -45 $OUT/tick_breakdown.csv
python3 tools/pmc_summary.py $OUT/k1_FETCH_SIZE $OUT/k1_WRITE_SIZE > $OUT/k1_pmc_summary.txt
for j in 1 2 3 4 5 6 7; do echo "== conv$j: $(tail -1 $OUT/conv${j}_a.log)"; python3 tools/pmc_summary.py $OUT/conv${j}_a $OUT/conv${j}_b; done > $OUT/conv_pmc_summary.txt
# keep the merge small: drop raw traces, keep summaries + stats
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
