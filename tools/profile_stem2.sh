#!/bin/bash
# PMC passes on the fused stem kernel (counters in their own passes, no trace domains with --pmc)
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
OUT=$ROOT/gpurun_out/prof_stem2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $ROOT/tools/bench_stem2.py fused > $OUT/a.log 2>&1; echo "a rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $ROOT/tools/bench_stem2.py fused > $OUT/b.log 2>&1; echo "b rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c -- python3 $ROOT/tools/bench_stem2.py fused > $OUT/c.log 2>&1; echo "c rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/d -- python3 $ROOT/tools/bench_stem2.py fused > $OUT/d.log 2>&1; echo "d rc=$?"
cd $ROOT
python3 tools/pmc_summary.py $OUT/a $OUT/b $OUT/c $OUT/d | tee $OUT/summary.txt
