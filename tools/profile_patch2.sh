ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
OUT=$ROOT/gpurun_out/prof_p66
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $ROOT/tools/bench_one.py 64 64 3 1 80 32 66 > $OUT/a.log 2>&1; echo "a rc=$?"; tail -1 $OUT/a.log
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $ROOT/tools/bench_one.py 64 64 3 1 80 32 66 > $OUT/b.log 2>&1; echo "b rc=$?"
cd $ROOT
python3 tools/pmc_summary.py $OUT/a $OUT/b | tee $OUT/summary.txt
