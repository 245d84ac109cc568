# rocprofv3 passes for the round: kernel-trace stats, then PMC counters in separate runs (never combined with traces)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/prof_raw
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd $R
# WARMUP control steps are stepped but not counted (PROF_SKIP_STEPS of prof_summarize.py): the profiled launches are the workload as bench.py times it
export PROF_SKIP_STEPS=${WARMUP:-3}
CMD="python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-3} --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?"
for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" \
           "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "sq2:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_FLAT" \
           "sq3:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$?"
done
find $OUT -name "*.csv" | head -30
python tools/prof_summarize.py $OUT $R/gpurun_out/${TAG:-prof}_summary
tail -3 $OUT/trace.log
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG:-prof}_kernel_stats.csv 2>/dev/null
