cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=/tmp/prof_raw; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out; cd $R
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_ic -o pmc -- $CMD > $OUT/l1.log 2>&1; echo rc=$?
rocprofv3 --pmc SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_ic2 -o pmc -- $CMD > $OUT/l2.log 2>&1; echo rc=$?
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_STALL SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_ic3 -o pmc -- $CMD > $OUT/l3.log 2>&1; echo rc=$?
python tools/prof_summarize.py $OUT $R/gpurun_out/r01e_icache_summary | grep -v "^ \+0\.\|calls="
