# Round 6, VERDICT r5 item 1 step A: which shared resource bounds the body-body-contact step kernel?  Counter passes of
# `bench.py --self-collision` at 8 and 6 resident envs per CU (vector memory = scratch traffic, instruction cache, L1 / L2 behaviour, wait split),
# then lone-wave runs.  PMC passes are their own runs (never combined with traces).  EPWS="8 6"  PASSES="..." to narrow.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
LIBARG=${SMPLSIM_HIP_LIB:+SMPLSIM_HIP_LIB=$SMPLSIM_HIP_LIB}
CMD="python bench.py --self-collision --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-reference-contact-set ${BENCH_ARGS:-}"
TAG=${TAG:-r06_selfcol_bound}
for e in ${EPWS:-8 6}; do
  OUT=/tmp/prof_raw_$e; rm -rf $OUT; mkdir -p $OUT
  export SS_ENVS_PER_WG=$e
  $CMD 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('epw $e bench: value', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['launch'])"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?"
  i=0
  for ctrs in \
    "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
    "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVES" \
    "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_TC_STALL" \
    "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum" \
    "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TOTAL_WRITE_sum" \
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
    "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    case " ${PASSES:-1 2 3 4 5 6 7 8 9 10} " in *" $i "*) ;; *) continue;; esac
    rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$i -o pmc -- $CMD > $OUT/pmc_$i.log 2>&1; echo "epw $e pmc $i rc=$? ($ctrs)"
    [ -n "$(find $OUT/pmc_$i -name '*counter_collection.csv' | head -1)" ] || tail -5 $OUT/pmc_$i.log
  done
  unset SS_ENVS_PER_WG
  python tools/prof_summarize.py $OUT $R/gpurun_out/${TAG}_epw$e > /dev/null 2>&1
  grep -A80 "PMC step" $R/gpurun_out/${TAG}_epw$e.txt | head -90
done
