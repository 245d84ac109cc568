# effective shader clock of the body-body-contact step launch: GRBM_GUI_ACTIVE / 8 XCDs / launch duration, for the libraries in LIBS
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for L in ${LIBS:-smplsim_amd/libsmplsim_hip.so smplsim_amd/variants/libsmplsim_hip_r05.so}; do
  OUT=/tmp/clk_$$; rm -rf $OUT; mkdir -p $OUT
  CMD="python bench.py --self-collision --steps 12 --warmup 3 --no-cpu-baseline --no-reference-contact-set ${BENCH_ARGS:-}"
  SMPLSIM_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
  SMPLSIM_HIP_LIB=$L rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_1 -o pmc -- $CMD > $OUT/pmc.log 2>&1
  python tools/prof_summarize.py $OUT $OUT/sum > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open("$OUT/sum.json"))
st=d["step_launches"]; p=[v for k,v in d["pmc"].items() if k.startswith("step")][0]
g=p["GRBM_GUI_ACTIVE"]["mean_per_dispatch"]/8
print("$L", "launch us", round(st["avg_us"],1), "cycles", round(g), "clock GHz", round(g/st["avg_us"]/1e3,3), {k:round(v["mean_per_dispatch"]/1e6,1) for k,v in p.items()})
PY
done
