# body-body contacts: step-launch time of library variants early / late in the rollout and in short / sustained runs (phase of the rollout vs
# what a sustained run does to the clock).  VARIANTS="base c2 r05"
cd $GRAFT_REPO_ROOT
for cfg in "3 12" "60 12" "60 200"; do set -- $cfg
for v in ${VARIANTS:-base r05}; do
  if [ $v = base ]; then L=smplsim_amd/libsmplsim_hip.so; else L=smplsim_amd/variants/libsmplsim_hip_$v.so; fi
  SMPLSIM_HIP_LIB=$L python bench.py --self-collision --steps $2 --warmup $1 --no-cpu-baseline --no-reference-contact-set 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('warmup $1 steps $2 %-6s' % '$v', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'value', round(d['value']), 'iters', round(d['config']['mean_newton_iters_per_step'],2))"
done; done
