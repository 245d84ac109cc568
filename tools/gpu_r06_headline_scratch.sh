# is the headline kernel's scratch (248 B per lane at 168 VGPRs) on the heaviest env's critical chain, and what is its share of
# WRITE_SIZE?  The same source with launch bounds of 512 threads (-DSS_MAX_THREADS=512: 239 VGPRs, NO scratch, 8 envs per CU) against the shipped
# 768-thread build: lone waves (one env per CU: nothing but the env's own chain), then the full chip, then WRITE_SIZE / FETCH_SIZE of both.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in base t512; do
  if [ $v = base ]; then L=smplsim_amd/libsmplsim_hip.so; else L=smplsim_amd/variants/libsmplsim_hip_$v.so; fi
  echo "== $v"
  SMPLSIM_HIP_LIB=$L python tools/gpu_lone.py 2>/dev/null
  for seed in 1234 77; do
  SMPLSIM_HIP_LIB=$L python bench.py --steps 300 --warmup 20 --seed $seed --no-cpu-baseline --no-reference-contact-set 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v bench seed $seed: value', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['launch'], 'iters', round(d['config']['mean_newton_iters_per_step'],3))"
  done
  for c in WRITE_SIZE FETCH_SIZE; do
    OUT=/tmp/hs_$v_$c; rm -rf $OUT; mkdir -p $OUT
    SMPLSIM_HIP_LIB=$L rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_1 -o pmc -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-reference-contact-set > $OUT/log 2>&1
    python tools/prof_summarize.py $OUT $OUT/sum > /dev/null 2>&1
    python -c "
import json
d=json.load(open('$OUT/sum.json')); p=[v for k,v in d['pmc'].items() if k.startswith('step')][0]
print('$v $c (KiB per step launch, raw):', {k: round(v['mean_per_dispatch'],1) for k,v in p.items()})"
  done
done
