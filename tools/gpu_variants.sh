# same-box A/B of prebuilt library variants (tools/build_variant.sh): bench kernel time per variant, interleaved repeats
mkdir -p gpurun_out
for rep in 1 2; do
for v in ${VARIANTS}; do
  SMPLSIM_HIP_LIB=$PWD/smplsim_amd/variants/libsmplsim_hip_$v.so python bench.py --workload ${WORKLOAD:-smpl} --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep$rep value',round(d['value']),'ms/step',round(d['ms_per_step'],4),'kernel_ms',round(d['roofline']['kernel_ms'],4),d['config']['launch'],'iters',round(d['config']['mean_newton_iters_per_step'],2))" | tee -a gpurun_out/variants.log
done; done
