# same-box A/B of prebuilt library variants on the self-collision workload
mkdir -p gpurun_out
for rep in 1 2; do
for v in ${VARIANTS}; do
  SMPLSIM_HIP_LIB=$PWD/smplsim_amd/variants/libsmplsim_hip_$v.so timeout -k 5 120 python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --self-collision 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep$rep value',round(d['value']),'ms/step',round(d['ms_per_step'],4))" | tee -a gpurun_out/variants_sc.log
done; done
