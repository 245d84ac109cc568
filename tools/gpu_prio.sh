for sh in 9 5 4 3; do
  SS_HIPCC_OPT="-O3 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -DSS_PRIO_SHIFT=$sh" python -c "from smplsim_amd import _lib; _lib.build(force=True)" 2>/dev/null
  echo "== SS_PRIO_SHIFT=$sh"
  for i in 1 2; do python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))"; done
done
