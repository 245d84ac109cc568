# A/B of hipcc scheduler flags on the bench workload
for opt in ${OPTS}; do
  SS_HIPCC_OPT="${opt//,/ }" python -c "from smplsim_amd import _lib; _lib.build(force=True)" 2>/dev/null
  echo "== $opt"
  for w in ${WORKLOADS:-smpl smpl}; do SS_HIPCC_OPT="${opt//,/ }" python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w ms/step',round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))"; done
done
