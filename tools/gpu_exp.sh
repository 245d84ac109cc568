for mt in 512 640 704; do
  SS_HIPCC_OPT="-O3 -DSS_MAX_THREADS=$mt" python -c "from smplsim_amd import _lib; _lib.build(force=True)" 2>/dev/null
  echo "== SS_MAX_THREADS=$mt"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],3), d['config']['launch'])"
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload smplx 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('smplx ms/step',round(d['ms_per_step'],3), d['config']['launch'])"
