for tol in 4e-7f 3e-6f 3e-5f; do
  SS_HIPCC_OPT="-O3 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -DSS_MOVE_TOL=$tol" python -c "from smplsim_amd import _lib; _lib.build(force=True)" 2>/dev/null
  echo "== SS_MOVE_TOL=$tol"
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), d['config']['mean_newton_iters_per_step'], d['config']['newton_iters_p50_p99_max'])"
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1
done
