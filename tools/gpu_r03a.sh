# r03a: MuJoCo termination rule (cap 100) against the r02 library (cap 8, "moving" rule), same box; then the GPU tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('$1', 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'iters mean', round(c['mean_newton_iters_per_step'],2), c.get('newton_iters_p50_p99_max'), 'resets', c['bad_state_resets_total'])"; }
for rep in 1 2; do
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | show "new(cap100)"
  SMPLSIM_HIP_LIB=smplsim_amd/variants/libsmplsim_hip_r02.so python bench.py --steps 300 --warmup 20 --no-cpu-baseline --newton-iters 8 2>/dev/null | show "r02(cap8)"
done
SMPLSIM_HIP_LIB=smplsim_amd/variants/libsmplsim_hip_r02.so python bench.py --steps 300 --warmup 20 --no-cpu-baseline --newton-iters 100 2>/dev/null | show "r02(cap100)"
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --newton-iters 8 2>/dev/null | show "new(cap8)"
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --workload getup 2>/dev/null | show "new getup"
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --workload smplx 2>/dev/null | show "new smplx"
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --self-collision 2>/dev/null | show "new selfcol"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -25
