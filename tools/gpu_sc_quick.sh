# self-collision quick check: the two GPU self-collision tests + the bench line with body-body contacts on
set -x
mkdir -p gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "self_col or selfcol or native_mjcf or fused_imitation" 2>&1 | tail -5
timeout -k 5 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --self-collision 2>/dev/null > gpurun_out/bench_selfcol.json < /dev/null
python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_selfcol.json')); print('selfcol', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['mean_newton_iters_per_step'], d['config']['envs_with_body_body_contact_frac'])
PY
