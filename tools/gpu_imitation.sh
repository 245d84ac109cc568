# motion-library / imitation row on the GPU: parity tests, the imitation bench line, and a regression check of the headline
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --workload imitation --steps ${STEPS:-300} --warmup 10 2>gpurun_out/bench_imitation.err > gpurun_out/bench_imitation.json; echo "imitation rc=$?"
tail -3 gpurun_out/bench_imitation.err
python - <<'P'
import json
d=json.load(open('gpurun_out/bench_imitation.json'))
print(round(d['value']),'env-steps/s',round(d['ms_per_step'],3),'ms/step', {k:d['config'][k] for k in ('mean_reward','episodes_ended','obs_finite','step_kernel_ms','load_motions_s (upload + cook)','cook')}, d['roofline']['kernel_ms'], d['roofline']['achieved'], d.get('cpu_baseline',{}).get('value'))
P
timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_quick.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('smpl value',round(d['value']),'ms/step',round(d['ms_per_step'],3),'kernel_ms',round(d['roofline']['kernel_ms'],3),d['config']['launch'])"
