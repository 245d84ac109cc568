mkdir -p gpurun_out
cat /sys/fs/cgroup/cpu.max; nproc
for w in getup smplx; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload $w 2>gpurun_out/bench_$w.err > gpurun_out/bench_$w.json; echo "$w rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/bench_$w.json')); print('$w', round(d['value']), 'env-steps/s', round(d['ms_per_step'],2),'ms/step', d['config']['launch'], d['config']['newton_iters_p50_p99_max'], d['config']['obs_finite'])"
done
timeout 900 python bench.py > gpurun_out/bench_full.json 2>gpurun_out/bench_full.err; echo "full rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_full.json')); print(round(d['value']), d['ms_per_step'], d['roofline'], d['cpu_baseline'])"
