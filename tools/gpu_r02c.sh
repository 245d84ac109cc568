# round-2 GPU validation C: tests, headline, imitation fused vs unfused, launch count of one imitation step (rocprofv3 kernel trace)
set -x
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|ERROR|^E " gpurun_out/pytest_gpu.log | tail -15
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_smpl.json; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_smpl.json')); print('headline', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'])
PY
for u in "" "--unfused"; do
timeout 600 python bench.py --workload imitation --steps 200 --warmup 20 --no-cpu-baseline $u 2>/dev/null > gpurun_out/bench_imitation$u.json; python - <<PY
import json; d=json.load(open('gpurun_out/bench_imitation$u.json')); print('imitation $u', round(d['value']), d['ms_per_step'], d['config']['env_step_ms (events around env.step)'], d['config']['episodes_ended'], d['config']['mean_reward'])
PY
done
cd /tmp && export TMPDIR=/tmp
for u in "" "--unfused"; do
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/imtrace$u -o im -- python $GRAFT_REPO_ROOT/bench.py --workload imitation --steps 100 --warmup 10 --no-cpu-baseline $u > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/imtrace$u -name "*kernel_stats.csv" | head -1)
echo "== kernel stats $u"; head -25 $f | cut -c1-150
done
