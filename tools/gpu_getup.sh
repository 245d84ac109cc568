mkdir -p gpurun_out
T="timeout -k 5"
$T 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "getup or Getup or fall or Fall or autoreset" > gpurun_out/getup_test.log 2>&1 < /dev/null; tail -2 gpurun_out/getup_test.log
for w in getup smpl; do
$T 300 python bench.py --workload $w --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null < /dev/null > gpurun_out/bench_$w.json
python - <<PY
import json; d=json.load(open('gpurun_out/bench_$w.json')); print('$w', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'])
PY
done
