mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_quick.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value',round(d['value']),'ms/step',round(d['ms_per_step'],3),'kernel_ms',round(d['roofline']['kernel_ms'],3),d['config']['launch'],'iters',d['config']['mean_newton_iters_per_step'])"
