set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|ERROR|^E " gpurun_out/pytest_gpu.log | tail -15
for args in "" "--self-collision"; do
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench [$args] value',round(d['value']),'ms/step',round(d['ms_per_step'],4),'kernel_ms',round(d['roofline']['kernel_ms'],4),d['config']['launch'],'iters',round(d['config']['mean_newton_iters_per_step'],2),'selfc frac',d['config']['envs_with_body_body_contact_frac'],'mean',d['config']['mean_body_body_contacts'],'resets',d['config']['bad_state_resets_total'])"
done
