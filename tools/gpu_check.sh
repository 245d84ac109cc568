# what a gpurun call executes to validate the tree: smoke, the GPU tests, the default bench line (as the driver runs it)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; python -c "
import json
d=json.load(open('gpurun_out/bench_default.json')); c=d['config']
print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'), d['roofline'].get('pmc_source'))
print('reference_contact_set', {k: c['reference_contact_set'][k] for k in ('value','ms_per_step','kernel_ms','frac','envs_per_cu','truncated_mj_step_frac')})
print('parity_probe', c['parity_probe']); print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
