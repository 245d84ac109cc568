cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('%-14s' % '$1', 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), c['launch'])"; }
for rep in 1 2; do
for e in 12 11 10 9; do
  SS_ENVS_PER_WG=$e python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-reference-contact-set 2>/dev/null | show "epw$e"
done
done
