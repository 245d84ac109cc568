# rocprofv3 passes for the motion-library / imitation kernels: kernel-trace stats, then FETCH_SIZE / WRITE_SIZE in their own runs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/prof_motion
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd $R
CMD="python bench.py --workload imitation --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $CMD > $OUT/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
python - <<'P'
import csv, glob, json, os, collections
out = {}
for f in glob.glob('/tmp/prof_motion/trace/**/*kernel_trace.csv', recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for r in csv.DictReader(open(f)):
        d = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
        a = agg[r['Kernel_Name']]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    out['kernel_trace'] = [{'kernel': k[:110], 'calls': a[0], 'total_ms': a[1] / 1e6, 'avg_us': a[1] / a[0] / 1e3, 'min_us': a[2] / 1e3, 'max_us': a[3] / 1e3,
                            'pct': 100 * a[1] / tot} for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])][:25]
pm = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'/tmp/prof_motion/pmc_{c}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if 'ss_motion' in r['Kernel_Name'] or 'ss_imitation' in r['Kernel_Name']:
                a = agg[r['Kernel_Name'][:60]]; a[0] += 1; a[1] += float(r['Counter_Value'])
        for k, a in agg.items():
            pm[k][c + '_kib_raw_mean'] = a[1] / a[0]; pm[k]['dispatches'] = a[0]
for k, v in pm.items():   # MI355X_MICROARCH.md HBM section: FETCH_SIZE on gfx950 reports half of the bytes -> double it; both in KiB
    if 'FETCH_SIZE_kib_raw_mean' in v and 'WRITE_SIZE_kib_raw_mean' in v:
        v['hbm_bytes_per_launch'] = (2 * v['FETCH_SIZE_kib_raw_mean'] + v['WRITE_SIZE_kib_raw_mean']) * 1024
out['pmc'] = pm
R = os.environ['GRAFT_REPO_ROOT']
json.dump(out, open(f'{R}/gpurun_out/motion_prof_summary.json', 'w'), indent=1)
for k in out.get('kernel_trace', []): print(k)
print(json.dumps(pm, indent=1))
P
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/motion_kernel_stats.csv 2>/dev/null
tail -2 $OUT/trace.log
