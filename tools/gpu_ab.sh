# A/B of the launch bound (waves per workgroup / VGPR cap): time + HBM traffic counters per setting
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
for mt in ${MTS:-512 704}; do
  SS_HIPCC_OPT="-O3 -DSS_MAX_THREADS=$mt" python -c "from smplsim_amd import _lib; _lib.build(force=True)" 2>/dev/null
  echo "== SS_MAX_THREADS=$mt"
  for i in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],3), 'kernel_ms', d['roofline'].get('kernel_ms'), d['config']['launch'])"
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/ab_$c
    rocprofv3 --pmc $c --output-format csv -d /tmp/ab_$c -o pmc -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/ab.log 2>&1
    python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/ab_$c/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'ss_env_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    v.sort(); print(k,'n',len(v),'median KiB',v[len(v)//2],'max',v[-1])
PY
  done
done
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
