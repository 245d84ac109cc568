# kernel trace of the serial sampler (one 4096-env batch, bf16 MFMA policy): where the 1.87 ms per control step go
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=/tmp/prof_sampler; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out; cd $R
GS=${GS:-0} HORIZON=${HORIZON:-30} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python tools/gpu_sampler.py > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_sampler/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    a = agg[r['Kernel_Name'][:90]]; a[0] += 1; a[1] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3
tot = sum(a[1] for a in agg.values())
out = open('gpurun_out/r05_sampler_kernels.txt', 'w')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    line = f"{100*a[1]/tot:6.2f}%  calls={a[0]:5d}  avg={a[1]/a[0]:9.1f} us  total={a[1]/1e3:9.2f} ms  {k}"
    print(line); out.write(line + "\n")
# busy vs wall between first and last kernel
st = sorted((float(r['Start_Timestamp']), float(r['End_Timestamp'])) for r in rows)
busy, cur_s, cur_e = 0.0, st[0][0], st[0][1]
for s, e in st[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
line = f"kernels: {len(rows)}  union of kernel time {busy/1e6:.1f} ms over a span of {(st[-1][1]-st[0][0])/1e6:.1f} ms"
print(line); out.write(line + "\n")
PY
