# kernel trace of the PPO epoch (tools/train_ppo.py): top kernels by time.  ARGS="--mfma-inference --mfma-update"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=/tmp/ppo_prof; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python tools/train_ppo.py ${ARGS:---mfma-inference --mfma-update} --epochs ${EPOCHS:-3} > $OUT/log 2>&1
tail -2 $OUT/log
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    a = agg[r["Kernel_Name"][:110]]; a[0] += 1; a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print("total kernel time ms", round(tot / 1e6, 1))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:${TOP:-28}]:
    print(f"{100 * a[1] / tot:6.2f}%  calls={a[0]:6d}  avg={a[1] / a[0] / 1e3:9.1f} us  total={a[1] / 1e6:8.2f} ms  {k}")
PY
