cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=/tmp/gemm_pmc; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd $R
for grp in "a:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "b:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  timeout -k 5 120 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/$name -o pmc -- python tools/gpu_gemm_one.py > $OUT/$name.log 2>&1 < /dev/null; echo "pmc $name rc=$?"
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/tmp/gemm_pmc/*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ss_linear' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items(): print(f"{k:28s} mean/dispatch {sum(v)/len(v):.4g}  (n={len(v)})")
PY
