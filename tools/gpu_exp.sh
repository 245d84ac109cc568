rocprofv3 -L 2>/dev/null | grep -i -E "IFETCH|ICACHE|INST_CACHE|SQC_" | head -30
for opt in "-O3" "-Os" "-O2" "-O3 -mllvm -amdgpu-unroll-threshold-private=0 -fno-unroll-loops"; do
  SS_HIPCC_OPT="$opt" python -c "from smplsim_amd import _lib; _lib.build(force=True)" 2>/dev/null
  echo "== $opt: $(/opt/rocm/lib/llvm/bin/llvm-readelf -s --wide smplsim_amd/libsmplsim_hip.so 2>/dev/null | grep -c xx)"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],3), d['config']['launch'])"
done
