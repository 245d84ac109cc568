# round-2 final measurement set (every command bounded, stdin closed): GPU tests, smoke, bench lines of every workload,
# rocprofv3 kernel trace + PMC passes of the headline, stage ticks, lone-wave pricing
mkdir -p gpurun_out
T="timeout -k 5"
$T 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
$T 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 < /dev/null; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
$T 600 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/bench.err < /dev/null; echo "bench default rc=$?"
$T 900 python bench.py --steps 1000 --warmup 20 > gpurun_out/r02_bench_smpl4096.json 2>> gpurun_out/bench.err < /dev/null; echo "bench smpl rc=$?"
for w in getup smplx imitation; do
  $T 600 python bench.py --workload $w --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/r02_bench_${w}.json 2>> gpurun_out/bench.err < /dev/null; echo "bench $w rc=$?"
done
$T 600 python bench.py --workload imitation --steps 300 --warmup 20 --no-cpu-baseline --unfused > gpurun_out/r02_bench_imitation_unfused.json 2>> gpurun_out/bench.err < /dev/null
$T 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --self-collision > gpurun_out/r02_bench_smpl4096_selfcollision.json 2>> gpurun_out/bench.err < /dev/null; echo "bench selfcol rc=$?"
TAG=r02 WORKLOAD=smpl ENVS_PER_GPU=4096 $T 900 bash tools/gpu_prof.sh > gpurun_out/prof.log 2>&1 < /dev/null; echo "prof rc=$?"
$T 300 bash tools/gpu_stage.sh > gpurun_out/r02_stage_ticks.txt 2>&1 < /dev/null
$T 300 python tools/gpu_lone.py 2>&1 < /dev/null | grep -v amdgpu.ids > gpurun_out/r02_lone_wave.txt
tail -3 gpurun_out/bench.err
for f in gpurun_out/r02_bench_*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f')); print('$f', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'))
except Exception as e: print('$f', 'unreadable', e)"; done
