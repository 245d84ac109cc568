# round-2 measurement set: bench lines of every workload, rocprofv3 kernel trace + PMC passes of the headline, stage ticks
set -x
mkdir -p gpurun_out
timeout 900 python bench.py --steps 1000 --warmup 20 > gpurun_out/r02_bench_smpl4096.json 2> gpurun_out/bench.err; echo "bench smpl rc=$?"
for w in getup smplx imitation; do
  timeout 600 python bench.py --workload $w --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/r02_bench_${w}.json 2>> gpurun_out/bench.err; echo "bench $w rc=$?"
done
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --self-collision > gpurun_out/r02_bench_smpl4096_selfcollision.json 2>> gpurun_out/bench.err; echo "bench selfcol rc=$?"
TAG=r02 WORKLOAD=smpl ENVS_PER_GPU=4096 bash tools/gpu_prof.sh > gpurun_out/prof.log 2>&1; echo "prof rc=$?"
bash tools/gpu_stage.sh > gpurun_out/r02_stage_ticks.txt 2>&1
python tools/gpu_selfcol.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_selfcollision_cost.txt
python tools/gpu_lone.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_lone_wave.txt
tail -5 gpurun_out/bench.err
for f in gpurun_out/r02_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'))"; done
