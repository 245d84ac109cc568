# round-4 measurement set: bench lines of every workload, rocprofv3 kernel trace + PMC passes per workload (the sources of
# profiles/pmc_summary_<workload>.json that bench.py's roofline object cites), MLP kernel PMC, stage ticks, lone-wave timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --steps 1000 --warmup 20 > gpurun_out/r05_bench_smpl4096.json 2> gpurun_out/bench.err; echo "bench smpl rc=$?"
for w in getup smplx imitation; do
  timeout 600 python bench.py --workload $w --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/r05_bench_${w}.json 2>> gpurun_out/bench.err; echo "bench $w rc=$?"
done
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --self-collision > gpurun_out/r05_bench_smpl4096_selfcollision.json 2>> gpurun_out/bench.err; echo "bench selfcol rc=$?"
if [ -n "$BENCH_ONLY" ]; then   # only the bench lines (they cite profiles/pmc_summary_*.json as committed)
  for f in gpurun_out/r05_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))"; done
  exit 0
fi
TAG=r05_smpl WORKLOAD=smpl ENVS_PER_GPU=4096 BENCH_ARGS="--no-reference-contact-set" bash tools/gpu_prof.sh > gpurun_out/prof_smpl.log 2>&1; echo "prof smpl rc=$?"
TAG=r05_getup WORKLOAD=getup ENVS_PER_GPU=4096 BENCH_ARGS="--workload getup --no-reference-contact-set" bash tools/gpu_prof.sh > gpurun_out/prof_getup.log 2>&1; echo "prof getup rc=$?"
TAG=r05_smplx WORKLOAD=smplx ENVS_PER_GPU=4096 BENCH_ARGS="--workload smplx --no-reference-contact-set" bash tools/gpu_prof.sh > gpurun_out/prof_smplx.log 2>&1; echo "prof smplx rc=$?"
TAG=r05_imitation WORKLOAD=imitation ENVS_PER_GPU=1024 BENCH_ARGS="--workload imitation" bash tools/gpu_prof.sh > gpurun_out/prof_imitation.log 2>&1; echo "prof imitation rc=$?"
TAG=r05_smpl_selfcollision WORKLOAD=smpl_selfcollision ENVS_PER_GPU=4096 BENCH_ARGS="--self-collision" bash tools/gpu_prof.sh > gpurun_out/prof_selfcol.log 2>&1; echo "prof selfcol rc=$?"
bash tools/gpu_gemm_pmc.sh > gpurun_out/r05_mlp_gemm_pmc.txt 2>&1
python tools/gpu_mlp.py > gpurun_out/r05_mlp_inference.txt 2>&1
SELFCOLS="0 1" bash tools/gpu_stage.sh > gpurun_out/r05_stage_ticks.txt 2>&1
python tools/gpu_lone.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_lone_wave.txt
bash tools/gpu_sc_profile.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_selfcol_stage_lone.txt
tail -5 gpurun_out/bench.err
for f in gpurun_out/r05_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))"; done
ls gpurun_out/*pmc_summary.json
