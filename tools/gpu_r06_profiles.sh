# round-6 measurement set on the FINAL tree (every profile is taken on the TIMED regime: WARMUP control steps are stepped but not counted,
# VERDICT r5 item 5): bench lines of every workload, rocprofv3 kernel trace + PMC passes per workload (the sources of
# profiles/pmc_summary_<workload>.json, which carry the tree's source hash: bench.py refuses a block taken on another tree), lone-wave timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
WARMUP=20 TAG=r06_smpl WORKLOAD=smpl ENVS_PER_GPU=4096 BENCH_ARGS="--no-reference-contact-set" bash tools/gpu_prof.sh > gpurun_out/prof_smpl.log 2>&1; echo "prof smpl rc=$?"
WARMUP=60 TAG=r06_smpl_selfcollision WORKLOAD=smpl_selfcollision ENVS_PER_GPU=4096 BENCH_ARGS="--self-collision" bash tools/gpu_prof.sh > gpurun_out/prof_selfcol.log 2>&1; echo "prof selfcol rc=$?"
WARMUP=20 TAG=r06_smplx WORKLOAD=smplx ENVS_PER_GPU=4096 BENCH_ARGS="--workload smplx --no-reference-contact-set" bash tools/gpu_prof.sh > gpurun_out/prof_smplx.log 2>&1; echo "prof smplx rc=$?"
WARMUP=150 STEPS=60 TAG=r06_getup WORKLOAD=getup ENVS_PER_GPU=4096 BENCH_ARGS="--workload getup --no-reference-contact-set" bash tools/gpu_prof.sh > gpurun_out/prof_getup.log 2>&1; echo "prof getup rc=$?"
WARMUP=60 TAG=r06_imitation WORKLOAD=imitation ENVS_PER_GPU=1024 BENCH_ARGS="--workload imitation" bash tools/gpu_prof.sh > gpurun_out/prof_imitation.log 2>&1; echo "prof imitation rc=$?"
# the summaries go where bench.py looks for them BEFORE the bench lines are taken
for w in smpl smpl_selfcollision smplx getup imitation; do cp gpurun_out/r06_${w}_summary_pmc_summary.json profiles/pmc_summary_${w}.json 2>/dev/null; done
timeout 900 python bench.py --steps 1000 --warmup 20 > gpurun_out/r06_bench_smpl4096.json 2> gpurun_out/bench.err; echo "bench smpl rc=$?"
timeout 600 python bench.py > gpurun_out/r06_bench_default.json 2>> gpurun_out/bench.err; echo "bench default rc=$?"
for w in getup smplx imitation; do
  timeout 600 python bench.py --workload $w --steps 300 --warmup 60 --no-cpu-baseline > gpurun_out/r06_bench_${w}.json 2>> gpurun_out/bench.err; echo "bench $w rc=$?"
done
timeout 600 python bench.py --steps 200 --warmup 60 --no-cpu-baseline --self-collision > gpurun_out/r06_bench_smpl4096_selfcollision.json 2>> gpurun_out/bench.err; echo "bench selfcol rc=$?"
timeout 600 python bench.py --envs-total 512 --steps 300 --warmup 20 --no-cpu-baseline --no-reference-contact-set > gpurun_out/r06_bench_smpl512_one_eighth_of_strong_scaling.json 2>> gpurun_out/bench.err; echo "bench 512 rc=$?"
python tools/gpu_lone.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_lone_wave.txt
python tools/gpu_mlp.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_mlp_inference.txt
GS=2 python tools/gpu_sampler.py 2>&1 | tail -1 > gpurun_out/r06_sampler.json
tail -5 gpurun_out/bench.err
for f in gpurun_out/r06_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']; print('$f', round(d['value']), round(d['ms_per_step'],3), r.get('kernel_ms'), r.get('frac'), 'pmc_stale', r.get('pmc_stale'), 'traffic', r.get('traffic'), d['config'].get('launch'))"; done
ls gpurun_out/*pmc_summary.json
