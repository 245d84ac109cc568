# round-2 first GPU pass: smoke, GPU tests (incl. per-sample parity), headline bench, rocprofv3 trace + PMC passes
set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
tail -4 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -rA > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|ERROR" gpurun_out/pytest_gpu.log | tail -15
timeout 600 python bench.py --steps 300 --warmup 20 > gpurun_out/bench_smpl.json 2> gpurun_out/bench_smpl.err; echo "bench rc=$?"
cat gpurun_out/bench_smpl.json; tail -3 gpurun_out/bench_smpl.err
TAG=r02a WORKLOAD=smpl ENVS_PER_GPU=4096 bash tools/gpu_prof.sh > gpurun_out/prof.log 2>&1; echo "prof rc=$?"
tail -40 gpurun_out/prof.log
