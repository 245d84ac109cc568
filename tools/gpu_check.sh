set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" 
tail -8 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc=$?"
cat gpurun_out/bench1.json; tail -3 gpurun_out/bench1.err
