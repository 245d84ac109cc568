# r03b: GPU tests at the new solver settings, the default bench line as the driver runs it, stage profile of the headline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -40
python bench.py > gpurun_out/r03b_bench_default.json 2> gpurun_out/r03b_bench_default.err; tail -c 3000 gpurun_out/r03b_bench_default.json
SELFCOLS=0 bash tools/gpu_stage.sh
python tools/gpu_lone.py 2>/dev/null
