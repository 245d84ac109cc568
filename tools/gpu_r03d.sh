cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SELFCOLS="1" bash tools/gpu_stage.sh
python - <<'PY'
import sys, time, os, torch
sys.path.insert(0, os.getcwd())
from smplsim_amd.batch import SMPLSimVecEnv
def run(tag, N, steps=20, **kw):
    env = SMPLSimVecEnv(N, autoreset=True, seed=1234, self_collision=True, **kw)
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    env.reset()
    for _ in range(20): env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
    torch.cuda.synchronize(); t0 = time.perf_counter(); mx = 0; nc = 0
    for _ in range(steps):
        env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
        mx += env.solver_iters.float().max().item(); nc += (env.self_contacts == 8).float().mean().item()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{tag:30s} N={N:5d} {1e3*dt:7.3f} ms/step max iters/step {mx/steps:6.1f} mean {env.solver_iters.float().mean().item():.1f} contacts mean {env.self_contacts.float().mean().item():.2f} at-cap frac {nc/steps:.3f} {env.launch_info()}")
run("selfcol lone waves", 64)
run("selfcol lone waves", 256)
run("selfcol lone maxit1", 256, newton_iters=1)
run("selfcol full chip", 4096)
PY
