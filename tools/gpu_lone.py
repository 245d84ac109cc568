"""Lone-wave latency: few envs (<= 1 wave per CU) so that no two waves share a CU."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.batch import SMPLSimVecEnv
def run(tag, N, steps=20, act=1.0, **kw):
    env = SMPLSimVecEnv(N, autoreset=True, seed=1234, **kw)
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    env.reset()
    for _ in range(30): env.step((torch.rand(N, 69, generator=g, device=env.device) * 2 - 1) * act)
    torch.cuda.synchronize(); t0 = time.perf_counter(); its = 0
    for _ in range(steps):
        env.step((torch.rand(N, 69, generator=g, device=env.device) * 2 - 1) * act)
        its += env.solver_iters.float().max().item()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{tag:44s} N={N:5d} {1e3*dt:7.3f} ms/step  max iters/step {its/steps:6.1f}  mean {env.solver_iters.float().mean().item():.1f}")
run("lone waves, full-range actions", 64)
run("lone waves, full-range actions", 256)
run("lone waves, maxit 1", 256, newton_iters=1)
run("lone waves, tiny actions (standing, few iters)", 256, act=0.02)
run("full chip", 4096)
run("full chip, tiny actions", 4096, act=0.02)
