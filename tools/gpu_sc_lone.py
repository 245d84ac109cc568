"""Body-body contacts: lone-wave latency (<= 1 env per CU) and full-chip step time of the library SMPLSIM_HIP_LIB points at."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.batch import SMPLSimVecEnv
def run(tag, N, steps=20, act=1.0, **kw):
    env = SMPLSimVecEnv(N, autoreset=True, seed=1234, self_collision=True, **kw)
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    env.reset()
    for _ in range(30): env.step((torch.rand(N, 69, generator=g, device=env.device) * 2 - 1) * act)
    torch.cuda.synchronize(); t0 = time.perf_counter(); its = 0; mx = 0
    for _ in range(steps):
        env.step((torch.rand(N, 69, generator=g, device=env.device) * 2 - 1) * act)
        its += env.solver_iters.float().max().item(); mx = max(mx, env.self_contacts.max().item())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{os.environ.get('TAG','')} {tag:22s} N={N:5d} {1e3*dt:7.3f} ms/step  max iters/step {its/steps:6.1f}  mean {env.solver_iters.float().mean().item():.1f} contacts mean {env.self_contacts.float().mean().item():.2f} max {mx}", flush=True)
run("lone waves", 256)
run("lone waves maxit1", 256, newton_iters=1)
run("lone, 4 per CU", 1024)
run("full chip", 4096)
