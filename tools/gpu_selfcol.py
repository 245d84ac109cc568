"""Cost of the body-body contact path: ms per 4096-env step with self_collision off / on at several action amplitudes
(amplitude 0.02 = standing, no body-body contacts: the fixed cost of the pair table scan + the larger LDS slice)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.batch import SMPLSimVecEnv
def run(N, act, sc, steps=40, warm=25):
    env = SMPLSimVecEnv(N, autoreset=True, seed=1234, self_collision=sc)
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    env.reset()
    for _ in range(warm): env.step((torch.rand(N, 69, generator=g, device=env.device) * 2 - 1) * act)
    torch.cuda.synchronize(); t0 = time.perf_counter(); its = 0; frac = 0; mx = 0
    for _ in range(steps):
        env.step((torch.rand(N, 69, generator=g, device=env.device) * 2 - 1) * act)
        its += env.solver_iters.float().mean().item(); frac += (env.self_contacts > 0).float().mean().item(); mx = max(mx, env.self_contacts.max().item())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"N={N:5d} amp {act:4.2f} self_collision {int(sc)}: {1e3*dt:8.3f} ms/step  {N/dt/1e6:6.3f} M env-steps/s  mean iters {its/steps:6.1f}  envs with body-body contact {frac/steps:5.3f} max {mx}  {env.launch_info()}", flush=True)
N = int(os.environ.get("NENV", "4096"))
for act in (0.02, 0.3, 1.0):
    for sc in (False, True):
        run(N, act, sc)
