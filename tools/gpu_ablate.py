"""Ablation timing on the GPU: which parts of the step cost what (no rebuild; uses cfg knobs)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.batch import SMPLSimVecEnv
N = 4096
def run(tag, steps=20, **kw):
    env = SMPLSimVecEnv(N, autoreset=True, seed=1234, **kw)
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    env.reset()
    for _ in range(3): env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{tag:40s} {1e3*dt:7.3f} ms/step  iters/step {env.solver_iters.float().mean().item():.1f}")
    return env
run("baseline uhc_pd maxit 8")
run("maxit 1", newton_iters=1)
run("maxit 2", newton_iters=2)
run("control_mode=pd (no SPD solve)", control_mode="pd")
run("pd + maxit 1", control_mode="pd", newton_iters=1)
env = run("substeps=1 per step (control_freq_inv=1)", control_freq_inv=1)
run("substeps=5", control_freq_inv=5)
# kinematics only
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): env.kinematics()
torch.cuda.synchronize(); print("kinematics launch (1 position-only forward)", 1e3 * (time.perf_counter() - t0) / 20, "ms")
