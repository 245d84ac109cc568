"""How much of the step launch is the straggler tail?  Launch time versus the cap on Newton iterations per mj_step
(diagnostic only: a cap below convergence changes the results)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.batch import SMPLSimVecEnv
N = 4096
for cap, lpt in ((8, True), (8, False), (4, True), (2, True), (1, True)):
    env = SMPLSimVecEnv(N, autoreset=True, seed=1234, newton_iters=cap)
    env.lpt_order = lpt
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    env.reset()
    for _ in range(10):
        env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for a, b in ev:
        env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1, _events=(a, b))
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    it = env.solver_iters.float()
    print(f"newton cap {cap} lpt {lpt}: step launch {ms:.3f} ms, iters mean {it.mean():.1f} max {it.max():.0f}, "
          f"ideal-balanced bound = mean work share: {ms * (it.mean() + 15 * 1.4) / (it.max() + 15 * 1.4):.2f}")
