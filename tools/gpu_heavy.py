"""What do the envs look like that need the most Newton iterations (they set the launch time)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.batch import SMPLSimVecEnv
N = 4096
env = SMPLSimVecEnv(N, autoreset=True, seed=1234)
g = torch.Generator(device=env.device); g.manual_seed(1234)
env.reset()
for _ in range(60):
    w0 = env.nwarn.clone()
    env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
it = env.solver_iters
order = torch.argsort(it, descending=True)
touch = env.touch
ntouch = torch.tensor([bin(int(touch[i, 0].item()) & 0xFFFFFF).count("1") for i in range(N)], device=env.device)
dw = env.nwarn - w0
for name, idx in (("top 40 by iterations", order[:40]), ("median 40", order[N // 2 - 20:N // 2 + 20]), ("all", order)):
    print(f"{name}: iters {it[idx].float().mean():.1f}  root z {env.qpos[idx, 2].mean():.2f}  max|qvel| median {env.qvel[idx].abs().max(dim=1).values.median():.1f} "
          f" bodies touching floor {ntouch[idx].float().mean():.1f}  autoreset this step {dw[idx].float().mean():.2f}  cur_t {env.cur_t[idx].float().mean():.0f}")
