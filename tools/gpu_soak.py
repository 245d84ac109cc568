"""Soak: many control steps of the bench workload; finiteness, bookkeeping and launch-time stability."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.batch import SMPLSimVecEnv
N, K = 4096, int(os.environ.get("STEPS", "6000"))
for task, init in (("HumanoidEnv", "Default"), ("HumanoidGetup", "Fall")):
    env = SMPLSimVecEnv(N, task=task, state_init=init, autoreset=True, seed=7)
    g = torch.Generator(device=env.device); g.manual_seed(7)
    env.reset()
    worst, t_blocks = 0.0, []
    for blk in range(K // 500):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500):
            obs, rew, term, trunc, _ = env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
        torch.cuda.synchronize(); t_blocks.append((time.perf_counter() - t0) / 500 * 1e3)
        assert torch.isfinite(obs).all() and torch.isfinite(env.qpos).all() and torch.isfinite(env.qvel).all() and torch.isfinite(rew).all()
        qn = env.qpos[:, 3:7].norm(dim=1)
        assert (qn - 1).abs().max() < 1e-3
        assert int(env.cur_t.max()) <= 301 and int(env.cur_t.min()) >= 0
        worst = max(worst, float(env.qvel.abs().max()))
    print(f"{task}/{init}: {K} steps ok, ms/step per 500-block min {min(t_blocks):.3f} max {max(t_blocks):.3f}, autoresets {int(env.nwarn.sum())}, max |qvel| seen {worst:.3g}")
