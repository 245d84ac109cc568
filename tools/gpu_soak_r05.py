import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from smplsim_amd.batch import ShardModel, SMPLSimVecEnv
N, K = 4096, 3000
for name, kw in (("smplx", dict(model=ShardModel(humanoid="smplx_humanoid"))), ("smpl selfcol", dict(self_collision=True))):
    env = SMPLSimVecEnv(N, autoreset=True, seed=7, **kw)
    g = torch.Generator(device=env.device); g.manual_seed(7)
    env.reset(); tb = []
    for blk in range(K // 500):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500):
            obs, rew, term, trunc, _ = env.step(torch.rand(N, env.nu, generator=g, device=env.device) * 2 - 1)
        torch.cuda.synchronize(); tb.append((time.perf_counter() - t0) / 500 * 1e3)
        assert torch.isfinite(obs).all() and torch.isfinite(env.qpos).all() and torch.isfinite(env.qvel).all()
        assert (env.qpos[:, 3:7].norm(dim=1) - 1).abs().max() < 1e-3 and int(env.cur_t.max()) <= 301
    print(f"{name}: {K} steps ok, ms/step per 500-block min {min(tb):.3f} max {max(tb):.3f}, bad-state resets {int(env.nwarn.sum())}")
