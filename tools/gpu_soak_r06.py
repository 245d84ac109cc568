"""Soak of the shipped tree, round 6: body-body contacts on the matrix-core dense solve (SMPL, and the 52-body SMPL-X whose coupled sets reach
dozens of bodies: matrix rows beyond 64, the shared block), 4096 / 1024 envs, uniform(-1,1) actions, in-launch autoreset."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from smplsim_amd.batch import ShardModel, SMPLSimVecEnv
for name, N, K, kw in (("smpl + body-body contacts", 4096, 3000, dict(self_collision=True)),
                       ("smplx + body-body contacts", 1024, 600, dict(self_collision=True, model=ShardModel(humanoid="smplx_humanoid"))),
                       ("getup / Fall + body-body contacts", 4096, 1000, dict(self_collision=True, task="HumanoidGetup", state_init="Fall"))):
    env = SMPLSimVecEnv(N, autoreset=True, seed=7, **kw)
    g = torch.Generator(device=env.device); g.manual_seed(7)
    env.reset(); tb = []; mc = 0
    blk_n = 500 if K >= 1000 else 200
    for blk in range(K // blk_n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(blk_n):
            obs, rew, term, trunc, _ = env.step(torch.rand(N, env.nu, generator=g, device=env.device) * 2 - 1)
        torch.cuda.synchronize(); tb.append((time.perf_counter() - t0) / blk_n * 1e3)
        mc = max(mc, int(env.self_contacts.max()))
        assert torch.isfinite(obs).all() and torch.isfinite(env.qpos).all() and torch.isfinite(env.qvel).all()
        assert (env.qpos[:, 3:7].norm(dim=1) - 1).abs().max() < 1e-3 and int(env.cur_t.max()) <= 301
    print(f"{name}: {N} envs x {K} steps ok, ms/step per {blk_n}-block min {min(tb):.3f} max {max(tb):.3f}, bad-state resets {int(env.nwarn.sum())}, most body-body contacts at a block's end {mc}", flush=True)
