"""Which pre-step quantities predict an env's Newton-iteration count of the coming control step?  (hand-out order study)
Saves per-workload feature / target arrays to gpurun_out/lpt_features_<workload>.npz for offline fitting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smplsim_amd.batch import SMPLSimVecEnv, ShardModel
def collect(tag, N, steps, **kw):
    env = SMPLSimVecEnv(N, autoreset=True, seed=1234, **kw)
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    env.reset()
    X, Y = [], []
    prev = torch.zeros(N, device=env.device); prev2 = prev.clone()
    for t in range(steps):
        a = torch.rand(N, env.nu, generator=g, device=env.device) * 2 - 1
        pop = torch.zeros(N, device=env.device)
        for w in range(2):
            x = env.touch[:, w].to(torch.int64) & 0xFFFFFFFF
            for b in range(32): pop += ((x >> b) & 1).float()
        feats = torch.stack([prev, prev2, pop, env.qpos[:, 2], env.qvel.abs().amax(1).clamp(max=1e6), env.qacc_warm.abs().amax(1).clamp(max=1e12),
                             env.qvel[:, :6].abs().amax(1).clamp(max=1e6), env.cur_t.float()], 1)
        env.step(a)
        it = env.solver_iters.float()
        if t >= 12:
            X.append(feats.cpu().numpy()); Y.append(it.cpu().numpy())
        prev2 = prev; prev = it
    np.savez_compressed(f"gpurun_out/lpt_features_{tag}.npz", X=np.stack(X), Y=np.stack(Y))
    print(tag, "saved", np.stack(X).shape)
collect("smpl", 4096, 72)
collect("getup", 4096, 40, task="HumanoidGetup", state_init="Fall")
collect("smplx", 4096, 40, model=ShardModel(humanoid="smplx_humanoid"))
collect("smpl_selfcol", 4096, 30, self_collision=True)
