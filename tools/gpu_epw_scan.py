#!/usr/bin/env python3
"""Resident envs per CU against throughput (ss_set_launch_geometry): what one more resident env is worth on a workload.
WORKLOAD=smplx|smpl  SELFCOL=0|1  EPWS="3,4,5" """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd._lib import lib
from smplsim_amd.batch import ShardModel, SMPLSimVecEnv, _check
W = os.environ.get("WORKLOAD", "smplx"); SC = os.environ.get("SELFCOL", "0") == "1"
N, steps = 4096, int(os.environ.get("STEPS", "60"))
model = ShardModel(humanoid="smplx_humanoid" if W == "smplx" else "smpl_humanoid", device=0)
for epw in [int(x) for x in os.environ.get("EPWS", "3,4,5").split(",")]:
    env = SMPLSimVecEnv(N, model=model, autoreset=True, seed=1234, self_collision=SC)
    _check(lib().ss_set_launch_geometry(env.handle, epw, 0))
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    env.reset(); a = torch.empty(N, env.nu, device=env.device)
    for _ in range(20): env.step(a.uniform_(-1, 1, generator=g))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): env.step(a.uniform_(-1, 1, generator=g))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{W} selfcol={int(SC)} envs per CU {epw}: {N*steps/dt:,.0f} env-steps/s  {1e3*dt/steps:.3f} ms/step", flush=True)
    env.close()
