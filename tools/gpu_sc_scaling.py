#!/usr/bin/env python3
"""Self-collision launch time against the batch size (is the launch bound by its heaviest env's chain or by capacity?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from smplsim_amd.batch import SMPLSimVecEnv

for sc in (True, False):
    for N in (64, 256, 1024, 4096):
        env = SMPLSimVecEnv(N, autoreset=True, seed=1234, self_collision=sc)
        g = torch.Generator(device=env.device); g.manual_seed(1234)
        env.reset()
        for _ in range(40):
            env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
        torch.cuda.synchronize()
        acts = [torch.rand(N, 69, generator=g, device=env.device) * 2 - 1 for _ in range(40)]
        its, nsc = [], []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for a in acts:
            env.step(a)
            its.append(env.solver_iters.max().item()); nsc.append(env.self_contacts.float().mean().item())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
        it = env.solver_iters.float()
        print(f"selfcol={int(sc)} N={N:5d}  {dt*1e3:7.3f} ms/step  {N/dt/1e3:8.1f} k env-steps/s   iters mean {it.mean().item():5.1f} max(avg over steps) {sum(its)/len(its):6.1f}  self contacts mean {sum(nsc)/len(nsc):.2f}")
        del env
