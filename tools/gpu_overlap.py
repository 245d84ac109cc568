"""(See also smplsim_amd/pipeline.py, the packaged form of this.)  Does overlapping the straggler tail of one sub-batch with the body of another help?  G independent sub-batches
(SMPLSimVecEnv of 4096/G envs each, own torch stream), stepped round-robin; time per step of all 4096 envs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.batch import SMPLSimVecEnv
N = 4096
for G in (1, 2, 4):
    envs = [SMPLSimVecEnv(N // G, autoreset=True, seed=1234 + i) for i in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    gens = []
    for e, s in zip(envs, streams):
        g = torch.Generator(device=e.device); g.manual_seed(99); gens.append(g)
        with torch.cuda.stream(s):
            e.reset()
    def step_all():
        for e, s, g in zip(envs, streams, gens):
            with torch.cuda.stream(s):
                e.step(torch.rand(e.num_envs, 69, generator=g, device=e.device) * 2 - 1)
    for _ in range(20):
        step_all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        step_all()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"G={G}: {1e3 * dt / K:.3f} ms per step of {N} envs -> {N * K / dt:,.0f} env-steps/s")
    del envs
