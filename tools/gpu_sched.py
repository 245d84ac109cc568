"""Scheduling experiment: LPT order and multi-stream sub-batches (same total 4096 envs, same work)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.batch import SMPLSimVecEnv
N = 4096
def run(tag, nshard, lpt, steps=30):
    envs = [SMPLSimVecEnv(N // nshard, autoreset=True, seed=1234 + i, lpt_order=lpt) for i in range(nshard)]
    streams = [torch.cuda.Stream() for _ in range(nshard)]
    gens = []
    for i, e in enumerate(envs):
        g = torch.Generator(device=e.device); g.manual_seed(99 + i); gens.append(g); e.reset()
    def step_all():
        for e, s, g in zip(envs, streams, gens):
            with torch.cuda.stream(s):
                e.step(torch.rand(e.num_envs, 69, generator=g, device=e.device) * 2 - 1)
    for _ in range(5): step_all()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step_all()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{tag:34s} {1e3*dt:7.3f} ms/step  -> {N/dt:9.0f} env-steps/s")
run("1 stream, natural order", 1, False)
run("1 stream, LPT order", 1, True)
run("2 streams, LPT", 2, True)
run("4 streams, LPT", 4, True)
run("8 streams, LPT", 8, True)
run("4 streams, natural", 4, False)
