"""Sub-batches in flight: G independent SMPLSimVecEnv shards of the 4096-env headline workload, each on its own stream, stepped
round-robin (PipelinedVecEnv): the tail of one sub-batch's launch overlaps the body of the next.  Each sub-batch keeps the
vector-env barrier over its own envs only, so this is a sampler's figure, not bench.py's headline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.pipeline import PipelinedVecEnv
N = 4096
for G in (1, 2, 4, 8):
    pipe = PipelinedVecEnv(N, sub_batches=G, seed=1234)
    gens = [torch.Generator(device=pipe.device) for _ in range(G)]
    for i, g in enumerate(gens): g.manual_seed(1234 + i)
    bufs = [torch.empty(N // G, pipe.nu, device=pipe.device) for _ in range(G)]
    pipe.reset(); pipe.synchronize()
    def run(steps):
        for _ in range(steps):
            for g in range(G):
                with pipe.stream(g):
                    a = bufs[g].uniform_(-1.0, 1.0, generator=gens[g])
                pipe.step_async(g, a)
    run(20); pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(300); pipe.synchronize(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"sub-batches {G}: {N * 300 / dt:,.0f} env-steps/s  ({1e3 * dt / 300:.3f} ms per step of all {N} envs)")
    del pipe
