#!/usr/bin/env python3
"""Do the step launches of several batches on several streams overlap?  4 x 1024 SMPL envs: sequential on one stream versus
fanned out over 4 streams and joined every step (the pattern of smplsim_amd/shapes.py)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from smplsim_amd.batch import SMPLSimVecEnv, ShardModel, _check, lib
K, n = 4, 1024
shared = os.environ.get("SHARED_MODEL", "1") == "1"
m = ShardModel(device=0)
envs = [SMPLSimVecEnv(n, model=m if shared else ShardModel(device=0), seed=g) for g in range(K)]
for e in envs:
    _check(lib().ss_set_launch_geometry(e.handle, 12, int(os.environ.get("MAX_WGS", 64)))); e.reset()
streams = [torch.cuda.Stream() for _ in range(K)]
acts = [torch.rand(n, 69, device="cuda") * 2 - 1 for _ in range(K)]
big = torch.rand(K * n, 69, device="cuda") * 2 - 1
def run(multi, steps=100, sliced=False, fresh=False):
    global big
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        if fresh:
            big = torch.rand(K * n, 69, device="cuda") * 2 - 1
        if multi:
            main = torch.cuda.current_stream(); ready = main.record_event()
            for g, (e, s, a) in enumerate(zip(envs, streams, acts)):
                s.wait_event(ready)
                with torch.cuda.stream(s):
                    e.step(big[g * n:(g + 1) * n] if sliced else a)
                main.wait_event(s.record_event())
        else:
            for e, a in zip(envs, acts):
                e.step(a)
    torch.cuda.synchronize(); return round(1e3 * (time.perf_counter() - t0) / steps, 3)
print("shared model", shared, "| one stream", run(False), "| 4 streams", run(True), "| 4 streams, sliced actions", run(True, sliced=True),
      "| + fresh rand", run(True, sliced=True, fresh=True))
