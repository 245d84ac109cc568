#!/usr/bin/env python3
"""Cost of body-shape variation on one GPU: 4096 SMPL envs stepped as K shape groups (one ss_batch + stream per group,
joined every step) versus one batch.  Same workload as bench.py (uniform(-1,1) actions, fused autoreset)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from smplsim_amd.mjcf_writer import scaled_xml_str
from smplsim_amd.shapes import ShapeVariedVecEnv

N, STEPS = 4096, int(os.environ.get("STEPS", 200))
SINGLE = os.environ.get("SINGLE_LAUNCH", "1") == "1"
for K in ((1, 2, 16, 256, 4096) if SINGLE else (1, 2, 4, 8, 16, 32)):
    # identical geometry in every group (but K separately compiled models / batches / streams): isolates the cost of grouping —
    # really different shapes change the workload itself (a scaled body starts above or inside the floor at the fixed reset height)
    xmls = [scaled_xml_str("smpl_humanoid", 1.0) for g in range(K)]
    env = ShapeVariedVecEnv(xmls, N // K, seed=0, single_launch=SINGLE)
    g = torch.Generator(device=env.device); g.manual_seed(1)
    env.reset()
    for _ in range(10):
        env.step(torch.rand(N, env.nu, generator=g, device=env.device) * 2 - 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(STEPS):
        env.step(torch.rand(N, env.nu, generator=g, device=env.device) * 2 - 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"single_launch": SINGLE, "shapes": K, "envs_per_shape": N // K, "env_steps_per_s": round(N * STEPS / dt), "ms_per_step": round(1e3 * dt / STEPS, 3),
                      "obs_finite": bool(torch.isfinite(env.obs_buf).all())}))
    for e in env.envs:
        e.close()
