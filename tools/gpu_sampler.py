#!/usr/bin/env python3
"""Sampler throughput = stepper + policy inference: AgentPPO.sample (one 4096-env batch, policy then step, serially on
one stream) against AgentPPO.sample_pipelined over G sub-batches on G streams (same job, bit-identical rollouts), reference MLP
2048-1536-1024-1024-512-512, bf16 MFMA policy kernels.  Prints env-steps/s, the host's issue time per step and an equality check."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd.agents.ppo import AgentPPO, PPOConfig
from smplsim_amd.batch import SMPLSimVecEnv
from smplsim_amd.pipeline import PipelinedVecEnv

N = int(os.environ.get("NENV", "4096")); T = int(os.environ.get("HORIZON", "50")); TASK = os.environ.get("TASK", "HumanoidSpeed")
kw = dict(task=TASK, seed=0)
cfg = PPOConfig(min_batch_size=N * T, mfma_inference=os.environ.get("TORCH_POLICY") is None)
res = {"envs": N, "horizon": T, "task": TASK, "policy": "bf16 MFMA kernels" if cfg.mfma_inference else "torch fp32"}

def run(agent, fn, reps=3):
    fn(); torch.cuda.synchronize()                                   # warm-up (allocator, module load)
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b = fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        r = (N * T / (t2 - t0), 1e3 * (t1 - t0) / T, 1e3 * (t2 - t0) / T)
        best = r if best is None or r[0] > best[0] else best
    return best, b

a1 = AgentPPO(SMPLSimVecEnv(N, **kw), cfg, seed=0)
(r, b1) = run(a1, a1.sample)
res["serial"] = {"env_steps_per_s": round(r[0]), "host_issue_ms_per_step": round(r[1], 3), "ms_per_step": round(r[2], 3)}
a1.env.close()
for G in [int(x) for x in os.environ.get("GS", "2,4,8").split(",") if int(x) > 1]:
    pipe = PipelinedVecEnv(N, sub_batches=G, **kw)
    if os.environ.get("PIPE_EPW"):                                 # envs per workgroup of every sub-batch: small enough for two workgroups
        from smplsim_amd._lib import lib                           # (of different sub-batches) to share a CU's LDS
        from smplsim_amd.batch import _check
        for e in pipe.envs:
            _check(lib().ss_set_launch_geometry(e.handle, int(os.environ["PIPE_EPW"]), 0))
    a2 = AgentPPO(pipe, cfg, seed=0)
    (r, b2) = run(a2, lambda: a2.sample_pipelined(pipe))
    res[f"pipelined_G{G}" + ("_epw" + os.environ["PIPE_EPW"] if os.environ.get("PIPE_EPW") else "")] = {"env_steps_per_s": round(r[0]), "host_issue_ms_per_step": round(r[1], 3), "ms_per_step": round(r[2], 3),
                              "rollout_equals_serial": bool(all(torch.equal(b1[k], b2[k]) for k in b1))}
    pipe.close()
print(json.dumps(res))
