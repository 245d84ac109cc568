"""Pipelined sub-batches on one GPU: G independent SMPLSimVecEnv shards, each on its own HIP stream.

A step launch ends when its slowest env ends (heavy-tailed Newton iteration counts), so a single batch leaves the GPU
mostly idle during the tail of every launch.  With G sub-batches stepped round-robin — sub-batch g+1 is launched while
the tail of sub-batch g is still running, and the policy / action generation of one sub-batch overlaps the stepping
of the others — the tails are hidden (+12 % env-steps/s at G = 4, 4096 SMPL envs, profiles/r01v_straggler_tail.txt).

This relaxes the per-step barrier over ALL envs that the reference's vector env has (gym.vector.AsyncVectorEnv.step
waits for every worker, examples/benchmark.py:78-116): each sub-batch keeps the barrier over its own envs only.
bench.py therefore does not use it; it is for samplers that evaluate their policy per sub-batch.
"""
import torch

from .batch import SMPLSimVecEnv


class PipelinedVecEnv:
    def __init__(self, num_envs, sub_batches=4, device=0, seed=0, **env_kw):
        assert num_envs % sub_batches == 0
        self.num_envs, self.sub_batches = num_envs, sub_batches
        n = num_envs // sub_batches
        self.envs = [SMPLSimVecEnv(n, device=device, seed=seed + 1000 * g, **env_kw) for g in range(sub_batches)]
        self.streams = [torch.cuda.Stream(device=self.envs[0].device) for _ in range(sub_batches)]
        self.device, self.nu, self.obs_size = self.envs[0].device, self.envs[0].nu, self.envs[0].obs_size

    def reset(self):
        out = []
        for env, s in zip(self.envs, self.streams):
            with torch.cuda.stream(s):
                out.append(env.reset()[0])
        return out

    def step_async(self, g, actions, task_rand=None):
        """Enqueue one control step of sub-batch g on its stream; returns that sub-batch's (obs, rew, term, trunc, info)
        tensors, valid once the stream has reached this point (torch ops issued under `stream(g)` are ordered after it)."""
        with torch.cuda.stream(self.streams[g]):
            return self.envs[g].step(actions, task_rand)

    def stream(self, g):
        return torch.cuda.stream(self.streams[g])

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
