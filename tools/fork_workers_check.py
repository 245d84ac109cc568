#!/usr/bin/env python3
"""Build a HumanoidSpeed env, THEN fork two worker processes that step it (the reference sampler's pattern, agents/agent.py:121-145),
then step it in the parent too: every process creates the env's device state in its own HIP context; trajectories must agree."""
import multiprocessing as mp
import os
import queue
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from smplsim_amd.config import default_cfg
from smplsim_amd.envs import HumanoidSpeed

env = HumanoidSpeed(default_cfg("HumanoidSpeed"))               # no GPU work yet
assert env._vec_obj is None and env.observation_space.shape[0] == env.get_obs_size()


def work(q, seed):
    np.random.seed(7)
    obs, _ = env.reset(seed=seed)
    out = [obs]
    rs = np.random.default_rng(3)
    for _ in range(3):
        out.append(env.step(rs.uniform(-0.3, 0.3, 69).astype(np.float32))[0])
    q.put(np.stack(out))


if __name__ == "__main__":
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    ps = [ctx.Process(target=work, args=(q, 5)) for _ in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=200) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0, p.exitcode
    assert np.array_equal(res[0], res[1]) and np.isfinite(res[0]).all()
    q2 = queue.Queue()
    work(q2, 5)                                                  # the parent itself, after the children: its own lazy creation
    assert np.array_equal(q2.get(), res[0])
    assert env.get_obs_size() == env._vec.obs_size
    print("fork workers ok", res[0].shape)
