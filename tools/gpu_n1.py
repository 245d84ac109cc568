import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from smplsim_amd.config import default_cfg
from smplsim_amd.envs import HumanoidSpeed
cfg = default_cfg("HumanoidSpeed")
env = HumanoidSpeed(cfg)
obs, info = env.reset(seed=54)
rs = np.random.default_rng(0)
for _ in range(20):
    env.step(rs.uniform(-1, 1, 69))
t0 = time.perf_counter()
n = 1000
for i in range(n):
    o, r, te, tr, info = env.step(rs.uniform(-1, 1, 69))
    if te or tr:
        env.reset()
dt = time.perf_counter() - t0
print("N=1 gym env (speed task): %.0f steps/s, %.3f ms/step" % (n / dt, 1e3 * dt / n))
