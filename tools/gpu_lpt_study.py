"""How much of the step launch is hand-out order?  For the headline workload: launch time with (a) no order, (b) the shipped
order = last step's Newton counts, (c) the PERFECT order = this step's own counts (the step is run twice from the same state),
(d) max(last two steps).  Times by HIP events around the step launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from smplsim_amd.batch import SMPLSimVecEnv, _check, _ptr
from smplsim_amd._lib import lib

N = 4096
env = SMPLSimVecEnv(N, autoreset=False, seed=1234, lpt_order=False)
g = torch.Generator(device=env.device); g.manual_seed(1234)
env.reset()
fields = ("qpos", "qvel", "qpos_prev", "qvel_prev", "qacc_warm", "cur_t", "task_state", "nwarn", "body_vel", "touch")
def snap(): return {k: getattr(env, k).clone() for k in fields}
def restore(s):
    for k, v in s.items(): getattr(env, k).copy_(v)
def timed_step(a, order):
    _check(lib().ss_set_order(env.handle, _ptr(order) if order is not None else None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.step(a); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
res = {k: [] for k in ("none", "last", "perfect", "max2", "reverse_perfect")}
prev = torch.zeros(N, dtype=torch.int32, device=env.device); prev2 = prev.clone()
for t in range(60):
    a = torch.rand(N, 69, generator=g, device=env.device) * 2 - 1
    s = snap()
    t_none = timed_step(a, None); true_it = env.solver_iters.clone()
    if t >= 20:
        res["none"].append(t_none)
        for name, key in (("last", prev), ("perfect", true_it), ("max2", torch.maximum(prev, prev2)), ("reverse_perfect", -true_it)):
            restore(s)
            order = torch.argsort(key, descending=True, stable=True).to(torch.int32)
            res[name].append(timed_step(a, order))
    # autoreset by hand (bad states are reset in-kernel); episodes here never end (base task, 300 steps)
    prev2 = prev; prev = true_it
for k, v in res.items():
    print(f"{k:16s} mean {np.mean(v):.4f} ms  (n={len(v)})")
print("iters mean", float(true_it.float().mean()), "max", int(true_it.max()))
