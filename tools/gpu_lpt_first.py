"""Hand-out order from the step's own first mj_steps: how well do the Newton counts of the first k of the 15 mj_steps of a control step
order the whole step?  (The step is run repeatedly from the same state; SELFCOL=1 for the body-body-contact workload.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smplsim_amd.batch import SMPLSimVecEnv, _check, _ptr
from smplsim_amd._lib import lib
SC = os.environ.get("SELFCOL", "0") == "1"
N = 4096
env = SMPLSimVecEnv(N, autoreset=True, seed=1234, lpt_order=False, self_collision=SC)
g = torch.Generator(device=env.device); g.manual_seed(1234)
env.reset()
fields = ("qpos", "qvel", "qpos_prev", "qvel_prev", "qacc_warm", "cur_t", "task_state", "nwarn", "body_vel", "touch", "self_contacts", "solver_iters")
def snap(): return {k: getattr(env, k).clone() for k in fields}
def restore(s):
    for k, v in s.items(): getattr(env, k).copy_(v)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
def set_order(o): _check(lib().ss_set_order(env.handle, _ptr(o) if o is not None else None))
def popc(x):
    x = x.to(torch.int64) & 0xFFFFFFFF
    c = torch.zeros_like(x)
    for i in range(32): c += (x >> i) & 1
    return c
def shipped_key(s):
    tc = (popc(s["touch"][:, 0]) + popc(s["touch"][:, 1])).float()
    am = s["qacc_warm"].abs().nan_to_num(1e12).amax(1).clamp(max=1e12); vm = s["qvel"].abs().nan_to_num(1e6).amax(1).clamp(max=1e6)
    return s["solver_iters"].float() + 6 * tc + 8 * torch.log1p(am) + 8 * torch.log1p(vm)
res = {}
def add(k, v): res.setdefault(k, []).append(v)
for t in range(int(os.environ.get("STEPS", "90"))):
    a = torch.rand(N, 69, generator=g, device=env.device) * 2 - 1
    s = snap()
    if t >= 60:
        k0 = shipped_key(s); o0 = torch.argsort(k0, descending=True, stable=True).to(torch.int32)
        set_order(o0); add("shipped", timed(lambda: env.step(a))); true_it = env.solver_iters.clone().float(); s_after = snap()
        restore(s); set_order(torch.argsort(true_it, descending=True, stable=True).to(torch.int32)); add("perfect", timed(lambda: env.step(a)))
        for kf in (1, 2, 3, 5):
            restore(s); set_order(o0); tA = timed(lambda: env.substep(a, kf)); itk = env.solver_iters.clone().float()
            restore(s); set_order(torch.argsort(itk, descending=True, stable=True).to(torch.int32)); tF = timed(lambda: env.step(a))
            add(f"first{kf}: phase A alone", tA); add(f"first{kf}: whole step in that order", tF)
            add(f"first{kf}: estimate A + (15-k)/15 of whole", tA + tF * (15 - kf) / 15)
            restore(s); set_order(torch.argsort(itk + 0.25 * k0, descending=True, stable=True).to(torch.int32)); add(f"first{kf}+.25 shipped: whole step", timed(lambda: env.step(a)))
        restore(s_after)
    else:
        set_order(torch.argsort(shipped_key(s), descending=True, stable=True).to(torch.int32)); env.step(a)
for k, v in res.items():
    print(f"{k:44s} mean {np.mean(v):.4f} ms  (n={len(v)})")
