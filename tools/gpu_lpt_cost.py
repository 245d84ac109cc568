"""Hand-out order keyed by the env's own measured cycles of the LAST control step instead of its Newton count (study build -DSS_COST_KEY:
cycles >> 10 in the truncation counter's array).  SELFCOL=1 for the body-body-contact workload."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smplsim_amd import _cabi, _lib
_lib._LIB = _cabi.bind_mlp(_cabi.bind(C.CDLL(os.environ["SS_COST_LIB"])))
from smplsim_amd.batch import SMPLSimVecEnv, _check, _ptr
from smplsim_amd._lib import lib
SC = os.environ.get("SELFCOL", "1") == "1"
N = 4096
env = SMPLSimVecEnv(N, autoreset=True, seed=1234, lpt_order=False, self_collision=SC)
g = torch.Generator(device=env.device); g.manual_seed(1234)
env.reset()
cost = torch.zeros(N, dtype=torch.int32, device=env.device)
_check(lib().ss_debug_self_truncation(env.handle, _ptr(cost)))
fields = ("qpos", "qvel", "qpos_prev", "qvel_prev", "qacc_warm", "cur_t", "task_state", "nwarn", "body_vel", "touch", "self_contacts", "solver_iters")
def snap(): return {k: getattr(env, k).clone() for k in fields}
def restore(s):
    for k, v in s.items(): getattr(env, k).copy_(v)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
def set_order(key): _check(lib().ss_set_order(env.handle, _ptr(torch.argsort(key, descending=True, stable=True).to(torch.int32))))
def popc(x):
    x = x.to(torch.int64) & 0xFFFFFFFF
    c = torch.zeros_like(x)
    for i in range(32): c += (x >> i) & 1
    return c
def feats(s):
    tc = (popc(s["touch"][:, 0]) + popc(s["touch"][:, 1])).float()
    am = s["qacc_warm"].abs().nan_to_num(1e12).amax(1).clamp(max=1e12); vm = s["qvel"].abs().nan_to_num(1e6).amax(1).clamp(max=1e6)
    return 6 * tc + 8 * torch.log1p(am) + 8 * torch.log1p(vm)
res = {}
def add(k, v): res.setdefault(k, []).append(v)
last_cost = torch.zeros(N, device=env.device)
corr = []
for t in range(int(os.environ.get("STEPS", "85"))):
    a = torch.rand(N, 69, generator=g, device=env.device) * 2 - 1
    s = snap(); f = feats(s); it = s["solver_iters"].float()
    if t >= 60:
        set_order(it + f); add("shipped key (iterations + features)", timed(lambda: env.step(a)))
        true_cost = cost.clone().float(); s_after = snap()
        for name, key in (("last step's cycles", last_cost), ("last cycles / 64 + features", last_cost / 64 + f), ("last cycles / 32 + features", last_cost / 32 + f),
                          ("last cycles / 64 + iterations + features", last_cost / 64 + it + f), ("this step's own cycles (perfect)", true_cost)):
            restore(s); set_order(key); add(name, timed(lambda: env.step(a)))
        corr.append(float(np.corrcoef(last_cost.cpu().numpy(), true_cost.cpu().numpy())[0, 1]))
        restore(s_after); last_cost = true_cost
    else:
        set_order(it + f); env.step(a); last_cost = cost.clone().float()
for k, v in res.items():
    print(f"{k:44s} mean {np.mean(v):.4f} ms  (n={len(v)})")
print("kilocycles per env: mean", float(last_cost.mean()), "max", float(last_cost.max()), " correlation last -> this step", np.mean(corr).round(3))
