"""Per-env cost of a control step with body-body contacts (profile build with -DSS_PROF_ENV: ticks in the truncation counter's array).
SS_PROF_LIB = the library; prints the heaviest envs of a few steps: Newton iterations, contacts, total and dense-part kiloticks."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smplsim_amd import _cabi, _lib
_lib._LIB = _cabi.bind_mlp(_cabi.bind(C.CDLL(os.environ["SS_PROF_LIB"])))
from smplsim_amd.batch import SMPLSimVecEnv, _check, _ptr
N = int(os.environ.get("NENV", "256"))
env = SMPLSimVecEnv(N, autoreset=True, seed=1234, self_collision=os.environ.get("SELFCOL", "1") == "1")
g = torch.Generator(device=env.device); g.manual_seed(1234)
env.reset()
rec = torch.zeros(N, dtype=torch.int32, device=env.device)
_check(_lib.lib().ss_debug_self_truncation(env.handle, _ptr(rec)))
for _ in range(int(os.environ.get("WARMUP", "30"))): env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
tot_all, den_all, it_all = [], [], []
for s in range(int(os.environ.get("STEPS", "12"))):
    a_ = torch.rand(N, 69, generator=g, device=env.device) * 2 - 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    env.step(a_, _events=(e0, e1))
    torch.cuda.synchronize()
    kms = e0.elapsed_time(e1)
    r = rec.cpu().numpy(); tot = r & 0xFFFF; den = (r >> 16) & 0x7FFF
    if os.environ.get("DENSE_PARTS"):                       # -DSS_PROF_ENV_DENSE build: tree part | contact rows | factorization | back substitution, 32 k ticks each
        parts = np.stack([r & 255, (r >> 8) & 255, (r >> 16) & 255, (r >> 24) & 127], 1) * 32.768
        top = parts.sum(1).argsort()[::-1][:4]
        it = env.solver_iters.cpu().numpy()
        print(os.environ.get("TAG", ""), "step", s, "largest dense solves (iters | tree, rows, factorization, back substitution kticks):", [(int(it[i]), parts[i].round().tolist()) for i in top], "mean parts", parts.mean(0).round(1).tolist())
        continue
    it = env.solver_iters.cpu().numpy(); nc = env.self_contacts.cpu().numpy()
    tot_all.append(tot.copy()); den_all.append(den.copy()); it_all.append(it.copy())
    top = tot.argsort()[::-1][:3]
    print(os.environ.get("TAG", ""), "step", s, "heaviest:", [(int(it[i]), int(nc[i]), int(tot[i]), int(den[i])) for i in top], "(iters, contacts at end, total kticks, dense kticks or 160-ns units)  step launch ms", round(kms, 3), " mean total", round(float(tot.mean()), 1), "mean dense", round(float(den.mean()), 1))
import numpy as np
T, D, I = np.concatenate(tot_all), np.concatenate(den_all), np.concatenate(it_all)
print(os.environ.get("TAG", ""), "all steps: mean total", T.mean().round(1), "dense", D.mean().round(1), "| max-env mean total", np.mean([t.max() for t in tot_all]).round(1),
      "| kticks per Newton iteration (fit):", np.polyfit(I, T, 1).round(2))
