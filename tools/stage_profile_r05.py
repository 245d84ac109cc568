#!/usr/bin/env python3
"""In-kernel stage timing: builds a -DSS_PROFILE variant of the library on the GPU box, runs the bench
workload for a few steps and prints shader-clock ticks per stage (summed over all waves, and per mj_step)."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from smplsim_amd import _cabi, _lib

prof_so = os.environ.get("SS_PROF_LIB") or os.path.join(ROOT, "gpurun_out", "libsmplsim_hip_prof.so")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
if not os.environ.get("SS_PROF_LIB"):                        # or a prebuilt -DSS_PROFILE variant (tools/build_variant.sh prof -DSS_PROFILE)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *_lib.DEFAULT_OPT.split(), "-std=c++17", "-shared", "-fPIC", "-DSS_PROFILE", *os.environ.get("SS_EXTRA", "").split(),
                       *[os.path.join(_lib.SRC_DIR, f) for f in ("smplsim_hip.hip", "smplsim_hip_sc.hip", "smplsim_hip_im.hip", "smplsim_motion.hip", "smplsim_mlp.hip")],
                       "-o", prof_so])
_lib._LIB = _cabi.bind(C.CDLL(prof_so))
from smplsim_amd.batch import SMPLSimVecEnv

N, steps = int(os.environ.get("NENV", "4096")), int(os.environ.get("STEPS", "30"))
SELF = bool(int(os.environ.get("SELFCOL", "0")))
env = SMPLSimVecEnv(N, autoreset=True, seed=1234, self_collision=SELF)
g = torch.Generator(device=env.device); g.manual_seed(1234)
env.reset()
WARM = int(os.environ.get("WARMUP", "0"))                   # control steps before the counted ones (the counters of the warm-up are subtracted)
for _ in range(WARM):
    env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
torch.cuda.synchronize()
out0 = (C.c_ulonglong * 64)()
_lib.lib().ss_debug_prof(env.handle, out0, 40)
for _ in range(steps):
    env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
rc = _lib.lib().ss_debug_prof(env.handle, out, 40)
for i in range(64): out[i] -= out0[i]
names = ["fwd_kin", "constraints", "newton_begin", "newton_prepare", "sc:broad phase", "aba_solve", "sc:pair function calls", "newton_finish",
         "spd_prepare", "spd_finish", "integrate", "misc", "aba:up_phase1", "aba:up_last_sync", "aba:up_phase2", "aba:down", "fk:chain sums (V, Ab)",
         "fk:prologue", "fk:level_sweep", "fk:body inertia + bias force", "fk:subtree_C", "fk:velocity products + chain sum", "prep:contactK", "prep:subtree", "prep:grad",
         "selfcol:pair functions", "selfcol:factor+base solve", "selfcol:Delassus columns", "selfcol:dense solve", "selfcol:final re-solve"]
tot = sum(out[i] for i in range(12))
iters = float(env.solver_iters.float().mean().item())
res = {"rc": rc, "total_ticks": tot, "mean_newton_iters": iters, "stages": {}}
nsub = N * (steps + 0) * 15
for i, nm in enumerate(names):
    res["stages"][nm] = {"pct": 100.0 * out[i] / tot, "ticks_per_mj_step_per_wave": out[i] / nsub}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "stage_profile.json"), "w"), indent=1)
