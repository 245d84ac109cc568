#!/usr/bin/env python3
"""MuJoCo's line search against the exact one on the benchmark's state distribution (CPU; VERDICT r4 item 5 / oracle.h MJ-(V9b)):
the oracle with OM_LS_EXACT and OM_LS_MUJOCO, and the float64 instantiation of the kernel (exact search, shipped settings), replaying
the same control steps.  Prints the table of DESIGN.md 4d."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_tools as P

rows = []
for name, hum, sc, (ne, ns) in (("2: SMPL, uniform(-1,1)", "smpl_humanoid", False, (48, 30)), ("2 + body-body contacts", "smpl_humanoid", True, (32, 24)),
                                ("4: SMPL-X", "smplx_humanoid", False, (12, 16))):
    kw = {"self_collision": True} if sc else {}
    pre, A, post = P.rollout_samples_emu(ne, ns, seed=21, skip=6, humanoid=hum, **kw)
    ex = P.oracle_step(pre, A, hum, self_collision=sc)
    mj = P.oracle_step(pre, A, hum, self_collision=sc, linesearch="mujoco")
    k64 = P.emu_step(pre, A, True, humanoid=hum, **kw)
    ok = (ex["nwarn"] == 0) & (mj["nwarn"] == 0) & (k64["nwarn"] == 0)
    e, ek = P.rel_err(mj, ex)[ok], P.rel_err(k64, mj)[ok]
    ls = mj["ls_stats"][ok].sum(axis=0)
    hist = lambda x: np.percentile(x, [50, 90, 99, 100]).astype(int).tolist()
    rows.append(f"| {name} | {ok.sum()} | {e[:, 0].max():.1e} / {e[:, 1].max():.1e} | {ek[:, 0].max():.1e} / {ek[:, 1].max():.1e} | "
                f"{(ex['iters'][ok] == mj['iters'][ok]).mean():.4f} | {hist(ex['iters'][ok])} / {hist(mj['iters'][ok])} / {hist(k64['iters'][ok])} | "
                f"{ls[0] / ls[1]:.2f} | {ls[2]} of {ls[1]} |")
    print(rows[-1], flush=True)
print()
print("| config (emulator-made samples, no bad-state reset) | samples | oracle, MuJoCo's search vs exact search: max relative difference per control step (qpos / qvel) | float64 kernel (exact search) vs oracle with MuJoCo's search | identical Newton counts per control step | Newton iterations per control step p50 / p90 / p99 / max: oracle exact / oracle MuJoCo's / float64 kernel | evaluations per search | searches out of ls_iterations |")
print("|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
