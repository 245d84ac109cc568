#!/usr/bin/env python3
"""True MuJoCo CPU baseline for the same workload (needs a `mujoco` wheel; not installable in the build container).

Runs the reference's own step semantics — 15 x (Stable-PD via SciPy Cholesky on mj_fullM + mujoco.mj_step) per control
step on the packaged SMPL fixture (reference humanoid_env.py:439-453, controllers.py:116-190) — single env in-process,
uniform(-1,1) actions, and prints env-steps/s plus the split between mj_step and the SPD solve.
"""
import sys
import time

import numpy as np


def main(steps=300):
    import mujoco
    from scipy.linalg import cho_factor, cho_solve
    sys.path.insert(0, ".")
    from smplsim_amd.gains import build_pd_tables
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import default_xml_str
    xml = default_xml_str()
    mc = compile_mjcf(xml)
    rng = {n: mc.jnt_range[6 + i] for i, n in enumerate(mc.joint_names)}
    kp, kd, lim, sc, of = build_pd_tables(mc.actuator_names, lambda n: rng[n])
    model = mujoco.MjModel.from_xml_string(xml); model.opt.timestep = 1.0 / 450
    data = mujoco.MjData(model)
    data.qpos[2] = 0.94; data.qpos[3:7] = 0.5
    mujoco.mj_forward(model, data)
    nv, dt = model.nv, model.opt.timestep
    M = np.zeros((nv, nv)); kpv, kdv = np.zeros(nv), np.zeros(nv); kpv[6:], kdv[6:] = kp, kd
    rs = np.random.default_rng(1234)
    t_step = t_spd = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        a = rs.uniform(-1, 1, model.nu)
        for _ in range(15):
            t1 = time.perf_counter()
            mujoco.mj_fullM(model, M, data.qM)
            perr = np.concatenate([np.zeros(6), data.qpos[7:] + data.qvel[6:] * dt - (a * sc + of)])
            acc = cho_solve(cho_factor(M + np.diag(kdv) * dt), -data.qfrc_bias - kpv * perr - kdv * data.qvel)
            data.ctrl[:] = np.clip(-kp * perr[6:] - kd * (data.qvel[6:] + acc[6:] * dt), -lim, lim)
            t2 = time.perf_counter()
            mujoco.mj_step(model, data)
            t3 = time.perf_counter()
            t_spd += t2 - t1; t_step += t3 - t2
        if i % 300 == 299:
            mujoco.mj_resetData(model, data); data.qpos[2] = 0.94; data.qpos[3:7] = 0.5; mujoco.mj_forward(model, data)
    el = time.perf_counter() - t0
    print({"mujoco": mujoco.__version__, "env_steps_per_s_1core": steps / el, "mj_step_us": 1e6 * t_step / (15 * steps),
           "spd_us": 1e6 * t_spd / (15 * steps), "autoresets": int(data.warning[mujoco.mjtWarning.mjWARN_BADQACC].number)})


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
