#!/usr/bin/env python3
"""Produce REAL MuJoCo golden vectors for the path (needs `pip install mujoco` — not available in the build
container, so the vectors this writes are not in the repo yet; the physics is "parity unpinned").

    python tools/dump_mujoco_golden.py tests/golden/mujoco_vectors.npz

For a seeded set of states it records what mujoco.mj_forward / mj_step produce on the packaged SMPL fixture with
opt.timestep = 1/450 (reference smpl_sim/envs/base_env.py:139-142): qM (dense), qfrc_bias, xpos, xquat, contact
list, efc rows, qacc, and 15-substep Stable-PD rollouts restated from reference controllers.py:116-190.
tests/test_oracle_vs_mujoco.py (skipped when the file is absent) then checks the oracle against it.
"""
import sys

import numpy as np


def main(out):
    import mujoco  # noqa: F401  (the whole point of this tool)
    from scipy.linalg import cho_factor, cho_solve
    sys.path.insert(0, ".")
    from smplsim_amd.gains import build_pd_tables
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import default_xml_str
    xml = default_xml_str()
    mc = compile_mjcf(xml)
    rng = {n: mc.jnt_range[6 + i] for i, n in enumerate(mc.joint_names)}
    kp, kd, lim, sc, of = build_pd_tables(mc.actuator_names, lambda n: rng[n])
    model = mujoco.MjModel.from_xml_string(xml)
    model.opt.timestep = 1.0 / 450
    data = mujoco.MjData(model)
    rs = np.random.default_rng(20240925)
    rec = dict(mass=model.body_mass[1:].copy(), inertia=model.body_inertia[1:].copy(),
               body_invweight0=model.body_invweight0[1:].copy(), dof_invweight0=model.dof_invweight0.copy(),
               qpos=[], qvel=[], ctrl=[], qM=[], bias=[], xpos=[], xquat=[], qacc=[], ncon=[], con_pos=[], con_dist=[],
               con_geom2=[], efc_force=[], roll_action=[], roll_qpos=[], roll_qvel=[])
    nv = model.nv
    for case in range(24):
        q = np.zeros(model.nq); q[2] = [0.94, 0.93, 0.3, 0.2, 0.15, 2.0][case % 6]
        quat = rs.normal(size=4) if case % 3 else np.array([.5, .5, .5, .5]); q[3:7] = quat / np.linalg.norm(quat)
        q[7:] = rs.uniform(-0.8, 0.8, nv - 6)
        v = rs.normal(size=nv) * (0.5 if case % 2 else 3.0)
        u = rs.normal(size=model.nu) * 20
        data.qpos[:], data.qvel[:], data.ctrl[:] = q, v, u
        data.qacc_warmstart[:] = 0
        mujoco.mj_forward(model, data)
        M = np.zeros((nv, nv)); mujoco.mj_fullM(model, M, data.qM)
        for k, val in (("qpos", q), ("qvel", v), ("ctrl", u), ("qM", M), ("bias", data.qfrc_bias.copy()),
                       ("xpos", data.xpos[1:].copy()), ("xquat", data.xquat[1:].copy()), ("qacc", data.qacc.copy()),
                       ("ncon", data.ncon), ("efc_force", np.pad(data.efc_force, (0, 600 - data.nefc))),
                       ("con_pos", np.pad(data.contact.pos, ((0, 100 - data.ncon), (0, 0)))),
                       ("con_dist", np.pad(data.contact.dist, (0, 100 - data.ncon))),
                       ("con_geom2", np.pad(data.contact.geom2, (0, 100 - data.ncon)))):
            rec[k].append(val)
        # one control step of the reference loop: 15 x (SPD on the stale qM / qfrc_bias, then mj_step)
        a = rs.uniform(-0.5, 0.5, model.nu)
        for _ in range(15):
            mujoco.mj_fullM(model, M, data.qM)
            kpv, kdv = np.zeros(nv), np.zeros(nv); kpv[6:], kdv[6:] = kp, kd
            perr = np.concatenate([np.zeros(6), data.qpos[7:] + data.qvel[6:] * model.opt.timestep - (a * sc + of)])
            acc = cho_solve(cho_factor(M + np.diag(kdv) * model.opt.timestep), -data.qfrc_bias - kpv * perr - kdv * data.qvel)
            tau = np.clip(-kp * perr[6:] - kd * (data.qvel[6:] + acc[6:] * model.opt.timestep), -lim, lim)
            data.ctrl[:] = tau
            mujoco.mj_step(model, data)
        rec["roll_action"].append(a); rec["roll_qpos"].append(data.qpos.copy()); rec["roll_qvel"].append(data.qvel.copy())
    np.savez_compressed(out, **{k: np.asarray(v) for k, v in rec.items()}, mujoco_version=mujoco.__version__)
    print("wrote", out, "mujoco", mujoco.__version__)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/mujoco_vectors.npz")
