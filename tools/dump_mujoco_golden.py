#!/usr/bin/env python3
"""Produce REAL MuJoCo golden vectors for the path, stage by stage (needs a `mujoco` wheel — there is none in the build
container and no network, so the file this writes is not in the repo yet and the physics stays "parity unpinned",
DESIGN.md §5; the moment a wheel exists:

    python tools/dump_mujoco_golden.py tests/golden/mujoco_vectors.npz && python -m pytest tests/test_oracle_vs_mujoco.py

Both packaged fixtures (SMPL: 24 bodies; SMPL-X/H layout: 52) with opt.timestep = 1/450 (reference
smpl_sim/envs/base_env.py:139-142), in two collision settings: "full" = the MJCF as it is (bodies collide with each other:
contype / conaffinity / excludes, smpl_humanoid.xml:5,24,231-242) and "floor" = body geoms' conaffinity cleared, so that only
the floor pairs remain (the scope of BASELINE.json's north_star).  Per seeded state, every stage of mj_forward that the oracle
restates separately:
    model constants   body_mass, body_inertia, body_ipos, body_iquat, body_invweight0, dof_invweight0, jnt_range
    kinematics        xpos, xquat, xipos
    inertia / bias    qM (dense, mj_fullM), qfrc_bias
    collision         ncon, contact geom1/geom2/pos/frame/dist (MuJoCo's order)
    constraint rows   nefc, efc_J (dense), efc_pos, efc_margin, efc_D, efc_R, efc_aref, efc_diagApprox
    solve             qacc_smooth, qacc (cold start), efc_force, qfrc_constraint, solver iterations
    step              qpos / qvel after one mj_step with zero ctrl, and after 15 x (Stable-PD torque restated from reference
                      controllers.py:116-190 on the stale qM / qfrc_bias, then mj_step) for a seeded action
"""
import sys

import numpy as np


def dump(mujoco, xml, name, floor_only, rec, n_cases, seed):
    from scipy.linalg import cho_factor, cho_solve
    from smplsim_amd.gains import build_pd_tables
    from smplsim_amd.mjcf import compile_mjcf
    mc = compile_mjcf(xml)
    rng = {n: mc.jnt_range[6 + i] for i, n in enumerate(mc.joint_names)}
    kp, kd, lim, sc, of = build_pd_tables(mc.actuator_names, lambda n: rng[n])
    model = mujoco.MjModel.from_xml_string(xml)
    model.opt.timestep = 1.0 / 450
    if floor_only:
        model.geom_conaffinity[1:] = 0                        # geom 0 is the floor: (floor, body) pairs still pass through body.contype
    data = mujoco.MjData(model)
    rs = np.random.default_rng(seed)
    nv, nu = model.nv, model.nu
    pre = f"{name}_{'floor' if floor_only else 'full'}_"
    rec[pre + "model"] = dict(body_mass=model.body_mass[1:].copy(), body_inertia=model.body_inertia[1:].copy(), body_ipos=model.body_ipos[1:].copy(),
                              body_iquat=model.body_iquat[1:].copy(), body_invweight0=model.body_invweight0[1:].copy(),
                              dof_invweight0=model.dof_invweight0.copy(), jnt_range=model.jnt_range.copy())
    keys = ("qpos qvel ctrl xpos xquat xipos qM qfrc_bias ncon con_geom1 con_geom2 con_pos con_frame con_dist nefc efc_J efc_pos efc_margin "
            "efc_D efc_R efc_aref efc_diagApprox qacc_smooth qacc efc_force qfrc_constraint solver_niter step_qpos step_qvel "
            "roll_action roll_qpos roll_qvel").split()
    out = {k: [] for k in keys}
    MAXC, MAXE = 200, 1000
    for case in range(n_cases):
        q = np.zeros(model.nq); q[2] = [0.94, 0.93, 0.3, 0.2, 0.15, 2.0][case % 6]
        quat = rs.normal(size=4) if case % 3 else np.array([.5, .5, .5, .5]); q[3:7] = quat / np.linalg.norm(quat)
        q[7:] = rs.uniform(-1.2 if case % 4 == 3 else -0.8, 1.2 if case % 4 == 3 else 0.8, nv - 6)      # every 4th: folded up (body-body contacts)
        v = rs.normal(size=nv) * (0.5 if case % 2 else 3.0)
        u = rs.normal(size=nu) * 20
        mujoco.mj_resetData(model, data)
        data.qpos[:], data.qvel[:], data.ctrl[:] = q, v, u
        mujoco.mj_forward(model, data)
        M = np.zeros((nv, nv)); mujoco.mj_fullM(model, M, data.qM)
        nc, ne = data.ncon, data.nefc
        J = np.zeros((MAXE, nv))
        J[:ne] = data.efc_J.reshape(ne, nv) if data.efc_J.size == ne * nv else 0.0     # dense Jacobian builds only
        padc = lambda x: np.pad(np.asarray(x, np.float64), ((0, MAXC - nc),) + ((0, 0),) * (np.ndim(x) - 1))
        pade = lambda x: np.pad(np.asarray(x, np.float64), (0, MAXE - ne))
        vals = dict(qpos=q, qvel=v, ctrl=u, xpos=data.xpos[1:].copy(), xquat=data.xquat[1:].copy(), xipos=data.xipos[1:].copy(), qM=M,
                    qfrc_bias=data.qfrc_bias.copy(), ncon=nc, con_geom1=padc(data.contact.geom1), con_geom2=padc(data.contact.geom2),
                    con_pos=padc(data.contact.pos), con_frame=padc(data.contact.frame), con_dist=padc(data.contact.dist), nefc=ne, efc_J=J,
                    efc_pos=pade(data.efc_pos), efc_margin=pade(data.efc_margin), efc_D=pade(data.efc_D), efc_R=pade(data.efc_R),
                    efc_aref=pade(data.efc_aref), efc_diagApprox=pade(data.efc_diagApprox), qacc_smooth=data.qacc_smooth.copy(),
                    qacc=data.qacc.copy(), efc_force=pade(data.efc_force), qfrc_constraint=data.qfrc_constraint.copy(),
                    solver_niter=int(np.sum(data.solver_niter)))
        data.ctrl[:] = 0
        mujoco.mj_step(model, data)
        vals.update(step_qpos=data.qpos.copy(), step_qvel=data.qvel.copy())
        # one control step of the reference loop from the same state: 15 x (SPD on the stale qM / qfrc_bias, then mj_step)
        mujoco.mj_resetData(model, data)
        data.qpos[:], data.qvel[:] = q, v * 0.2
        mujoco.mj_forward(model, data)
        a = rs.uniform(-0.5, 0.5, nu)
        for _ in range(15):
            mujoco.mj_fullM(model, M, data.qM)
            kpv, kdv = np.zeros(nv), np.zeros(nv); kpv[6:], kdv[6:] = kp, kd
            perr = np.concatenate([np.zeros(6), data.qpos[7:] + data.qvel[6:] * model.opt.timestep - (a * sc + of)])
            acc = cho_solve(cho_factor(M + np.diag(kdv) * model.opt.timestep), -data.qfrc_bias - kpv * perr - kdv * data.qvel)
            data.ctrl[:] = np.clip(-kp * perr[6:] - kd * (data.qvel[6:] + acc[6:] * model.opt.timestep), -lim, lim)
            mujoco.mj_step(model, data)
        vals.update(roll_action=a, roll_qpos=data.qpos.copy(), roll_qvel=data.qvel.copy())
        for k in keys:
            out[k].append(vals[k])
    for k in keys:
        rec[pre + k] = np.asarray(out[k])
    for k, v in rec.pop(pre + "model").items():
        rec[pre + "model_" + k] = v


def main(out):
    import mujoco  # noqa: F401  (the whole point of this tool)
    sys.path.insert(0, ".")
    from smplsim_amd.mjcf_writer import default_xml_str
    rec = {}
    for name, n in (("smpl_humanoid", 24), ("smplx_humanoid", 8)):
        for floor_only in (True, False):
            dump(mujoco, default_xml_str(name), name, floor_only, rec, n, 20240925)
    np.savez_compressed(out, mujoco_version=mujoco.__version__, **rec)
    print("wrote", out, "mujoco", mujoco.__version__)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/mujoco_vectors.npz")
