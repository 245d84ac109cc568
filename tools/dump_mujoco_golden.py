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
and, since round 4:
    stat              mjModel.stat.meaninertia (scale of the solver's termination test)
    pair functions    mjc_CapsuleCapsule / mjc_CapsuleBox / mjc_BoxBox on the SAME random geometry the oracle's pair functions are
                      tested on (tests/test_selfcol_emu.py::test_pair_functions_of_the_kernel_match_the_oracle): two free bodies
                      with one geom each, mj_forward, the contact list (count, positions, normals, distances)
    rollout           BASELINE config 2's workload itself: 64 envs x 1000 control steps of the reference's loop (Stable PD +
                      15 mj_steps, uniform(-1,1) actions, reset(Default) every 300 steps), both collision settings — MuJoCo's rate of
                      bad-state autoresets (mjWARN_BADQPOS / BADQVEL / BADQACC), the histogram of Newton iterations per mj_step
                      and of simultaneous contacts: what bench.py reports as bad_state_resets_total / newton_iters_p50_p99_max
                      for the kernel (3.9 % of the envs per control step reset; stragglers at 105-125 iterations)
and, since round 5 (oracle.h MJ-(V9b)):
    solve iterations  the `solver_niter` of the cold-started solve recorded above is compared with the oracle's count under MuJoCo's
                      line search restated (OM_LS_MUJOCO: bracketing + 1-D Newton, ls_tolerance 0.01, ls_iterations 50) and under the
                      exact search the HIP kernel uses: tests/test_oracle_vs_mujoco.py::test_stage_solve_iterations
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


PAIR_SEED, PAIR_TRIALS = 4, 600     # = tests/test_selfcol_emu.py::test_pair_functions_of_the_kernel_match_the_oracle


def pair_geometry(seed=PAIR_SEED, trials=PAIR_TRIALS):
    """The random touching geometry of the pair-function test: (kind, g1, g2) with capsule = (centre, axis, radius, half length),
    box = (centre, rotation with the box axes as columns, half sizes)."""
    from scipy.spatial.transform import Rotation as sRot
    rs = np.random.default_rng(seed)
    for trial in range(trials):
        kind = ("cc", "cb", "bb")[trial % 3]

        def geom(box, centre):
            Rm = sRot.random(random_state=rs.integers(1 << 30)).as_matrix()
            if box:
                return (centre, Rm, rs.uniform(0.03, 0.15, 3))
            return (centre, Rm[:, 2], rs.uniform(0.03, 0.08), rs.uniform(0.05, 0.2))
        g1 = geom(kind == "bb", np.zeros(3))
        g2 = geom(kind != "cc", rs.normal(size=3) * 0.12)
        yield trial, kind, g1, g2


def dump_pairs(mujoco, rec, margin=0.001):
    """MuJoCo's own pair functions on that geometry.  geom order = the order MuJoCo hands to the pair function (capsule before box)."""
    from scipy.spatial.transform import Rotation as sRot

    def quat_wxyz(Rm):
        q = sRot.from_matrix(Rm).as_quat()
        return q[[3, 0, 1, 2]]

    def frame_of_axis(a):                                     # a rotation whose z axis is the capsule axis
        z = a / np.linalg.norm(a)
        x = np.cross([0.0, 1.0, 0.0] if abs(z[1]) < 0.9 else [1.0, 0.0, 0.0], z); x /= np.linalg.norm(x)
        return np.stack([x, np.cross(z, x), z], 1)

    def body(name, g, box):
        if box:
            c, Rm, half = g
            q = quat_wxyz(Rm)
            geom = f'<geom type="box" size="{half[0]} {half[1]} {half[2]}"/>'
        else:
            c, a, r, h = g
            q = quat_wxyz(frame_of_axis(a))
            geom = f'<geom type="capsule" size="{r} {h}"/>'
        return f'<body name="{name}" pos="{c[0]} {c[1]} {c[2]}" quat="{q[0]} {q[1]} {q[2]} {q[3]}"><freejoint/>{geom}</body>'

    out = dict(trial=[], kind=[], ncon=[], pos=[], normal=[], dist=[])
    for trial, kind, g1, g2 in pair_geometry():
        xml = (f'<mujoco><option gravity="0 0 0"/><default><geom margin="{margin}" condim="3"/></default><worldbody>'
               f'{body("a", g1, kind == "bb")}{body("b", g2, kind != "cc")}</worldbody></mujoco>')
        m = mujoco.MjModel.from_xml_string(xml)
        d = mujoco.MjData(m)
        mujoco.mj_forward(m, d)
        n = d.ncon
        pos, nrm, dist = np.zeros((8, 3)), np.zeros((8, 3)), np.zeros(8)
        for i in range(min(n, 8)):
            c = d.contact[i]
            sign = 1.0 if c.geom1 == 0 else -1.0                   # normal from the FIRST geom of our convention (geom 0) to the second
            pos[i], nrm[i], dist[i] = c.pos, sign * np.asarray(c.frame[:3]), c.dist
        out["trial"].append(trial); out["kind"].append(("cc", "cb", "bb").index(kind)); out["ncon"].append(n)
        out["pos"].append(pos); out["normal"].append(nrm); out["dist"].append(dist)
    for k, v in out.items():
        rec["pairs_" + k] = np.asarray(v)
    rec["pairs_margin"] = margin


def dump_rollout(mujoco, xml, name, floor_only, rec, n_envs=64, n_steps=1000, seed=20240926):
    """BASELINE config 2 on MuJoCo itself: the reference's control step (StablePDController + 15 mj_steps, humanoid_env.py:439-453) under
    uniform(-1,1) actions; reset_sim(Default) every 300 steps like the base task's truncation."""
    from scipy.linalg import cho_factor, cho_solve
    from smplsim_amd.gains import build_pd_tables
    from smplsim_amd.mjcf import compile_mjcf
    mc = compile_mjcf(xml)
    rng = {n: mc.jnt_range[6 + i] for i, n in enumerate(mc.joint_names)}
    kp, kd, lim, sc, of = build_pd_tables(mc.actuator_names, lambda n: rng[n])
    model = mujoco.MjModel.from_xml_string(xml)
    model.opt.timestep = 1.0 / 450
    if floor_only:
        model.geom_conaffinity[1:] = 0
    nv = model.nv
    rs = np.random.default_rng(seed)
    W = mujoco.mjtWarning
    bad = (W.mjWARN_BADQPOS, W.mjWARN_BADQVEL, W.mjWARN_BADQACC)
    it_hist, con_hist, self_hist = np.zeros(128, np.int64), np.zeros(256, np.int64), np.zeros(256, np.int64)
    resets, env_steps, steps_with_reset, it_per_step = 0, 0, 0, []
    M = np.zeros((nv, nv))
    kpv, kdv = np.zeros(nv), np.zeros(nv); kpv[6:], kdv[6:] = kp, kd
    for e in range(n_envs):
        data = mujoco.MjData(model)

        def reset():
            mujoco.mj_resetData(model, data)
            data.qpos[:] = 0; data.qpos[2] = 0.94; data.qpos[3:7] = 0.5
            mujoco.mj_forward(model, data)
        reset()
        for t in range(n_steps):
            if t % 300 == 0 and t:
                reset()
            a = rs.uniform(-1, 1, model.nu)
            w0 = sum(int(data.warning[w].number) for w in bad)
            its = 0
            for _ in range(15):
                mujoco.mj_fullM(model, M, data.qM)
                perr = np.concatenate([np.zeros(6), data.qpos[7:] + data.qvel[6:] * model.opt.timestep - (a * sc + of)])
                acc = cho_solve(cho_factor(M + np.diag(kdv) * model.opt.timestep), -data.qfrc_bias - kpv * perr - kdv * data.qvel)
                data.ctrl[:] = np.clip(-kp * perr[6:] - kd * (data.qvel[6:] + acc[6:] * model.opt.timestep), -lim, lim)
                mujoco.mj_step(model, data)
                ni = int(np.sum(data.solver_niter))
                its += ni
                it_hist[min(ni, 127)] += 1
                con_hist[min(data.ncon, 255)] += 1
                self_hist[min(int(np.sum(np.asarray(data.contact.geom1[:data.ncon]) != 0)), 255)] += 1
            w1 = sum(int(data.warning[w].number) for w in bad)
            resets += w1 - w0; steps_with_reset += w1 > w0; env_steps += 1
            it_per_step.append(its)
    pre = f"rollout_{name}_{'floor' if floor_only else 'full'}_"
    rec[pre + "env_steps"] = env_steps
    rec[pre + "bad_state_resets"] = resets
    rec[pre + "env_steps_with_reset_frac"] = steps_with_reset / env_steps
    rec[pre + "newton_iters_per_mj_step_hist"] = it_hist
    rec[pre + "newton_iters_per_control_step"] = np.asarray(it_per_step, np.int32)
    rec[pre + "contacts_per_mj_step_hist"] = con_hist
    rec[pre + "body_body_contacts_per_mj_step_hist"] = self_hist


def main(out):
    import mujoco  # noqa: F401  (the whole point of this tool)
    sys.path.insert(0, ".")
    from smplsim_amd.mjcf_writer import default_xml_str
    rec = {}
    for name, n in (("smpl_humanoid", 24), ("smplx_humanoid", 8)):
        for floor_only in (True, False):
            dump(mujoco, default_xml_str(name), name, floor_only, rec, n, 20240925)
        rec[name + "_stat_meaninertia"] = float(mujoco.MjModel.from_xml_string(default_xml_str(name)).stat.meaninertia)
    dump_pairs(mujoco, rec)
    for floor_only in (True, False):
        dump_rollout(mujoco, default_xml_str("smpl_humanoid"), "smpl_humanoid", floor_only, rec)
    np.savez_compressed(out, mujoco_version=mujoco.__version__, **rec)
    print("wrote", out, "mujoco", mujoco.__version__)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/mujoco_vectors.npz")
