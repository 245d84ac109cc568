#!/usr/bin/env python3
"""Write a file with the SCHEMA of tools/dump_mujoco_golden.py's output, filled by the CPU oracle instead of MuJoCo — a plumbing check
of tests/test_oracle_vs_mujoco.py (every key it reads exists, every test body executes), used by
tests/test_oracle_golden.py::test_mujoco_pin_is_ready_to_fire.  It pins nothing: the oracle trivially agrees with itself.  The real file
needs a `mujoco` wheel (README.md "Pinning the physics")."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main(out, n_cases=4, rollout=(8, 150)):
    from dump_mujoco_golden import pair_geometry
    from helpers import oracle_model
    from oracle import oracle as O
    rec = {}
    for h in ("smpl_humanoid", "smplx_humanoid"):
        for c in ("floor", "full"):
            om = oracle_model(h, self_collision=(c == "full"))
            nv, nu = om.nv, om.nu
            pre = f"{h}_{c}_"
            rs = np.random.default_rng(20240925)
            keys = ("qpos qvel ctrl xpos xquat xipos qM qfrc_bias ncon con_geom1 con_geom2 con_pos con_dist nefc qacc_smooth qacc efc_force "
                    "qfrc_constraint solver_niter step_qpos step_qvel roll_action roll_qpos roll_qvel").split()
            acc = {k: [] for k in keys}
            d = O.OracleData(om)
            for case in range(n_cases):
                q = np.zeros(om.nq); q[2] = [0.94, 0.93, 0.3, 0.2][case % 4]
                quat = rs.normal(size=4) if case % 3 else np.array([.5, .5, .5, .5]); q[3:7] = quat / np.linalg.norm(quat)
                q[7:] = rs.uniform(-0.8, 0.8, nv - 6)
                v, u = rs.normal(size=nv) * 0.5, rs.normal(size=nu) * 20
                d.qpos = q; d.qvel = v; d.ctrl = u; d.warm = np.zeros(nv); d.forward()
                nc, ne = d.ncon, int(d.get(O.D_NEFC)[0])
                padc = lambda x: np.pad(np.asarray(x, np.float64), ((0, 200 - nc),) + ((0, 0),) * (np.ndim(x) - 1))
                vals = dict(qpos=q, qvel=v, ctrl=u, xpos=d.xpos, xquat=d.xquat, xipos=d.xipos, qM=d.M, qfrc_bias=d.bias, ncon=nc,
                            con_geom1=padc(d.con_body1 + 1), con_geom2=padc(d.con_body + 1), con_pos=padc(d.con_pos), con_dist=padc(d.con_dist),
                            nefc=ne, qacc_smooth=d.get(O.D_QACC_SMOOTH), qacc=d.qacc, efc_force=np.pad(d.get(O.D_EFC_FORCE), (0, 1000 - ne)),
                            qfrc_constraint=d.get(O.D_QFRC_CONSTRAINT), solver_niter=d.solver_iter)
                d.ctrl = np.zeros(nu); d.step()
                vals.update(step_qpos=d.qpos, step_qvel=d.qvel)
                d.qpos = q; d.qvel = v * 0.2; d.warm = np.zeros(nv); d.ctrl = np.zeros(nu); d.forward()
                a = rs.uniform(-0.5, 0.5, nu)
                for _ in range(15):
                    d.ctrl = d.spd_torque(a); d.step()
                vals.update(roll_action=a, roll_qpos=d.qpos, roll_qvel=d.qvel)
                for k in keys:
                    acc[k].append(vals[k])
            for k in keys:
                rec[pre + k] = np.asarray(acc[k])
            if c == "floor":
                rng_ = om.get(O.M_RANGE).reshape(-1, 2)
                rec[pre + "model_body_mass"] = om.get(O.M_MASS); rec[pre + "model_body_inertia"] = om.get(O.M_INERTIA).reshape(-1, 3)
                rec[pre + "model_body_ipos"] = om.get(O.M_IPOS).reshape(-1, 3); rec[pre + "model_body_invweight0"] = om.get(O.M_BODY_INVW).reshape(-1, 2)
                rec[pre + "model_dof_invweight0"] = om.get(O.M_DOF_INVW)
                rec[pre + "model_jnt_range"] = np.concatenate([np.zeros((1, 2)), rng_[6:]])
        rec[h + "_stat_meaninertia"] = float(oracle_model(h).get(O.M_MEANINERTIA)[0])
    margin = 0.001
    pr = dict(trial=[], kind=[], ncon=[], pos=[], normal=[], dist=[])
    for trial, kind, g1, g2 in pair_geometry():
        cs = O.narrow_phase(kind, g1, g2, margin)
        pos, nrm, dist = np.zeros((8, 3)), np.zeros((8, 3)), np.zeros(8)
        for i, (p, n, dd) in enumerate(cs):
            pos[i], nrm[i], dist[i] = p, n, dd
        pr["trial"].append(trial); pr["kind"].append(("cc", "cb", "bb").index(kind)); pr["ncon"].append(len(cs))
        pr["pos"].append(pos); pr["normal"].append(nrm); pr["dist"].append(dist)
    for k, v in pr.items():
        rec["pairs_" + k] = np.asarray(v)
    rec["pairs_margin"] = margin
    for c in ("floor", "full"):
        om = oracle_model(self_collision=(c == "full"))
        rs = np.random.default_rng(20240926)
        resets = steps = 0
        its = []
        for e in range(rollout[0]):
            env = O.OracleEnv(om); env.reset()
            for t in range(rollout[1]):
                nw0 = env.data.nwarn
                _, _, te, tu = env.step(rs.uniform(-1, 1, om.nu))
                resets += env.data.nwarn > nw0; steps += 1; its.append(15 * env.data.solver_iter)
                if te or tu:
                    env.reset()
        pre = f"rollout_smpl_humanoid_{c}_"
        rec[pre + "env_steps"] = steps; rec[pre + "bad_state_resets"] = resets; rec[pre + "env_steps_with_reset_frac"] = resets / steps
        rec[pre + "newton_iters_per_control_step"] = np.asarray(its, np.int32)
    np.savez_compressed(out, mujoco_version="oracle twin (schema check only)", **rec)
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1])
