"""Analytic / physical checks of the oracle's restated mj_step (the part of the path that has no
reference golden vectors — MuJoCo is absent): mass matrix, bias forces, momentum and
energy behaviour, contact generation, constraint optimality (KKT) and the env-level bookkeeping."""
import numpy as np
import pytest

from helpers import default_qpos, model_const, oracle_model
from oracle import oracle as O


def _rand_state(nq, rs, vel=3.0, z=50.0):
    q = np.zeros(nq)
    q[:3] = rs.normal(size=3); q[2] = z
    q[3:7] = rs.normal(size=4); q[3:7] /= np.linalg.norm(q[3:7])
    q[7:] = rs.uniform(-1.2, 1.2, nq - 7)
    return q, rs.normal(size=nq - 1) * vel


def test_default_pose_known_answers():
    om = oracle_model()
    d = O.OracleData(om)
    d.qpos = default_qpos(76); d.forward()
    assert d.ncon == 0
    # lowest foot-box face at +5.1 mm: box centre z - half height along world z
    # hand sums of the MJCF offsets: left foot boxes bottom at 0.94-0.9349, right ankle box at 0.94-0.9367
    names = model_const().body_names
    assert abs(_box_lowest(d, names.index("L_Ankle")) - 0.0051) < 1e-9
    assert abs(_box_lowest(d, names.index("L_Toe")) - 0.0051) < 1e-9
    assert abs(_box_lowest(d, names.index("R_Ankle")) - 0.0033) < 1e-9
    assert np.allclose(d.qacc[:3], [0, 0, -9.81], atol=1e-9)          # free fall
    assert np.abs(d.qacc[3:]).max() < 1e-6


def _mats(q):
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([np.stack([1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], -1),
                     np.stack([2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)], -1),
                     np.stack([2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)], -1)], -2)


def _box_lowest(d, b):
    mc = model_const()
    R = _mats(d.xquat)[b]
    c = d.xpos[b] + R @ mc.geom_pos[b]
    corners = [(R @ ((2 * np.array(s) - 1) * mc.geom_size[b]))[2] for s in np.ndindex(2, 2, 2)]
    return c[2] + min(corners)


def test_mass_matrix_is_kinetic_energy_hessian():
    """q'^T M q' must equal sum_b m|v_c|^2 + w^T I w computed from finite-differenced kinematics."""
    om, mc = oracle_model(), model_const()
    rs = np.random.default_rng(3)
    d, d2 = O.OracleData(om), O.OracleData(om)
    q, v = _rand_state(76, rs)
    d.qpos = q; d.qvel = v; d.forward()
    M = d.M
    assert np.abs(M - M.T).max() == 0 and np.linalg.eigvalsh(M).min() > 0.009
    # integrate positions by a tiny dt with the oracle's own integrator semantics (root quat: local omega)
    eps = 1e-6
    q2 = q.copy(); q2[:3] += eps * v[:3]; q2[7:] += eps * v[6:]
    w = v[3:6]; ang = eps * np.linalg.norm(w); ax = w / np.linalg.norm(w)
    dq = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
    a = q[3:7]
    q2[3:7] = [a[0]*dq[0]-a[1]*dq[1]-a[2]*dq[2]-a[3]*dq[3], a[0]*dq[1]+a[1]*dq[0]+a[2]*dq[3]-a[3]*dq[2],
               a[0]*dq[2]-a[1]*dq[3]+a[2]*dq[0]+a[3]*dq[1], a[0]*dq[3]+a[1]*dq[2]-a[2]*dq[1]+a[3]*dq[0]]
    d2.qpos = q2; d2.kinematics()
    vc = (d2.xipos - d.xipos) / eps
    ke = 0.5 * (mc.body_mass * (vc ** 2).sum(1)).sum()
    R = _mats(d.xquat)
    Ri = _mats(mc.body_iquat)
    for b in range(24):
        Rw = R[b] @ Ri[b]
        ke += 0.5 * d.angvel[b] @ (Rw @ np.diag(mc.body_inertia[b]) @ Rw.T) @ d.angvel[b]
    ke += 0.5 * (mc.dof_armature * v ** 2).sum()
    assert abs(0.5 * v @ M @ v - ke) / ke < 1e-5


def test_free_flight_energy_drift_is_first_order_and_momentum_conserved():
    mc = model_const()
    drift, perr = [], []
    for scale, n in ((1.0, 90), (0.5, 180)):
        om = oracle_model(timestep=scale / 450)
        d = O.OracleData(om)
        q, v = _rand_state(76, np.random.default_rng(1))
        d.qpos = q; d.qvel = v; d.forward()
        e0 = d.get(O.D_ENERGY).sum()
        p0 = (mc.body_mass[:, None] * _com_vel(d)).sum(0)
        for _ in range(n):
            d.step()
        d.forward()
        drift.append(d.get(O.D_ENERGY).sum() - e0)
        p1 = (mc.body_mass[:, None] * _com_vel(d)).sum(0)
        g_impulse = np.array([0, 0, -9.81 * mc.total_mass * n * scale / 450])
        perr.append(np.linalg.norm(p1 - p0 - g_impulse))
    # explicit Euler in generalized coordinates: both errors are O(dt) globally and halve with dt
    assert abs(drift[0]) < 5e-4 * 35000 and abs(drift[1] / drift[0] - 0.5) < 0.05
    assert perr[0] < 0.01 * 400 and abs(perr[1] / perr[0] - 0.5) < 0.1


def _com_vel(d):
    mc = model_const()
    rc = d.xipos - d.xpos
    return d.linvel + np.cross(d.angvel, rc)


def test_contacts_and_kkt_conditions():
    """Drop the humanoid onto the floor; at every step the solver's answer must satisfy the KKT
    conditions of MuJoCo's convex problem: M(a - a_s) = J^T f, f = -D*min(0, J a - aref) >= 0."""
    om = oracle_model()
    d = O.OracleData(om)
    q = default_qpos(76); q[2] = 0.93
    d.qpos = q
    seen = 0
    for i in range(200):
        d.step()
        ncon = d.ncon
        if ncon:
            seen += 1
            M, a, a_s = d.M, d.qacc, d.get(O.D_QACC_SMOOTH)
            fc = d.get(O.D_QFRC_CONSTRAINT)
            res = M @ (a - a_s) - fc
            assert np.abs(res).max() < 1e-7 * (1 + np.abs(fc).max())
            f = d.get(O.D_EFC_FORCE)
            assert (f >= 0).all() and len(f) == int(d.get(O.D_NEFC)[0])
            # all contact points within margin of the plane, on feet only while standing
            assert (d.get(O.D_CON_DIST) <= 0.001 + 1e-12).all()
    assert seen > 100
    # standing on its feet after 200 substeps of zero torque? it collapses, but the floor must hold it
    assert d.qpos[2] > 0.05 and np.isfinite(d.qpos).all()
    # resting contact force ~ weight once (nearly) static is checked on a single box-like settle:
    total_normal = 0.0
    for _ in range(600):
        d.step()
    fc = d.get(O.D_QFRC_CONSTRAINT)
    assert fc[2] > 0                                       # net upward constraint force on the root z dof


def test_plane_box_contact_rule_first_four_corners():
    om, mc = oracle_model(), model_const()
    d = O.OracleData(om)
    q = default_qpos(76); q[2] = 0.9349                    # feet just touching (within margin)
    d.qpos = q; d.forward()
    bodies = d.get(O.D_CON_BODY).astype(int)
    names = [mc.body_names[b] for b in bodies]
    assert set(names) <= {"L_Ankle", "R_Ankle", "L_Toe", "R_Toe"} and d.ncon >= 4
    fr = d.get(O.D_CON_FRAME).reshape(-1, 9)
    assert np.allclose(fr[:, :3], [0, 0, 1]) and np.allclose(fr[:, 3:6], [0, 1, 0]) and np.allclose(fr[:, 6:], [-1, 0, 0])
    pos, dist = d.get(O.D_CON_POS).reshape(-1, 3), d.get(O.D_CON_DIST)
    assert np.allclose(pos[:, 2], dist / 2, atol=1e-12)    # contact point at half penetration


def test_joint_limit_rows_activate():
    om = oracle_model()
    d = O.OracleData(om)
    q = default_qpos(76); q[2] = 5.0; q[7 + 20] = np.pi + 0.05
    d.qpos = q; d.forward()
    assert int(d.get(O.D_NEFC)[0]) == 1 and d.get(O.D_EFC_FORCE)[0] > 0
    assert d.qacc[6 + 20] < -1.0                            # pushed back inside the range


def test_autoreset_on_bad_state():
    om = oracle_model()
    d = O.OracleData(om)
    q = default_qpos(76); d.qpos = q
    v = np.zeros(75); v[10] = 1e11; d.qvel = v
    d.step()
    assert d.get(O.D_SOLVER_ITER)[1] == 1
    assert np.abs(d.qvel).max() < 100.0 and np.allclose(d.qpos[3:7], [1, 0, 0, 0], atol=1e-2)


def test_env_layer_bookkeeping():
    om = oracle_model()
    env = O.OracleEnv(om, task=O.TASK_GETUP, state_init=O.INIT_FALL)
    rs = np.random.default_rng(0)
    obs = env.reset(fall_actions=rs.uniform(size=(3, 69)), task_rand=[0.25, 0.5])
    assert obs.shape == (290,) and np.isfinite(obs).all()
    t = env.get_task()
    assert t[0] == 0 and np.isclose(t[1], 0.5 + 0.7 * 0.25) and t[2] == 0 + 100 + 50 and t[3] == 60
    assert obs[-1] == np.float32(t[1]) and env.data.qpos[2] < 0.5
    for i in range(61):
        obs, r, term, trunc = env.step(rs.uniform(-0.2, 0.2, 69), task_rand=[0.5, 0.5])
        if i < 60:
            assert not term and not trunc                    # recovery grace (humanoid_getup.py:60-72)
    assert term                                              # lying on the floor: illegal contacts
    assert 0 < r <= 1
    base = O.OracleEnv(om)
    base.reset()
    for i in range(302):
        obs, r, term, trunc = base.step(np.zeros(69))
        assert r == 0 and not term and trunc == (i + 1 > 300)
