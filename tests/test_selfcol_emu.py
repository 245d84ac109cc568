"""Body-body contacts in the kernel (SELFCOL instantiation, ss_env_cfg.self_collision) on the wavefront
emulator against the float64 oracle: contact counts, the constrained acceleration of states with arms folded into the
torso / legs crossed / lying on the floor (float64 build: 1e-10 — pair functions, contact building and the solve (articulated-body
recursion outside the coupled set, dense block system inside it) are the oracle's problem exactly; float32 build: 2e-5),
teacher-forced control steps, crowded states (every contact is kept, like MuJoCo: the oracle is UNCAPPED) and the benchmark
distribution per sample.  GPU twins: test_gpu_parity.py."""
import numpy as np
import pytest

import parity_tools as P
from helpers import FEET, default_qpos, model_const, oracle_model, pd_tables
from oracle import oracle as O
from wave_emu import emu


def _states(n, seed, min_self=1, humanoid="smpl_humanoid"):
    """Random large joint angles in the air or just above the floor, kept when the oracle reports body-body contacts."""
    om = oracle_model(humanoid, self_collision=True)
    mc = model_const(humanoid)
    rs = np.random.default_rng(seed)
    d = O.OracleData(om)
    Q, V, T = [], [], []
    while len(Q) < n:
        q = default_qpos(mc.nq); q[2] = rs.choice([5.0, 0.35]); q[7:] = rs.uniform(-1.5, 1.5, mc.nq - 7)
        v, t = rs.normal(size=mc.nv) * 0.5, rs.normal(size=mc.nu) * 5
        d.qpos = q; d.qvel = v; d.ctrl = t; d.forward()
        if d.nself >= min_self:
            Q.append(q); V.append(v); T.append(t)
    return om, np.array(Q), np.array(V), np.array(T)


# float32: 3e-5 of the largest acceleration.  Over 40 such states the error is 1.5e-6 in the median, 3.9e-6 at the 90th percentile and 6e-5 at
# most — the same for the round 5 solver (3 x 3 blocks) and the round 6 one (scalar tiles on the matrix core): the tail is the states'
# conditioning, the summation order only moves it (sample 8 of these ten: 1.97e-5 before, 2.01e-5 now)
@pytest.mark.parametrize("f64,tol", [(True, 1e-10), (False, 3e-5)])
def test_constrained_acceleration_with_body_body_contacts(f64, tol):
    mc = model_const()
    om, Q, V, T = _states(10, 2)
    eb = emu.EmuBatch(mc, pd_tables(mc), len(Q), legal_bodies=FEET, f64=f64, self_collision=True)
    eb.set_state(Q, V)
    M, bias, qacc = eb.debug_forward(T)
    d = O.OracleData(om)
    seen = set()
    for i in range(len(Q)):
        d.qpos = Q[i]; d.qvel = V[i]; d.ctrl = T[i]; d.warm = np.zeros(75); d.forward()
        assert eb.self_contacts[i] == d.nself
        assert np.abs(qacc[i] - d.qacc).max() < tol * np.abs(d.qacc).max(), (i, d.ncon, d.nself)
        seen.add((d.nself > 0, d.ncon > d.nself))
    assert (True, True) in seen and (True, False) in seen      # with and without simultaneous floor contacts
    # without the flag the same states take the floor-only path: the acceleration differs
    eb0 = emu.EmuBatch(mc, pd_tables(mc), len(Q), legal_bodies=FEET, f64=f64)
    eb0.set_state(Q, V)
    assert np.abs(eb0.debug_forward(T)[2] - qacc).max() > 1e-2 and (eb0.self_contacts == 0).all()


def _kernel_narrow_phase(kind, g1, g2, margin, f64):
    """The kernel's pair functions (ss_selfcol.h) on raw geometry through the emulator's test hook."""
    import ctypes as C
    L = emu.lib(f64)
    flat = lambda g: np.concatenate([np.ravel(np.asarray(x, dtype=np.float64)) for x in g])
    inp = np.ascontiguousarray(np.concatenate([flat(g1), flat(g2), [margin], np.zeros(4)]))
    out = np.zeros(1 + 7 * 8)
    n = L.ss_emu_narrow_phase({"cc": 0, "cb": 1, "bb": 2}[kind], inp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return [(out[1 + 7 * i:4 + 7 * i].copy(), out[4 + 7 * i:7 + 7 * i].copy(), float(out[7 + 7 * i])) for i in range(n)]


@pytest.mark.parametrize("f64,tol", [(True, 1e-10), (False, 3e-5)])
def test_pair_functions_of_the_kernel_match_the_oracle(f64, tol):
    """capsule-capsule / capsule-box / box-box of ss_selfcol.h against the oracle's twins on random touching geometry, plus the
    crossed-boxes known answer (face manifold = the overlap rectangle, no vertex of either box inside the other)."""
    from scipy.spatial.transform import Rotation as sRot
    rs = np.random.default_rng(4)
    seen = {"cc": 0, "cb": 0, "bb": 0}
    multi = 0
    for trial in range(600):
        kind = ("cc", "cb", "bb")[trial % 3]
        def geom(box, centre):
            Rm = sRot.random(random_state=rs.integers(1 << 30)).as_matrix()
            if box:
                return (centre, Rm, rs.uniform(0.03, 0.15, 3))
            return (centre, Rm[:, 2], rs.uniform(0.03, 0.08), rs.uniform(0.05, 0.2))
        g1 = geom(kind == "bb", np.zeros(3))
        g2 = geom(kind != "cc", rs.normal(size=3) * 0.12)
        margin = 0.001
        ref = O.narrow_phase(kind, g1, g2, margin)
        if not ref or min(abs(c[2] - margin) for c in ref) < 1e-4:     # borderline at the margin: float32 may decide otherwise
            continue
        got = _kernel_narrow_phase(kind, g1, g2, margin, f64)
        assert len(got) == len(ref), (kind, trial, len(got), len(ref))
        for (p, n, d), (p2, n2, d2) in zip(sorted(ref, key=lambda c: tuple(np.round(c[0], 4))), sorted(got, key=lambda c: tuple(np.round(c[0], 4)))):
            assert np.abs(p - p2).max() < tol and np.abs(n - n2).max() < tol and abs(d - d2) < tol, (kind, trial)
        seen[kind] += 1
        multi += len(ref) > 1
    assert min(seen.values()) >= 15 and multi >= 10, (seen, multi)
    lower, I3 = ([0, 0, 0], np.eye(3), [0.3, 0.05, 0.05]), np.eye(3)
    for yaw in (0.0, 0.3):
        Rz = sRot.from_euler("z", yaw).as_matrix()
        got = _kernel_narrow_phase("bb", lower, ([0, 0, 0.09], Rz, [0.05, 0.3, 0.05]), 0.001, f64)
        assert len(got) == 4 and all(abs(c[2] + 0.01) < tol and np.abs(c[1] - [0, 0, 1]).max() < tol for c in got)


def test_crowded_states_keep_every_contact():
    """MuJoCo keeps every contact of the narrow phase; so does the kernel (rounds 2-3 kept the deepest 8): states with 9 .. 40
    simultaneous body-body contacts follow the UNCAPPED oracle, the count is the oracle's and no mj_step is flagged as truncated."""
    mc = model_const()
    om_all = oracle_model(self_collision=True)
    rs = np.random.default_rng(9)
    da = O.OracleData(om_all)
    found = []
    for _ in range(6000):
        q = default_qpos(76); q[2] = 5.0; q[7:] = rs.uniform(-2.2, 2.2, 69)
        da.qpos = q; da.qvel = np.zeros(75); da.ctrl = np.zeros(69); da.forward()
        if da.nself > 8 + 4 * len(found):
            found.append(q)
            if len(found) == 5:
                break
    assert len(found) >= 3
    Q = np.array(found)
    eb = emu.EmuBatch(mc, pd_tables(mc), len(Q), legal_bodies=FEET, f64=True, self_collision=True)
    eb.set_state(Q, np.zeros((len(Q), 75)))
    qacc = eb.debug_forward(np.zeros((len(Q), 69)))[2]
    most = 0
    for i, q in enumerate(Q):
        da.qpos = q; da.qvel = np.zeros(75); da.ctrl = np.zeros(69); da.warm = np.zeros(75); da.forward()
        assert eb.self_contacts[i] == da.nself
        assert np.abs(qacc[i] - da.qacc).max() < 1e-9 * np.abs(da.qacc).max(), (i, da.nself)
        most = max(most, da.nself)
    assert most >= 16
    # the truncation counter (ss_debug_self_truncation) stays at zero over whole control steps from these states
    import ctypes as C
    eb2 = emu.EmuBatch(mc, pd_tables(mc), 2, legal_bodies=FEET, f64=True, self_collision=True)
    eb2.set_state(np.stack([Q[-1], default_qpos(76)]), np.zeros((2, 75)))
    cnt = np.zeros(2, np.int32)
    eb2._chk(eb2.L.ss_debug_self_truncation(eb2.batch, cnt.ctypes.data_as(C.c_void_p)))
    eb2.cur_t[:] = 0
    eb2.step(np.zeros((2, 69)))
    assert cnt[0] == 0 and cnt[1] == 0


def test_teacher_forced_control_steps_with_self_collision():
    """Actions that fold the arms through the torso and cross the legs (uniform(-1,1) targets up to +-pi): 10 control steps
    of the float32 kernel, each from the oracle's state, against the oracle with the same contact set."""
    mc = model_const()
    om = oracle_model(self_collision=True)
    oenv = O.OracleEnv(om)
    eb = emu.EmuBatch(mc, pd_tables(mc), 1, legal_bodies=FEET, self_collision=True)
    assert np.abs(oenv.reset() - eb.reset()[0]).max() < 1e-6
    rs = np.random.default_rng(5)
    worst, with_self = np.zeros(3), 0
    for i in range(10):
        eb.set_state(oenv.data.qpos[None], oenv.data.qvel[None], eb.qpos_prev, eb.qvel_prev)
        a = rs.uniform(-1, 1, 69)
        o_ref, r, te, tu = oenv.step(a)
        obs, rew, term, trunc = eb.step(a[None])
        with_self += oenv.data.nself > 0
        assert eb.self_contacts[0] == oenv.data.nself
        scale = max(1.0, np.abs(oenv.data.qvel).max())
        worst = np.maximum(worst, [np.abs(eb.qpos[0] - oenv.data.qpos).max() / scale, np.abs(eb.qvel[0] - oenv.data.qvel).max() / scale,
                                   np.abs(obs[0] - o_ref).max() / scale])
    assert with_self >= 5
    assert worst[0] < 2e-5 and worst[1] < 2e-3 and worst[2] < 2e-3, worst


def test_benchmark_distribution_per_sample_with_self_collision():
    """The per-sample triage of test_parity_f64.py with body-body contacts on (69% of the benchmark's control steps end with
    at least one): formulation, resets, precision and cap gap under the same bounds."""
    from test_parity_f64 import _check
    pre, A, post = P.rollout_samples_emu(12, 16, seed=3, skip=6, self_collision=True)
    r = P.triage(pre, A, post, n_perturb=4, self_collision=True)
    _check(r, "smpl uniform(-1,1), self-collision", 90)
    assert (r["nself"] > 0).mean() > 0.3


def test_smplx_with_self_collision():
    """52 bodies, 1265 candidate pairs (20 broad-phase rounds of 64 lanes)."""
    mc = model_const("smplx_humanoid")
    om, Q, V, T = _states(3, 4, humanoid="smplx_humanoid")
    eb = emu.EmuBatch(mc, pd_tables(mc), len(Q), legal_bodies=FEET, f64=True, self_collision=True)
    eb.set_state(Q, V)
    qacc = eb.debug_forward(T)[2]
    d = O.OracleData(om)
    for i in range(len(Q)):
        d.qpos = Q[i]; d.qvel = V[i]; d.ctrl = T[i]; d.warm = np.zeros(mc.nv); d.forward()
        assert eb.self_contacts[i] == d.nself
        assert np.abs(qacc[i] - d.qacc).max() < 1e-9 * np.abs(d.qacc).max()


def test_self_collision_with_per_env_body_shapes():
    """Body-body contacts + ss_model_create_shapes (the reference's has_shape_variation humanoids collide with themselves too): the
    pair functions read the geom table of the env's own shape.  Every env of the mixed batch equals the same env in a single-shape
    self-collision batch of its shape, bit for bit, and the thick-limbed shape collides where the thin one does not."""
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import scaled_xml_str
    xmls = [scaled_xml_str("smpl_humanoid", 1.0), scaled_xml_str("smpl_humanoid", 0.9, {"L_Knee": 1.1, "R_Knee": 1.1}),
            scaled_xml_str("smpl_humanoid", 1.1, {"Chest": 0.9, "L_Elbow": 1.2})]
    mcs = [compile_mjcf(x) for x in xmls]
    tabs = pd_tables(mcs[0])
    sid = np.array([0, 1, 2, 2, 1, 0], np.int32)
    n = len(sid)
    eb = emu.EmuBatch(mcs[0], tabs, n, legal_bodies=FEET, shape_mcs=mcs, shape_id=sid, self_collision=True)
    solos = [(np.nonzero(sid == s)[0], emu.EmuBatch(mcs[s], tabs, int((sid == s).sum()), legal_bodies=FEET, self_collision=True)) for s in range(3)]
    obs0 = eb.reset()
    for idx, so in solos:
        assert np.array_equal(so.reset(), obs0[idx])
    rs = np.random.default_rng(4)
    seen = 0
    for k in range(6):
        act = np.repeat(rs.uniform(-1, 1, (1, 69)), n, axis=0)               # the same violent action for every shape
        obs, rew, term, trunc = eb.step(act)
        for idx, so in solos:
            o2, r2, t2, u2 = so.step(act[idx])
            assert np.array_equal(o2, obs[idx]) and np.array_equal(so.qpos, eb.qpos[idx]) and np.array_equal(so.self_contacts, eb.self_contacts[idx]), k
        seen += int(eb.self_contacts.sum())
    assert seen > 0
    om = oracle_model(self_collision=True)            # shape 0 is the packaged fixture: its oracle applies
    oenv = O.OracleEnv(om)
    oenv.reset()
    e0 = emu.EmuBatch(mcs[0], tabs, 2, legal_bodies=FEET, shape_mcs=mcs, shape_id=np.array([0, 2], np.int32), self_collision=True)
    e0.reset()
    for k in range(4):
        a = rs.uniform(-1, 1, 69)
        e0.set_state(np.stack([oenv.data.qpos, e0.qpos[1]]), np.stack([oenv.data.qvel, e0.qvel[1]]), e0.qpos_prev, e0.qvel_prev)
        oenv.step(a); e0.step(np.stack([a, a]))
        assert e0.self_contacts[0] == oenv.data.nself and np.abs(e0.qpos[0] - oenv.data.qpos).max() < 2e-5 * max(1.0, np.abs(oenv.data.qvel).max())


def test_smplx_crowded_contact_states_follow_the_oracle():
    """SMPL-X with body-body contacts on its benchmark distribution (52 bodies, 1265 candidate pairs: the finger capsules make
    states with dozens of simultaneous contacts and coupled sets of 20-40 bodies): the float64 kernel against the UNCAPPED oracle,
    both converged, per sample.  (Rounds 2-3 kept the deepest 8 contacts on both sides.)"""
    pre, A, post = P.rollout_samples_emu(24, 14, seed=7, skip=4, humanoid="smplx_humanoid", self_collision=True)
    kw = dict(humanoid="smplx_humanoid", task_state=pre.get("task"), cur_t=pre.get("cur_t"), task_rand=pre.get("task_rand"), self_collision=True)
    orc = P.oracle_step(pre, A, "smplx_humanoid", self_collision=True, solver="converged")
    f64 = P.emu_step(pre, A, True, solver_tolerance=1e-30, **kw)
    ok = (orc["nwarn"] == 0) & (f64["nwarn"] == 0)
    e = P.rel_err(f64, orc)
    assert ok.sum() >= 200 and (f64["nself"][ok] > 8).sum() >= 40          # crowded states are in the sample
    # (all but a handful of samples agree to 1e-12; the largest moves between 1e-10 and 6e-9 with the order in which a node's children
    # are added — rounding, amplified by 15 mj_steps of a state with dozens of contacts)
    assert e[ok].max() < 2e-8 and np.quantile(e[ok].max(1), 0.98) < 1e-10, (np.flatnonzero(ok & (e.max(1) > 1e-9)), e[ok].max(axis=0), np.quantile(e[ok].max(1), [0.5, 0.9, 0.98]))
