"""Edge cases of the C ABI and the kernel driver, run on the wavefront emulator (CPU): ragged batch sizes,
empty / full reset masks, non-finite actions, error paths, state round trips, episode bookkeeping."""
import ctypes as C

import numpy as np
import pytest

from helpers import FEET, default_qpos, model_const, oracle_model, pd_tables
from oracle import oracle as O
from smplsim_amd import _cabi
from wave_emu import emu


def _batch(n, **kw):
    mc = model_const()
    return emu.EmuBatch(mc, pd_tables(mc), n, legal_bodies=FEET, **kw)


@pytest.mark.parametrize("n", [1, 5, 9])
def test_ragged_batch_sizes_are_env_independent(n):
    """Every env of a batch evolves exactly as it would alone (no cross-env coupling, any N)."""
    rs = np.random.default_rng(n)
    acts = rs.uniform(-0.4, 0.4, (3, n, 69))
    eb = _batch(n)
    eb.reset()
    for a in acts:
        eb.step(a)
    solo = _batch(1)
    solo.reset()
    for a in acts:
        solo.step(a[n - 1:n])
    assert np.array_equal(eb.qpos[n - 1], solo.qpos[0]) and np.array_equal(eb.obs[n - 1], solo.obs[0])


def test_reset_masks_empty_and_full():
    eb = _batch(3)
    eb.reset()
    eb.step(np.full((3, 69), 0.2))
    q = eb.qpos.copy(); t = eb.cur_t.copy()
    eb.reset(mask=[0, 0, 0])
    assert np.array_equal(eb.qpos, q) and np.array_equal(eb.cur_t, t)
    eb.reset(mask=[1, 1, 1])
    assert np.allclose(eb.qpos[:, 2], 0.94) and (eb.cur_t == 0).all()
    assert np.array_equal(eb.qpos, eb.qpos_prev) and np.array_equal(eb.qvel, eb.qvel_prev)


def test_non_finite_action_triggers_mujoco_style_autoreset():
    eb = _batch(2)
    eb.reset()
    a = np.zeros((2, 69)); a[0, 3] = np.nan
    obs, rew, term, trunc = eb.step(a)
    assert eb.nwarn[0] >= 1 and eb.nwarn[1] == 0
    assert np.isfinite(eb.qpos).all() and np.isfinite(eb.qvel).all() and np.isfinite(obs[1]).all()


def test_error_paths_return_status_and_message():
    L = emu.lib()
    mc = model_const()
    desc, keep = _cabi.make_model_desc(mc, *pd_tables(mc))
    model = C.c_void_p()
    assert L.ss_model_create(C.byref(desc), 0, C.byref(model)) == 0
    cfg = _cabi.make_env_cfg()
    st = _cabi.State(4)                                        # every buffer NULL
    batch = C.c_void_p()
    assert L.ss_batch_create(model, C.byref(cfg), C.byref(st), C.byref(batch)) == -1
    assert b"ss_state buffer" in L.ss_last_error()
    eb = _batch(1, state_init=_cabi.INIT_FALL)
    with pytest.raises(RuntimeError, match="fall_actions"):
        eb.reset()                                             # StateInit.Fall without its random draws
    bad = _cabi.make_env_cfg(self_obs_v=3)
    assert L.ss_batch_create(model, C.byref(bad), C.byref(st), C.byref(batch)) == -1
    # a model whose bodies are not in depth-first order is rejected by the table builder
    par = mc.body_parent.copy(); par[5] = 9                    # R_Hip under Torso, which comes later
    desc2, keep2 = _cabi.make_model_desc(mc, *pd_tables(mc))
    arr = np.ascontiguousarray(par, np.int32); desc2.body_parent = arr.ctypes.data_as(C.c_void_p)
    m2 = C.c_void_p()
    assert L.ss_model_create(C.byref(desc2), 0, C.byref(m2)) == -1 and L.ss_last_error()
    L.ss_model_destroy(model)


def test_state_round_trip_replays_bit_identically():
    """get/set of the full state (incl. the stale-forward source and the warm start) reproduces a run exactly."""
    rs = np.random.default_rng(8)
    acts = rs.uniform(-0.5, 0.5, (4, 1, 69))
    a = _batch(1); a.reset()
    a.step(acts[0]); a.step(acts[1])
    snap = [x.copy() for x in (a.qpos, a.qvel, a.qpos_prev, a.qvel_prev, a.qacc_warm, a.cur_t, a.task)]
    a.step(acts[2]); a.step(acts[3])
    b = _batch(1)
    b.qpos[:], b.qvel[:], b.qpos_prev[:], b.qvel_prev[:], b.qacc_warm[:], b.cur_t[:], b.task[:] = snap
    b.step(acts[2]); b.step(acts[3])
    assert np.array_equal(a.qpos, b.qpos) and np.array_equal(a.qvel, b.qvel) and np.array_equal(a.obs, b.obs)


def test_speed_task_termination_and_truncation_flags_match_oracle():
    """Let the humanoid collapse (zero actions): illegal floor contacts must terminate the speed task at the
    same control step as the oracle; episode_length truncation uses the strict `>` of the reference."""
    om = oracle_model()
    eb = _batch(1, task=_cabi.TASK_SPEED, episode_length=40)
    env = O.OracleEnv(om, task=O.TASK_SPEED, episode_length=40)
    tr = [0.3, 0.6, 0.0, 0.0]
    env.reset(task_rand=tr); eb.reset(task_rand=[tr])
    rs = np.random.default_rng(0)
    first_term = None
    for i in range(45):
        a = rs.uniform(-0.3, 0.3, 69)                          # the rollout of test_kernel_emu: on the floor by step ~30
        eb.set_state(env.data.qpos, env.data.qvel, eb.qpos_prev, eb.qvel_prev)
        o_ref, r, te, tu = env.step(a, task_rand=tr)
        o, r2, te2, tu2 = eb.step(a[None], task_rand=[tr])
        assert (te, tu) == (bool(te2[0]), bool(tu2[0])), i
        assert abs(r - r2[0]) < 1e-4
        if te and first_term is None:
            first_term = i
        assert tu == (i + 1 > 40)
    assert first_term is not None and first_term > 5


@pytest.mark.parametrize("task", [0, 1, 3])
def test_fused_autoreset_equals_step_plus_masked_reset(task):
    """ss_step_autoreset (one launch) against ss_step followed by a masked ss_reset: bit-identical state, observations and
    flags, through speed-task terminations (falls) and a short episode_length that truncates."""
    kw = dict(task=task, episode_length=5, tar_dist_max=1.0) if task else dict(task=task, episode_length=5)
    a_env, b_env = _batch(3, **kw), _batch(3, **kw)
    rs = np.random.default_rng(11 + task)
    tr0 = rs.uniform(size=(3, 4))
    a_env.reset(task_rand=tr0); b_env.reset(task_rand=tr0)
    ended = 0
    for t in range(14):
        act = rs.uniform(-1, 1, (3, 69)) * (1.0 if t % 2 else 0.3)
        tr, tr2 = rs.uniform(size=(3, 4)), rs.uniform(size=(3, 4))
        obs_a, rew_a, te_a, tu_a = a_env.step(act, task_rand=tr)
        done = te_a | tu_a
        next_a = obs_a.copy()
        if done.any():
            next_a = a_env.reset(mask=done.astype(np.uint8), task_rand=tr2)
        obs_b, next_b, rew_b, te_b, tu_b = b_env.step_autoreset(act, task_rand=tr, reset_task_rand=tr2)
        ended += int(done.sum())
        assert np.array_equal(te_a, te_b) and np.array_equal(tu_a, tu_b) and np.array_equal(rew_a, rew_b)
        assert np.array_equal(obs_a, obs_b) and np.array_equal(next_a, next_b)
        for f in ("qpos", "qvel", "qpos_prev", "qvel_prev", "qacc_warm", "cur_t", "task", "touch", "body_vel"):
            assert np.array_equal(getattr(a_env, f), getattr(b_env, f)), (t, f)
    assert ended >= 4


def test_fused_fall_autoreset_equals_step_plus_masked_fall_reset():
    """StateInit.Fall (config 3, getup): the 45 warm-up mj_steps of the reset inside the step launch (ss_set_fall_actions +
    ss_step_autoreset) against ss_step + masked ss_reset with the same draws: bit-identical."""
    import ctypes as C
    kw = dict(task=2, state_init=1, episode_length=4, recovery_steps=1)
    a_env, b_env = _batch(3, **kw), _batch(3, **kw)
    rs = np.random.default_rng(5)
    fa0, tr0 = rs.uniform(size=(3, 3, 69)), rs.uniform(size=(3, 4))
    assert np.array_equal(a_env.reset(fall_actions=fa0, task_rand=tr0), b_env.reset(fall_actions=fa0, task_rand=tr0))
    fall = np.zeros((3, 3, 69), np.float32)
    b_env._chk(emu.lib().ss_set_fall_actions(b_env.batch, fall.ctypes.data_as(C.c_void_p)))
    ended = 0
    for t in range(10):
        act = rs.uniform(-1, 1, (3, 69))
        tr, tr2, fa = rs.uniform(size=(3, 4)), rs.uniform(size=(3, 4)), rs.uniform(size=(3, 3, 69)).astype(np.float32)
        obs_a, rew_a, te_a, tu_a = a_env.step(act, task_rand=tr)
        done = te_a | tu_a
        next_a = obs_a.copy()
        if done.any():
            next_a = a_env.reset(mask=done.astype(np.uint8), fall_actions=fa, task_rand=tr2)
        fall[:] = fa
        obs_b, next_b, rew_b, te_b, tu_b = b_env.step_autoreset(act, task_rand=tr, reset_task_rand=tr2)
        ended += int(done.sum())
        assert np.array_equal(te_a, te_b) and np.array_equal(tu_a, tu_b) and np.array_equal(rew_a, rew_b)
        assert np.array_equal(obs_a, obs_b) and np.array_equal(next_a, next_b), t
        for f in ("qpos", "qvel", "qpos_prev", "qvel_prev", "qacc_warm", "cur_t", "task", "touch", "body_vel"):
            assert np.array_equal(getattr(a_env, f), getattr(b_env, f)), (t, f)
    assert ended >= 3


def test_device_side_longest_first_schedule_is_a_sorted_permutation_and_changes_nothing():
    eb, ref = _batch(5), _batch(5)
    rs = np.random.default_rng(21)
    eb.reset(); ref.reset()
    for t in range(3):
        act = rs.uniform(-1, 1, (5, 69))
        eb._chk(emu.lib().ss_schedule_longest_first(eb.batch, None))
        eb.step(act); ref.step(act)
        assert np.array_equal(eb.qpos, ref.qpos) and np.array_equal(eb.obs, ref.obs)      # the hand-out order is only a hint
    assert emu.lib().ss_schedule_longest_first(None, None) != 0


def test_step_autoreset_error_paths():
    import ctypes as C
    eb = _batch(2, state_init=1)                              # StateInit.Fall: the in-launch reset is refused
    a = np.zeros((2, 69), np.float32); obs2 = np.zeros_like(eb.obs)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    rc = emu.lib().ss_step_autoreset(eb.batch, p(a), None, None, p(eb.obs), p(obs2), p(eb.reward), p(eb.terminated), p(eb.truncated), None)
    assert rc != 0 and b"ss_set_fall_actions" in emu.lib().ss_last_error()       # Fall: needs its draws first
    ex = _batch(2, state_init=2)
    rc = emu.lib().ss_step_autoreset(ex.batch, p(a), None, None, p(ex.obs), p(obs2), p(ex.reward), p(ex.terminated), p(ex.truncated), None)
    assert rc != 0 and b"External" in emu.lib().ss_last_error()
    ok = _batch(2)
    rc = emu.lib().ss_step_autoreset(ok.batch, p(a), None, None, p(ok.obs), None, p(ok.reward), p(ok.terminated), p(ok.truncated), None)
    assert rc != 0                                            # obs_next is mandatory


def test_per_env_body_shapes_equal_single_shape_batches_and_track_their_oracles():
    """ss_model_create_shapes: 3 body shapes, 7 envs with mixed shape ids, one launch.  Every env must be bit-identical to
    the same env in a single-shape batch of its shape (the shape tables are the only thing that changes), and follow the
    oracle compiled from ITS MJCF."""
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import scaled_xml_str
    xmls = [scaled_xml_str("smpl_humanoid", 1.0), scaled_xml_str("smpl_humanoid", 0.9, {"L_Knee": 1.1, "R_Knee": 1.1}),
            scaled_xml_str("smpl_humanoid", 1.08, {"Chest": 0.9, "L_Elbow": 1.2})]
    mcs = [compile_mjcf(x) for x in xmls]
    tabs = pd_tables(mcs[0])
    sid = np.array([0, 1, 2, 2, 1, 0, 1], np.int32)
    n = len(sid)
    eb = emu.EmuBatch(mcs[0], tabs, n, legal_bodies=FEET, shape_mcs=mcs, shape_id=sid, task=_cabi.TASK_SPEED, episode_length=3)
    rs = np.random.default_rng(8)
    tr = rs.uniform(size=(n, 4))
    obs0 = eb.reset(task_rand=tr)
    solos = []
    for s in range(3):
        idx = np.nonzero(sid == s)[0]
        so = emu.EmuBatch(mcs[s], tabs, len(idx), legal_bodies=FEET, task=_cabi.TASK_SPEED, episode_length=3)
        assert np.array_equal(so.reset(task_rand=tr[idx]), obs0[idx])
        solos.append((idx, so))
    oenvs = []
    for i in range(n):
        om = O.OracleModel(xmls[sid[i]], *tabs, legal_bodies=FEET, timestep=1.0 / 450)
        oe = O.OracleEnv(om, task=O.TASK_SPEED, episode_length=3)
        assert np.abs(oe.reset(task_rand=tr[i]) - obs0[i]).max() < 1e-5
        oenvs.append(oe)
    assert np.abs(obs0[0] - obs0[1]).max() > 1e-3                       # the shapes really differ
    for k in range(3):
        act = rs.uniform(-0.4, 0.4, (n, 69))
        tr = rs.uniform(size=(n, 4))
        obs, rew, term, trunc = eb.step(act, task_rand=tr)
        for idx, so in solos:
            o2, r2, t2, u2 = so.step(act[idx], task_rand=tr[idx])
            assert np.array_equal(o2, obs[idx]) and np.array_equal(so.qpos, eb.qpos[idx]) and np.array_equal(r2, rew[idx])
        for i, oe in enumerate(oenvs):
            o_ref, r, te, tu = oe.step(act[i], task_rand=tr[i])
            assert np.abs(oe.data.qpos - eb.qpos[i]).max() < 2e-4 and np.abs(o_ref - obs[i]).max() < 5e-3, (k, i)
            assert (te, tu) == (bool(term[i]), bool(trunc[i]))
    # the shape of an env may change between launches (a new body at reset): swap two envs' shapes and reset them
    sid2 = sid.copy(); sid2[0], sid2[1] = sid[1], sid[0]
    eb.shape_id[:] = sid2
    m = np.zeros(n, np.uint8); m[:2] = 1
    ob = eb.reset(mask=m, task_rand=tr)
    assert np.array_equal(ob[0, 1:70], obs0[1, 1:70]) and np.array_equal(ob[1, 1:70], obs0[0, 1:70])   # body positions of the other shape


def test_shaped_model_error_paths():
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import scaled_xml_str
    mc0, mc1 = compile_mjcf(scaled_xml_str("smpl_humanoid", 1.0)), compile_mjcf(scaled_xml_str("smpl_humanoid", 0.9))
    tabs = pd_tables(mc0)
    with pytest.raises(RuntimeError, match="shape_id"):
        emu.EmuBatch(mc0, tabs, 2, legal_bodies=FEET, shape_mcs=[mc0, mc1], shape_id=None)
    mcx = model_const("smplx_humanoid")
    descs = (_cabi.ModelDesc * 2)()
    descs[0], k0 = _cabi.make_model_desc(mc0, *tabs, legal_bodies=FEET)
    descs[1], k1 = _cabi.make_model_desc(mcx, *pd_tables(mcx), legal_bodies=FEET)
    h = C.c_void_p()
    assert emu.lib().ss_model_create_shapes(descs, 2, 0, C.byref(h)) == -1
    assert b"differs from shape 0" in emu.lib().ss_last_error()
    t2 = [np.array(t, dtype=np.float64).copy() for t in tabs]
    t2[0][3] *= 2.0                                                     # a different gain is not a body shape
    descs[1], k2 = _cabi.make_model_desc(mc1, *t2, legal_bodies=FEET)
    assert emu.lib().ss_model_create_shapes(descs, 2, 0, C.byref(h)) == -1


@pytest.mark.parametrize("humanoid", ["smpl_humanoid", "smplx_humanoid"])
def test_fixed_layout_instantiation_is_bit_identical_to_the_generic_one(humanoid, monkeypatch):
    """The instantiations with compile-time dimensions / LDS layout (HdrFixedT<24,5>, <52,10>, ss_hdr.h) against the generic kernel
    (the emulator's launcher mirrors the GPU's choice; SS_EMU_GENERIC forces the generic one): same bits after contact-rich steps."""
    from helpers import FEET, model_const, pd_tables
    mc = model_const(humanoid)
    rs = np.random.default_rng(4)
    acts = rs.uniform(-1, 1, (3, 4, mc.nu))
    outs = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("SS_EMU_GENERIC", "1")
        eb = emu.EmuBatch(mc, pd_tables(mc), 4, legal_bodies=FEET, task=1)
        eb.reset(task_rand=np.full((4, 4), 0.5))
        q = eb.qpos.copy(); q[:, 2] = 0.5                     # dropped onto the floor: contacts, limits, many Newton iterations
        eb.set_state(q, eb.qvel.copy())
        for a in acts:
            obs, rew, term, trunc = eb.step(a, np.full((4, 4), 0.25))
        outs.append((eb.qpos.copy(), eb.qvel.copy(), obs, rew, eb.solver_iters.copy()))
    for x, y in zip(*outs):
        assert np.array_equal(x, y)
    assert outs[0][4].max() > 15                            # contacts were active (one Newton iteration per mj_step otherwise)


@pytest.mark.parametrize("generic", [False, True])
def test_aliased_lds_layout_is_bit_identical_to_the_plain_one(generic, monkeypatch):
    """Round 5: the SMPL-X size class keeps a body's (W, y) rows in the slot of its generalized inertia and moves the forward pass's
    scratch, the contact records and R, r into the idle An | Aown | IA stretch (ss_hdr.h make_layout(alias_w): 6 resident envs per CU
    instead of 5).  Same arithmetic, other addresses: the results must be the plain layout's (SS_NO_ALIAS_LAYOUT, read when the model is
    built) bit for bit — fixed-layout and generic instantiation, through contact-rich steps, a reset and the velocity sensors (obs v2)."""
    import ctypes as C
    from helpers import FEET, model_const, pd_tables
    mc = model_const("smplx_humanoid")
    rs = np.random.default_rng(11)
    acts = rs.uniform(-1, 1, (4, 3, mc.nu))
    if generic:
        monkeypatch.setenv("SS_EMU_GENERIC", "1")
    outs, per_wg = [], []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("SS_NO_ALIAS_LAYOUT", "1")
        eb = emu.EmuBatch(mc, pd_tables(mc), 3, legal_bodies=FEET, task=1, self_obs_v=2)
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        eb.L.ss_launch_info(eb.batch, C.byref(a), C.byref(b), C.byref(c)); per_wg.append(a.value)
        eb.reset(task_rand=np.full((3, 4), 0.5))
        q = eb.qpos.copy(); q[:, 2] = 0.45                    # dropped onto the floor: contacts, limits, many Newton iterations
        eb.set_state(q, eb.qvel.copy())
        rec = []
        for t, act in enumerate(acts):
            obs, rew, term, trunc = eb.step(act, np.full((3, 4), 0.25))
            rec += [eb.qpos.copy(), eb.qvel.copy(), obs.copy(), rew.copy(), eb.solver_iters.copy(), eb.body_vel.copy()]
            if t == 1:
                rec.append(eb.reset(mask=np.array([1, 0, 0], np.uint8), task_rand=np.full((3, 4), 0.3)))
        outs.append(rec)
    assert per_wg == [7, 5]                                   # what the layout (and the lean tables) are for
    for x, y in zip(*outs):
        assert np.array_equal(x, y)
    assert max(r.max() for r in outs[0][4::6] if r.dtype == np.int32) > 15


@pytest.mark.parametrize("f64,tol", [(True, 1e-9), (False, 2e-4)])
def test_power_usage_output_matches_the_reference_definition(f64, tol):
    """HumanoidEnv.curr_power_usage (reference humanoid_env.py:443-451): per mj_step |qfrc_actuator[6:] * qvel[6:]| with the torque
    of that mj_step and the velocity after it — the optional by-product of the step launch (ss_set_power_output) against the
    oracle doing what the reference's loop does."""
    from oracle import oracle as O
    mc = model_const()
    om = oracle_model()
    rs = np.random.default_rng(12)
    n = 3
    eb = emu.EmuBatch(mc, pd_tables(mc), n, legal_bodies=FEET, f64=f64)
    eb.reset()
    P = eb.set_power_output()
    acts = rs.uniform(-0.6, 0.6, (n, mc.nu))
    pre_q, pre_v, pre_w = eb.qpos.astype(np.float64), eb.qvel.astype(np.float64), eb.qacc_warm.astype(np.float64)
    pq, pv = eb.qpos_prev.astype(np.float64), eb.qvel_prev.astype(np.float64)
    eb.step(acts)
    assert P.shape == (n, 15, 69) and np.abs(P).max() > 1.0
    for i in range(n):
        d = O.OracleData(om)
        d.qpos = pq[i]; d.qvel = pv[i]; d.forward()
        d.qpos = pre_q[i]; d.qvel = pre_v[i]; d.warm = pre_w[i]
        for s_ in range(15):
            tau = d.spd_torque(acts[i]); d.ctrl = tau; d.step()
            ref = np.abs(tau * d.qvel[6:])
            assert np.abs(P[i, s_] - ref).max() < tol * max(1.0, np.abs(ref).max()), (i, s_)
    # off again: the buffer keeps its contents
    eb._chk(eb.L.ss_set_power_output(eb.batch, None))
    keep = P.copy(); eb.step(acts)
    assert np.array_equal(P, keep)
