"""Kernel logic (float32, the exact source the GPU runs) against the float64 oracle, executed on the
64-fiber wavefront emulator — the only way to exercise the kernel in the GPU-less build container.
The GPU parity tests proper are in test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest

from helpers import FEET, default_qpos, model_const, oracle_model, pd_tables
from oracle import oracle as O
from wave_emu import emu


def _batch(n, **kw):
    mc = model_const()
    return emu.EmuBatch(mc, pd_tables(mc), n, legal_bodies=FEET, **kw)


def _states(n, seed):
    rs = np.random.default_rng(seed)
    Q, V = [], []
    for i in range(n):
        q = default_qpos(76)
        q[2] = [0.93, 0.3, 0.25, 0.2, 2.0, 0.15][i % 6]
        if i % 6 in (2, 3, 5):
            q[3:7] = rs.normal(size=4); q[3:7] /= np.linalg.norm(q[3:7])
        q[7:] = rs.uniform(-0.8, 0.8, 69)
        if i % 6 == 4:
            q[7 + 5] = 3.3; q[7 + 30] = -3.2               # beyond the joint limits
        Q.append(q); V.append(rs.normal(size=75) * (0.5 if i % 2 else 3.0))
    return np.array(Q), np.array(V)


def test_forward_dynamics_pieces_match_oracle():
    om = oracle_model()
    Q, V = _states(6, 11)
    eb = _batch(6)
    eb.set_state(Q, V)
    rs = np.random.default_rng(1)
    tq = rs.normal(size=(6, 69)) * 20
    xpos, xmat = eb.kinematics()
    M, bias, qacc = eb.debug_forward(tq)
    for i in range(6):
        d = O.OracleData(om); d.qpos = Q[i]; d.qvel = V[i]; d.ctrl = tq[i]; d.forward()
        assert np.abs(xpos[i] - d.xpos).max() < 2e-6
        assert np.abs(M[i] - d.M).max() < 2e-6 * np.abs(d.M).max()
        assert np.abs(bias[i] - d.bias).max() < 2e-6 * max(1.0, np.abs(d.bias).max())
        assert np.abs(qacc[i] - d.qacc).max() < 5e-5 * np.abs(d.qacc).max(), (i, d.ncon)
        touch = sum(1 << b for b in range(24) if d.touch[b])
        assert (int(eb.touch[i, 0]) & 0xFFFFFFFF) == touch


def test_free_running_rollout_tracks_oracle():
    """40 control steps (600 mj_steps) from Default with moderate actions: the humanoid falls and hits
    the floor; float32 kernel vs float64 oracle without any re-synchronisation."""
    eb = _batch(1)
    env = O.OracleEnv(oracle_model())
    assert np.abs(env.reset() - eb.reset()[0]).max() < 1e-6
    rs = np.random.default_rng(0)
    worst = np.zeros(3)
    for i in range(40):
        a = rs.uniform(-0.3, 0.3, 69)
        o_ref, r, te, tu = env.step(a)
        o_emu, r2, te2, tu2 = eb.step(a[None])
        worst = np.maximum(worst, [np.abs(eb.qpos[0] - env.data.qpos).max(), np.abs(eb.qvel[0] - env.data.qvel).max(),
                                   np.abs(o_ref - o_emu[0]).max()])
        assert (te, tu) == (bool(te2[0]), bool(tu2[0])) and r2[0] == 0
    assert worst[0] < 2e-4 and worst[1] < 5e-3 and worst[2] < 5e-3, worst


@pytest.mark.parametrize("task,init", [(O.TASK_SPEED, O.INIT_DEFAULT), (O.TASK_GETUP, O.INIT_FALL), (O.TASK_REACH, O.INIT_DEFAULT)])
def test_task_envs_teacher_forced(task, init):
    """Speed / getup tasks incl. the Fall reset (45 warm-up mj_steps): per-step map with the kernel state
    re-synchronised to the oracle's every step (teacher forcing)."""
    om = oracle_model()
    kw = dict(reach_body=23, tar_dist_max=1.0, tar_height=(0.2, 2.0), height_change=(50, 100)) if task == O.TASK_REACH else {}
    eb = _batch(1, task=task, state_init=init, **kw)
    env = O.OracleEnv(om, task=task, state_init=init, **kw)
    rs = np.random.default_rng(3)
    fa, tr = rs.uniform(size=(3, 69)), rs.uniform(size=4)
    o_ref = env.reset(fall_actions=fa, task_rand=tr)
    o_emu = eb.reset(fall_actions=fa[None], task_rand=tr[None])[0]
    assert np.abs(eb.qpos[0] - env.data.qpos).max() < 2e-4
    assert np.abs(o_ref - o_emu).max() < 5e-3
    t = env.get_task()
    assert np.allclose(eb.task[0], [t[1], t[7], t[8], t[2]] if task == O.TASK_REACH else [t[1], t[2], t[3], 0])
    for i in range(12):
        # teacher forcing: the oracle's current state + the state of its last forward + warm start
        eb.set_state(env.data.qpos, env.data.qvel, eb.qpos_prev, eb.qvel_prev)
        a, tr = rs.uniform(-0.5, 0.5, 69), rs.uniform(size=4)
        o_ref, r, te, tu = env.step(a, task_rand=tr)
        o_emu, r2, te2, tu2 = eb.step(a[None], task_rand=tr[None])
        assert np.abs(eb.qpos[0] - env.data.qpos).max() < 1e-4
        assert np.abs(eb.qvel[0] - env.data.qvel).max() < 5e-3
        assert np.abs(o_ref - o_emu[0]).max() < 5e-3
        assert abs(r - r2[0]) < 1e-4 and (te, tu) == (bool(te2[0]), bool(tu2[0]))


def test_obs_v2_matches():
    """self_obs_v=2: body velocities come from the sensors of the LAST mj_forward (stale by design)."""
    eb = _batch(1, self_obs_v=2)
    env = O.OracleEnv(oracle_model(), self_obs_v=2)
    assert np.abs(env.reset() - eb.reset()[0]).max() < 1e-6
    rs = np.random.default_rng(4)
    for i in range(6):
        a = rs.uniform(-0.3, 0.3, 69)
        o_ref, *_ = env.step(a)
        o_emu, *_ = eb.step(a[None])
        assert o_emu.shape[1] == 358 and np.abs(o_ref - o_emu[0]).max() < 5e-3
        assert np.abs(eb.body_vel[0, :, :3] - env.data.linvel).max() < 5e-3


@pytest.mark.parametrize("mode", [1, 2])
def test_pd_and_torque_controllers_substep(mode):
    """`pd` (explicit PD, reference controllers.py:335-346) amplifies errors ~15x per mj_step on the
    armature-dominated links with the stablepd gains, so it is checked at substep granularity."""
    om = oracle_model()
    eb = _batch(1, control_mode=mode, power_scale=1.0)
    d = O.OracleData(om)
    Q, V = _states(1, 5)
    Q[0, 2] = 1.5
    d.qpos = Q[0]; d.qvel = V[0] * 0.1; d.forward()
    eb.set_state(Q, V * 0.1)
    a = np.random.default_rng(mode).uniform(-0.5, 0.5, 69)
    for s in range(2):
        d.ctrl = d.ctrl_torque(a, mode=mode, power_scale=1.0); d.step()
    eb.substep(a[None], 2)
    vmax = max(1.0, np.abs(d.qvel).max())
    assert np.abs(eb.qvel[0] - d.qvel).max() < 1e-4 * vmax
    assert np.abs(eb.qpos[0] - d.qpos).max() < 1e-5 * vmax


@pytest.mark.parametrize("mode,name", [(3, "simple_pid"), (4, "default")])
def test_simple_pid_and_default_controllers_substep(mode, name):
    """`simple_pid` (stateful SimplePID, reference controllers.py:193-262, gains jkp/10, 1, jkd/10) and `default`
    (ctrl = action): two launches of two mj_steps each, so the PID state has to survive in the ss_state buffers."""
    mc = model_const()
    om = oracle_model(control_mode=name)
    eb = emu.EmuBatch(mc, pd_tables(mc, control_mode=name), 1, legal_bodies=FEET, control_mode=mode)
    d = O.OracleData(om); d.set_pid_dt(15.0 / 450)
    Q, V = _states(1, 5)
    Q[0, 2] = 1.5
    d.qpos = Q[0]; d.qvel = V[0] * 0.1; d.forward()
    eb.set_state(Q, V * 0.1)
    rs = np.random.default_rng(mode)
    for launch in range(2):
        a = rs.uniform(-0.5, 0.5, 69) * (1.0 if mode == 3 else 40.0)      # `default` takes raw torques
        for s_ in range(2):
            d.ctrl = d.ctrl_torque(a, mode=mode); d.step()
        eb.substep(a[None], 2)
        vmax = max(1.0, np.abs(d.qvel).max())
        assert np.abs(eb.qvel[0] - d.qvel).max() < 1e-4 * vmax
        assert np.abs(eb.qpos[0] - d.qpos).max() < 1e-5 * vmax
    if mode == 3:
        assert eb.pid_started[0] == 1 and np.abs(eb.pid_integral).max() > 0


def test_autoreset_on_bad_state_and_masked_reset():
    eb = _batch(2)
    eb.reset()
    eb.qvel[0, 10] = np.inf
    eb.qvel_prev[:] = eb.qvel
    eb.step(np.zeros((2, 69)))
    assert eb.nwarn[0] == 1 and eb.nwarn[1] == 0 and np.isfinite(eb.qpos).all()
    before = eb.qpos.copy()
    eb.reset(mask=[0, 1])
    assert np.array_equal(eb.qpos[0], before[0]) and np.allclose(eb.qpos[1, 3:7], 0.5)
    assert eb.cur_t[0] == 1 and eb.cur_t[1] == 0


def test_smplx_layout_runs_and_matches():
    mc = model_const("smplx_humanoid")
    om = oracle_model("smplx_humanoid")
    eb = emu.EmuBatch(mc, pd_tables(mc), 1, legal_bodies=FEET)
    env = O.OracleEnv(om)
    assert np.abs(env.reset() - eb.reset()[0]).max() < 1e-6 and eb.obs_size == 625
    rs = np.random.default_rng(2)
    for i in range(3):
        a = rs.uniform(-0.2, 0.2, 153)
        o_ref, *_ = env.step(a)
        o_emu, *_ = eb.step(a[None])
        assert np.abs(eb.qpos[0] - env.data.qpos).max() < 1e-4
        assert np.abs(o_ref - o_emu[0]).max() < 5e-3


def test_tree_solve_keeps_float32_accuracy_at_the_trunk_joints():
    """One articulated-body solve (airborne, torque-dominated: a single Newton iteration = H x = tau), float32 kernel against its
    float64 instantiation, per dof group.  The elimination tree is rooted at the Spine: the Torso / Spine / Chest joints sit next to the
    root with the legs on one side, their joint-space 3x3 D is poorly conditioned, and W = U D^-1 has to satisfy W D = U to rounding
    (Ldl3 in ss_kernel.h).  With an explicit cofactor inverse these joints came out at 1.7e-6 / 4.2e-6 / 3.1e-6 on these states
    (medians, relative to the largest acceleration of the sample) and the free joint at 5.1e-7; with the substitution solve
    0.8e-6 / 2.3e-6 / 1.4e-6 and 2.4e-7 (profiles/r03_centred_elimination.md)."""
    n = 24
    Q, V = _states(n, 5)
    Q[:, 2] += 5.0                                             # no floor contact
    tq = np.random.default_rng(2).uniform(-1, 1, (n, 69)) * 20000
    acc = {}
    for f64 in (True, False):
        eb = _batch(n, f64=f64)
        eb.set_state(Q, 0 * V)
        acc[f64] = eb.debug_forward(tq)[2].astype(np.float64)
    scale = np.maximum(1.0, np.abs(acc[True]).max(1))
    err = np.abs(acc[False] - acc[True]) / scale[:, None]
    groups = {"root": slice(0, 6), "hips": np.r_[6:9, 18:21], "Torso": slice(30, 33), "Spine": slice(33, 36), "Chest": slice(36, 39),
              "leaves": np.r_[9:18, 21:30, 48:60, 63:75]}
    med = {k: float(np.median(err[:, s].max(1))) for k, s in groups.items()}
    print(med)
    assert med["Torso"] < 1.3e-6 and med["Spine"] < 3.2e-6 and med["Chest"] < 2.3e-6, med
    assert med["root"] < 4e-7 and med["hips"] < 1.0e-6 and med["leaves"] < 5e-6, med
