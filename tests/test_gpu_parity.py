"""GPU parity tests proper (-m gpu): the HIP path, through the product API and the C ABI, against the
float64 CPU oracle on the same seeded inputs, plus size-independent properties at the benchmark size."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from helpers import FEET, default_qpos, model_const, oracle_model  # noqa: E402
from oracle import oracle as O  # noqa: E402

# Stated tolerances (float32 kernel vs float64 oracle), per control step = 15 mj_steps, for the gentle-action tests (the
# benchmark's violent states have their own, conditioning-scaled per-sample test below).  Kept at <= 2-3x the maxima measured
# on the MI355X (gpurun_out/parity_measured.json): teacher-forced qpos 3.4e-6, qvel 8.5e-4, obs 8.5e-4 (the
# observation carries qvel entries unscaled, so it inherits the velocity tolerance), reward 2.6e-7; SURVEY §8d's budget was
# 1e-5 / 2e-3 / 1e-4 (positions) / 1e-5.  Free-running rollouts accumulate: 40 control steps with floor impacts measured 4.4e-5.
TOL_QPOS, TOL_QVEL, TOL_OBS, TOL_REW = 1e-5, 2e-3, 2e-3, 1e-6
TOL_QPOS_FREE = 1e-4


@pytest.fixture(scope="module")
def vec():
    from smplsim_amd.batch import SMPLSimVecEnv
    return SMPLSimVecEnv


def _np(t):
    return t.detach().cpu().numpy()


def _record(name, **vals):
    """Measured maxima of a run, appended to gpurun_out/parity_measured.json (the source of DESIGN.md's parity table and of
    the tolerances above, which are kept at <= 2x these)."""
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_measured.json")
        cur = json.load(open(path)) if os.path.exists(path) else {}
        def conv(v):
            if isinstance(v, (list, tuple)) and any(isinstance(x, (list, tuple)) for x in v):
                return [conv(x) for x in v]                      # nested records (e.g. the listed outliers)
            return float(v) if np.ndim(v) == 0 else [float(x) for x in np.ravel(v)]
        cur[name] = {k: conv(v) for k, v in vals.items()}
        json.dump(cur, open(path, "w"), indent=1)
    except OSError:
        pass
    print("measured", name, vals)


def test_extension_is_loaded_not_a_fallback():
    from smplsim_amd import _lib
    L = _lib.lib()
    assert L._name.endswith("libsmplsim_hip.so")
    with open("/proc/self/maps") as f:
        assert "libsmplsim_hip.so" in f.read()


def test_forward_pieces_match_oracle(vec):
    from test_kernel_emu import _states
    om, mc = oracle_model(), model_const()
    Q, V = _states(12, 11)
    env = vec(12, autoreset=False)
    env.set_state(Q, V)
    rs = np.random.default_rng(1)
    tq = rs.normal(size=(12, 69)) * 20
    xpos, xmat = env.kinematics()
    Me, bias, qacc = env.debug_forward(torch.tensor(tq, device=env.device))
    torch.cuda.synchronize()
    xpos, Me, bias, qacc = _np(xpos), _np(Me), _np(bias), _np(qacc)
    for i in range(12):
        d = O.OracleData(om); d.qpos = Q[i]; d.qvel = V[i]; d.ctrl = tq[i]; d.forward()
        assert np.abs(xpos[i] - d.xpos).max() < 2e-6
        assert np.abs(Me[i] - d.M).max() < 2e-6 * np.abs(d.M).max()
        assert np.abs(bias[i] - d.bias).max() < 2e-6 * max(1.0, np.abs(d.bias).max())
        assert np.abs(qacc[i] - d.qacc).max() < 5e-5 * np.abs(d.qacc).max(), (i, d.ncon)
        touch = sum(1 << b for b in range(24) if d.touch[b])
        assert (int(env.touch[i, 0].item()) & 0xFFFFFFFF) == touch


def test_free_running_rollout_tracks_oracle(vec):
    env = vec(8, autoreset=False)
    obs, _ = env.reset()
    oenv = O.OracleEnv(oracle_model())
    assert np.abs(oenv.reset() - _np(obs)[0]).max() < 1e-6
    rs = np.random.default_rng(0)
    worst = np.zeros(3)
    for i in range(40):
        a = rs.uniform(-0.3, 0.3, 69)
        obs, rew, term, trunc, _ = env.step(torch.tensor(np.tile(a, (8, 1)), device=env.device, dtype=torch.float32))
        o_ref, r, te, tu = oenv.step(a)
        worst = np.maximum(worst, [np.abs(_np(env.qpos)[0] - oenv.data.qpos).max(),
                                   np.abs(_np(env.qvel)[0] - oenv.data.qvel).max(), np.abs(_np(obs)[0] - o_ref).max()])
        assert (te, tu) == (bool(term[0]), bool(trunc[0]))
    assert torch.equal(env.qpos[0], env.qpos[7])             # identical inputs -> bit-identical envs
    _record("free_running_40_steps_smpl", qpos=worst[0], qvel=worst[1], obs=worst[2])
    assert worst[0] < TOL_QPOS_FREE and worst[1] < TOL_QVEL and worst[2] < TOL_OBS, worst


@pytest.mark.parametrize("task,init", [("HumanoidSpeed", "Default"), ("HumanoidGetup", "Fall"), ("HumanoidReach", "Default")])
def test_task_envs_teacher_forced(vec, task, init):
    from smplsim_amd import _cabi
    om = oracle_model()
    kw = dict(tar_dist_max=1.0, tar_height=(0.2, 2.0), height_change=(50, 100)) if task == "HumanoidReach" else {}
    env = vec(4, task=task, state_init=init, autoreset=False, **kw)
    oenv = O.OracleEnv(om, task=_cabi.TASKS[task], state_init=_cabi.STATE_INITS[init], reach_body=23, **kw)
    rs = np.random.default_rng(3)
    fa, tr = rs.uniform(size=(3, 69)), rs.uniform(size=4)
    dev = env.device
    T = lambda x: torch.tensor(np.tile(np.asarray(x)[None], (4,) + (1,) * np.asarray(x).ndim), device=dev, dtype=torch.float32)
    o_ref = oenv.reset(fall_actions=fa, task_rand=tr)
    obs, _ = env.reset(fall_actions=T(fa), task_rand=T(tr))
    assert np.abs(_np(env.qpos)[0] - oenv.data.qpos).max() < TOL_QPOS
    assert np.abs(o_ref - _np(obs)[0]).max() < TOL_OBS
    worst = np.zeros(4)
    for i in range(12):
        env.set_state(np.tile(oenv.data.qpos, (4, 1)), np.tile(oenv.data.qvel, (4, 1)), env.qpos_prev, env.qvel_prev)
        a, tr = rs.uniform(-0.5, 0.5, 69), rs.uniform(size=4)
        o_ref, r, te, tu = oenv.step(a, task_rand=tr)
        obs, rew, term, trunc, _ = env.step(T(a), task_rand=T(tr))
        e = [np.abs(_np(env.qpos)[0] - oenv.data.qpos).max(), np.abs(_np(env.qvel)[0] - oenv.data.qvel).max(),
             np.abs(o_ref - _np(obs)[0]).max(), abs(r - float(rew[0]))]
        worst = np.maximum(worst, e)
        assert e[0] < TOL_QPOS and e[1] < TOL_QVEL and e[2] < TOL_OBS and e[3] < TOL_REW, (i, e)
        assert (te, tu) == (bool(term[0]), bool(trunc[0]))
    _record(f"teacher_forced_{task}_{init}", qpos=worst[0], qvel=worst[1], obs=worst[2], reward=worst[3])


def test_native_mjcf_entry_on_gpu(vec):
    """ss_model_create_from_mjcf (the library's own MJCF compiler + gain tables): the same stepping, bit for bit, as the model
    the Python host compiler describes."""
    from smplsim_amd.batch import ShardModel
    a = vec(3, model=ShardModel(humanoid="smplx_humanoid"), autoreset=False, seed=5)
    b = vec(3, model=ShardModel(humanoid="smplx_humanoid", compiler="native"), autoreset=False, seed=5)
    assert torch.equal(a.reset()[0], b.reset()[0])
    g = torch.Generator().manual_seed(0)
    for _ in range(3):
        act = (torch.rand(3, 153, generator=g) - 0.5).to(a.device)
        ra, rb = a.step(act), b.step(act)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(a.qpos, b.qpos)


def test_smplx_layout(vec):
    from smplsim_amd.batch import ShardModel
    env = vec(4, model=ShardModel(humanoid="smplx_humanoid"), autoreset=False)
    oenv = O.OracleEnv(oracle_model("smplx_humanoid"))
    obs, _ = env.reset()
    assert env.obs_size == 625 and np.abs(oenv.reset() - _np(obs)[0]).max() < 1e-6
    rs = np.random.default_rng(2)
    for i in range(3):
        a = rs.uniform(-0.2, 0.2, 153)
        o_ref, *_ = oenv.step(a)
        obs, *_ = env.step(torch.tensor(np.tile(a, (4, 1)), device=env.device, dtype=torch.float32))
        assert np.abs(_np(env.qpos)[0] - oenv.data.qpos).max() < TOL_QPOS
        assert np.abs(o_ref - _np(obs)[0]).max() < TOL_OBS


def test_smplx_free_running_with_floor_contact(vec):
    """SMPL-X layout (52 bodies, two tree-level passes of the articulated-body sweeps), 20 control steps from Default:
    the model settles onto its feet, so contacts, limits and the Newton solve are all exercised."""
    from smplsim_amd.batch import ShardModel
    env = vec(2, model=ShardModel(humanoid="smplx_humanoid"), autoreset=False)
    oenv = O.OracleEnv(oracle_model("smplx_humanoid"))
    obs, _ = env.reset(); oenv.reset()
    rs = np.random.default_rng(7)
    worst = np.zeros(2)
    for i in range(20):
        a = rs.uniform(-0.3, 0.3, 153)
        o_ref, *_ = oenv.step(a)
        obs, *_ = env.step(torch.tensor(np.tile(a, (2, 1)), device=env.device, dtype=torch.float32))
        worst = np.maximum(worst, [np.abs(_np(env.qpos)[0] - oenv.data.qpos).max(), np.abs(o_ref - _np(obs)[0]).max()])
    assert int(env.touch[0, 0].item()) != 0 or int(env.touch[0, 1].item()) != 0     # it does stand on the floor
    _record("free_running_20_steps_smplx", qpos=worst[0], obs=worst[1])
    assert worst[0] < TOL_QPOS_FREE and worst[1] < TOL_OBS, worst


def test_obs_v2_on_gpu(vec):
    """self_obs_v=2 (reference humanoid_env.py:637-687): per-body velocities from the sensors of the last mj_forward."""
    env = vec(2, self_obs_v=2, autoreset=False)
    oenv = O.OracleEnv(oracle_model(), self_obs_v=2)
    obs, _ = env.reset()
    assert env.obs_size == 358 and np.abs(oenv.reset() - _np(obs)[0]).max() < 1e-6
    rs = np.random.default_rng(4)
    for i in range(8):
        a = rs.uniform(-0.3, 0.3, 69)
        o_ref, *_ = oenv.step(a)
        obs, *_ = env.step(torch.tensor(np.tile(a, (2, 1)), device=env.device, dtype=torch.float32))
        assert np.abs(o_ref - _np(obs)[0]).max() < TOL_OBS
        assert np.abs(_np(env.body_vel)[0, :, :3] - oenv.data.linvel).max() < TOL_QVEL


@pytest.mark.parametrize("mode", ["simple_pid", "default"])
def test_simple_pid_and_default_controllers_on_gpu(vec, mode):
    """`simple_pid` (stateful, reference controllers.py:193-262) and `default` (ctrl = action) through the product API:
    two launches of two mj_steps, the PID state persists in the caller-owned pid_* buffers between them."""
    from smplsim_amd import _cabi
    from test_kernel_emu import _states
    env = vec(2, control_mode=mode, autoreset=False)
    m = _cabi.CONTROL_MODES[mode]
    d = O.OracleData(oracle_model(control_mode=mode)); d.set_pid_dt(15.0 / 450)
    Q, V = _states(1, 5)
    Q[0, 2] = 1.5
    d.qpos = Q[0]; d.qvel = V[0] * 0.1; d.forward()
    env.set_state(np.tile(Q, (2, 1)), np.tile(V * 0.1, (2, 1)))
    rs = np.random.default_rng(m)
    for launch in range(2):
        a = rs.uniform(-0.5, 0.5, 69) * (1.0 if mode == "simple_pid" else 40.0)
        for s_ in range(2):
            d.ctrl = d.ctrl_torque(a, mode=m); d.step()
        env.substep(torch.tensor(np.tile(a, (2, 1)), device=env.device, dtype=torch.float32), 2)
        torch.cuda.synchronize()
        vmax = max(1.0, np.abs(d.qvel).max())
        assert np.abs(_np(env.qvel)[0] - d.qvel).max() < 1e-4 * vmax
        assert np.abs(_np(env.qpos)[0] - d.qpos).max() < 1e-5 * vmax
    if mode == "simple_pid":
        assert int(env.pid_started[0].item()) == 1 and float(env.pid_integral.abs().max().item()) > 0


@pytest.mark.parametrize("mode", ["pd", "torque"])
def test_pd_and_torque_controllers_on_gpu(vec, mode):
    """control_mode pd / torque (reference controllers.py:6-47,265-349) at substep granularity: explicit PD with the
    stablepd gains amplifies differences ~15x per mj_step on the armature-dominated links."""
    from smplsim_amd import _cabi
    from test_kernel_emu import _states
    from smplsim_amd.batch import ShardModel
    om = oracle_model()
    # tables with the stablepd torque limits for both modes: the reference leaves torque_lim at zero in `torque`
    # mode (humanoid_env.py:341-349 only fills it for pd / uhc_pd), which would make this check vacuous
    env = vec(2, model=ShardModel(control_mode="uhc_pd"), control_mode=mode, power_scale=1.0, autoreset=False)
    m = _cabi.CONTROL_MODES[mode]
    d = O.OracleData(om)
    Q, V = _states(1, 5)
    Q[0, 2] = 1.5
    d.qpos = Q[0]; d.qvel = V[0] * 0.1; d.forward()
    env.set_state(np.tile(Q, (2, 1)), np.tile(V * 0.1, (2, 1)))
    a = np.random.default_rng(m).uniform(-0.5, 0.5, 69)
    for s_ in range(2):
        d.ctrl = d.ctrl_torque(a, mode=m, power_scale=1.0); d.step()
    env.substep(torch.tensor(np.tile(a, (2, 1)), device=env.device, dtype=torch.float32), 2)
    torch.cuda.synchronize()
    vmax = max(1.0, np.abs(d.qvel).max())
    assert np.abs(_np(env.qvel)[0] - d.qvel).max() < 1e-4 * vmax
    assert np.abs(_np(env.qpos)[0] - d.qpos).max() < 1e-5 * vmax


@pytest.mark.parametrize("shape", ["chain15", "comb9", "comb17"])
def test_synthetic_trees_on_gpu(vec, shape):
    """Deep chain (15 bodies: the elimination tree is rooted at its middle, 7 levels), wide comb (9 nodes in one level: two 8-node
    passes, the large kernel variant) and a comb of 17 branches (rooted at a branch: 16 nodes in the widest level)."""
    from smplsim_amd.batch import ShardModel
    from smplsim_amd.mjcf_writer import table_to_mjcf
    from test_synthetic_trees_emu import _chain, _comb
    xml = table_to_mjcf(_chain(14) if shape == "chain15" else _comb(9 if shape == "comb9" else 17))
    root_z = 2.8 if shape == "chain15" else 0.40
    from smplsim_amd.mjcf import compile_mjcf
    mc = compile_mjcf(xml)
    nu = mc.nu
    tables = (np.full(nu, 60.0), np.full(nu, 6.0), np.full(nu, 40.0), np.full(nu, 2.0), np.zeros(nu))
    legal = tuple(mc.body_names)
    env = vec(3, model=ShardModel(xml=xml, contact_bodies=legal, tables=tables), autoreset=False, reach_body=mc.body_names[-1])
    om = O.OracleModel(xml, *tables, legal_bodies=legal)
    rs = np.random.default_rng(len(shape))
    q = np.zeros(mc.nq); q[2] = root_z; q[3] = 1.0
    q[7:] = rs.uniform(-0.3, 0.3, mc.nq - 7)
    v = rs.normal(size=mc.nv) * 0.3
    env.set_state(np.tile(q, (3, 1)), np.tile(v, (3, 1)))
    d = O.OracleData(om); d.qpos = q; d.qvel = v; d.ctrl = np.zeros(nu); d.forward()
    M, bias, qacc = env.debug_forward(torch.zeros(3, nu, device=env.device))
    torch.cuda.synchronize()
    assert np.abs(_np(M)[0] - d.M).max() < 5e-6 * np.abs(d.M).max()
    assert np.abs(_np(qacc)[0] - d.qacc).max() < 2e-4 * max(1.0, np.abs(d.qacc).max())
    a = rs.uniform(-0.3, 0.3, nu)
    for s_ in range(6):
        d.ctrl = d.spd_torque(a); d.step()
    env.substep(torch.tensor(np.tile(a, (3, 1)), device=env.device, dtype=torch.float32), 6)
    torch.cuda.synchronize()
    vmax = max(1.0, np.abs(d.qvel).max())
    assert np.abs(_np(env.qpos)[0] - d.qpos).max() < 2e-5 * vmax and np.abs(_np(env.qvel)[0] - d.qvel).max() < 5e-4 * vmax


def test_benchmark_size_properties(vec):
    """4096 envs, full-range random actions (BASELINE config 2): size-independent properties —
    finite state, unit root quaternions, deterministic replay, yaw invariance of the observation,
    episode bookkeeping with device-side autoreset."""
    N = 4096
    env = vec(N, autoreset=True, seed=1)
    env.reset()
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    acts = [torch.rand(N, 69, generator=g, device=env.device) * 2 - 1 for _ in range(12)]
    for a in acts:
        obs, rew, term, trunc, info = env.step(a)
    torch.cuda.synchronize()
    q1, o1 = env.qpos.clone(), obs.clone()
    assert torch.isfinite(env.qpos).all() and torch.isfinite(env.qvel).all() and torch.isfinite(obs).all()
    assert (env.qpos[:, 3:7].norm(dim=1) - 1).abs().max() < 1e-4
    assert int(env.cur_t.max()) == 12 and not term.any()
    env2 = vec(N, autoreset=True, seed=1)
    env2.reset()
    for a in acts:
        obs2, *_ = env2.step(a)
    torch.cuda.synchronize()
    assert torch.equal(q1, env2.qpos) and torch.equal(o1, obs2)      # bit-reproducible


@pytest.mark.parametrize("obs_v", [1, 2])
def test_proprioception_is_yaw_invariant(vec, obs_v):
    """The reference's commented check (humanoid_env.py:497-504): the observation of a state and of the same state turned
    about the vertical (root position anywhere in the plane) is the same vector.  The world-frame root linear velocity turns
    with the body; the body-frame angular velocity and the joint velocities are yaw-free already."""
    n = 16
    rs = np.random.default_rng(obs_v)
    q = np.tile(default_qpos(76), (n, 1)); v = np.zeros((n, 75))
    q[:, 7:] = rs.uniform(-0.6, 0.6, (1, 69)); v[:] = rs.normal(size=(1, 75))
    base = rs.normal(size=4); base /= np.linalg.norm(base)
    for i in range(n):
        th = 0.0 if i == 0 else rs.uniform(-np.pi, np.pi)
        yaw = np.array([np.cos(th / 2), 0, 0, np.sin(th / 2)])
        w1, x1, y1, z1 = yaw; w2, x2, y2, z2 = base
        q[i, 3:7] = [w1*w2 - x1*x2 - y1*y2 - z1*z2, w1*x2 + x1*w2 + y1*z2 - z1*y2, w1*y2 - x1*z2 + y1*w2 + z1*x2, w1*z2 + x1*y2 - y1*x2 + z1*w2]
        c, s_ = np.cos(th), np.sin(th)
        v[i, 0], v[i, 1] = c * v[0, 0] - s_ * v[0, 1], s_ * v[0, 0] + c * v[0, 1]
        if i:
            q[i, :2] = rs.uniform(-5, 5, 2)
    env = vec(n, state_init="External", self_obs_v=obs_v, autoreset=False)
    env.set_state(q, v)
    obs, _ = env.reset()                                     # reset_sim(): mj_forward + compute_proprioception on the caller's state
    torch.cuda.synchronize()
    o = _np(obs)
    d = np.abs(o[1:] - o[:1]).max(axis=0)
    if obs_v == 1:                                           # the body-frame root angular velocity, rotated like a world vector
        d[217:219] = 0.0                                     # by the reference (see the emulator twin in test_parity_f64.py)
    d = d.max()
    _record(f"yaw_invariance_obs_v{obs_v}", max_abs_diff=d)
    assert d < 2e-5 * max(1.0, np.abs(o[0]).max()), d


def test_episode_truncation_and_autoreset(vec):
    env = vec(64, autoreset=True, episode_length=5)
    env.reset()
    z = torch.zeros(64, 69, device=env.device)
    for i in range(6):
        obs, rew, term, trunc, info = env.step(z)
    torch.cuda.synchronize()
    assert trunc.all() and int(env.cur_t.max()) == 0          # cur_t 6 > 5 -> truncated -> reset in the same call
    assert torch.allclose(env.qpos[:, 2], torch.full((64,), 0.94, device=env.device))
    assert "final_observation" in info and (info["final_observation"][:, 0] - 0.94).abs().max() > 1e-4


def test_fused_autoreset_equals_two_launch_path(vec):
    """ss_step_autoreset (default of SMPLSimVecEnv for StateInit.Default) against ss_step + masked ss_reset on the GPU."""
    kw = dict(task="HumanoidSpeed", episode_length=6, seed=5)
    a_env, b_env = vec(64, fused_autoreset=True, **kw), vec(64, fused_autoreset=False, **kw)
    g = torch.Generator(device=a_env.device); g.manual_seed(2)
    a_env.reset(); b_env.reset()
    ended = 0
    for t in range(20):
        act = torch.rand(64, 69, generator=g, device=a_env.device) * 2 - 1
        oa, ra, tea, tua, ia = a_env.step(act)
        ob, rb, teb, tub, ib = b_env.step(act)
        ended += int((tea | tua).sum())
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(tea, teb) and torch.equal(tua, tub)
        assert torch.equal(ia["final_observation"], ib["final_observation"])
        assert torch.equal(a_env.qpos, b_env.qpos) and torch.equal(a_env.cur_t, b_env.cur_t) and torch.equal(a_env.task_state, b_env.task_state)
    assert ended > 64


@pytest.mark.parametrize("which", ["smpl", "getup", "smplx", "smpl_selfcol", "getup_selfcol", "smplx_selfcol"])
def test_per_sample_parity_on_the_benchmark_distribution(vec, which):
    """The benchmark's own states, per sample (GPU twin of test_parity_f64.py): envs driven by full-range uniform(-1,1)
    actions (thrown around, lying on the floor with ~10 bodies in contact, some diverging), the GPU at its SHIPPED solver
    settings.  At several control steps the envs with the most Newton iterations (the stragglers of the launch) plus a random
    sample are replayed from the GPU's own pre-step state by the float64 oracle at MuJoCo's solver settings (and converged), the
    float64 instantiation of the kernel (shipped settings / converged) and its input-perturbed twins; asserted per sample:
    formulation (f64 kernel vs oracle, both converged) <= 1e-9, identical bad-state resets, solver rule (f64 kernel at the shipped
    settings vs the oracle at MuJoCo's) within the stated per-step tolerance on >= 99.5% of the samples (the rest listed),
    precision (GPU float32 vs f64 kernel) <= TOL_STEP * max(1, cond / COND_REF), the fraction of samples on which the GPU is within
    TOL_STEP of the oracle at MuJoCo's settings, and observation / reward of the GPU against the f64 kernel.
    BASELINE configs 2 (smpl), 3 (getup, StateInit.Fall) and 4 (smplx), each also with the reference MJCF's body-body contacts."""
    import parity_tools as P
    from test_parity_f64 import COND_FLOOR, K_ROUND, f32_bound
    from smplsim_amd.batch import ShardModel
    base = which.replace("_selfcol", "")
    selfcol = which.endswith("_selfcol")
    N, steps, every, per_step = {"smpl": (4096, 36, 5, (24, 60)), "getup": (1024, 12, 4, (12, 40)), "smplx": (1024, 24, 6, (8, 20)),
                                 "smpl_selfcol": (2048, 24, 4, (16, 40)), "getup_selfcol": (1024, 12, 4, (8, 24)),
                                 "smplx_selfcol": (512, 18, 6, (4, 10))}[which]
    humanoid = "smplx_humanoid" if base == "smplx" else "smpl_humanoid"
    kw = dict(task="HumanoidGetup", state_init="Fall") if base == "getup" else {}
    tkw = {"task": "HumanoidGetup", "state_init": 1} if base == "getup" else {}
    if selfcol:
        kw["self_collision"] = True; tkw["self_collision"] = True
    env = vec(N, model=ShardModel(humanoid=humanoid), autoreset=True, seed=11, **kw)
    g = torch.Generator(device=env.device); g.manual_seed(11)
    env.reset()
    rs = np.random.default_rng(0)
    keys = ("qpos", "qvel", "qpos_prev", "qvel_prev", "qacc_warm")
    pres, posts, acts = [], [], []
    for t in range(steps):
        act = torch.rand(N, env.nu, generator=g, device=env.device) * 2 - 1
        sample = t % every == every - 1
        if sample:
            pre = {k: _np(getattr(env, k)).astype(np.float64) for k in keys}
            pre["task"], pre["cur_t"], nw0 = _np(env.task_state).astype(np.float64), _np(env.cur_t).copy(), _np(env.nwarn).copy()
        tr = torch.rand(N, 4, generator=g, device=env.device)
        obs, rew, term, trunc, info = env.step(act, task_rand=tr)
        if not sample:
            continue
        torch.cuda.synchronize()
        it = _np(env.solver_iters)
        alive = ~(_np(term) | _np(trunc))
        heavy = np.argsort(-it)[:per_step[0]]                                 # the launch's stragglers
        pick = np.unique(np.concatenate([heavy, rs.choice(N, per_step[1], replace=False)]))
        pick = pick[alive[pick]]
        pre["task_rand"] = _np(tr).astype(np.float64)
        pres.append({k: v[pick] for k, v in pre.items()}); acts.append(_np(act).astype(np.float64)[pick])
        final = info.get("final_observation", obs)
        posts.append(dict(qpos=_np(env.qpos).astype(np.float64)[pick], qvel=_np(env.qvel).astype(np.float64)[pick],
                          obs=_np(final).astype(np.float64)[pick], reward=_np(rew).astype(np.float64)[pick],
                          nwarn=(_np(env.nwarn) - nw0)[pick], iters=it[pick]))
    pre = {k: np.concatenate([p[k] for p in pres]) for k in pres[0]}
    post = {k: np.concatenate([p[k] for p in posts]) for k in posts[0]}
    A = np.concatenate(acts)
    r = P.triage(pre, A, post, humanoid=humanoid, n_perturb=4 if base == "smplx" and selfcol else 8, **tkw)
    ok = ~r["reset"]
    cond = np.maximum(r["cond"], COND_FLOOR)
    ratio64 = r["formulation"] / (cond * P.EPS64)
    bound32 = f32_bound(r["cond"])
    rule_ok, f32_ok = P.within_tol(r["solver_rule"]), P.within_tol(r["f32_vs_oracle"])
    for k in ("formulation", "solver_rule", "oracle_rule", "precision", "f32_vs_oracle", "cond"):
        print(P.summarize(k, r[k], ok))
    outside = [(int(i), r["f32_vs_oracle"][i].tolist(), r["cond"][i].tolist()) for i in np.flatnonzero(ok & ~f32_ok)]
    print(f"solver rule within {P.TOL_STEP}: {rule_ok[ok].mean():.4f}; outside {r['solver_rule'][ok & ~rule_ok].tolist()}")
    print(f"GPU float32 vs oracle at MuJoCo's settings within {P.TOL_STEP}: {f32_ok[ok].mean():.4f}; outside (sample, error, cond): {outside}")
    try:                                                         # per-sample record for choosing / checking the gate (tests/test_parity_f64.py COND_REF)
        out_ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out_, exist_ok=True)
        np.savez_compressed(os.path.join(out_, f"parity_samples_{which}.npz"), precision=r["precision"], cond=r["cond"], ok=ok, f32_vs_oracle=r["f32_vs_oracle"],
                            obs=r["obs"], reward=r["reward"], vscale=r["vscale"])
    except OSError:
        pass
    _record("benchmark_distribution_" + which, samples=len(ok), resets=int((~ok).sum()), max_newton_iters=int(post["iters"].max()),
            mean_newton_iters=float(post["iters"].mean()), mean_newton_iters_f64_kernel=float(r["iters"].mean()),
            formulation_max=r["formulation"][ok].max(axis=0), solver_rule_max=r["solver_rule"][ok].max(axis=0),
            solver_rule_within_tol_frac=float(rule_ok[ok].mean()), oracle_mujoco_vs_converged_max=r["oracle_rule"][ok].max(axis=0),
            precision_median=np.median(r["precision"][ok], axis=0),
            precision_p99=np.quantile(r["precision"][ok], 0.99, axis=0), precision_max=r["precision"][ok].max(axis=0),
            precision_over_bound_max=(r["precision"][ok] / bound32[ok]).max(axis=0),
            f32_vs_oracle_within_tol_frac=float(f32_ok[ok].mean()), f32_vs_oracle_outside=outside,
            f32_vs_oracle_max=r["f32_vs_oracle"][ok].max(axis=0), obs_max=r["obs"][ok].max(), reward_max=r["reward"][ok].max())
    assert ok.sum() >= {"smpl": 400, "getup": 100, "smplx": 80, "smpl_selfcol": 200, "getup_selfcol": 60, "smplx_selfcol": 30}[which], ok.sum()
    if selfcol:
        assert (r["nself"][ok] > 0).mean() > 0.3                 # most of these samples do have body-body contacts
    # MuJoCo's bad-state autoreset (|qpos|, |qvel|, |qacc| > 1e10 or NaN): the oracle and the float64 kernel must take it on the
    # same samples — except that a trajectory which is blowing up crosses 1e10 one mj_step earlier or later depending on
    # rounding (these states double per step), so a sample in a few hundred may flip; such samples are excluded from the
    # comparisons above either way ("reset" = any implementation reset)
    assert (~r["resets_agree"]).sum() <= max(1, len(r["resets_agree"]) // 100), (~r["resets_agree"]).sum()
    _record("benchmark_distribution_resets_" + which, **P.check_reset_rates(r, which))   # rate per implementation, within 20 % of the oracle's
    assert (r["formulation"][ok] <= np.maximum(1e-9, K_ROUND * cond[ok] * P.EPS64)).all(), r["formulation"][ok].max(axis=0)
    assert (ratio64[ok] <= K_ROUND).all(), ratio64[ok].max(axis=0)
    assert rule_ok[ok].mean() >= 0.995, (rule_ok[ok].mean(), r["solver_rule"][ok & ~rule_ok])
    assert (r["precision"][ok] <= bound32[ok]).all(), (r["precision"][ok] / bound32[ok]).max(axis=0)
    med, p90 = np.median(r["precision"][ok], axis=0), np.quantile(r["precision"][ok], 0.9, axis=0)
    assert med[0] < 5e-7 and med[1] < 5e-5 and p90[0] < 5e-6 and p90[1] < 5e-4, (med, p90)
    # measured 0.973 - 1.0 on the six configurations (profiles/r04_parity_measured.json); the stragglers are over-represented here
    assert (~f32_ok[ok]).sum() <= max(1, int(0.02 * ok.sum())), (f32_ok[ok].mean(), outside)
    worst = r["precision"].max(axis=1)
    # (measured on all six configurations: obs error / state error <= 1.0 for every sample whose state error is below 1 % of the velocity scale;
    #  a sample that is diverging — state error 6 % of the scale at a condition number of 9e5 — showed 6.6: the observation's rotations are
    #  not linear over such a distance.  Those samples get the looser factor.)
    lin = worst <= 1e-2
    assert (r["obs"][ok & lin] <= 4 * worst[ok & lin] + 1e-5).all()
    assert (r["obs"][ok & ~lin] <= 16 * worst[ok & ~lin]).all()
    assert (r["reward"][ok] <= 2 * r["precision"][ok, 0] * r["vscale"][ok] + 1e-6).all()


def test_self_collision_on_gpu(vec):
    """Body-body contacts (ss_env_cfg.self_collision) on the GPU: contact counts and constrained accelerations of
    folded-up states against the oracle, then teacher-forced control steps under full-range actions; the flag off is the old path."""
    from test_selfcol_emu import _states
    om, Q, V, T = _states(16, 2)
    env = vec(16, autoreset=False, self_collision=True)
    env.set_state(Q, V)
    M, bias, qacc = env.debug_forward(torch.tensor(T, device=env.device))
    torch.cuda.synchronize()
    d = O.OracleData(om)
    worst = 0.0
    for i in range(16):
        d.qpos = Q[i]; d.qvel = V[i]; d.ctrl = T[i]; d.warm = np.zeros(75); d.forward()
        assert int(env.self_contacts[i]) == d.nself
        e = np.abs(_np(qacc)[i] - d.qacc).max() / np.abs(d.qacc).max()
        worst = max(worst, e)
        assert e < 5e-5, (i, d.ncon, d.nself, e)
    env0 = vec(16, autoreset=False)
    env0.set_state(Q, V)
    assert (env0.debug_forward(torch.tensor(T, device=env.device))[2] - qacc).abs().max() > 1e-2 and int(env0.self_contacts.sum()) == 0
    oenv = O.OracleEnv(om)
    env = vec(2, autoreset=False, self_collision=True)
    obs, _ = env.reset(); oenv.reset()
    rs = np.random.default_rng(5)
    w2, with_self = np.zeros(3), 0
    for i in range(12):
        env.set_state(np.tile(oenv.data.qpos, (2, 1)), np.tile(oenv.data.qvel, (2, 1)), env.qpos_prev, env.qvel_prev)
        a = rs.uniform(-1, 1, 69)
        o_ref, r, te, tu = oenv.step(a)
        obs, rew, term, trunc, _ = env.step(torch.tensor(np.tile(a, (2, 1)), device=env.device, dtype=torch.float32))
        with_self += oenv.data.nself > 0
        assert int(env.self_contacts[0]) == oenv.data.nself
        scale = max(1.0, np.abs(oenv.data.qvel).max())
        w2 = np.maximum(w2, [np.abs(_np(env.qpos)[0] - oenv.data.qpos).max() / scale, np.abs(_np(env.qvel)[0] - oenv.data.qvel).max() / scale,
                             np.abs(_np(obs)[0] - o_ref).max() / scale])
    _record("self_collision", qacc_rel=worst, qpos=w2[0], qvel=w2[1], obs=w2[2], steps_with_body_body_contact=with_self)
    assert with_self >= 5 and w2[0] < 2e-5 and w2[1] < 2e-3 and w2[2] < 2e-3, w2


def test_pipelined_sub_batches_are_the_single_batch_bit_for_bit(vec):
    """PipelinedVecEnv(N, G, seed): G sub-batches on G streams, stepped round-robin, ARE the job SMPLSimVecEnv(N, seed) — the master
    generator hands every env the draws it gets in the single batch — bit for bit through autoresets (GPU twin of the emulator test)."""
    from smplsim_amd.pipeline import PipelinedVecEnv
    kw = dict(seed=3, task="HumanoidSpeed", episode_length=6)
    pipe, one = PipelinedVecEnv(256, sub_batches=4, **kw), vec(256, **kw)
    pipe.reset(); one.reset()
    pipe.synchronize()
    assert torch.equal(one.obs_buf, torch.cat([e.obs_buf for e in pipe.envs]))
    gen = torch.Generator(device=pipe.device); gen.manual_seed(7)
    for t in range(10):
        a = torch.rand(256, 69, generator=gen, device=pipe.device) * 2 - 1
        torch.cuda.synchronize()
        outs = [pipe.step_async(g, a[pipe.rows(g)]) for g in range(4)]
        o1, r1, te1, tu1, i1 = one.step(a)
        pipe.synchronize(); torch.cuda.synchronize()
        assert torch.equal(o1, torch.cat([o[0] for o in outs])) and torch.equal(r1, torch.cat([o[1] for o in outs]))
        assert torch.equal(te1, torch.cat([o[2] for o in outs])) and torch.equal(tu1, torch.cat([o[3] for o in outs]))
        assert torch.equal(i1["final_observation"], torch.cat([o[4]["final_observation"] for o in outs]))
        assert torch.equal(one.qpos, torch.cat([e.qpos for e in pipe.envs])) and torch.equal(one.qvel, torch.cat([e.qvel for e in pipe.envs]))
    assert int(one.cur_t.max()) <= 7                            # the short episodes were truncated and reset inside the step launches
    pipe.close(); one.close()


def test_pipelined_sampler_equals_the_serial_sampler_on_gpu(vec):
    """AgentPPO.sample_pipelined (sub-batches on their own streams: one sub-batch's bf16 MFMA policy forward fills the tail of another's
    step launch) returns the rollout of AgentPPO.sample bit for bit: the MFMA kernels' row results do not depend on the row count."""
    from smplsim_amd.agents.ppo import AgentPPO, PPOConfig
    from smplsim_amd.pipeline import PipelinedVecEnv
    N, G, T = 512, 4, 6
    kw = dict(task="HumanoidSpeed", episode_length=4, seed=9)
    # (the bf16 MFMA policy only: torch / hipBLASLt picks its GEMM kernel by the row count, its row results move by a rounding between
    # 512 and 128 rows, and the contact-rich rollout amplifies that to O(1) within a few steps — measured 1.5 in the observations)
    cfg = PPOConfig(hidden=(256, 128), min_batch_size=N * T, mfma_inference=True)
    a1 = AgentPPO(vec(N, **kw), cfg, seed=4)
    pipe = PipelinedVecEnv(N, sub_batches=G, **kw)
    a2 = AgentPPO(pipe, cfg, seed=4)
    for _ in range(2):
        b1, b2 = a1.sample(), a2.sample_pipelined(pipe)
        torch.cuda.synchronize()
        assert set(b1) == set(b2)
        for k in b1:
            assert torch.equal(b1[k], b2[k]), (k, (b1[k] - b2[k]).abs().max())
    assert (1.0 - b1["not_done"]).sum() >= N
    a1.env.close(); pipe.close()


@pytest.mark.parametrize("mode", ["one_action", "fresh_actions"])
def test_gym_style_single_env_matches_oracle(mode):
    """The reference's single-env surface (HumanoidEnv(cfg).reset/step, numpy in/out), BASELINE config 1: 120 control steps against
    the oracle with the reference's contact set (every body-body contact kept), each step from the oracle's COMPLETE pre-step state
    (qpos, qvel, the stale state of the controller's M / bias, the warm start) at the stated per-step tolerances (TOL_QPOS / TOL_QVEL /
    TOL_OBS relative to the velocity scale).  Both ways config 1 is driven: examples/benchmark.py:100 samples ONE action and reuses it
    for every rep ("one_action": the humanoid folds up under constant full-range targets and stays in body-body contact), BASELINE.json's
    config 1 says random-action steps ("fresh_actions").  Every step outside the tolerance is triaged (tests/gym_parity.py: float64
    kernel vs oracle, conditioning from input-perturbed twins, contact-set equality) and must be EXPLAINED by its conditioning — a
    count alone is not accepted — and at most 3 / 6 of the 120 steps may be outside at all."""
    import gym_parity as G
    import smpl_sim.envs.tasks as tasks                       # the reference's import path
    from smplsim_amd.config import default_cfg
    env = tasks.HumanoidEnv(default_cfg("HumanoidEnv"))
    assert env.self_collision                                  # body-body contacts on, like the reference's MuJoCo model
    assert env.observation_space.shape == (289,) and env.action_space.shape == (69,) and env.actuator_names[0] == "L_Hip_x"
    assert env.curr_power_usage == []                          # recorded from the first access on (humanoid_env.py:443-451)
    rec = G.run(env, mode, to_np=_np)
    pw = env.curr_power_usage
    assert len(pw) == 15 and pw[0].shape == (69,) and all((p >= 0).all() for p in pw) and max(p.max() for p in pw) > 0.1
    _record("gym_single_env_" + mode, **{k: v for k, v in rec.items() if k != "outside"},
            outside_tolerance=[[o["step"]] + o["error"] + [o["body_body_contacts"], o["newton_iters_control_step"]] + o["cond"] + o["precision_over_bound"]
                               + [float(o["contact_sets_equal"])] for o in rec["outside"]],
            outside_causes=[[float(o["step"]), float(not o["cause"].startswith("UNEXPLAINED"))] for o in rec["outside"]])
    import json
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"gym_triage_{mode}.json"), "w"), indent=1, default=float)
    G.check(rec, mode)
    env.close()


def test_benchmark_harness_shape_runs():
    """evaluate_env of reference examples/benchmark.py:97-116, re-stated on the vector env."""
    import time
    from smplsim_amd.config import default_cfg
    from smplsim_amd.envs import SMPLSimGymVecEnv
    env = SMPLSimGymVecEnv(default_cfg("HumanoidEnv"), 64)
    num_envs = env.num_envs if hasattr(env, "num_envs") else 1
    action = env.action_space.sample()
    times = []
    for _ in range(3):
        t0 = time.perf_counter(); env.reset(seed=54); times.append(time.perf_counter() - t0)
    times = []
    for _ in range(5):
        t0 = time.perf_counter(); out = env.step(actions=action); times.append(time.perf_counter() - t0)
    sps = num_envs * 5 / np.sum(times)
    assert out[0].shape == (64, 289) and np.isfinite(out[0]).all() and sps > 0
    env.close()
    # the single env the way benchmark.py:87-116 times it (one sampled action, reset(seed=54) / step(action=...) reps): BASELINE config 1's
    # plumbing figure on this box (numpy in / out, one launch + one device-to-host copy per step; the reference contact set is on)
    import smpl_sim.envs.tasks as tasks
    e1 = tasks.HumanoidEnv(default_cfg("HumanoidEnv"))
    action = e1.action_space.sample()
    e1.reset(seed=54)
    for _ in range(10):
        e1.step(action=action)
    t0 = time.perf_counter()
    for _ in range(200):
        e1.step(action=action)
    dt = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(50):
        e1.reset(seed=54)
    dr = (time.perf_counter() - t0) / 50
    _record("reference_harness_restated", vector64_step_sps=sps, single_env_step_avg_time_s=dt, single_env_step_sps=1.0 / dt, single_env_reset_avg_time_s=dr)
    e1.close()


@pytest.mark.parametrize("n", [1, 5, 4097])
def test_ragged_batch_sizes(vec, n):
    """Any N (not a multiple of the 8 envs per workgroup; more envs than resident waves): last env == env alone."""
    rs = np.random.default_rng(n)
    acts = torch.tensor(rs.uniform(-0.4, 0.4, (3, 1, 69)), dtype=torch.float32, device="cuda:0")
    env = vec(n, autoreset=False)
    env.reset()
    for a in acts:
        env.step(a.expand(n, 69).contiguous())
    solo = vec(1, autoreset=False)
    solo.reset()
    for a in acts:
        solo.step(a.contiguous())
    torch.cuda.synchronize()
    assert torch.equal(env.qpos[n - 1], solo.qpos[0]) and torch.equal(env.obs_buf[n - 1], solo.obs_buf[0])
    assert torch.equal(env.qpos[0], solo.qpos[0])


def test_nan_action_autoreset_and_empty_mask(vec):
    env = vec(16, autoreset=False)
    env.reset()
    a = torch.zeros(16, 69, device=env.device); a[3, 7] = float("nan")
    env.step(a)
    torch.cuda.synchronize()
    assert int(env.nwarn[3]) >= 1 and int(env.nwarn.sum()) == int(env.nwarn[3])
    assert torch.isfinite(env.qpos).all() and torch.isfinite(env.qvel).all()
    q = env.qpos.clone()
    env.reset(mask=torch.zeros(16, dtype=torch.uint8, device=env.device))
    torch.cuda.synchronize()
    assert torch.equal(q, env.qpos)


def test_c_abi_error_paths_on_gpu():
    import ctypes as C
    from smplsim_amd import _cabi, _lib
    L = _lib.lib()
    st = _cabi.State(4)
    batch = C.c_void_p()
    assert L.ss_batch_create(None, None, C.byref(st), C.byref(batch)) == -1 and L.ss_last_error()
    assert L.ss_step(None, None, None, None, None, None, None, None) == -1


# ---------------------------------------------------------------- motion library / imitation
@pytest.mark.parametrize("filt", [True, False])
def test_motion_cook_on_gpu_matches_reference_vectors(filt):
    import test_motion_lib as T
    lib = T.make_lib(None, filt, device=0)
    torch.cuda.synchronize()
    T.check_cooked(lib, "f_" if filt else "n_")


def test_motion_lookup_on_gpu_matches_oracle_and_reference():
    import test_motion_lib as T
    lib = T.make_lib(None, device=0)
    T.check_blended(lib, np.random.default_rng(5), n=1000)
    st = lib.get_motion_state_intervaled(T.G["q_ids"], T.G["q_times"], offset=T.G["q_offset"])
    for k in ("root_pos", "root_rot", "root_vel", "xpos", "xquat", "body_vel", "qpos"):
        tol = 2e-3 if "vel" in k else 2e-5
        assert np.abs(_np(st[k]) - T.G["iv_" + k].reshape(st[k].shape)).max() < tol, k


def test_imitation_step_on_gpu_matches_oracle():
    import test_motion_lib as T
    from smplsim_amd import _lib
    lib = T.make_lib(None, device=0)
    T.check_imitation(lib, _lib.lib(), np.random.default_rng(11), n=37, device="cuda")
    T.check_imitation(lib, _lib.lib(), np.random.default_rng(12), n=1000, device="cuda")


def test_imitation_env_rollout_on_gpu_tracks_oracle():
    """Reference-state init + PD replay of the clip for a few control steps: the simulator state against the float64 oracle
    env started from the same state, the task observation / reward against motion_oracle on the GPU's own body state."""
    import test_motion_lib as T
    from oracle import motion_oracle as mo
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    lib = T.make_lib(None, device=0)
    n, J = 6, 24
    env = SMPLSimImitationVecEnv(n, lib, autoreset=False, seed=3)
    ids = np.array([0, 1, 2, 0, 1, 2], np.int32)
    t0 = np.array([0.1, 0.2, 0.3, 0.55, 0.4, 0.7], np.float32)
    env.offset[:, 2] = 0.05                                  # the synthetic clips are not height-fixed: lift them off the floor
    obs0, _ = env.reset(motion_ids=ids, start_times=t0)
    torch.cuda.synchronize()
    assert obs0.shape == (n, env.base.obs_size + 24 * J) and torch.isfinite(obs0).all()
    arr = T.lib_arrays(lib)
    off = _np(env.offset).astype(np.float64)
    oenvs = []
    for i in range(n):
        oe = O.OracleEnv(oracle_model(), state_init=O.INIT_EXTERNAL, self_obs_v=2, episode_length=10 ** 6)
        oe.data.qpos = _np(env.base.qpos)[i].astype(np.float64); oe.data.qvel = _np(env.base.qvel)[i].astype(np.float64)
        assert np.abs(oe.reset() - _np(obs0)[i, :env.base.obs_size]).max() < 2e-4
        oenvs.append(oe)
    want0 = mo.motion_state(arr, ids, t0.astype(np.float64), off)
    assert np.abs(_np(env.base.qpos)[:, :3] - want0["root_pos"]).max() < 1e-4
    for k in range(4):
        act = env.reference_actions()
        obs, rew, term, trunc, info = env.step(act)
        torch.cuda.synchronize()
        a = _np(act).astype(np.float64)
        for i, oe in enumerate(oenvs):
            oe.step(a[i])
            assert np.abs(oe.data.qpos - _np(env.base.qpos)[i]).max() < 5e-4, (k, i)
        times = t0 + np.float32((k + 1) * env.dt)
        assert np.abs(_np(env.times) - times).max() < 1e-6 and not _np(env.truncated).any()
        kx, km = env.base.kinematics()                       # the step launch's by-product == a separate mj_kinematics launch
        assert torch.equal(kx, env.xpos) and torch.equal(km, env.xmat)
        xpos, xmat, bv = _np(env.xpos).astype(np.float64), _np(env.xmat).astype(np.float64), _np(env.base.body_vel).astype(np.float64)
        quat = mo.matrix_to_quaternion(xmat.reshape(n, J, 3, 3))
        ref = mo.motion_state(arr, ids, times.astype(np.float64), off)
        fut = mo.motion_state(arr, ids, (times + np.float32(env.dt)).astype(np.float64), off)
        want_obs = mo.imitation_obs(xpos, quat, bv[..., :3], bv[..., 3:], fut["rg_pos"], fut["rb_rot"], fut["body_vel"], fut["body_ang_vel"])
        want_rew, _ = mo.imitation_reward(xpos, quat, bv[..., :3], bv[..., 3:], ref["rg_pos"], ref["rb_rot"], ref["body_vel"], ref["body_ang_vel"])
        assert np.abs(_np(obs)[:, env.base.obs_size:] - want_obs).max() < 5e-4
        assert np.abs(_np(rew) - want_rew).max() < 5e-5
        assert not trunc.any()


def test_imitation_env_autoreset_and_truncation_on_gpu():
    import test_motion_lib as T
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    lib = T.make_lib(None, device=0)
    env = SMPLSimImitationVecEnv(64, lib, seed=1)
    env.offset[:, 2] = 0.05
    obs, _ = env.reset()
    ended = torch.zeros(64, dtype=torch.bool, device=env.device)
    for k in range(45):                                      # the longest clip is 61 frames at 60 fps = 30 control steps
        obs, rew, term, trunc, info = env.step(env.reference_actions())
        done = term | trunc
        ended |= done
        # envs that just ended were re-initialised on a clip: time restarts, the final observation is kept for the learner
        assert (env.base.cur_t[done] == 0).all() and (env.base.cur_t[~done] > 0).all()
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and (rew >= 0).all() and (rew <= 1.0001).all()
        assert "final_observation" in info
    torch.cuda.synchronize()
    assert ended.all()
    assert ((env.start_times + env.base.cur_t * env.dt) <= env.motion_len + 1e-5).all()


@pytest.mark.parametrize("humanoid", ["smpl_humanoid", "smplx_humanoid"])
def test_fused_imitation_step_on_gpu_equals_the_launch_sequence(humanoid):
    """ss_imitation_step_fused (the whole imitation control step in one launch) against the six-launch sequence it replaces, with
    re-initialisations: identical flags / clip assignments, outputs to float32 round-off (two compilations of the same
    functions: -Os inside the stepper, -O3 in the motion kernels)."""
    import test_motion_lib as T
    from smplsim_amd.batch import ShardModel
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    if humanoid == "smpl_humanoid":
        lib, nu = T.make_lib(None, device=0), 69
    else:
        lib, nu = T.smplx_lib(None, device=0)[0], 153
    n = 96
    envs = [SMPLSimImitationVecEnv(n, lib, model=ShardModel(humanoid=humanoid, device=0), seed=5, fused=f, termination_distance=0.15) for f in (True, False)]
    assert envs[0].fused and not envs[1].fused
    for e in envs:
        e.offset[:, 2] = 0.05
    o = [e.reset()[0].clone() for e in envs]
    assert torch.equal(envs[0].motion_ids, envs[1].motion_ids) and torch.equal(envs[0].start_times, envs[1].start_times)
    assert (o[0] - o[1]).abs().max() < 1e-5
    g = torch.Generator().manual_seed(1)
    resets = 0
    for k in range(12):
        act = envs[0].reference_actions() if k % 3 else ((torch.rand(n, nu, generator=g) - 0.5) * 1.5).to(envs[0].device)
        (o1, r1, te1, tr1, i1), (o2, r2, te2, tr2, i2) = [e.step(act.clone()) for e in envs]
        torch.cuda.synchronize()
        assert torch.equal(te1, te2) and torch.equal(tr1, tr2), k
        assert torch.equal(envs[0].motion_ids, envs[1].motion_ids) and torch.equal(envs[0].start_times, envs[1].start_times), k
        assert torch.equal(envs[0].base.cur_t, envs[1].base.cur_t)
        # observation: TOL_OBS, the bound of the parity tests — the two step kernels are separate compilations of run_env, and the
        # velocity entries of the observation amplify their round-off after violent actions (measured 7e-4)
        assert (r1 - r2).abs().max() < 1e-5 and (o1 - o2).abs().max() < TOL_OBS and (i1["final_observation"] - i2["final_observation"]).abs().max() < TOL_OBS, k
        assert (envs[0].base.qpos - envs[1].base.qpos).abs().max() < TOL_QPOS
        envs[1].base.qpos.copy_(envs[0].base.qpos); envs[1].base.qvel.copy_(envs[0].base.qvel)      # keep round-off from accumulating into a flag flip
        envs[1].base.qpos_prev.copy_(envs[0].base.qpos_prev); envs[1].base.qvel_prev.copy_(envs[0].base.qvel_prev)
        envs[1].base.qacc_warm.copy_(envs[0].base.qacc_warm)
        resets += int((te1 | tr1).sum())
    assert resets >= 20


def test_fused_imitation_step_with_body_body_contacts_on_gpu():
    """SELFCOL + IMIT instantiation: the one-launch imitation step with body-body contacts against the launch sequence."""
    import test_motion_lib as T
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    lib = T.make_lib(None, device=0)
    n = 64
    envs = [SMPLSimImitationVecEnv(n, lib, seed=9, fused=f, termination_distance=0.2, self_collision=True) for f in (True, False)]
    assert envs[0].fused and not envs[1].fused
    for e in envs:
        e.offset[:, 2] = 0.05
    o = [e.reset()[0].clone() for e in envs]
    assert (o[0] - o[1]).abs().max() < 1e-5
    g = torch.Generator().manual_seed(4)
    contacts = resets = 0
    for k in range(6):
        act = ((torch.rand(n, 69, generator=g) - 0.5) * 2.0).to(envs[0].device)
        (o1, r1, te1, tr1, i1), (o2, r2, te2, tr2, i2) = [e.step(act.clone()) for e in envs]
        torch.cuda.synchronize()
        assert torch.equal(te1, te2) and torch.equal(tr1, tr2) and torch.equal(envs[0].motion_ids, envs[1].motion_ids), k
        assert torch.equal(envs[0].base.self_contacts, envs[1].base.self_contacts)
        assert (r1 - r2).abs().max() < 1e-5 and (o1 - o2).abs().max() < 2 * TOL_OBS and (envs[0].base.qpos - envs[1].base.qpos).abs().max() < TOL_QPOS_FREE, k
        for f in ("qpos", "qvel", "qpos_prev", "qvel_prev", "qacc_warm"):
            getattr(envs[1].base, f).copy_(getattr(envs[0].base, f))
        contacts += int(envs[0].base.self_contacts.sum()); resets += int((te1 | tr1).sum())
    assert contacts > 0 and resets > 0


def test_motion_lib_52_body_skeleton_on_gpu():
    import test_motion_lib as T
    from smplsim_amd import _lib
    lib, sk, clips = T.smplx_lib(None, device=0)
    T.check_smplx(lib, sk, clips, _lib.lib(), device="cuda")


def test_imitation_evaluate_scores_a_pd_replay_on_gpu():
    import test_motion_lib as T
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    lib = T.make_lib(None, device=0)
    env = SMPLSimImitationVecEnv(6, lib, seed=1, termination_distance=10.0)      # never terminates: all frames are scored
    env.offset[:, 2] = 0.05
    r = env.evaluate()
    assert r["success_rate"] == 1.0 and r["num_clips"] == 6 and r["frames_scored"] > 50
    for k in ("mpjpe_g", "mpjpe_l", "mpjpe_pa", "vel_dist", "accel_dist"):
        assert np.isfinite(r[k]) and r[k] >= 0
    assert r["mpjpe_pa"] <= r["mpjpe_l"] + 1e-3 <= r["mpjpe_g"] + 2e-3              # alignment can only reduce the error
    env2 = SMPLSimImitationVecEnv(6, lib, seed=1, termination_distance=1e-4)      # terminates at once
    env2.offset[:, 2] = 0.05
    assert env2.evaluate()["success_rate"] == 0.0


def test_smplx_imitation_env_on_gpu():
    """52-body model: the BODYOUT instantiation of the SMPL-X step kernel + the 64-lanes-per-env imitation kernel."""
    import test_motion_lib as T
    from oracle import motion_oracle as mo
    from smplsim_amd.batch import ShardModel
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    lib, sk, clips = T.smplx_lib(None, device=0)
    n, J = 5, 52
    env = SMPLSimImitationVecEnv(n, lib, model=ShardModel(humanoid="smplx_humanoid", device=0), autoreset=False, seed=2)
    ids = np.array([0, 1, 0, 1, 0], np.int32)
    t0 = np.array([0.0, 0.1, 0.3, 0.2, 0.5], np.float32)
    obs, _ = env.reset(motion_ids=ids, start_times=t0)
    assert obs.shape == (n, env.base.obs_size + 24 * J)
    om = oracle_model("smplx_humanoid")
    oenvs = []
    for i in range(n):
        oe = O.OracleEnv(om, state_init=O.INIT_EXTERNAL, self_obs_v=2, episode_length=10 ** 6)
        oe.data.qpos = _np(env.base.qpos)[i].astype(np.float64); oe.data.qvel = _np(env.base.qvel)[i].astype(np.float64)
        assert np.abs(oe.reset() - _np(obs)[i, :env.base.obs_size]).max() < 5e-4
        oenvs.append(oe)
    arr = T.lib_arrays(lib)
    for k in range(2):
        act = env.reference_actions()
        obs, rew, term, trunc, _ = env.step(act)
        torch.cuda.synchronize()
        for i, oe in enumerate(oenvs):
            oe.step(_np(act)[i].astype(np.float64))
            assert np.abs(oe.data.qpos - _np(env.base.qpos)[i]).max() < 1e-3, (k, i)
        kx, km = env.base.kinematics()
        assert torch.equal(kx, env.xpos) and torch.equal(km, env.xmat)
        times = (t0 + np.float32((k + 1) * env.dt)).astype(np.float64)
        xpos, bv = _np(env.xpos).astype(np.float64), _np(env.base.body_vel).astype(np.float64)
        quat = mo.matrix_to_quaternion(_np(env.xmat).astype(np.float64).reshape(n, J, 3, 3))
        ref, fut = mo.motion_state(arr, ids, times), mo.motion_state(arr, ids, times + np.float32(env.dt))
        want_obs = mo.imitation_obs(xpos, quat, bv[..., :3], bv[..., 3:], fut["rg_pos"], fut["rb_rot"], fut["body_vel"], fut["body_ang_vel"])
        want_rew, _ = mo.imitation_reward(xpos, quat, bv[..., :3], bv[..., 3:], ref["rg_pos"], ref["rb_rot"], ref["body_vel"], ref["body_ang_vel"])
        assert np.abs(_np(obs)[:, env.base.obs_size:] - want_obs).max() < 1e-3 and np.abs(_np(rew) - want_rew).max() < 1e-4


def test_self_collision_with_per_env_shapes_on_gpu(vec):
    """Body-body contacts with one geom table per body shape (SHAPED + SELFCOL instantiation): every env of the mixed batch
    against the same env in a single-shape self-collision batch (another instantiation of the same kernel source: round-off)."""
    from smplsim_amd.batch import ShardModel
    from smplsim_amd.mjcf_writer import scaled_xml_str
    xmls = [scaled_xml_str("smpl_humanoid", 1.0), scaled_xml_str("smpl_humanoid", 0.9, {"L_Knee": 1.1, "R_Knee": 1.1}),
            scaled_xml_str("smpl_humanoid", 1.1, {"Chest": 0.9, "L_Elbow": 1.2})]
    sid = torch.tensor([0, 1, 2, 2, 1, 0, 0, 1, 2])
    mixed = vec(9, model=ShardModel(xmls=xmls), shape_id=sid, autoreset=False, self_collision=True)
    solos = [(torch.nonzero(sid == s)[:, 0].to(mixed.device), vec(3, model=ShardModel(xml=xmls[s]), autoreset=False, self_collision=True)) for s in range(3)]
    obs0 = mixed.reset()[0]
    for idx, so in solos:
        assert (so.reset()[0] - obs0[idx]).abs().max() < 1e-6
    g = torch.Generator().manual_seed(3)
    seen = 0
    nwarn_seen = mixed.nwarn.clone()
    for k in range(6):
        act = (torch.rand(1, 69, generator=g) * 2 - 1).repeat(9, 1).to(mixed.device)
        obs = mixed.step(act)[0]
        for idx, so in solos:
            o2 = so.step(act[idx])[0]
            assert torch.equal(so.self_contacts, mixed.self_contacts[idx]) or bool((mixed.nwarn[idx] != nwarn_seen[idx]).any()), k
            # two instantiations (fixed / runtime layout): round-off, relative to the velocity scale; an env whose state blew up inside
            # the step (MuJoCo's bad-state reset, counted in nwarn) crosses the 1e10 threshold one mj_step earlier or later depending on
            # rounding and is not compared (as in the per-sample tests)
            calm = (mixed.nwarn[idx] == nwarn_seen[idx]) & (so.nwarn == nwarn_seen[idx])
            nwarn_seen[idx] = mixed.nwarn[idx]; so.nwarn.copy_(mixed.nwarn[idx])
            if bool(calm.any()):
                vmax = max(1.0, float(mixed.qvel[idx][calm].abs().max()))
                assert (so.qpos - mixed.qpos[idx])[calm].abs().max() < TOL_QPOS_FREE and (o2 - obs[idx])[calm].abs().max() < 2 * TOL_OBS * vmax, (k, vmax)
            so.qpos.copy_(mixed.qpos[idx]); so.qvel.copy_(mixed.qvel[idx]); so.qacc_warm.copy_(mixed.qacc_warm[idx])
            so.qpos_prev.copy_(mixed.qpos_prev[idx]); so.qvel_prev.copy_(mixed.qvel_prev[idx])
        seen += int(mixed.self_contacts.sum())
    assert seen > 0


def test_shape_varied_env_groups_match_their_oracles_and_standalone_envs(vec):
    """Two body shapes in one shard: every group against the oracle built from ITS MJCF, and bit-identical
    to the same group stepped as a standalone env (the grouping / streams change nothing)."""
    from helpers import FEET, pd_tables
    from smplsim_amd.batch import ShardModel
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import scaled_xml_str
    from smplsim_amd.shapes import ShapeVariedVecEnv
    xmls = [scaled_xml_str("smpl_humanoid", 1.0), scaled_xml_str("smpl_humanoid", 0.85, {"L_Knee": 1.15, "R_Knee": 1.15, "Chest": 0.9})]
    env = ShapeVariedVecEnv(xmls, 4, autoreset=False, seed=0, single_launch=False)
    one = ShapeVariedVecEnv(xmls, 4, autoreset=False, seed=0)              # per-env shapes in one launch
    solo = [vec(4, model=ShardModel(xml=x), autoreset=False, seed=1000 * g) for g, x in enumerate(xmls)]
    obs, _ = env.reset()
    assert torch.equal(one.reset()[0], obs)
    for s in solo:
        s.reset()
    oenvs = []
    for x in xmls:
        mc = compile_mjcf(x)
        om = O.OracleModel(x, *pd_tables(mc), legal_bodies=FEET, timestep=1.0 / 450)
        oe = O.OracleEnv(om)
        oenvs.append(oe)
    torch.cuda.synchronize()
    assert np.abs(oenvs[0].reset() - _np(obs)[0]).max() < 1e-5 and np.abs(oenvs[1].reset() - _np(obs)[4]).max() < 1e-5
    assert np.abs(_np(obs)[0] - _np(obs)[4]).max() > 1e-3                 # the shapes really differ
    rs = np.random.default_rng(4)
    for k in range(6):
        a = rs.uniform(-0.3, 0.3, (2, 69))
        act = torch.tensor(np.repeat(a, 4, axis=0), device=env.device, dtype=torch.float32)
        obs, rew, term, trunc, info = env.step(act)
        obs1, rew1 = one.step(act)[:2]
        assert torch.equal(obs1, obs) and torch.equal(rew1, rew) and torch.equal(one.state()[0], env.state()[0])
        for g, s in enumerate(solo):
            s.step(act[4 * g:4 * g + 4])
        torch.cuda.synchronize()
        qpos, qvel = env.state()
        for g in range(2):
            o_ref, r, te, tu = oenvs[g].step(a[g])
            assert np.abs(_np(qpos)[4 * g] - oenvs[g].data.qpos).max() < TOL_QPOS_FREE and np.abs(_np(obs)[4 * g] - o_ref).max() < TOL_OBS, (k, g)
            assert torch.equal(qpos[4 * g:4 * g + 4], solo[g].qpos) and torch.equal(obs[4 * g:4 * g + 4], solo[g].obs_buf)
    assert obs.shape == (8, env.obs_size) and rew.shape == (8,)


def test_imitation_with_per_env_shapes_and_per_clip_offsets():
    """PHC-style setup: every env has its own body shape and tracks a clip cooked with THAT shape's joint offsets.  After the
    reference-state init the simulator's body positions (kinematics of the shaped model) must coincide with the clip's
    (forward kinematics of the motion library with per-clip offsets) — two independent FK implementations, two tables."""
    import test_motion_lib as T
    from smplsim_amd.batch import ShardModel
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    from smplsim_amd.mjcf_writer import scaled_xml_str
    from smplsim_amd.motion_lib import MotionLibSMPL, Skeleton
    xmls = [scaled_xml_str("smpl_humanoid", 1.0), scaled_xml_str("smpl_humanoid", 0.9, {"L_Knee": 1.1, "R_Knee": 1.1}),
            scaled_xml_str("smpl_humanoid", 1.1, {"Chest": 0.9})]
    model = ShardModel(xmls=xmls, device=0)
    sks = [Skeleton.from_model_const(mc) for mc in model.mcs]
    lib = MotionLibSMPL(T.clip_dict(), sks[0], device=0)
    lib.load_motions(random_sample=False, offsets=np.stack([sk.offsets for sk in sks]))       # clip m <-> shape m
    n = 9
    ids = np.arange(n, dtype=np.int32) % 3
    env = SMPLSimImitationVecEnv(n, lib, model=model, shape_id=ids, autoreset=False, seed=0)
    env.offset[:, 2] = 0.3
    # on exact frames (blend 0): between frames the clip's qpos (Euler angles interpolated linearly) and its body positions
    # (interpolated linearly themselves) are not the same pose, in the reference as here
    frame = np.array([3, 6, 9, 4, 7, 10, 5, 8, 11])
    t0 = (frame / np.asarray(T.G["fps"])[ids]).astype(np.float32)
    env.reset(motion_ids=ids, start_times=t0)
    ref = lib.get_motion_state(ids, t0, offset=env.offset)
    torch.cuda.synchronize()
    # 5e-4: the motion library rounds the joint offsets to 5 decimals like the reference's update_model, the model does not
    assert np.abs(_np(env.xpos) - _np(ref["rg_pos"])).max() < 5e-4
    d01 = np.abs(_np(lib.gts)[0] - _np(lib.gts)[int(lib.length_starts[1])]).max()
    assert d01 > 1e-2                                        # the shapes / clips really differ
    obs, rew, term, trunc, _ = env.step(env.reference_actions())
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert (rew[torch.as_tensor(ids == 0, device=rew.device)] > 0.5).all()                       # the gentle clip: one PD step from its own state stays close to it


def test_external_init_refuses_autoreset(vec):
    with pytest.raises(ValueError, match="External"):
        vec(4, state_init="External")
    env = vec(4, state_init="External", autoreset=False)
    env.set_state(np.tile(default_qpos(76), (4, 1)), np.zeros((4, 75)))
    obs, _ = env.reset()
    ref = vec(4, autoreset=False)
    assert torch.equal(ref.reset()[0], obs)              # the Default pose written by hand == StateInit Default


# ---------------------------------------------------------------------------------------------- policy inference on the matrix cores
@pytest.mark.parametrize("M,N,K,act", [(4096, 2048, 320, "silu"), (1000, 512, 1024, "tanh"), (37, 69, 512, "none"), (256, 1536, 2048, "relu"),
                                       (16000, 1024, 512, "silu")])       # (the last: 252 tiles of 256 x 256 -> the big-batch kernel, ragged M)
def test_linear_bf16_mfma_kernel(M, N, K, act):
    """ss_linear_bf16 against torch on the same bf16-rounded operands with fp32 accumulation: asymmetric operands (a row / column
    swap or a wrong fragment layout cannot pass), ragged M and N, both tile widths, every epilogue."""
    import ctypes as C
    from smplsim_amd import _cabi
    from smplsim_amd._lib import lib
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 0.5 + torch.linspace(-1, 1, K)[None, :] * 0.3).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5 + torch.linspace(-1, 1, N)[:, None] * 0.02).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    ref = x.float() @ w.float().T + b
    ref = {"silu": torch.nn.functional.silu, "tanh": torch.tanh, "relu": torch.relu, "none": lambda t: t}[act](ref)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    y32 = torch.full((M, N + 3), 7.0, device="cuda")
    assert lib().ss_linear_bf16(ptr(x), ptr(w), ptr(b), ptr(y32), M, N, K, N + 3, _cabi.ACTIVATIONS[act], 1, st) == 0
    torch.cuda.synchronize()
    assert (y32[:, N:] == 7.0).all()                                 # nothing written beyond column N
    err = (y32[:, :N] - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err        # fp32 accumulation in another order
    y16 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    assert lib().ss_linear_bf16(ptr(x), ptr(w), ptr(b), ptr(y16), M, N, K, N, _cabi.ACTIVATIONS[act], 0, st) == 0
    torch.cuda.synchronize()
    assert (y16.float() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())
    assert lib().ss_linear_bf16(ptr(x), ptr(w), ptr(b), ptr(y16), M, N, K - 1, N, 0, 0, st) == -1     # K must be a multiple of 32


@pytest.mark.parametrize("M,N,K,act", [(1024, 1536, 2048, "silu"), (300, 192, 128, "tanh"), (2048, 64, 512, "none"), (1000, 1000, 64, "relu")])
def test_linear_bf16_train_kernel_outputs(M, N, K, act):
    """ss_linear_bf16_train, bf16 form: result, TRANSPOSED result and activation derivative of one launch against torch on the same bf16
    operands (asymmetric: a row / column swap cannot pass), with and without the multiplying operand, ragged M and N."""
    import ctypes as C
    from smplsim_amd import _cabi
    from smplsim_amd._lib import lib
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 0.5 + torch.linspace(-1, 1, K)[None, :] * 0.3).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5 + torch.linspace(-1, 1, N)[:, None] * 0.02).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    mul = (torch.rand(M, N, generator=g) + 0.5).to(torch.bfloat16).cuda()
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fn = {"silu": torch.nn.functional.silu, "tanh": torch.tanh, "relu": torch.relu, "none": lambda t: t}[act]
    for use_mul in (False, True):
        z = x.float() @ w.float().T + b
        if use_mul:
            z = z * mul.float()
        zz = z.clone().requires_grad_(True)
        ref = fn(zz)
        dref, = torch.autograd.grad(ref.sum(), zz)
        ldt = (M + 7) // 8 * 8
        y = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); yt = torch.zeros(N, ldt, dtype=torch.bfloat16, device="cuda")
        d = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        assert lib().ss_linear_bf16_train(ptr(x), ptr(w), ptr(b), ptr(mul if use_mul else None), ptr(y), ptr(yt), ptr(d), M, N, K, N, ldt,
                                          _cabi.ACTIVATIONS[act], 0, st) == 0
        torch.cuda.synchronize()
        tol = 1e-2 * max(1.0, ref.abs().max().item())
        assert (y.float() - ref.detach()).abs().max().item() < tol
        assert torch.equal(yt[:, :M], y.t())                            # the transposed copy is the same rounding of the same numbers
        assert (d.float() - dref).abs().max().item() < 2e-2
    assert lib().ss_linear_bf16_train(ptr(x), ptr(w), ptr(b), None, ptr(y), None, None, M, N, K - 32, N, 0, 0, 0, st) == -1   # K: multiples of 64


@pytest.mark.parametrize("M,N,K,act", [(2048, 1536, 2048, "silu"), (300, 260, 128, "tanh"), (4096, 512, 1024, "none"), (1000, 1000, 384, "relu")])
def test_gemm256_kernel_outputs(M, N, K, act, monkeypatch):
    """The 256 x 256 macro-tile kernel behind ss_linear_bf16_train (csrc/ss_gemm256.h: staggered wave rows, copies in flight across barriers):
    the same three outputs against torch on asymmetric bf16 operands, ragged M and N, the shortest K loop it takes (two K tiles), and the same
    bits from every one of 20 launches (a copy that lands after its reader shows up as a launch that differs)."""
    import ctypes as C
    from smplsim_amd import _cabi
    from smplsim_amd._lib import lib
    monkeypatch.setenv("SS_MLP_TRAIN_256", "1")
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 0.5 + torch.linspace(-1, 1, K)[None, :] * 0.3).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5 + torch.linspace(-1, 1, N)[:, None] * 0.02).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    mul = (torch.rand(M, N, generator=g) + 0.5).to(torch.bfloat16).cuda()
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fn = {"silu": torch.nn.functional.silu, "tanh": torch.tanh, "relu": torch.relu, "none": lambda t: t}[act]
    ldt = (M + 7) // 8 * 8
    for use_mul in (False, True):
        z = x.float() @ w.float().T + b
        if use_mul:
            z = z * mul.float()
        zz = z.clone().requires_grad_(True)
        ref = fn(zz)
        dref, = torch.autograd.grad(ref.sum(), zz)
        first = None
        for rep in range(20):
            y = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); yt = torch.zeros(N, ldt, dtype=torch.bfloat16, device="cuda")
            d = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            assert lib().ss_linear_bf16_train(ptr(x), ptr(w), ptr(b), ptr(mul if use_mul else None), ptr(y), ptr(yt), ptr(d), M, N, K, N, ldt,
                                              _cabi.ACTIVATIONS[act], 0, st) == 0
            torch.cuda.synchronize()
            if first is None:
                first = (y.clone(), yt.clone(), d.clone())
                tol = 1e-2 * max(1.0, ref.abs().max().item())
                assert (y.float() - ref.detach()).abs().max().item() < tol
                assert torch.equal(yt[:, :M], y.t())
                assert (d.float() - dref).abs().max().item() < 2e-2
            else:
                assert torch.equal(y, first[0]) and torch.equal(yt, first[1]) and torch.equal(d, first[2]), rep
    # the accumulating form: K split into shares of an even number of tiles, against the fp32 product
    yf = torch.zeros(M, N + 3, device="cuda")
    assert lib().ss_linear_bf16_train(ptr(x), ptr(w), None, None, ptr(yf), None, None, M, N, K, N + 3, 0, 0, 1, st) == 0
    torch.cuda.synchronize()
    ref = x.float() @ w.float().T
    assert (yf[:, N:] == 0).all()
    assert (yf[:, :N] - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(2048, 1536, 53248), (69, 512, 4096 * 3), (1024, 320, 8192), (1024, 1600, 53248), (512, 576, 53248 + 128)])   # (the last two: 256-tile kernel, K shares rounded to even tile counts)
def test_linear_bf16_train_kernel_split_k_accumulates(M, N, K):
    """The accumulating fp32 form (weight gradients: few outputs, the batch as K): the K split's partial sums meet in the output by
    atomics; against the fp32 product of the same bf16 operands.  Accumulates: a second call doubles the result."""
    import ctypes as C
    from smplsim_amd._lib import lib
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.5 + torch.linspace(-1, 1, N)[:, None] * 0.1).to(torch.bfloat16).cuda()
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    y = torch.zeros(M, N + 5, device="cuda")
    assert lib().ss_linear_bf16_train(ptr(x), ptr(w), None, None, ptr(y), None, None, M, N, K, N + 5, 0, 0, 1, st) == 0
    torch.cuda.synchronize()
    ref = x.float() @ w.float().T
    assert (y[:, N:] == 0).all()
    assert (y[:, :N] - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    assert lib().ss_linear_bf16_train(ptr(x), ptr(w), None, None, ptr(y), None, None, M, N, K, N + 5, 0, 0, 1, st) == 0
    torch.cuda.synchronize()
    assert (y[:, :N] - 2 * ref).abs().max().item() < 4e-3 * ref.abs().max().item()


@pytest.mark.parametrize("Mb,n_out,n_in,ldz,ldh", [(53248, 1536, 2048, 1536, 2048), (1024, 72, 520, 128, 576), (4096, 1000, 264, 1024, 320), (256, 8, 8, 8, 8)])
def test_wgrad_bf16_reads_both_operands_untransposed(Mb, n_out, n_in, ldz, ldh):
    """ss_wgrad_bf16: dW += dZ^T h with dZ [Mb, ldz] and h [Mb, ldh] as they lie (contraction over their rows, fragments by the LDS transpose read) against
    the fp32 product of the same bf16 operands; asymmetric operands (a swap of the two cannot pass), widths that are not multiples of the 256-wide tile, row strides
    wider than the used columns (what lies beyond them must not leak in), the K split's shares rounded to even tile counts, accumulation into dW."""
    import ctypes as C
    from smplsim_amd._lib import lib
    g = torch.Generator().manual_seed(Mb + n_out + n_in)
    dz = (torch.randn(Mb, ldz, generator=g) * 0.5 + torch.linspace(-1, 1, ldz)[None, :] * 0.2).to(torch.bfloat16).cuda()
    h = (torch.randn(Mb, ldh, generator=g) * 0.5 + torch.linspace(1, -1, Mb)[:, None] * 0.2).to(torch.bfloat16).cuda()
    ptr = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dw = torch.zeros(n_out, n_in + 3, device="cuda")
    assert lib().ss_wgrad_bf16(ptr(dz), ptr(h), ptr(dw), Mb, n_out, n_in, ldz, ldh, n_in + 3, st) == 0
    torch.cuda.synchronize()
    ref = dz[:, :n_out].float().t() @ h[:, :n_in].float()
    assert (dw[:, n_in:] == 0).all()
    assert (dw[:, :n_in] - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())
    assert lib().ss_wgrad_bf16(ptr(dz), ptr(h), ptr(dw), Mb, n_out, n_in, ldz, ldh, n_in + 3, st) == 0
    torch.cuda.synchronize()
    assert (dw[:, :n_in] - 2 * ref).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())
    assert lib().ss_wgrad_bf16(ptr(dz), ptr(h), ptr(dw), Mb - 64, n_out, n_in, ldz, ldh, n_in + 3, st) == -1      # the batch: multiples of 128 rows


def test_linear_bf16_dx_result_and_column_sums():
    """ss_linear_bf16_dx: y = (x W^T) * mul in bf16 and the column sums of the fp32 result from the same launch (the bias gradient of the layer below), against torch;
    ragged M; refused (not silently computed without the sums) on shapes the 256 x 256 kernel does not serve."""
    import ctypes as C
    from smplsim_amd._lib import lib
    M, N, K = 5000, 768, 384
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5 + torch.linspace(-1, 1, N)[:, None] * 0.02).to(torch.bfloat16).cuda()
    mul = (torch.rand(M, N, generator=g) + 0.5).to(torch.bfloat16).cuda()
    ptr = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    y = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); cs = torch.zeros(N, device="cuda")
    assert lib().ss_linear_bf16_dx(ptr(x), ptr(w), ptr(mul), ptr(y), ptr(cs), M, N, K, N, st) == 0
    torch.cuda.synchronize()
    ref = (x.float() @ w.float().T) * mul.float()
    assert (y.float() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())
    assert (cs - ref.sum(0)).abs().max().item() < 2e-3 * max(1.0, ref.sum(0).abs().max().item())
    assert lib().ss_linear_bf16_dx(ptr(x), ptr(w), ptr(mul), ptr(y), ptr(cs), 1000, N, K, N, st) == -1


def test_fused_mlp_train_gradients_match_autograd():
    """learning.fused_train.FusedMLPTrain (forward + backward on ss_linear_bf16_train) against torch autograd over the same layers in
    fp32: outputs and every parameter gradient agree to bf16 round-off through the stack (relative to the gradient's own scale), on the
    PPO losses' shapes: the policy's 69-wide head and the value's 1-wide head, a batch that is not a multiple of 64."""
    from smplsim_amd.learning.fused_train import FusedMLPTrain
    from smplsim_amd.learning.networks import MLP
    torch.manual_seed(0)
    for out_dim in (69, 1):
        net = MLP(292, (512, 256, 256), "silu").cuda()
        head = torch.nn.Linear(256, out_dim).cuda()
        x = torch.randn(1000, 292, device="cuda")
        target = torch.randn(1000, out_dim, device="cuda")
        fused = FusedMLPTrain(net.affine_layers, head, "silu")
        params = [p for l in list(net.affine_layers) + [head] for p in (l.weight, l.bias)]
        y_ref = head(net(x))
        g_ref = torch.autograd.grad((y_ref - target).pow(2).mean(), params)
        y = fused(x)
        g = torch.autograd.grad((y - target).pow(2).mean(), params)
        assert (y - y_ref).abs().max().item() < 3e-2 * max(1.0, y_ref.abs().max().item())
        for a, b_, p in zip(g, g_ref, params):
            assert a.shape == p.shape
            rel = (a - b_).norm().item() / max(b_.norm().item(), 1e-12)
            assert rel < 3e-2, (out_dim, tuple(p.shape), rel)


def test_ppo_update_on_the_librarys_gemm_follows_the_torch_update():
    """AgentPPO(mfma_update=True): one update_params on the same rollout as the fp32 torch update — the losses agree to bf16 round-off
    and the parameters move the same way (cosine of the two parameter steps > 0.98 for both networks)."""
    from smplsim_amd.agents.ppo import AgentPPO, PPOConfig
    from smplsim_amd.batch import SMPLSimVecEnv
    kw = dict(hidden=(256, 128, 128), min_batch_size=256 * 8, opt_num_epochs=2)
    env = SMPLSimVecEnv(256, task="HumanoidSpeed", autoreset=True, seed=3)
    a0 = AgentPPO(env, PPOConfig(**kw), seed=1)
    a1 = AgentPPO(env, PPOConfig(mfma_update=True, **kw), seed=1)
    a1.policy_net.load_state_dict(a0.policy_net.state_dict()); a1.value_net.load_state_dict(a0.value_net.state_dict())
    batch = a0.sample()
    before = [p.detach().clone() for p in list(a0.policy_net.parameters()) + list(a0.value_net.parameters())]
    i0 = a0.update_params({k: v.clone() for k, v in batch.items()})
    i1 = a1.update_params({k: v.clone() for k, v in batch.items()})
    assert abs(float(i0["value_loss"]) - float(i1["value_loss"])) < 3e-2 * max(1.0, abs(float(i0["value_loss"])))
    assert abs(float(i0["surr_loss"]) - float(i1["surr_loss"])) < 3e-2
    p0 = list(a0.policy_net.parameters()) + list(a0.value_net.parameters()); p1 = list(a1.policy_net.parameters()) + list(a1.value_net.parameters())
    d0 = torch.cat([(p - b).flatten() for p, b in zip(p0, before) if p.requires_grad])
    d1 = torch.cat([(p - b).flatten() for p, b in zip(p1, before) if p.requires_grad])
    cos = float((d0 * d1).sum() / (d0.norm() * d1.norm()))
    assert cos > 0.9, cos                                               # (Adam's first steps are sign-like: small gradient entries may flip)
    env.close()


def test_fused_policy_inference_matches_the_torch_policy():
    """FusedPolicyInference (obs clamp + RunningNorm + 7 fused bf16 layers) against PolicyGaussian in fp32: the action means agree
    to bf16 round-off through the 7 layers, with the normalisation on (n > 0) and off (n = 0), on a strided observation tensor."""
    from smplsim_amd.learning.fast_policy import FusedPolicyInference
    from smplsim_amd.learning.networks import PolicyGaussian
    torch.manual_seed(0)
    pol = PolicyGaussian(289, 69).cuda().eval()
    big = torch.randn(1500, 300, device="cuda") * 3.0
    obs = big[:, 5:294]                                              # row stride 300
    fast = FusedPolicyInference(pol, (-5.0, 5.0))
    for trained in (False, True):
        if trained:
            pol.train(); pol.norm(obs.clamp(-5, 5)); pol.eval()     # running statistics from one batch
            with torch.no_grad():
                for l in pol.net.affine_layers:
                    l.weight.mul_(1.5)
            fast.refresh()
        with torch.no_grad():
            want = pol.select_action(obs.clamp(-5, 5), mean_action=True)
        got = fast.select_action(obs, mean_action=True)
        torch.cuda.synchronize()
        scale = want.abs().max().item()
        assert (got - want).abs().max().item() < 0.03 * scale + 1e-3, ((got - want).abs().max().item(), scale)
    g1 = torch.Generator(device="cuda").manual_seed(5); g2 = torch.Generator(device="cuda").manual_seed(5)
    a = fast.select_action(obs, generator=g1)
    n = torch.randn(a.shape, device="cuda", generator=g2)
    assert torch.allclose(a, fast.mean(obs) + pol.action_log_std.exp() * n, atol=1e-6)
    # the behaviour policy's own log-density of the drawn action (what the PPO ratio divides by when the sampler runs here)
    g3 = torch.Generator(device="cuda").manual_seed(5)
    a2, logp = fast.select_action(obs, generator=g3, return_log_prob=True)
    z = (a2 - fast.mean(obs)) * torch.exp(-pol.action_log_std)
    want_lp = (-0.5 * z * z - pol.action_log_std - 0.5 * np.log(2 * np.pi)).sum(1, keepdim=True)
    assert torch.equal(a2, a) and logp.shape == (1500, 1) and torch.allclose(logp, want_lp, atol=2e-3)


def test_gaussian_sample_kernel_matches_the_torch_expressions():
    """ss_gaussian_sample (the sampler's Gaussian head in one launch) against what it replaces: PolicyGaussian.select_action's draw
    (reference policy_gaussian.py:25-41 -> DiagGaussian.sample), Agent.preprocess_actions' clip (agents/agent.py:153-161) and
    normal_log_density of the draw — on a strided destination (a row block of a rollout tensor), M not a multiple of 4."""
    import math
    from smplsim_amd.learning.fast_policy import FusedPolicyInference
    from smplsim_amd.learning.networks import PolicyGaussian
    torch.manual_seed(3)
    pol = PolicyGaussian(289, 69, (64, 32)).cuda().eval()
    with torch.no_grad():
        pol.action_log_std.copy_(torch.linspace(-3.0, -0.5, 69, device="cuda")[None])
    fast = FusedPolicyInference(pol)
    M = 1001
    mean, noise = torch.randn(M, 69, device="cuda"), torch.randn(M, 69, device="cuda")
    rollout = torch.zeros(3, 2 * M, 69, device="cuda")
    a_env, logp = torch.zeros(M, 69, device="cuda"), torch.zeros(M, 1, device="cuda")
    fast.sample_into(mean, noise, rollout[1, M:], a_env, (-1.0, 1.0), logp)
    torch.cuda.synchronize()
    ls = pol.action_log_std.detach()
    ref = mean + ls.exp() * noise
    assert torch.allclose(rollout[1, M:], ref, rtol=0, atol=2e-6) and float(rollout[0].abs().max()) == 0 and float(rollout[1, :M].abs().max()) == 0
    assert torch.equal(a_env, rollout[1, M:].clamp(-1.0, 1.0))
    ref_lp = (-0.5 * noise.pow(2) - 0.5 * math.log(2.0 * math.pi) - ls).sum(1, keepdim=True)
    assert torch.allclose(logp, ref_lp, rtol=1e-5, atol=1e-4)


def test_env_built_before_fork_runs_in_worker_processes():
    """The reference's sampler builds the env once and forks its workers (agents/agent.py:121-145, num_threads > 1).  HumanoidEnv
    creates its device state lazily in the process that uses it: two forked workers and then the parent step the same env object,
    each in its own HIP context, and get the same trajectory (run in a fresh interpreter: this one has a HIP context already)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fork_workers_check.py")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fork workers ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
