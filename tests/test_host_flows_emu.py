"""The Python host logic of the batched envs (smplsim_amd/batch.py, imitation.py, shapes.py) run end to end in the GPU-less
container: the `emu_backend` fixture (tests/conftest.py) monkeypatches the package's library handle, device choice and
stream lookup onto the CPU emulator build of the same C ABI and host tensors — the package itself has no backend switch.
The GPU twins are in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from helpers import oracle_model
from oracle import oracle as O


def test_vec_env_autoreset_flows_on_the_emulator(emu_backend):
    from smplsim_amd.batch import SMPLSimVecEnv
    n = 5
    fused = SMPLSimVecEnv(n, task="HumanoidSpeed", episode_length=2, seed=3)                       # in-launch Default autoreset
    plain = SMPLSimVecEnv(n, task="HumanoidSpeed", episode_length=2, seed=3, fused_autoreset=False)  # step + masked reset
    assert fused.device.type == "cpu" and fused._fused_autoreset and not plain._fused_autoreset
    o1, _ = fused.reset(); o2, _ = plain.reset()
    assert torch.equal(o1, o2)
    oenv = O.OracleEnv(oracle_model(), task=O.TASK_SPEED, episode_length=2)
    rs = np.random.default_rng(0)
    for k in range(4):
        a = torch.tensor(rs.uniform(-0.3, 0.3, (n, 69)), dtype=torch.float32)
        tr = torch.rand(n, 4, generator=torch.Generator().manual_seed(k))
        ob1, r1, t1, u1, i1 = fused.step(a, task_rand=tr)
        ob2, r2, t2, u2, i2 = plain.step(a, task_rand=tr)
        assert torch.equal(r1, r2) and torch.equal(t1, t2) and torch.equal(u1, u2)
        assert torch.equal(i1["final_observation"], i2["final_observation"])
        assert bool(u1.all()) == (k % 3 == 2)                   # cur_t > 2 on every third step, then the envs start over
    assert int(fused.cur_t.max()) <= 3


@pytest.mark.parametrize("fused", [True, False])
def test_imitation_env_end_to_end_on_the_emulator(emu_backend, fused):
    import test_motion_lib as T
    from oracle import motion_oracle as mo
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    lib = T.make_lib(emu_backend)
    n, J = 4, 24
    env = SMPLSimImitationVecEnv(n, lib, seed=2, fused=fused)
    assert env.fused == fused
    env.offset[:, 2] = 0.05
    ids = np.array([0, 1, 2, 0], np.int32)
    t0 = np.array([0.1, 0.2, 0.3, 1.25], np.float32)           # env 3 starts one step before the end of its 1.3 s clip
    obs, _ = env.reset(motion_ids=ids, start_times=t0)
    assert obs.shape == (n, env.base.obs_size + 24 * J)
    arr = T.lib_arrays(lib)
    act = env.reference_actions()
    pre_ids = env.motion_ids.clone()
    obs, rew, term, trunc, info = env.step(act)
    # reward / flags of the step against the oracle on the body state the step launch wrote (before the re-initialisation)
    times = t0 + np.float32(env.dt)
    ref = mo.motion_state(arr, ids, times.astype(np.float64), env.offset.numpy().astype(np.float64))
    assert trunc.tolist() == [False, False, False, True] and "final_observation" in info
    assert np.isfinite(rew.numpy()).all() and (rew.numpy() > 0).all()
    done = (term | trunc).numpy()
    assert (env.base.cur_t.numpy()[done] == 0).all() and (env.base.cur_t.numpy()[~done] == 1).all()
    # finished envs were re-initialised on a (new) clip: their simulator state is that clip's state at the new start time
    st = lib.get_motion_state(env.motion_ids, env.start_times, offset=env.offset, with_qpos=True)
    assert np.allclose(env.base.qpos.numpy()[done], st["qpos"].numpy()[done], atol=1e-6)
    assert torch.equal(env.motion_ids[torch.as_tensor(~done)], pre_ids[torch.as_tensor(~done)])
    # envs that continue still hold the step's body state: their reward must be the oracle's on that state
    xpos, bv = env.xpos.numpy().astype(np.float64), env.base.body_vel.numpy().astype(np.float64)
    quat = mo.matrix_to_quaternion(env.xmat.numpy().astype(np.float64).reshape(n, J, 3, 3))
    want, _ = mo.imitation_reward(xpos, quat, bv[..., :3], bv[..., 3:], ref["rg_pos"], ref["rb_rot"], ref["body_vel"], ref["body_ang_vel"])
    assert np.abs(rew.numpy() - want)[~done].max() < 5e-5
    # the observation handed to the policy has the self part of the base env and the task part written in place
    assert torch.isfinite(obs).all() and (fused or torch.equal(obs[:, :env.self_obs_size], env.base.obs_buf))


def test_fused_imitation_step_equals_the_launch_sequence_it_replaces(emu_backend):
    """ss_imitation_step_fused (one launch) against ss_step -> ss_imitation_step -> resample -> state_at -> ss_reset ->
    ss_imitation_step, over steps in which envs run off their clips and drift away from them: every output identical."""
    import test_motion_lib as T
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    lib = T.make_lib(emu_backend)
    n = 5
    envs = [SMPLSimImitationVecEnv(n, lib, seed=7, fused=f, termination_distance=0.12) for f in (True, False)]
    ids = np.array([0, 1, 2, 0, 1], np.int32)
    t0 = np.array([0.1, 0.2, 0.3, 1.2, 0.0], np.float32)
    outs = [e.reset(motion_ids=ids, start_times=t0)[0].clone() for e in envs]
    assert torch.equal(*outs)
    g = torch.Generator().manual_seed(0)
    resets = 0
    for k in range(6):
        act = envs[0].reference_actions() if k % 2 == 0 else (torch.rand(n, 69, generator=g) - 0.5) * 2.0   # wild actions: early termination
        res = [e.step(act.clone()) for e in envs]
        (o1, r1, te1, tr1, i1), (o2, r2, te2, tr2, i2) = res
        assert torch.equal(te1, te2) and torch.equal(tr1, tr2) and torch.equal(r1, r2), k
        assert torch.equal(i1["reward_parts"], i2["reward_parts"]) and torch.equal(i1["final_observation"], i2["final_observation"]), k
        assert torch.equal(o1, o2), k
        for f in ("motion_ids", "start_times"):
            assert torch.equal(getattr(envs[0], f), getattr(envs[1], f)), (k, f)
        for f in ("qpos", "qvel", "cur_t", "qacc_warm"):
            assert torch.equal(getattr(envs[0].base, f), getattr(envs[1].base, f)), (k, f)
        resets += int((te1 | tr1).sum())
    assert resets >= 3                                          # clip ends and early terminations both happened
    # evaluation-style stepping: no re-initialisation, finished envs keep going
    for e in envs:
        e.autoreset = False
    act = envs[0].reference_actions()
    res = [e.step(act.clone()) for e in envs]
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(envs[0].base.cur_t, envs[1].base.cur_t)


def test_imitation_bind_error_paths(emu_backend):
    import ctypes as C
    import test_motion_lib as T
    from smplsim_amd import _cabi
    from smplsim_amd.batch import SMPLSimVecEnv
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    lib = T.make_lib(emu_backend)
    L = emu_backend
    env = SMPLSimImitationVecEnv(2, lib, seed=0)                  # bound (fused)
    io = env_io = _cabi.ImitationIO()
    plain = SMPLSimVecEnv(2, autoreset=False)                      # task base but StateInit Default, no body outputs
    assert L.ss_imitation_bind(plain.handle, C.byref(io)) == -1
    assert L.ss_imitation_step_fused(plain.handle, C.c_void_p(plain.qpos.data_ptr()), None, None) == -1 and b"bind" in L.ss_batch_last_error(plain.handle)
    assert L.ss_imitation_bind(env.base.handle, None) == -1
    assert L.ss_imitation_bind(env.base.handle, C.byref(env_io)) == -1            # null motion data
    # a too small row stride is refused
    from smplsim_amd.batch import _ptr
    bad = _cabi.ImitationIO(C.pointer(lib.data), env.cfg, *[_ptr(t) for t in (env.motion_ids, env.start_times, env.offset, lib.sampling_cdf)],
                            env.dt, 1, _ptr(env.obs_final), _ptr(env.obs_buf), 10, _ptr(env.rew_buf), _ptr(env.reward_parts), _ptr(env.terminated), _ptr(env.truncated))
    assert L.ss_imitation_bind(env.base.handle, C.byref(bad)) == -1 and b"obs_stride" in L.ss_batch_last_error(env.base.handle)


def test_per_env_shapes_through_the_python_api_on_the_emulator(emu_backend):
    from smplsim_amd.batch import ShardModel, SMPLSimVecEnv
    from smplsim_amd.mjcf_writer import scaled_xml_str
    from smplsim_amd.shapes import ShapeVariedVecEnv
    xmls = [scaled_xml_str("smpl_humanoid", 1.0), scaled_xml_str("smpl_humanoid", 0.9, {"L_Knee": 1.1})]
    env = ShapeVariedVecEnv(xmls, [2, 1], autoreset=False, seed=0)
    assert env.num_envs == 3 and env.shape_id.tolist() == [0, 0, 1]
    obs, _ = env.reset()
    solo = SMPLSimVecEnv(1, model=ShardModel(xml=xmls[1]), autoreset=False)
    assert torch.equal(solo.reset()[0][0], obs[2]) and not torch.equal(obs[0], obs[2])
    a = torch.zeros(3, 69)
    o2 = env.step(a)[0]
    assert torch.equal(solo.step(a[:1])[0][0], o2[2])
    with pytest.raises(ValueError, match="shape_id"):
        SMPLSimVecEnv(2, model=ShardModel(xmls=xmls), shape_id=[0, 5])


def test_per_env_shapes_on_the_52_body_layout(emu_backend):
    """The SMPL-X size class with per-env body shapes: the SHAPED generic kernel on the aliased LDS layout with lean tables (round 5: dof
    constants in global memory, per-shape inverse weights and body offsets from the shape tables) — every env equals the single-shape env
    of its own MJCF bit for bit, through floor contacts."""
    from smplsim_amd.batch import ShardModel, SMPLSimVecEnv
    from smplsim_amd.mjcf_writer import scaled_xml_str
    from smplsim_amd.shapes import ShapeVariedVecEnv
    xmls = [scaled_xml_str("smplx_humanoid", 1.0), scaled_xml_str("smplx_humanoid", 0.92, {"L_Knee": 1.08})]
    env = ShapeVariedVecEnv(xmls, [1, 2], autoreset=False, seed=0)
    assert env.num_envs == 3 and env.shape_id.tolist() == [0, 1, 1]
    solos = [SMPLSimVecEnv(1, model=ShardModel(xml=x), autoreset=False) for x in xmls]
    obs, _ = env.reset()
    so = [e.reset()[0][0].clone() for e in solos]
    assert torch.equal(obs[0], so[0]) and torch.equal(obs[1], so[1]) and torch.equal(obs[2], so[1]) and not torch.equal(obs[0], obs[1])
    for e in [env.single] + solos:                                # dropped onto the floor: contacts, limits, Newton iterations
        q = e.qpos.clone(); q[:, 2] = 0.4
        e.set_state(q, e.qvel.clone())
    g = torch.Generator(); g.manual_seed(2)
    for _ in range(2):
        a = torch.rand(3, env.nu, generator=g) * 1.2 - 0.6
        o = env.step(a)[0]
        o0, o1 = solos[0].step(a[:1])[0][0], solos[1].step(a[1:2])[0][0]
        assert torch.equal(o[0], o0) and torch.equal(o[1], o1)
    assert int(env.single.solver_iters.max()) > 15


def test_native_mjcf_compiler_through_the_python_api_on_the_emulator(emu_backend):
    """ShardModel(compiler="native") = ss_model_create_from_mjcf: the same env, bit for bit, as the Python compiler's model."""
    from smplsim_amd.batch import ShardModel, SMPLSimVecEnv
    a = SMPLSimVecEnv(2, model=ShardModel(), autoreset=False, seed=3)
    b = SMPLSimVecEnv(2, model=ShardModel(compiler="native"), autoreset=False, seed=3)
    assert torch.equal(a.reset()[0], b.reset()[0])
    act = torch.linspace(-0.4, 0.4, 2 * 69).reshape(2, 69)
    ra, rb = a.step(act), b.step(act)
    assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(a.qvel, b.qvel)


def test_package_has_no_backend_switch():
    """The emulator is reachable only through the monkeypatches of the fixture: outside it the package holds no CPU device
    or alternative library, and its loader exposes nothing to bind one."""
    from smplsim_amd import _lib, batch
    assert not hasattr(_lib, "use_test_backend") and not hasattr(_lib, "test_device")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            batch._shard_device(0)


def test_fused_imitation_step_with_per_env_shapes(emu_backend):
    """PHC-style setup (every env its own body shape, clips cooked with that shape's offsets): the one-launch imitation step on the
    SHAPED instantiation against the launch sequence, bit for bit, through re-initialisations."""
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
    lib.load_motions(random_sample=False, offsets=np.stack([sk.offsets for sk in sks]))
    n = 6
    sid = np.arange(n, dtype=np.int32) % 3
    envs = [SMPLSimImitationVecEnv(n, lib, model=model, shape_id=sid, seed=4, fused=f, termination_distance=0.15) for f in (True, False)]
    assert envs[0].fused and not envs[1].fused
    for e in envs:
        e.offset[:, 2] = 0.3
    t0 = np.array([0.1, 0.2, 0.3, 1.2, 0.0, 0.5], np.float32)
    o = [e.reset(motion_ids=sid, start_times=t0)[0].clone() for e in envs]
    assert torch.equal(*o)
    g = torch.Generator().manual_seed(2)
    resets = 0
    for k in range(5):
        act = envs[0].reference_actions() if k % 2 == 0 else (torch.rand(n, 69, generator=g) - 0.5) * 2.0
        (o1, r1, te1, tr1, i1), (o2, r2, te2, tr2, i2) = [e.step(act.clone()) for e in envs]
        assert torch.equal(te1, te2) and torch.equal(tr1, tr2) and torch.equal(r1, r2) and torch.equal(o1, o2), k
        assert torch.equal(i1["final_observation"], i2["final_observation"]) and torch.equal(envs[0].base.qpos, envs[1].base.qpos)
        assert torch.equal(envs[0].motion_ids, envs[1].motion_ids)
        resets += int((te1 | tr1).sum())
    assert resets >= 2


def test_fused_imitation_step_with_body_body_contacts(emu_backend):
    """The one-launch imitation step on the self-collision instantiation against the launch sequence (wild actions fold the arms
    through the torso: body-body contacts and early terminations), bit for bit."""
    import test_motion_lib as T
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    lib = T.make_lib(emu_backend)
    n = 4
    envs = [SMPLSimImitationVecEnv(n, lib, seed=9, fused=f, termination_distance=0.2, self_collision=True) for f in (True, False)]
    assert envs[0].fused and envs[0].base.self_collision
    for e in envs:
        e.offset[:, 2] = 0.05
    ids, t0 = np.array([0, 1, 2, 0], np.int32), np.array([0.1, 0.2, 0.3, 1.2], np.float32)
    assert torch.equal(*[e.reset(motion_ids=ids, start_times=t0)[0].clone() for e in envs])
    g = torch.Generator().manual_seed(4)
    contacts = 0
    for k in range(5):
        act = (torch.rand(n, 69, generator=g) - 0.5) * 2.0
        (o1, r1, te1, tr1, i1), (o2, r2, te2, tr2, i2) = [e.step(act.clone()) for e in envs]
        assert torch.equal(te1, te2) and torch.equal(tr1, tr2) and torch.equal(r1, r2) and torch.equal(o1, o2), k
        assert torch.equal(envs[0].base.qpos, envs[1].base.qpos) and torch.equal(envs[0].base.self_contacts, envs[1].base.self_contacts)
        contacts += int(envs[0].base.self_contacts.sum())
    assert contacts > 0


class _HostStream:
    """Stand-in for a HIP stream on the emulator: everything is synchronous there."""
    def wait_event(self, ev): pass
    def synchronize(self): pass


@pytest.fixture()
def host_streams(monkeypatch):
    import contextlib
    from smplsim_amd import pipeline
    monkeypatch.setattr(pipeline, "_make_stream", lambda device: _HostStream())
    monkeypatch.setattr(pipeline, "_stream_ctx", lambda stream: contextlib.nullcontext())
    monkeypatch.setattr(pipeline, "_record_event", lambda device: None)


@pytest.mark.parametrize("task,init", [("HumanoidSpeed", "Default"), ("HumanoidGetup", "Fall")])
def test_pipelined_sub_batches_are_the_single_batch_bit_for_bit(emu_backend, host_streams, task, init):
    """pipeline.PipelinedVecEnv(N, G, seed) is the SAME job as SMPLSimVecEnv(N, seed): the master generator draws every step's task
    targets / Fall-reset actions for all N envs in the single batch's order and hands the sub-batches row slices, so state,
    observations, rewards and flags agree bit for bit through autoresets (short episodes force them)."""
    from smplsim_amd.batch import SMPLSimVecEnv
    from smplsim_amd.pipeline import PipelinedVecEnv
    N, G = 6, 3
    kw = dict(task=task, state_init=init, episode_length=4, seed=5, recovery_steps=2)    # (getup: no reset during the recovery grace)
    one, pipe = SMPLSimVecEnv(N, **kw), PipelinedVecEnv(N, sub_batches=G, **kw)
    o1, _ = one.reset()
    op = pipe.reset()
    assert torch.equal(o1, torch.cat(op))
    g = torch.Generator(); g.manual_seed(1)
    ended = 0
    for t in range(10):
        a = torch.rand(N, one.nu, generator=g) * 2 - 1
        o1, r1, te1, tu1, i1 = one.step(a)
        outs = [pipe.step_async(k, a[pipe.rows(k)].contiguous()) for k in range(G)]
        assert torch.equal(o1, torch.cat([o[0] for o in outs])) and torch.equal(r1, torch.cat([o[1] for o in outs]))
        assert torch.equal(te1, torch.cat([o[2] for o in outs])) and torch.equal(tu1, torch.cat([o[3] for o in outs]))
        assert torch.equal(i1["final_observation"], torch.cat([o[4]["final_observation"] for o in outs]))
        assert torch.equal(one.qpos, torch.cat([e.qpos for e in pipe.envs])) and torch.equal(one.task_state, torch.cat([e.task_state for e in pipe.envs]))
        ended += int((te1 | tu1).sum())
    assert ended >= N, ended                                     # every env went through at least one in-launch reset
    one.close(); pipe.close()


def test_pipelined_sub_batches_keep_the_jobs_body_shapes(emu_backend, host_streams):
    """A multi-shape model: env i of the job has shape i % num_shapes in BOTH forms (the sub-batches used to restart the
    default table at 0, so with n % num_shapes != 0 env i got shape (i - g n) % num_shapes); an explicit job-wide shape_id is sliced."""
    from smplsim_amd.batch import ShardModel, SMPLSimVecEnv
    from smplsim_amd.mjcf_writer import scaled_xml_str
    from smplsim_amd.pipeline import PipelinedVecEnv
    xmls = [scaled_xml_str("smpl_humanoid", 1.0), scaled_xml_str("smpl_humanoid", 0.9, {"L_Knee": 1.1})]
    N, G = 6, 2                                                  # n = 3 per sub-batch, 2 shapes: 3 % 2 != 0
    kw = dict(task="HumanoidSpeed", episode_length=3, seed=5)
    model = ShardModel(xmls=xmls, device=0)
    one, pipe = SMPLSimVecEnv(N, model=model, **kw), PipelinedVecEnv(N, sub_batches=G, model=model, **kw)
    assert torch.cat([e.shape_id for e in pipe.envs]).tolist() == one.shape_id.tolist() == [0, 1, 0, 1, 0, 1]
    assert torch.equal(one.reset()[0], torch.cat(pipe.reset()))
    g = torch.Generator(); g.manual_seed(2)
    for t in range(5):
        a = torch.rand(N, one.nu, generator=g) * 2 - 1
        o1 = one.step(a)[0]
        outs = [pipe.step_async(k, a[pipe.rows(k)].contiguous()) for k in range(G)]
        assert torch.equal(o1, torch.cat([o[0] for o in outs])) and torch.equal(one.qpos, torch.cat([e.qpos for e in pipe.envs]))
    sid = [1, 1, 0, 0, 0, 1]
    pipe2 = PipelinedVecEnv(N, sub_batches=G, model=model, shape_id=sid, **kw)
    assert torch.cat([e.shape_id for e in pipe2.envs]).tolist() == sid
    with pytest.raises(ValueError, match="shape_id"):
        PipelinedVecEnv(N, sub_batches=G, model=model, shape_id=[0, 1, 0], **kw)
    one.close(); pipe.close(); pipe2.close()


def test_pipelined_draws_are_consumed_once_per_sub_batch_in_any_order(emu_backend, host_streams):
    """One control step's draws serve every sub-batch exactly once, in whatever order the caller steps them; stepping one twice, or
    drawing again before all have stepped, raises instead of silently leaving the single batch's random sequence."""
    from smplsim_amd.batch import SMPLSimVecEnv
    from smplsim_amd.pipeline import PipelinedVecEnv
    N, G = 6, 3
    kw = dict(task="HumanoidSpeed", episode_length=3, seed=5)
    one, pipe = SMPLSimVecEnv(N, **kw), PipelinedVecEnv(N, sub_batches=G, **kw)
    one.reset(); pipe.reset()
    g = torch.Generator(); g.manual_seed(1)
    for order in ([2, 0, 1], [1, 2, 0], [0, 1, 2], [2, 1, 0]):
        a = torch.rand(N, one.nu, generator=g) * 2 - 1
        o1 = one.step(a)[0]
        outs = {k: pipe.step_async(k, a[pipe.rows(k)].contiguous()) for k in order}
        assert torch.equal(o1, torch.cat([outs[k][0] for k in range(G)]))
    a = torch.rand(N, one.nu, generator=g) * 2 - 1
    pipe.step_async(1, a[pipe.rows(1)].contiguous())
    with pytest.raises(RuntimeError, match="stepped twice"):
        pipe.step_async(1, a[pipe.rows(1)].contiguous())
    with pytest.raises(RuntimeError, match="have not stepped"):
        pipe.draw_step_inputs()
    one.close(); pipe.close()


def test_pipelined_sampler_equals_the_serial_sampler(emu_backend, host_streams):
    """AgentPPO.sample_pipelined over G sub-batches returns the rollout of AgentPPO.sample over the single batch: same policy draws
    (the step's noise is drawn for all N envs and sliced), same env inputs (master generator), same tensors."""
    from smplsim_amd.agents.ppo import AgentPPO, PPOConfig
    from smplsim_amd.batch import SMPLSimVecEnv
    from smplsim_amd.pipeline import PipelinedVecEnv
    N, G, T = 8, 2, 5
    kw = dict(task="HumanoidSpeed", episode_length=3, seed=9)
    cfg = PPOConfig(hidden=(32, 32), min_batch_size=N * T)
    a1 = AgentPPO(SMPLSimVecEnv(N, **kw), cfg, seed=4)
    pipe = PipelinedVecEnv(N, sub_batches=G, **kw)
    a2 = AgentPPO(pipe, cfg, seed=4)
    for round_ in range(2):                                      # the second call continues from the kept observations
        b1, b2 = a1.sample(), a2.sample_pipelined(pipe)
        assert set(b1) == set(b2)
        for k in b1:
            if k in ("states", "actions", "last_state"):         # through the torch policy on the host: row results of a matmul may depend
                assert torch.allclose(b1[k], b2[k], rtol=0, atol=1e-6), k   # on the row count by a rounding (the MFMA kernels' do not: GPU test)
            else:
                assert torch.equal(b1[k], b2[k]), k
    assert (1.0 - b1["not_done"]).sum() >= N
    a1.env.close(); pipe.close()
