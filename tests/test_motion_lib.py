"""Motion library (BASELINE config 4): oracle vs the reference's golden vectors, and the kernel source
(ss_motion.h, run on the CPU emulator) vs both.  GPU twins of the kernel tests live in test_gpu_parity.py.

Tolerances: everything is float32 arithmetic on both sides.  Positions / quaternions / Euler angles 2e-5 absolute;
linear velocities 2e-3 (differences of float32 positions divided by dt <= 1/30); angular velocities 5e-2 rad/s
unfiltered, 2e-2 filtered (the reference's angle = acos(2 w^2 - 1) of a float32 quaternion carries ~3e-4 rad of
rounding noise per frame, times fps).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import motion_oracle as mo  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "motion_vectors.npz"))
TOL = dict(global_translation=2e-5, global_rotation=2e-5, local_rotation=2e-5, dof_pos=2e-5, qpos=2e-5,
           global_velocity=2e-3, global_root_velocity=2e-3, dof_vels=2e-3,
           global_angular_velocity=5e-2, global_root_angular_velocity=5e-2, qvel=5e-2)
NAMES = dict(global_translation="gts", global_rotation="grs", local_rotation="lrs", global_velocity="gvs", global_angular_velocity="gavs",
             dof_pos="dof_pos", dof_vels="dvs", qpos="qpos", qvel="qvel")


def clip_dict():
    nf = G["num_frames"]
    st = np.concatenate([[0], np.cumsum(nf)])
    return {f"clip{m}": dict(pose_aa=G["pose_aa"][st[m]:st[m + 1]].reshape(nf[m], 72), trans=G["trans"][st[m]:st[m + 1]], fps=float(G["fps"][m]))
            for m in range(len(nf))}


def make_lib(clib, filter_vel=True, device="cpu", **kw):
    from smplsim_amd.motion_lib import MotionLibSMPL, Skeleton
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import default_xml_str
    mc = compile_mjcf(default_xml_str("smpl_humanoid"))
    sk = Skeleton.from_model_const(mc)
    assert (sk.smpl_2_mujoco == G["smpl_2_mujoco"]).all() and (sk.parents == G["parents"]).all()
    assert np.abs(sk.offsets - G["offsets"]).max() < 1e-6        # fixture MJCF == what the golden generator parsed
    # clib given = the caller holds the emu_backend fixture (tests/conftest.py): the package is patched onto the emulator
    lib = MotionLibSMPL(clip_dict(), sk, filter_vel=filter_vel, device=0 if clib is not None else device, **kw)
    lib.load_motions(random_sample=False)
    return lib


def check_cooked(lib, pre):
    for k, attr in NAMES.items():
        got = getattr(lib, attr).cpu().numpy()
        ref = G[pre + k].reshape(got.shape)
        if k.endswith("rotation"):
            err = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1)).max()
        else:
            err = np.abs(got - ref).max()
        assert err < TOL[k], (pre, k, err)


# ------------------------------------------------------------------ oracle pinned by the reference's outputs
@pytest.mark.parametrize("filt", [True, False])
def test_oracle_cook_matches_reference_fk_batch(filt):
    nf = G["num_frames"]
    st = np.concatenate([[0], np.cumsum(nf)])
    pre = "f_" if filt else "n_"
    for m in range(len(nf)):
        s, e = st[m], st[m + 1]
        r = mo.cook(G["pose_aa"][s:e], G["trans"][s:e], G["offsets"], G["parents"], G["smpl_2_mujoco"], 1 / float(G["fps"][m]), filt)
        for k, v in r.items():
            ref = G[pre + k][s:e]
            assert np.abs(v.reshape(ref.shape) - ref).max() < TOL[k] * 0.2, (m, k)


def test_oracle_fix_continuous_dof_is_exercised_by_the_golden_clips():
    raw = mo.matrix_to_euler_xyz(mo.quaternion_to_matrix(mo.axis_angle_to_quaternion(G["pose_aa"].astype(np.float64)))[:, G["smpl_2_mujoco"]])[:, 1:]
    assert (np.abs(raw - G["f_dof_pos"]) > 1e-3).sum() > 100


def test_oracle_frame_lookup_and_slerp_match_reference():
    nf = G["num_frames"].astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(nf)[:-1]])
    dt = (1 / G["fps"]).astype(np.float32)
    L = (1 / G["fps"] * (nf - 1)).astype(np.float32)
    ids, t = G["q_ids"], G["q_times"]
    fl = mo.intervaled_frame(t, L[ids], nf[ids], dt[ids]) + starts[ids]
    assert np.abs(G["f_global_translation"][fl] + G["q_offset"][:, None] - G["iv_xpos"]).max() == 0
    assert np.abs(G["f_qpos"][fl] - G["iv_qpos"]).max() == 0
    i0, i1, bl = mo.calc_frame_blend(t, L[ids], nf[ids], dt[ids])
    assert (i0 == np.floor(G["q_idx0"])).all()
    s = mo.slerp(G["sl_q0"].astype(np.float64), G["sl_q1"].astype(np.float64), G["sl_t"].astype(np.float64))
    assert np.abs(s - G["sl_out"]).max() < 2e-4          # float32 acos/sqrt conditioning of the reference near cos = 1


# ------------------------------------------------------------------ kernel source on the CPU emulator
@pytest.fixture()
def emu_lib(emu_backend):
    return emu_backend


@pytest.mark.parametrize("filt", [True, False])
def test_emu_cook_matches_reference(emu_lib, filt):
    check_cooked(make_lib(emu_lib, filt), "f_" if filt else "n_")


def test_emu_intervaled_lookup_matches_reference(emu_lib):
    lib = make_lib(emu_lib)
    st = lib.get_motion_state_intervaled(G["q_ids"], G["q_times"], offset=G["q_offset"])
    # the lookup is a gather: compare with the same gather of the emulator-cooked arrays at the reference's frame choice
    nf = G["num_frames"].astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(nf)[:-1]])
    ids = G["q_ids"]
    dt = (1 / G["fps"]).astype(np.float32)
    L = (1 / G["fps"] * (nf - 1)).astype(np.float32)
    fl = mo.intervaled_frame(G["q_times"], L[ids], nf[ids], dt[ids]) + starts[ids]
    assert np.array_equal(st["xpos"].numpy(), lib.gts.numpy()[fl] + G["q_offset"][:, None])
    assert np.array_equal(st["xquat"].numpy(), lib.grs.numpy()[fl])
    assert np.array_equal(st["qpos"].numpy(), lib.qpos.numpy()[fl])      # the reference does not offset qpos (:341-355)
    assert np.array_equal(st["qvel"].numpy(), lib.qvel.numpy()[fl])
    assert np.array_equal(st["dof_pos"].numpy(), lib.dof_pos.numpy()[fl])
    assert np.array_equal(st["body_vel"].numpy(), lib.gvs.numpy()[fl])
    # and with the reference's own outputs
    for k in ("root_pos", "root_rot", "root_vel", "xpos", "xquat", "body_vel", "qpos"):
        tol = 2e-3 if "vel" in k else 2e-5
        assert np.abs(st[k].numpy() - G["iv_" + k].reshape(st[k].shape)).max() < tol, k


def lib_arrays(lib):
    nf = lib._motion_num_frames
    return dict(gts=lib.gts.cpu().numpy().astype(np.float64), grs=lib.grs.cpu().numpy().astype(np.float64), gvs=lib.gvs.cpu().numpy().astype(np.float64),
                gavs=lib.gavs.cpu().numpy().astype(np.float64), dof_pos=lib.dof_pos.cpu().numpy().astype(np.float64), dvs=lib.dvs.cpu().numpy().astype(np.float64),
                length_starts=lib.length_starts, num_frames=nf, dt=lib._motion_dt.astype(np.float64), lengths=lib._motion_lengths.astype(np.float64))


def check_blended(lib, rs, n=200):
    M = lib.num_current_motions()
    ids = rs.integers(0, M, size=n)
    times = (rs.uniform(-0.1, 1.1, size=n) * lib._motion_lengths[ids]).astype(np.float32)
    off = rs.normal(size=(n, 3)).astype(np.float32)
    got = lib.get_motion_state(ids, times, offset=off, with_qpos=True)
    want = mo.motion_state(lib_arrays(lib), ids, times.astype(np.float64), off.astype(np.float64))
    for k, v in want.items():
        g = got[k].cpu().numpy()
        if k.endswith("rot"):
            err = np.minimum(np.abs(g - v).max(-1), np.abs(g + v).max(-1)).max()
        else:
            err = np.abs(g - v.reshape(g.shape)).max()
        # float32 blend weight (error ~1e-5) times the frame-to-frame change, which is 100s of rad/s for the velocities here
        assert err < (2e-4 if k.endswith("rot") else 5e-3 if "vel" in k else 1e-4), (k, err)
    i0, _, _ = mo.calc_frame_blend(times, lib._motion_lengths[ids], lib._motion_num_frames[ids], lib._motion_dt[ids])
    sel = np.abs(times / lib._motion_dt[ids] - np.round(times / lib._motion_dt[ids])) > 1e-3      # away from frame boundaries
    assert np.array_equal(got["motion_aa"].cpu().numpy()[sel], lib._motion_aa[(i0 + lib.length_starts[ids])[sel]])
    # qpos/qvel of the blended state: root pose + Euler dofs, body-frame root angular velocity
    qp, qv = got["qpos"].cpu().numpy(), got["qvel"].cpu().numpy()
    assert np.abs(qp[:, :3] - want["root_pos"]).max() < 1e-4 and np.abs(qp[:, 7:] - want["dof_pos"]).max() < 1e-4
    R = mo.quaternion_to_matrix(want["root_rot"])
    assert np.abs(qv[:, 3:6] - np.einsum("nba,nb->na", R, want["root_ang_vel"])).max() < 5e-3
    assert np.abs(qv[:, 6:] - want["dof_vel"]).max() < 5e-3


def test_emu_blended_lookup_matches_oracle(emu_lib):
    check_blended(make_lib(emu_lib), np.random.default_rng(5))


def random_sim_state(rs, n, J):
    pos = rs.normal(size=(n, J, 3)) * 0.5 + np.array([0, 0, 0.9])
    q = rs.normal(size=(n, J, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    vel = rs.normal(size=(n, J, 6))
    return pos, q, vel


def check_imitation(lib, clib, rs, n=37, device="cpu", stream=None):
    import ctypes as C
    from smplsim_amd import _cabi
    J = 24
    ids = rs.integers(0, lib.num_current_motions(), size=n).astype(np.int32)
    times = (rs.uniform(0, 0.9, size=n) * lib._motion_lengths[ids]).astype(np.float32)
    off = rs.normal(size=(n, 3)).astype(np.float32) * 0.1
    # simulated humanoid = reference pose at `times` + noise, so that every reward term is in its sensitive range
    ref = mo.motion_state(lib_arrays(lib), ids, times.astype(np.float64), off.astype(np.float64))
    pos = ref["rg_pos"] + rs.normal(size=(n, J, 3)) * 0.05
    dq = mo.axis_angle_to_quaternion(rs.normal(size=(n, J, 3)) * 0.2)
    quat = mo.quat_mul(dq, ref["rb_rot"])
    vel = np.concatenate([ref["body_vel"] + rs.normal(size=(n, J, 3)), ref["body_ang_vel"] + rs.normal(size=(n, J, 3))], -1)
    pos[: n // 4] += rs.normal(size=(n // 4, 1, 3)) * 0.4          # some envs far off: termination
    xmat = mo.quaternion_to_matrix(quat).reshape(n, J, 9)
    cfg = _cabi.ImitationCfg(100.0, 10.0, 0.1, 0.1, 0.5, 0.3, 0.1, 0.1, 0.25, 1.0 / 30)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(device)  # noqa: E731
    d_ids, d_times, d_off, d_pos, d_mat, d_vel = t(ids, torch.int32), t(times), t(off), t(pos), t(xmat), t(vel)
    obs = torch.zeros(n, 24 * J, device=device); rew = torch.zeros(n, device=device); parts = torch.zeros(n, 4, device=device)
    term = torch.zeros(n, dtype=torch.uint8, device=device)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    trunc = torch.zeros(n, dtype=torch.uint8, device=device)
    rc = clib.ss_imitation_step(C.byref(lib.data), C.byref(cfg), p(d_ids), p(d_times), None, p(d_off), None, n, p(d_pos), p(d_mat), p(d_vel),
                                p(obs), 24 * J, p(rew), p(parts), p(term), p(trunc), stream)
    assert rc == 0, clib.ss_last_error()
    if device != "cpu":
        torch.cuda.synchronize()
    nxt = mo.motion_state(lib_arrays(lib), ids, (times + np.float32(1.0 / 30)).astype(np.float64), off.astype(np.float64))
    want_obs = mo.imitation_obs(pos, quat, vel[..., :3], vel[..., 3:], nxt["rg_pos"], nxt["rb_rot"], nxt["body_vel"], nxt["body_ang_vel"])
    want_rew, want_parts = mo.imitation_reward(pos, quat, vel[..., :3], vel[..., 3:], ref["rg_pos"], ref["rb_rot"], ref["body_vel"], ref["body_ang_vel"])
    want_term = mo.imitation_reset(pos, ref["rg_pos"], 0.25)
    assert np.abs(obs.cpu().numpy() - want_obs).max() < 2e-4
    assert np.abs(parts.cpu().numpy() - want_parts).max() < 2e-5
    assert np.abs(rew.cpu().numpy() - want_rew).max() < 2e-5
    assert want_parts.min() < 0.5 < want_parts.max()
    margin = np.abs(np.linalg.norm(pos - ref["rg_pos"], axis=-1).mean(-1) - 0.25) > 1e-4
    assert (term.cpu().numpy().astype(bool) == want_term)[margin].all() and want_term.any() and not want_term.all()
    assert np.array_equal(trunc.cpu().numpy().astype(bool), times + np.float32(1.0 / 30) >= lib._motion_lengths[ids])
    # masked launch with a row stride: only the selected envs' observation rows are written, nothing else
    mask = torch.as_tensor((np.arange(n) % 3 == 0).astype(np.uint8)).to(device)
    wide = torch.full((n, 24 * J + 7), -5.0, device=device)
    rc = clib.ss_imitation_step(C.byref(lib.data), C.byref(cfg), p(d_ids), p(d_times), None, p(d_off), p(mask), n, p(d_pos), p(d_mat), p(d_vel),
                                C.c_void_p(wide.data_ptr() + 4 * 7), 24 * J + 7, None, None, None, None, stream)
    assert rc == 0
    if device != "cpu":
        torch.cuda.synchronize()
    wide, mk = wide.cpu().numpy(), mask.cpu().numpy().astype(bool)
    assert (wide[:, :7] == -5.0).all() and (wide[~mk] == -5.0).all() and np.array_equal(wide[mk][:, 7:], obs.cpu().numpy()[mk])


def test_emu_imitation_step_matches_oracle(emu_lib):
    check_imitation(make_lib(emu_lib), emu_lib, np.random.default_rng(11))


def test_motion_api_error_paths(emu_lib):
    import ctypes as C
    from smplsim_amd import _cabi
    lib = make_lib(emu_lib)
    par = np.array([-1, 0, 0, 1], np.int32)          # body 3's parent (1) is not on the chain of body 2: not depth-first
    perm = np.arange(4, dtype=np.int32)
    sk = _cabi.Skeleton(4, par.ctypes.data_as(C.c_void_p), perm.ctypes.data_as(C.c_void_p))
    assert emu_lib.ss_motion_cook(C.byref(sk), C.byref(lib.data), 1, None) == -1
    assert b"depth-first" in emu_lib.ss_last_error()
    par2 = np.array(G["parents"], np.int32)
    bad = np.zeros(24, np.int32)
    sk = _cabi.Skeleton(24, par2.ctypes.data_as(C.c_void_p), bad.ctypes.data_as(C.c_void_p))
    assert emu_lib.ss_motion_cook(C.byref(sk), C.byref(lib.data), 1, None) == -1
    assert b"permutation" in emu_lib.ss_last_error()
    st = _cabi.MotionState()
    assert emu_lib.ss_motion_state_at(C.byref(lib.data), None, None, None, None, 4, 0, C.byref(st), None) == -1
    assert emu_lib.ss_motion_state_at(C.byref(lib.data), C.c_void_p(1), C.c_void_p(1), None, None, 0, 0, C.byref(st), None) == -1
    assert emu_lib.ss_motion_resample(C.byref(lib.data), None, None, None, 0.0, 4, None, None, None) == -1
    assert emu_lib.ss_imitation_step(C.byref(lib.data), C.byref(_cabi.ImitationCfg()), C.c_void_p(1), C.c_void_p(1), None, None, None, 4, C.c_void_p(1),
                                     C.c_void_p(1), C.c_void_p(1), C.c_void_p(1), 24 * 24 - 1, None, None, None, None, None) == -1
    assert b"obs_stride" in emu_lib.ss_last_error()


def test_motion_lib_needs_gpu_without_test_hook():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from smplsim_amd.motion_lib import MotionLibSMPL, Skeleton
    sk = Skeleton([f"b{i}" for i in range(3)], [-1, 0, 1], np.zeros((3, 3)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        MotionLibSMPL({"a": dict(pose_aa=np.zeros((4, 9)), trans=np.zeros((4, 3)))}, sk)


def test_sampling_and_history_interface(emu_lib):
    lib = make_lib(emu_lib)
    ids = lib.sample_motions(50)
    assert ids.dtype == torch.int32 and int(ids.min()) >= 0 and int(ids.max()) < 3
    t = lib.sample_time(ids)
    assert (t >= 0).all() and (t <= lib.get_motion_length(ids)).all()
    assert lib.num_current_motions() == 3 and lib.num_all_motions() == 3
    assert abs(lib.get_total_length() - float(((G["num_frames"] - 1) / G["fps"]).sum())) < 1e-4
    lib.update_soft_sampling_weight(["clip1"])
    assert lib._sampling_prob[1] == 1.0
    hist = lib.get_termination_history()
    lib.update_hard_sampling_weight([])
    assert np.allclose(lib._sampling_prob, 1 / 3)
    lib.set_termination_history(hist)
    assert lib._sampling_prob[1] == 1.0
    assert list(lib.get_motion_num_steps()) == [int(n * 30 / f) for n, f in zip(G["num_frames"], G["fps"])]


# ------------------------------------------------------------------ reference-state init + imitation rollout (emulator)
def test_emu_external_state_init_and_imitation_rollout_track_oracle(emu_lib):
    """ss_reset with StateInit External on a clip's qpos/qvel, then 3 control steps replaying the clip through the PD
    controller: simulator state vs the float64 oracle env, task obs / reward vs motion_oracle on the emulator's own state."""
    import ctypes as C
    from helpers import FEET, model_const, oracle_model, pd_tables
    from oracle import oracle as O
    from smplsim_amd import _cabi
    from wave_emu import emu
    lib = make_lib(emu_lib)
    mc = model_const()
    n, J = 3, 24
    eb = emu.EmuBatch(mc, pd_tables(mc), n, legal_bodies=FEET, state_init=_cabi.INIT_EXTERNAL, self_obs_v=2, episode_length=10 ** 6)
    ids = np.array([0, 1, 0], np.int32)
    t0 = np.array([0.1, 0.2, 0.55], np.float32)
    st = lib.get_motion_state(ids, t0, with_qpos=True)
    eb.qpos[:], eb.qvel[:] = st["qpos"].numpy(), st["qvel"].numpy()
    eb.qpos[:, 2] += 0.05                                    # lift the clips (they were not height-fixed) off the floor
    eb.set_body_outputs()
    obs0 = eb.reset()
    xp, xm = eb.kinematics()
    assert np.array_equal(xp, eb.xpos_out) and np.array_equal(xm, eb.xmat_out)      # by-product of the reset launch
    oenvs = []
    for i in range(n):
        oe = O.OracleEnv(oracle_model(), state_init=O.INIT_EXTERNAL, self_obs_v=2, episode_length=10 ** 6)
        oe.data.qpos = eb.qpos[i].astype(np.float64); oe.data.qvel = eb.qvel[i].astype(np.float64)
        assert np.abs(oe.reset() - obs0[i]).max() < 2e-4
        oenvs.append(oe)
    assert (eb.cur_t == 0).all()
    dt = 15 / 450.0
    cfg = _cabi.ImitationCfg(100.0, 10.0, 0.1, 0.1, 0.5, 0.3, 0.1, 0.1, 0.25, dt)
    arr = lib_arrays(lib)
    for k in range(3):
        nxt = lib.get_motion_state(ids, t0 + np.float32((k + 1) * dt))
        act = np.clip(nxt["dof_pos"].numpy() / np.pi, -1, 1)
        eb.step(act)
        for i, oe in enumerate(oenvs):
            oe.step(act[i].astype(np.float64))
            assert np.abs(oe.data.qpos - eb.qpos[i]).max() < 5e-4, (k, i)
        xpos, xmat = eb.kinematics()
        assert np.array_equal(xpos, eb.xpos_out) and np.array_equal(xmat, eb.xmat_out)  # by-product of the step launch
        times = (t0 + np.float32((k + 1) * dt)).astype(np.float32)
        obs = np.zeros((n, 24 * J), np.float32); rew = np.zeros(n, np.float32); parts = np.zeros((n, 4), np.float32); term = np.zeros(n, np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        assert emu_lib.ss_imitation_step(C.byref(lib.data), C.byref(cfg), p(ids), p(t0), p(eb.cur_t), None, None, n, p(xpos), p(xmat),
                                         p(eb.body_vel), p(obs), 24 * J, p(rew), p(parts), p(term), None, None) == 0
        quat = mo.matrix_to_quaternion(xmat.reshape(n, J, 3, 3).astype(np.float64))
        ref = mo.motion_state(arr, ids, times.astype(np.float64))
        fut = mo.motion_state(arr, ids, (times + np.float32(dt)).astype(np.float64))
        bv = eb.body_vel.astype(np.float64)
        want_obs = mo.imitation_obs(xpos, quat, bv[..., :3], bv[..., 3:], fut["rg_pos"], fut["rb_rot"], fut["body_vel"], fut["body_ang_vel"])
        want_rew, _ = mo.imitation_reward(xpos, quat, bv[..., :3], bv[..., 3:], ref["rg_pos"], ref["rb_rot"], ref["body_vel"], ref["body_ang_vel"])
        assert np.abs(obs - want_obs).max() < 5e-4 and np.abs(rew - want_rew).max() < 5e-5


def test_emu_resample_and_masked_state_write(emu_lib):
    lib = make_lib(emu_lib)
    n = 4000
    ids = torch.full((n,), -1, dtype=torch.int32)
    t0 = torch.full((n,), -1.0)
    mask = torch.as_tensor((np.arange(n) % 4 != 0).astype(np.uint8))
    lib.set_termination_history({"termination_history": np.array([1.0, 0.0, 3.0]), "failed_keys": []})   # p = (1/4, 0, 3/4)
    lib.load_motions(random_sample=False)
    lib.resample(mask, ids, t0, truncate_time=0.1, generator=torch.Generator().manual_seed(3))
    mk = mask.numpy().astype(bool)
    assert (ids.numpy()[~mk] == -1).all() and (t0.numpy()[~mk] == -1).all()
    got = ids.numpy()[mk]
    assert set(np.unique(got)) == {0, 2} and abs((got == 2).mean() - 0.75) < 0.03
    L = lib._motion_lengths[got]
    assert (t0.numpy()[mk] >= 0).all() and (t0.numpy()[mk] <= L - 0.1 + 1e-6).all() and t0.numpy()[mk].std() > 0.1
    # masked reference-state write straight into "simulator" tensors
    qpos, qvel = torch.full((n, 76), 7.0), torch.full((n, 75), 7.0)
    ids2 = torch.where(torch.as_tensor(mk), ids, torch.zeros_like(ids))
    lib.write_state(ids2, t0.clamp(min=0), None, mask, qpos, qvel)
    want = lib.get_motion_state(ids2, t0.clamp(min=0), with_qpos=True)
    assert (qpos[~torch.as_tensor(mk)] == 7.0).all() and torch.equal(qpos[torch.as_tensor(mk)], want["qpos"][torch.as_tensor(mk)])
    assert torch.equal(qvel[torch.as_tensor(mk)], want["qvel"][torch.as_tensor(mk)])


# ------------------------------------------------------------------ 52-body skeleton (SMPL-X/H layout): 64 lanes per env
def smplx_lib(clib, device="cpu"):
    from smplsim_amd.motion_lib import MotionLibSMPL, Skeleton
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import default_xml_str
    mc = compile_mjcf(default_xml_str("smplx_humanoid"))
    rs = np.random.default_rng(52)
    perm = rs.permutation(52)
    perm[list(perm).index(0)], perm[0] = perm[0], 0           # joint 0 stays the root, the rest in a scrambled "SMPL" order
    order = [mc.body_names[i] for i in perm]
    sk = Skeleton(mc.body_names, mc.body_parent, mc.body_pos, smpl_order_names=order)
    clips = {}
    for c, T in enumerate((33, 20)):
        t = np.arange(T)[:, None, None] / 30.0
        pose = rs.normal(size=(1, 52, 3)) * 0.4 + 0.5 * np.sin(2 * np.pi * rs.uniform(0.5, 2, size=(1, 52, 3)) * t)
        trans = np.stack([0.5 * t[:, 0, 0], 0 * t[:, 0, 0], 0.95 + 0 * t[:, 0, 0]], -1)
        clips[f"x{c}"] = dict(pose_aa=pose.reshape(T, -1).astype(np.float32), trans=trans.astype(np.float32), fps=30)
    lib = MotionLibSMPL(clips, sk, device=0 if clib is not None else device)
    lib.load_motions(random_sample=False)
    return lib, sk, clips


def check_smplx(lib, sk, clips, clib, device="cpu"):
    import ctypes as C
    from smplsim_amd import _cabi
    J = 52
    st = 0
    for c in clips.values():
        T = c["pose_aa"].shape[0]
        r = mo.cook(c["pose_aa"].reshape(T, J, 3), c["trans"], sk.offsets, sk.parents, sk.smpl_2_mujoco, 1 / 30, True)
        for k, attr in NAMES.items():
            got = getattr(lib, attr).cpu().numpy()[st:st + T]
            ref = r[k].reshape(got.shape)
            err = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1)).max() if k.endswith("rotation") else np.abs(got - ref).max()
            assert err < TOL[k], (k, err)
        st += T
    rs = np.random.default_rng(9)
    n = 11
    ids = rs.integers(0, 2, size=n).astype(np.int32)
    times = (rs.uniform(0, 0.9, size=n) * lib._motion_lengths[ids]).astype(np.float32)
    arr = lib_arrays(lib)
    ref = mo.motion_state(arr, ids, times.astype(np.float64))
    fut = mo.motion_state(arr, ids, (times + np.float32(1 / 30)).astype(np.float64))
    pos = ref["rg_pos"] + rs.normal(size=(n, J, 3)) * 0.05
    quat = mo.quat_mul(mo.axis_angle_to_quaternion(rs.normal(size=(n, J, 3)) * 0.2), ref["rb_rot"])
    vel = np.concatenate([ref["body_vel"] + rs.normal(size=(n, J, 3)), ref["body_ang_vel"] + rs.normal(size=(n, J, 3))], -1)
    cfg = _cabi.ImitationCfg(100.0, 10.0, 0.1, 0.1, 0.5, 0.3, 0.1, 0.1, 0.25, 1.0 / 30)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(device)  # noqa: E731
    d = [t(ids, torch.int32), t(times), t(pos), t(mo.quaternion_to_matrix(quat).reshape(n, J, 9)), t(vel)]
    obs = torch.zeros(n, 24 * J, device=device); rew = torch.zeros(n, device=device); term = torch.zeros(n, dtype=torch.uint8, device=device)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    assert clib.ss_imitation_step(C.byref(lib.data), C.byref(cfg), p(d[0]), p(d[1]), None, None, None, n, p(d[2]), p(d[3]), p(d[4]),
                                  p(obs), 24 * J, p(rew), None, p(term), None, None) == 0
    if device != "cpu":
        torch.cuda.synchronize()
    want_obs = mo.imitation_obs(pos, quat, vel[..., :3], vel[..., 3:], fut["rg_pos"], fut["rb_rot"], fut["body_vel"], fut["body_ang_vel"])
    want_rew, _ = mo.imitation_reward(pos, quat, vel[..., :3], vel[..., 3:], ref["rg_pos"], ref["rb_rot"], ref["body_vel"], ref["body_ang_vel"])
    assert np.abs(obs.cpu().numpy() - want_obs).max() < 2e-4 and np.abs(rew.cpu().numpy() - want_rew).max() < 2e-5


def test_emu_52_body_skeleton_cook_and_imitation(emu_lib):
    lib, sk, clips = smplx_lib(emu_lib)
    check_smplx(lib, sk, clips, emu_lib)


# ------------------------------------------------------------------ tracking metrics vs the reference's smpl_eval
def test_tracking_metrics_match_reference_smpl_eval():
    from smplsim_amd import metrics
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)  # noqa: E731
    m = metrics.compute_metrics_lite([t(G["ev_in_pred0"]), t(G["ev_in_pred1"])], [t(G["ev_in_gt0"]), t(G["ev_in_gt1"])],
                                     [t(G["ev_in_rpred0"]), t(G["ev_in_rpred1"])], [t(G["ev_in_rgt0"]), t(G["ev_in_rgt1"])])
    for k in ("mpjpe_g", "mpjpe_l", "mpjpe_pa", "accel_dist", "vel_dist", "rot_error"):
        ref = G["ev_" + k]
        assert m[k].shape == ref.shape, k
        assert np.abs(m[k].numpy() - ref).max() < 1e-8 * max(1.0, np.abs(ref).max()), k
    m32 = metrics.compute_metrics_lite([t(G["ev_in_pred0"]).float()], [t(G["ev_in_gt0"]).float()])
    assert np.abs(m32["mpjpe_pa"].numpy() - G["ev_mpjpe_pa"][:17]).max() < 1e-2      # millimetres, float32


def test_geom_height_fix_puts_the_first_frames_on_the_floor(emu_lib):
    from smplsim_amd.motion_lib import FixHeightMode
    raw = make_lib(emu_lib)
    lib = make_lib(emu_lib, fix_height=FixHeightMode.geom_fix)
    # lowest geom point per frame from the oracle-side restatement of the geoms (numpy, independent of the torch code)
    from helpers import model_const
    mc = model_const()
    gts, grs = lib.gts.numpy().astype(np.float64), lib.grs.numpy().astype(np.float64)
    R = mo.quaternion_to_matrix(grs)
    G = mo.quaternion_to_matrix(np.asarray(mc.geom_quat, np.float64))
    lows = np.full(gts.shape[:2], np.inf)
    for j in range(24):
        Rg = R[:, j] @ G[j]
        c = gts[:, j] + R[:, j] @ np.asarray(mc.geom_pos[j], np.float64)
        if mc.geom_type[j] == 0:
            corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * np.asarray(mc.geom_size[j])
            lows[:, j] = (c[:, None, 2] + np.einsum("fk,ck->fc", Rg[:, 2], corners)).min(1)
        else:
            r, hl = mc.geom_size[j][0], mc.geom_size[j][1]
            lows[:, j] = np.minimum(c[:, 2] + Rg[:, 2, 2] * hl, c[:, 2] - Rg[:, 2, 2] * hl) - r
    st = np.concatenate([[0], np.cumsum(np.asarray(lib._motion_num_frames))])
    for m in range(3):
        first = lows[st[m]:st[m] + 30].min()
        assert abs(first) < 2e-5, (m, first)
    # a pure z translation of positions; everything else untouched
    d = raw.gts.numpy() - lib.gts.numpy()
    assert np.abs(d[..., :2]).max() == 0 and np.abs(d[..., 2] - d[:, :1, 2]).max() < 1e-6
    assert np.array_equal(raw.gvs.numpy(), lib.gvs.numpy()) and np.array_equal(raw.grs.numpy(), lib.grs.numpy())
    assert np.abs((raw.qpos.numpy() - lib.qpos.numpy())[:, 2] - d[:, 0, 2]).max() < 1e-6


def test_qpos_to_pose_aa_inverts_the_cooked_qpos(emu_lib):
    from smplsim_amd.motion_lib import qpos_to_pose_aa
    lib = make_lib(emu_lib)
    root, aa = qpos_to_pose_aa(lib.qpos.double(), lib.skeleton)
    assert np.abs(root.numpy() - G["trans"]).max() < 1e-5
    # same rotations as the clips' pose_aa (compare as matrices: axis-angle is not unique beyond pi)
    want = mo.quaternion_to_matrix(mo.axis_angle_to_quaternion(G["pose_aa"].astype(np.float64)))
    got = mo.quaternion_to_matrix(mo.axis_angle_to_quaternion(aa.numpy()))
    assert np.abs(got - want).max() < 2e-5


def test_emu_very_short_clips_match_oracle(emu_lib):
    """2- and 3-frame clips between longer ones: every tap of the velocity filter is an edge sample."""
    from smplsim_amd.motion_lib import MotionLibSMPL, Skeleton
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import default_xml_str
    sk = Skeleton.from_model_const(compile_mjcf(default_xml_str("smpl_humanoid")))
    rs = np.random.default_rng(3)
    clips = {}
    for c, T in enumerate((2, 19, 3, 2, 17)):
        pose = (rs.normal(size=(1, 24, 3)) * 0.3 + 0.2 * np.arange(T)[:, None, None] * rs.normal(size=(1, 24, 3))).astype(np.float32)
        trans = np.cumsum(rs.normal(size=(T, 3)) * 0.02, 0).astype(np.float32) + np.array([0, 0, 0.9], np.float32)
        clips[f"s{c}"] = dict(pose_aa=pose.reshape(T, 72), trans=trans, fps=30)
    lib = MotionLibSMPL(clips, sk, device=0)
    lib.load_motions(random_sample=False)
    st = 0
    for c in clips.values():
        T = c["pose_aa"].shape[0]
        r = mo.cook(c["pose_aa"].reshape(T, 24, 3), c["trans"], sk.offsets, sk.parents, sk.smpl_2_mujoco, 1 / 30, True)
        for k, attr in NAMES.items():
            got = getattr(lib, attr).numpy()[st:st + T]
            ref = r[k].reshape(got.shape)
            err = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1)).max() if k.endswith("rotation") else np.abs(got - ref).max()
            assert err < TOL[k], (T, k, err)
        st += T
    # lookups at and beyond both ends of a 2-frame clip
    s = lib.get_motion_state(np.array([0, 0, 0, 3]), np.array([-1.0, 0.0, 5.0, 1 / 60], np.float32))
    assert np.allclose(s["rg_pos"][0].numpy(), lib.gts[0].numpy()) and np.allclose(s["rg_pos"][2].numpy(), lib.gts[1].numpy())
    f3 = int(lib.length_starts[3])
    assert np.allclose(s["rg_pos"][3].numpy(), 0.5 * (lib.gts[f3].numpy() + lib.gts[f3 + 1].numpy()), atol=1e-5)


class _FakeMeshParser:
    """Stand-in for the reference's SMPL_Parser (needs the licensed model files): a "mesh" of 4 vertices per joint, posed by the
    oracle's forward kinematics.  Only the interface fix_trans_height uses: get_joints_verts, lbs_weights, joint_names."""

    def __init__(self, sk):
        self.sk = sk
        self.joint_names = list(sk.smpl_order_names) if hasattr(sk, "smpl_order_names") else None
        J = len(sk.parents)
        self.lbs_weights = np.repeat(np.eye(J, dtype=np.float32), 4, axis=0)            # vertex 4 j .. 4 j + 3 belongs to joint j
        rs = np.random.default_rng(0)
        self.local = rs.uniform(-0.05, 0.05, (J, 4, 3))

    def get_joints_verts(self, pose_aa, betas, trans):
        pose, tr = np.asarray(pose_aa, np.float64), np.asarray(trans, np.float64)
        r = mo.cook(pose, tr, self.sk.offsets, self.sk.parents, self.sk.smpl_2_mujoco, 1 / 30, False)
        gt = r["global_translation"].reshape(len(pose), -1, 3)                            # MuJoCo body order
        verts = (gt[:, :, None, :] + self.local[None]).reshape(len(pose), -1, 3)
        return torch.as_tensor(verts, dtype=torch.float32), torch.as_tensor(gt, dtype=torch.float32)


def test_fix_trans_height_on_the_mesh(emu_lib):
    """FixHeightMode.full_fix / ankle_fix with mesh parsers (reference motion_lib_smpl.py:67-92): the first 30 frames' lowest
    vertex ends on the floor; ankle_fix ignores the vertices skinned to toes and hands and leaves 2.5 cm."""
    from smplsim_amd.motion_lib import FixHeightMode, MotionLibSMPL, Skeleton
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import default_xml_str
    mc = compile_mjcf(default_xml_str("smpl_humanoid"))
    sk = Skeleton.from_model_const(mc)
    parser = _FakeMeshParser(sk)
    parser.joint_names = list(mc.body_names)                                              # vertex owner index = MuJoCo body index here
    parsers = {"0": parser}
    clips = clip_dict()
    for mode, tol in ((FixHeightMode.full_fix, 0.0), (FixHeightMode.ankle_fix, -0.025)):
        lib = MotionLibSMPL(clips, sk, device=0, fix_height=mode, mesh_parsers=parsers)
        lib.load_motions(random_sample=False)
        st = 0
        for m, c in enumerate(clips.values()):
            T = c["pose_aa"].shape[0]
            tr = lib.qpos[st:st + T, :3].numpy()                                          # the fixed root translation (+ root offset)
            verts, _ = parser.get_joints_verts(c["pose_aa"].reshape(T, 24, 3)[:30], None, tr[:30] - sk.offsets[0])
            v = verts.numpy().reshape(min(T, 30), 24, 4, 3)
            if mode == FixHeightMode.ankle_fix:
                keep = [i for i, n in enumerate(mc.body_names) if n not in ("L_Toe", "R_Toe", "L_Hand", "R_Hand")]
                v = v[:, keep]
            assert abs(v[..., 2].min() - tol) < 2e-5, (mode, m, v[..., 2].min())
            st += T
    with pytest.raises(ValueError, match="mesh_parsers"):
        MotionLibSMPL(clips, sk, device=0, fix_height=FixHeightMode.full_fix)
