"""CPU restatement (NumPy, float64) of the reference's motion library for the imitation path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(smplsim_amd.motion_lib -> libsmplsim_hip.so) never does.

What it follows (reference file:line):
  * cook():  Humanoid_Batch.fk_batch / forward_kinematics_batch / _compute_velocity / _compute_angular_velocity,
             smpl_sim/smpllib/torch_smpl_humanoid_batch.py:118-228, with the conversions of
             smpl_sim/utils/pytorch3d_transforms.py (axis_angle_to_quaternion :546-569, quaternion_to_matrix :48-76,
             matrix_to_quaternion :139-185, matrix_to_euler_angles "XYZ" :331-364, fix_continous_dof :749-778,
             quat_mul_norm :729-734, quat_angle_axis :737-746) and scipy.ndimage.gaussian_filter1d(sigma=2, mode="nearest")
  * calc_frame_blend / motion_state / motion_state_intervaled: MotionLibBase, smpl_sim/smpllib/motion_lib_base.py:311-423,442-453
  * slerp: smpl_sim/utils/torch_utils.py:405-426
  * imitation_obs / imitation_reward / imitation_reset: NOT in the reference ("define the PHC tracking
    reward (absent from the reference)"); they restate the PHC formulation (Luo et al. 2023, "Perpetual Humanoid
    Control", tracking reward of eq. 3 and the v6 task observation) in the reference's wxyz / heading conventions
    (smpl_sim/utils/np_transform_utils.py).  Parity of these three is against this file only: "parity unpinned".

Pinned against tests/golden/motion_vectors.npz (outputs of the reference's own code, tests/golden/make_golden_motion.py)
for everything except the three imitation functions.
"""
import numpy as np

GAUSS_SIGMA = 2.0
GAUSS_RADIUS = 8            # int(truncate * sigma + 0.5), truncate = 4.0 (scipy default)


# ---------------------------------------------------------------- rotation conversions
def axis_angle_to_quaternion(aa):
    ang = np.linalg.norm(aa, axis=-1, keepdims=True)
    small = np.abs(ang) < 1e-6
    safe = np.where(small, 1.0, ang)
    k = np.where(small, 0.5 - ang * ang / 48.0, np.sin(0.5 * ang) / safe)
    return np.concatenate([np.cos(0.5 * ang), aa * k], axis=-1)


def quaternion_to_matrix(q):
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s = 2.0 / (q * q).sum(-1)
    m = np.stack([1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                  s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                  s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)], axis=-1)
    return m.reshape(q.shape[:-1] + (3, 3))


def matrix_to_quaternion(m):
    m00, m01, m02 = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    m10, m11, m12 = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    m20, m21, m22 = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    qa = np.sqrt(np.maximum(0.0, np.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)))
    cand = np.stack([np.stack([qa[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
                     np.stack([m21 - m12, qa[..., 1] ** 2, m10 + m01, m02 + m20], -1),
                     np.stack([m02 - m20, m10 + m01, qa[..., 2] ** 2, m12 + m21], -1),
                     np.stack([m10 - m01, m20 + m02, m21 + m12, qa[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * np.maximum(qa[..., None], 0.1))
    pick = qa.argmax(-1)
    return np.take_along_axis(cand, pick[..., None, None], axis=-2)[..., 0, :]


def matrix_to_euler_xyz(m):
    return np.stack([np.arctan2(-m[..., 1, 2], m[..., 2, 2]), np.arcsin(m[..., 0, 2]), np.arctan2(-m[..., 0, 1], m[..., 0, 0])], -1)


def fix_continuous_dof(dof):
    """dof [T,J,3] Euler angles; the reference's sequential flip fix (it skips the last frame and gives up after 2 tries)."""
    dof = dof.copy()
    T = dof.shape[0] - 1
    for t in range(1, T):
        diff = dof[t] - dof[t - 1]
        times = 0
        while np.abs(diff).max() >= 3:
            ch = np.abs(diff).sum(-1) >= 3
            c = dof[t][ch].copy()
            c[:, 0] = np.pi + c[:, 0]
            c[:, 1] = np.pi - c[:, 1]
            c[:, 2] = np.pi + c[:, 2]
            c[c > np.pi] -= 2 * np.pi
            c[c < -np.pi] += 2 * np.pi
            dof[t][ch] = c
            diff = dof[t] - dof[t - 1]
            times += 1
            if times > 1:
                break
    return dof


def quat_mul(a, b):
    w1, x1, y1, z1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    w2, x2, y2, z2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2, w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2], -1)


def quat_conj(q):
    return np.concatenate([q[..., :1], -q[..., 1:]], -1)


def quat_normalize(q):
    q = np.where(q[..., :1] < 0, -q, q)
    return q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), 1e-9)


def gauss_weights():
    x = np.arange(-GAUSS_RADIUS, GAUSS_RADIUS + 1)
    w = np.exp(-0.5 * x * x / (GAUSS_SIGMA * GAUSS_SIGMA))
    return w / w.sum()


def gauss_filter_time(v):
    """v [T,...]: correlate along axis 0 with the sigma-2 kernel, edge samples repeated ("nearest")."""
    T = v.shape[0]
    w = gauss_weights()
    out = np.zeros_like(v)
    for k in range(-GAUSS_RADIUS, GAUSS_RADIUS + 1):
        out += w[k + GAUSS_RADIUS] * v[np.clip(np.arange(T) + k, 0, T - 1)]
    return out


def cook(pose_aa, trans, offsets, parents, smpl_2_mujoco, dt, filter_vel=True):
    """One clip: pose_aa [T,J,3] (SMPL joint order), trans [T,3], offsets [J,3] (MuJoCo body order) -> the dict fk_batch returns."""
    pose_aa, trans, offsets = (np.asarray(a, np.float64) for a in (pose_aa, trans, offsets))
    T, J = pose_aa.shape[:2]
    pose_quat = axis_angle_to_quaternion(pose_aa)
    pose_mat = quaternion_to_matrix(pose_quat)[:, smpl_2_mujoco]
    root_pos = trans + offsets[0]
    wpos, wmat = [None] * J, [None] * J
    for j in range(J):
        p = parents[j]
        if p < 0:
            wpos[j], wmat[j] = root_pos, pose_mat[:, 0]
        else:
            wpos[j] = np.einsum("tab,b->ta", wmat[p], offsets[j]) + wpos[p]
            wmat[j] = wmat[p] @ pose_mat[:, j]
    gts, gmat = np.stack(wpos, 1), np.stack(wmat, 1)
    grs = matrix_to_quaternion(gmat)

    vel = (gts[1:] - gts[:-1]) / dt
    vel = np.concatenate([vel, vel[-1:]], 0)
    dq = np.zeros_like(grs)
    dq[..., 0] = 1.0
    dq[:-1] = quat_normalize(quat_mul(grs[1:], quat_conj(grs[:-1])))
    ang = np.arccos(np.clip(2 * dq[..., 0] ** 2 - 1, -1, 1))
    axis = dq[..., 1:] / np.maximum(np.linalg.norm(dq[..., 1:], axis=-1, keepdims=True), 1e-10)
    angvel = axis * ang[..., None] / dt
    if filter_vel:
        vel, angvel = gauss_filter_time(vel), gauss_filter_time(angvel)

    dof_pos = fix_continuous_dof(matrix_to_euler_xyz(pose_mat)[:, 1:])
    dof_vel = (dof_pos[1:] - dof_pos[:-1]) / dt
    dof_vel = np.concatenate([dof_vel, dof_vel[-1:]], 0)
    qpos = np.concatenate([root_pos, pose_quat[:, 0], dof_pos.reshape(T, -1)], -1)
    local_root_angvel = np.einsum("tba,tb->ta", gmat[:, 0], angvel[:, 0])
    qvel = np.concatenate([vel[:, 0], local_root_angvel, dof_vel.reshape(T, -1)], -1)
    return dict(global_translation=gts, global_rotation=grs, local_rotation=pose_quat, global_velocity=vel,
                global_angular_velocity=angvel, global_root_velocity=vel[:, 0], global_root_angular_velocity=angvel[:, 0],
                dof_pos=dof_pos, dof_vels=dof_vel, qpos=qpos, qvel=qvel)


# ---------------------------------------------------------------- frame lookup
def calc_frame_blend(time, length, num_frames, dt):
    """PHC semantics (integer frame numbers); the reference's NumPy version omits the integer cast (motion_lib_base.py:446)."""
    time = np.asarray(time, np.float64).copy()
    phase = np.clip(time / length, 0.0, 1.0)
    time[time < 0] = 0
    idx0 = np.floor(phase * (num_frames - 1)).astype(np.int64)
    idx1 = np.minimum(idx0 + 1, num_frames - 1)
    blend = np.clip((time - idx0 * dt) / dt, 0.0, 1.0)
    return idx0, idx1, blend


def slerp(q0, q1, t):
    cos_h = (q0 * q1).sum(-1)
    q1 = np.where((cos_h < 0)[..., None], -q1, q1)
    cos_h = np.abs(cos_h)[..., None]
    half = np.arccos(np.minimum(cos_h, 1.0))
    sin_h = np.sqrt(np.maximum(1.0 - cos_h * cos_h, 0.0))
    with np.errstate(divide="ignore", invalid="ignore"):
        new = (np.sin((1 - t) * half) / sin_h) * q0 + (np.sin(t * half) / sin_h) * q1
    new = np.where(np.abs(sin_h) < 0.001, 0.5 * q0 + 0.5 * q1, new)
    return np.where(np.abs(cos_h) >= 1, q0, new)


def motion_state(lib, motion_ids, times, offset=None):
    """lib: dict with gts, grs, gvs, gavs, dof_pos [F,J-1,3], dvs, length_starts, num_frames, dt, lengths (concatenated clips)."""
    i0, i1, bl = calc_frame_blend(times, lib["lengths"][motion_ids], lib["num_frames"][motion_ids], lib["dt"][motion_ids])
    f0, f1 = i0 + lib["length_starts"][motion_ids], i1 + lib["length_starts"][motion_ids]
    b1, b2 = bl[:, None], bl[:, None, None]
    pos = (1 - b2) * lib["gts"][f0] + b2 * lib["gts"][f1]
    if offset is not None:
        pos = pos + offset[:, None, :]
    vel = (1 - b2) * lib["gvs"][f0] + b2 * lib["gvs"][f1]
    angvel = (1 - b2) * lib["gavs"][f0] + b2 * lib["gavs"][f1]
    N = len(motion_ids)
    dp0, dp1 = lib["dof_pos"][f0].reshape(N, -1), lib["dof_pos"][f1].reshape(N, -1)
    dv0, dv1 = lib["dvs"][f0].reshape(N, -1), lib["dvs"][f1].reshape(N, -1)
    rot = slerp(lib["grs"][f0], lib["grs"][f1], b2)
    return dict(root_pos=pos[:, 0], root_rot=rot[:, 0], dof_pos=(1 - b1) * dp0 + b1 * dp1, root_vel=vel[:, 0],
                root_ang_vel=angvel[:, 0], dof_vel=(1 - b1) * dv0 + b1 * dv1, rg_pos=pos, rb_rot=rot, body_vel=vel, body_ang_vel=angvel)


def intervaled_frame(times, length, num_frames, dt):
    """Frame picked by get_motion_state_intervaled (:317-319) with the float frame numbers of the NumPy _calc_frame_blend."""
    time = np.asarray(times, np.float64).copy()
    phase = np.clip(time / length, 0.0, 1.0)
    time[time < 0] = 0
    idx0 = phase * (num_frames - 1)
    idx1 = np.minimum(idx0 + 1, num_frames - 1)
    blend = np.clip((time - idx0 * dt) / dt, 0.0, 1.0)
    return ((1.0 - blend) * idx0 + blend * idx1).astype(np.int64)


# ---------------------------------------------------------------- imitation task (PHC formulation; not in the reference)
def quat_rotate(q, v):
    w, qv = q[..., :1], q[..., 1:]
    return v * (2 * w * w - 1) + np.cross(qv, v) * w * 2 + qv * (qv * v).sum(-1, keepdims=True) * 2


def heading_quat_inv(q):
    """np_transform_utils.calc_heading_quat_inv of remove_base_rot(q) (the heading convention of the self observation)."""
    base = np.array([0.5, -0.5, -0.5, -0.5])
    q = quat_mul(q, np.broadcast_to(base, q.shape))
    x = quat_rotate(q, np.broadcast_to(np.array([1.0, 0, 0]), q.shape[:-1] + (3,)))
    h = np.arctan2(x[..., 1], x[..., 0])
    out = np.zeros(q.shape)
    out[..., 0], out[..., 3] = np.cos(-h / 2), np.sin(-h / 2)
    return out


def tan_norm(q):
    return np.concatenate([quat_rotate(q, np.broadcast_to(np.array([1.0, 0, 0]), q.shape[:-1] + (3,))),
                           quat_rotate(q, np.broadcast_to(np.array([0, 0, 1.0]), q.shape[:-1] + (3,)))], -1)


def imitation_obs(body_pos, body_rot, body_vel, body_ang_vel, ref_pos, ref_rot, ref_vel, ref_ang_vel):
    """All [N,J,*] (quaternions wxyz).  PHC's compute_imitation_observations_v6 with one future step:
    per body [dpos(3), drot tan-norm(6), dvel(3), dangvel(3), ref pos rel. root(3), ref rot(6)] in the heading frame -> [N, J*24]."""
    N, J = body_pos.shape[:2]
    hinv = np.repeat(heading_quat_inv(body_rot[:, 0])[:, None], J, 1)
    hq = quat_conj(hinv)
    dpos = quat_rotate(hinv, ref_pos - body_pos)
    drot = quat_mul(quat_mul(hinv, quat_mul(ref_rot, quat_conj(body_rot))), hq)
    dvel = quat_rotate(hinv, ref_vel - body_vel)
    dang = quat_rotate(hinv, ref_ang_vel - body_ang_vel)
    lpos = quat_rotate(hinv, ref_pos - body_pos[:, :1])
    lrot = quat_mul(hinv, ref_rot)
    return np.concatenate([dpos.reshape(N, -1), tan_norm(drot).reshape(N, -1), dvel.reshape(N, -1), dang.reshape(N, -1),
                           lpos.reshape(N, -1), tan_norm(lrot).reshape(N, -1)], -1)


REWARD_K = (100.0, 10.0, 0.1, 0.1)      # k_pos, k_rot, k_vel, k_ang_vel
REWARD_W = (0.5, 0.3, 0.1, 0.1)


def imitation_reward(body_pos, body_rot, body_vel, body_ang_vel, ref_pos, ref_rot, ref_vel, ref_ang_vel, k=REWARD_K, w=REWARD_W):
    e_pos = ((ref_pos - body_pos) ** 2).mean(-1).mean(-1)
    dq = quat_mul(ref_rot, quat_conj(body_rot))
    dq = dq / np.linalg.norm(dq, axis=-1, keepdims=True)
    ang = 2 * np.arccos(np.clip(np.abs(dq[..., 0]), 0, 1))
    e_rot = (ang ** 2).mean(-1)
    e_vel = ((ref_vel - body_vel) ** 2).mean(-1).mean(-1)
    e_ang = ((ref_ang_vel - body_ang_vel) ** 2).mean(-1).mean(-1)
    parts = np.stack([np.exp(-k[0] * e_pos), np.exp(-k[1] * e_rot), np.exp(-k[2] * e_vel), np.exp(-k[3] * e_ang)], -1)
    return (parts * np.array(w)).sum(-1), parts


def imitation_reset(body_pos, ref_pos, termination_distance=0.25):
    """terminated when the mean over bodies of the distance to the reference body exceeds the threshold (PHC training rule)."""
    return np.linalg.norm(body_pos - ref_pos, axis=-1).mean(-1) > termination_distance
