#!/usr/bin/env python3
"""Golden vectors for the motion-library row from the REFERENCE's own torch/NumPy code.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_motion.py
writes tests/golden/motion_vectors.npz (committed).  Functions called, unmodified:

  * Humanoid_Batch.fk_batch / forward_kinematics_batch / _compute_velocity / _compute_angular_velocity
                                                   smpl_sim/smpllib/torch_smpl_humanoid_batch.py:118-228
    (the object is made with __new__: its constructor needs the SMPL model files, which are not in the repo; the
     joint offsets it would derive from betas are replaced by the body positions of the SMPL fixture MJCF, rounded to
     5 decimals exactly as update_model does, :113)
  * pytorch3d_transforms.{axis_angle_to_quaternion, quaternion_to_matrix, matrix_to_quaternion,
    matrix_to_euler_angles, fix_continous_dof, quat_mul_norm, quat_angle_axis}   smpl_sim/utils/pytorch3d_transforms.py
  * MotionLibBase._calc_frame_blend / get_motion_state_intervaled          smpl_sim/smpllib/motion_lib_base.py:311-355,442-453
  * torch_utils.slerp                                                      smpl_sim/utils/torch_utils.py:405-426
  * smpl_eval.compute_metrics_lite (p_mpjpe, compute_error_vel/accel)      smpl_sim/smpllib/smpl_eval.py:58-138,298-339

MotionLibBase.get_motion_state (:359-423) cannot run as written (it indexes NumPy arrays with the float frame numbers
of _calc_frame_blend and calls Tensor.unsqueeze on NumPy arrays); its blend is pinned through slerp and the linear
interpolation formula only.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.jit  # noqa: F401  (must be imported before the auto-mock finder is installed)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

SMPL_PARENTS = [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 17, 11, 19, 20, 21, 22]


def smooth_clip(rs, T, amp, root_spin=0.0):
    """A smooth random pose sequence [T,24,3] (axis-angle, SMPL joint order) and root translation [T,3]."""
    base = rs.normal(size=(1, 24, 3)) * amp
    freq = rs.uniform(0.5, 2.0, size=(1, 24, 3))
    phase = rs.uniform(0, 2 * np.pi, size=(1, 24, 3))
    t = np.arange(T)[:, None, None] / 30.0
    pose = base + amp * np.sin(2 * np.pi * freq * t + phase)
    pose[:, 0, 2] += root_spin * t[:, 0, 0]
    trans = np.stack([0.8 * t[:, 0, 0], 0.1 * np.sin(t[:, 0, 0]), 0.9 + 0.02 * np.cos(3 * t[:, 0, 0])], axis=-1)
    return pose.astype(np.float32), trans.astype(np.float32)


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, mg.REF)
    mg._stub_modules()
    mg._shell_packages()

    class AttrDict(dict):                     # stand-in for easydict.EasyDict (not installed): a dict with attribute access
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    ed = types.ModuleType("easydict")
    ed.EasyDict = AttrDict
    sys.modules["easydict"] = ed
    sys.meta_path.append(mg._AutoMock())
    import smpl_sim.smpllib.torch_smpl_humanoid_batch as hb
    import smpl_sim.smpllib.motion_lib_base as mlb
    import smpl_sim.utils.torch_utils as tu
    from smpl_sim.smpllib.smpl_joint_names import SMPL_BONE_ORDER_NAMES, SMPL_MUJOCO_NAMES

    import xml.etree.ElementTree as ET
    root = ET.parse(os.path.join(mg.REF, "smpl_sim/data/assets/mjcf/smpl_humanoid.xml")).getroot()
    pos = {b.get("name"): [float(x) for x in b.get("pos").split()] for b in root.iter("body")}
    offsets = np.round(np.array([pos[n] for n in SMPL_MUJOCO_NAMES], np.float32), decimals=5)

    rs = np.random.default_rng(20260925)
    out = {"offsets": offsets, "parents": np.array(SMPL_PARENTS, np.int32),
           "smpl_2_mujoco": np.array([SMPL_BONE_ORDER_NAMES.index(n) for n in SMPL_MUJOCO_NAMES], np.int32)}

    clips = [smooth_clip(rs, 40, 0.35), smooth_clip(rs, 25, 0.6, root_spin=2.0), smooth_clip(rs, 61, 1.3, root_spin=-4.0)]
    fps = [30, 30, 60]
    out["num_frames"] = np.array([c[0].shape[0] for c in clips], np.int32)
    out["fps"] = np.array(fps, np.float32)
    out["pose_aa"] = np.concatenate([c[0] for c in clips])
    out["trans"] = np.concatenate([c[1] for c in clips])

    keys = ["global_translation", "global_rotation", "local_rotation", "global_root_velocity", "global_root_angular_velocity",
            "global_angular_velocity", "global_velocity", "dof_pos", "dof_vels", "qpos", "qvel"]
    for filt in (True, False):
        h = hb.Humanoid_Batch.__new__(hb.Humanoid_Batch)
        h._parents = SMPL_PARENTS
        h.smpl_2_mujoco = out["smpl_2_mujoco"].tolist()
        h._offsets = torch.from_numpy(offsets[None].copy())
        h.filter_vel = filt
        acc = {k: [] for k in keys}
        for (pose, trans), f in zip(clips, fps):
            h.dt = 1 / f
            r = h.fk_batch(torch.from_numpy(pose[None].copy()), torch.from_numpy(trans[None].copy()), return_full=True, count_offset=True)
            for k in keys:
                acc[k].append(np.asarray(r[k][0]))
        for k in keys:
            out[("f_" if filt else "n_") + k] = np.concatenate(acc[k]).astype(np.float32)

    # ---- MotionLibBase frame lookup on the cooked arrays (filtered variant), exactly as load_motions lays them out (:176-197)
    lib = mlb.MotionLibBase.__new__(mlb.MotionLibBase)
    nf = out["num_frames"].astype(np.int64)
    lib._motion_num_frames = nf
    lib._motion_dt = (1.0 / out["fps"]).astype(np.float32)
    lib._motion_lengths = (1.0 / out["fps"] * (nf - 1)).astype(np.float32)
    lib._motion_fps = out["fps"]
    lib.num_bodies = 24
    lib.gts, lib.grs, lib.gvs, lib.gavs = out["f_global_translation"], out["f_global_rotation"], out["f_global_velocity"], out["f_global_angular_velocity"]
    lib.dvs, lib.dof_pos, lib.qpos, lib.qvel = out["f_dof_vels"], out["f_dof_pos"], out["f_qpos"], out["f_qvel"]
    lib._motion_aa = out["pose_aa"]
    lib._motion_bodies = np.zeros((3, 17), np.float32)
    shifted = np.roll(nf, 1)
    shifted[0] = 0
    lib.length_starts = shifted.cumsum(0)
    ids = rs.integers(0, 3, size=64)
    times = (rs.uniform(-0.1, 1.15, size=64) * lib._motion_lengths[ids]).astype(np.float32)
    i0, i1, bl = lib._calc_frame_blend(times, lib._motion_lengths[ids], lib._motion_num_frames[ids], lib._motion_dt[ids])
    out["q_ids"], out["q_times"] = ids.astype(np.int32), times
    out["q_idx0"], out["q_idx1"], out["q_blend"] = i0, i1, bl
    offs = rs.normal(size=(64, 3)).astype(np.float32)
    st = lib.get_motion_state_intervaled(ids, times, offset=offs)
    out["q_offset"] = offs
    for k in ("root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "xpos", "xquat", "body_vel", "body_ang_vel", "qpos", "qvel"):
        out["iv_" + k] = np.asarray(st[k])

    # ---- slerp
    q0 = rs.normal(size=(256, 4)).astype(np.float32)
    q1 = rs.normal(size=(256, 4)).astype(np.float32)
    q0 /= np.linalg.norm(q0, axis=-1, keepdims=True)
    q1 /= np.linalg.norm(q1, axis=-1, keepdims=True)
    q1[:16] = q0[:16]                                   # identical (cos = 1)
    q1[16:32] = -q0[16:32]                              # antipodal representation of the same rotation
    q1[32:48] = q0[32:48] + 1e-4 * rs.normal(size=(16, 4)).astype(np.float32)
    q1[32:48] /= np.linalg.norm(q1[32:48], axis=-1, keepdims=True)
    t = rs.uniform(0, 1, size=(256, 1)).astype(np.float32)
    out["sl_q0"], out["sl_q1"], out["sl_t"] = q0, q1, t
    out["sl_out"] = tu.slerp(torch.from_numpy(q0), torch.from_numpy(q1), torch.from_numpy(t)).numpy()

    # ---- tracking metrics (smpl_sim/smpllib/smpl_eval.py:58-94): two sequences of different length
    import smpl_sim.smpllib.smpl_eval as ev
    pp, gp, pr, gr = [], [], [], []
    for T in (17, 9):
        gt = rs.normal(size=(T, 24, 3))
        pred = gt + 0.05 * rs.normal(size=(T, 24, 3)) + 0.02 * np.sin(np.arange(T))[:, None, None]
        qg = rs.normal(size=(T, 24, 4)); qg /= np.linalg.norm(qg, axis=-1, keepdims=True)
        qp = qg + 0.1 * rs.normal(size=(T, 24, 4)); qp /= np.linalg.norm(qp, axis=-1, keepdims=True)
        pp.append(pred); gp.append(gt); pr.append(qp); gr.append(qg)
    m = ev.compute_metrics_lite(pp, gp, pr, gr, use_tqdm=False)
    for k, v in m.items():
        out["ev_" + k] = np.asarray(v)
    for i in range(2):
        out[f"ev_in_pred{i}"], out[f"ev_in_gt{i}"], out[f"ev_in_rpred{i}"], out[f"ev_in_rgt{i}"] = pp[i], gp[i], pr[i], gr[i]

    path = os.path.join(HERE, "motion_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
