"""Tracking-quality metrics of the imitation task on the device.

Mirrors `compute_metrics_lite`, `p_mpjpe`, `compute_error_vel`, `compute_error_accel` of the reference
(smpl_sim/smpllib/smpl_eval.py:58-94,98-138,298-339) on torch tensors, so a rollout recorded in HBM is scored without
leaving the device (the reference loops over sequences in NumPy / SciPy on the host).  Sequences are [T, J, 3] positions in
metres (and [T, J, 4] xyzw quaternions, like the reference); results are in millimetres / radians, per frame.
"""
import torch


def compute_error_vel(joints_gt, joints_pred):
    """mean over joints of |finite-difference velocity error| -> [T-1]  (smpl_eval.py:331-339)"""
    v = (joints_pred[1:] - joints_pred[:-1]) - (joints_gt[1:] - joints_gt[:-1])
    return torch.linalg.norm(v, dim=2).mean(dim=1)


def compute_error_accel(joints_gt, joints_pred):
    """mean over joints of |second-difference error| -> [T-2]  (smpl_eval.py:298-323)"""
    a_gt = joints_gt[:-2] - 2 * joints_gt[1:-1] + joints_gt[2:]
    a_pr = joints_pred[:-2] - 2 * joints_pred[1:-1] + joints_pred[2:]
    return torch.linalg.norm(a_pr - a_gt, dim=2).mean(dim=1)


def p_mpjpe(predicted, target):
    """MPJPE after similarity (scale, rotation, translation) alignment per frame, [T, J]  (smpl_eval.py:98-138)"""
    muX, muY = target.mean(dim=1, keepdim=True), predicted.mean(dim=1, keepdim=True)
    X0, Y0 = target - muX, predicted - muY
    normX = torch.sqrt((X0 ** 2).sum(dim=(1, 2), keepdim=True))
    normY = torch.sqrt((Y0 ** 2).sum(dim=(1, 2), keepdim=True))
    X0, Y0 = X0 / normX, Y0 / normY
    H = X0.transpose(1, 2) @ Y0
    U, s, Vt = torch.linalg.svd(H)
    V = Vt.transpose(1, 2)
    R = V @ U.transpose(1, 2)
    sign = torch.sign(torch.linalg.det(R))                       # no reflections
    V = torch.cat([V[:, :, :-1], V[:, :, -1:] * sign[:, None, None]], dim=2)
    s = torch.cat([s[:, :-1], s[:, -1:] * sign[:, None]], dim=1)
    R = V @ U.transpose(1, 2)
    a = s.sum(dim=1, keepdim=True)[:, :, None] * normX / normY
    t = muX - a * (muY @ R)
    return torch.linalg.norm(a * (predicted @ R) + t - target, dim=2)


def rotation_error(rot_gt, rot_pred):
    """angle of q_gt * q_pred^-1 per body, xyzw quaternions [..., 4] -> [...]  (|rotvec| of smpl_eval.py:82)"""
    xg, wg = rot_gt[..., :3], rot_gt[..., 3:]
    xp, wp = -rot_pred[..., :3], rot_pred[..., 3:]
    w = wg * wp - (xg * xp).sum(-1, keepdim=True)
    v = wg * xp + wp * xg + torch.linalg.cross(xg, xp)
    return 2 * torch.atan2(torch.linalg.norm(v, dim=-1), w[..., 0].abs())


def compute_metrics_lite(pred_pos_all, gt_pos_all, pred_rot_all=None, gt_rot_all=None, root_idx=0, concatenate=True):
    """Per-sequence metrics (lists of [T,J,3] tensors in, dict of tensors out), names as in the reference."""
    out = {k: [] for k in ("mpjpe_g", "mpjpe_l", "mpjpe_pa", "accel_dist", "vel_dist")}
    if pred_rot_all is not None and gt_rot_all is not None:
        out["rot_error"] = []
    for i in range(len(pred_pos_all)):
        pred, gt = pred_pos_all[i], gt_pos_all[i]
        out["mpjpe_g"].append(torch.linalg.norm(gt - pred, dim=2) * 1000)
        out["vel_dist"].append(compute_error_vel(pred, gt) * 1000)
        out["accel_dist"].append(compute_error_accel(pred, gt) * 1000)
        pred_l, gt_l = pred - pred[:, [root_idx]], gt - gt[:, [root_idx]]
        out["mpjpe_pa"].append(p_mpjpe(pred_l, gt_l) * 1000)
        out["mpjpe_l"].append(torch.linalg.norm(pred_l - gt_l, dim=2) * 1000)
        if "rot_error" in out:
            out["rot_error"].append(rotation_error(gt_rot_all[i].reshape(-1, 4), pred_rot_all[i].reshape(-1, 4)))
    return {k: torch.cat(v) for k, v in out.items()} if concatenate else out
