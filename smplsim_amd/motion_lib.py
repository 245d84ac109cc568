"""Motion library on the device: the reference's MotionLibSMPL over libsmplsim_hip.so.

Mirrors `MotionLibBase` / `MotionLibSMPL` (reference smpl_sim/smpllib/motion_lib_base.py:38-458,
smpl_sim/smpllib/motion_lib_smpl.py:50-155): same clip format (the AMASS pickles: {key: {"pose_aa" [T,72|156],
"trans" [T,3], "fps"}}), same attribute names for the cooked arrays (gts, grs, lrs, gvs, gavs, grvs, gravs, dvs,
dof_pos, qpos, qvel, length_starts, _motion_lengths, ...), same sampling interface (load_motions, sample_motions,
sample_time, get_motion_state, get_motion_state_intervaled, termination-history based re-weighting).  Differences,
all on purpose:

  * the per-clip torch forward kinematics in worker processes (load_motion_with_skeleton, motion_lib_smpl.py:93-155)
    is three HIP launches over every frame of every selected clip (ss_motion_cook); the arrays are torch tensors in HBM;
  * joint offsets come from the compiled MJCF (body positions) instead of SMPL_Parser + betas — the SMPL model files
    are not redistributable; per-clip offsets ([M,J,3], e.g. from the user's own SMPL_Parser) can be passed in;
  * fix_trans_height (motion_lib_smpl.py:66-90) needs the SMPL mesh: FixHeightMode.no_fix is native, full_fix / ankle_fix
    need a `height_fix` callable supplied by the user (called per clip on the host); FixHeightMode.geom_fix (not in the
    reference) does full_fix on the device with the MJCF's collision geoms standing in for the mesh vertices.

There is no CPU path: without a GPU (and libsmplsim_hip.so) construction fails.
"""
import ctypes as C
import os
from enum import Enum

import numpy as np
import torch

from . import _cabi

SMPL_BONE_ORDER_NAMES = ["Pelvis", "L_Hip", "R_Hip", "Torso", "L_Knee", "R_Knee", "Spine", "L_Ankle", "R_Ankle", "Chest", "L_Toe",
                         "R_Toe", "Neck", "L_Thorax", "R_Thorax", "Head", "L_Shoulder", "R_Shoulder", "L_Elbow", "R_Elbow",
                         "L_Wrist", "R_Wrist", "L_Hand", "R_Hand"]


class FixHeightMode(Enum):          # motion_lib_base.py:28-31
    no_fix = 0
    full_fix = 1
    ankle_fix = 2
    geom_fix = 3                    # not in the reference: full_fix with the MJCF's collision geoms in place of the SMPL mesh


class Skeleton:
    """Body tree in MuJoCo (MJCF depth-first) order + the SMPL joint order permutation (Humanoid_Batch.__init__,
    torch_smpl_humanoid_batch.py:38-78)."""

    def __init__(self, body_names, parents, offsets, smpl_order_names=None, geoms=None):
        self.body_names = list(body_names)
        self.geoms = geoms          # (type [J], size [J,3], pos [J,3], quat [J,4]) of the collision geoms, for FixHeightMode.geom_fix
        self.parents = np.asarray(parents, np.int32)
        self.offsets = np.round(np.asarray(offsets, np.float32), decimals=5)        # update_model rounds to 5 decimals (:113)
        order = list(smpl_order_names) if smpl_order_names is not None else (
            SMPL_BONE_ORDER_NAMES if set(body_names) == set(SMPL_BONE_ORDER_NAMES) else list(body_names))
        self.smpl_2_mujoco = np.array([order.index(n) for n in self.body_names], np.int32)
        self.mujoco_2_smpl = np.array([self.body_names.index(n) for n in order], np.int32)
        self.num_joints = len(self.body_names)

    @classmethod
    def from_model_const(cls, mc, smpl_order_names=None):
        """From a compiled MJCF (smplsim_amd.mjcf.compile_mjcf): offsets = body positions in the parent frame."""
        return cls(mc.body_names, mc.body_parent, mc.body_pos, smpl_order_names,
                   geoms=(np.asarray(mc.geom_type), np.asarray(mc.geom_size, np.float32), np.asarray(mc.geom_pos, np.float32),
                          np.asarray(mc.geom_quat, np.float32)))


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def qpos_to_pose_aa(qpos, skeleton):
    """Simulator state -> SMPL pose: the inverse of the qpos the motion library cooks (Humanoid_Batch.qpos_to_pose_aa_torch /
    _numpy, torch_smpl_humanoid_batch.py:239-265).  qpos [B, 7+3(J-1)] (root pos, root quat wxyz, XYZ Euler hinges, MuJoCo body
    order) -> (root translation [B,3] without the root joint offset, pose_aa [B,J,3] in SMPL joint order).  Plain torch ops on
    whatever device qpos lives on."""
    B, J = qpos.shape[0], skeleton.num_joints
    e = qpos[:, 7:].reshape(B, J - 1, 3)
    # R = Rx Ry Rz (intrinsic XYZ), as a quaternion product qx * qy * qz
    hx, hy, hz = e[..., 0] / 2, e[..., 1] / 2, e[..., 2] / 2
    cx, sx, cy, sy, cz, sz = hx.cos(), hx.sin(), hy.cos(), hy.sin(), hz.cos(), hz.sin()
    qb = torch.stack([cx * cy * cz - sx * sy * sz, sx * cy * cz + cx * sy * sz, cx * sy * cz - sx * cy * sz, cx * cy * sz + sx * sy * cz], -1)
    quat = torch.cat([qpos[:, None, 3:7], qb], 1)                                   # [B,J,4] wxyz, MuJoCo order
    quat = quat / quat.norm(dim=-1, keepdim=True)
    quat = torch.where(quat[..., :1] < 0, -quat, quat)
    v = quat[..., 1:]
    n = v.norm(dim=-1, keepdim=True)
    ang = 2 * torch.atan2(n, quat[..., :1])
    aa = torch.where(n > 1e-8, v / n.clamp(min=1e-12) * ang, 2 * v)
    m2s = torch.as_tensor(skeleton.mujoco_2_smpl, device=qpos.device, dtype=torch.long)
    root = qpos[:, :3] - torch.as_tensor(skeleton.offsets[0], device=qpos.device, dtype=qpos.dtype)
    return root, aa[:, m2s]


class MotionLibSMPL:
    def __init__(self, motion_file, skeleton, device=0, fix_height=FixHeightMode.no_fix, min_length=-1, max_length=-1,
                 randomrize_heading=False, filter_vel=True, height_fix=None, seed=0, mesh_parsers=None):
        """motion_file: path of a joblib/pickle file, a directory of *.pkl files, or the dict itself.
        mesh_parsers: what the reference builds from the SMPL model files (motion_lib_smpl.py:48-64): {"0" | "1" | "2": parser} by
        gender with `get_joints_verts(pose_aa, betas, trans) -> (vertices, joints)`, `lbs_weights`, `joint_names`; with it
        FixHeightMode.full_fix / ankle_fix run `fix_trans_height` on the mesh like the reference (shape_params of load_motions give
        gender + betas per clip).  The parsers need the licensed SMPL files, so they are the caller's; `height_fix` (a plain
        callable) and FixHeightMode.geom_fix (collision geoms instead of the mesh, on the device) remain as alternatives."""
        from . import batch
        from ._lib import lib
        self.device = batch._shard_device(device.index if isinstance(device, torch.device) else device)   # raises without a GPU
        self._lib = lib()
        self.skeleton = skeleton
        self.fix_height, self.height_fix, self.mesh_parsers = fix_height, height_fix, mesh_parsers
        if fix_height in (FixHeightMode.full_fix, FixHeightMode.ankle_fix) and height_fix is None and mesh_parsers is None:
            raise ValueError("full_fix / ankle_fix need the SMPL mesh: pass mesh_parsers (the reference's SMPL parsers), a "
                             "height_fix(pose_aa, trans) callable, or use geom_fix")
        if fix_height == FixHeightMode.geom_fix and skeleton.geoms is None:
            raise ValueError("geom_fix needs the skeleton's collision geoms (Skeleton.from_model_const)")
        self.max_length, self.randomrize_heading, self.filter_vel = max_length, randomrize_heading, bool(filter_vel)
        self.rng = np.random.default_rng(seed)
        self.dtype = np.float32
        self.curr_failed_keys = []
        self.load_data(motion_file, min_length)
        self.setup_constants()
        self._num_motions = 0

    # ---- MotionLibBase.load_data / setup_constants (:51-89)
    def load_data(self, motion_file, min_length=-1):
        if isinstance(motion_file, dict):
            data = motion_file
        elif os.path.isfile(motion_file):
            import joblib
            data = joblib.load(motion_file)
        else:
            import glob
            import joblib
            data = {}
            for f in sorted(glob.glob(os.path.join(motion_file, "*.pkl"))):
                data.update(joblib.load(f))
        if min_length != -1:
            data = {k: v for k, v in data.items() if len(v["pose_aa"]) >= min_length}
        if not data:
            raise ValueError("no motion clips")
        self._motion_data_load = data
        self._motion_data_list = list(data.values())
        self._motion_data_keys = np.array(list(data.keys()))
        self._num_unique_motions = len(self._motion_data_list)

    def setup_constants(self):
        n = self._num_unique_motions
        self._curr_motion_ids = None
        self._termination_history = np.zeros(n)
        self._success_rate = np.zeros(n)
        self._sampling_history = np.zeros(n)
        self._sampling_prob = np.ones(n) / n
        self._sampling_batch_prob = None

    # ---- MotionLibBase.load_motions (:99-208)
    def load_motions(self, num_motions=None, shape_params=None, random_sample=True, start_idx=0, offsets=None, silent=True):
        """Select `num_motions` clips (default: one per entry of shape_params, as the reference does) and cook them.
        offsets: optional [M,J,3] per-clip joint offsets (MuJoCo order); default = the skeleton's for every clip."""
        if num_motions is None:
            num_motions = len(shape_params) if shape_params is not None else self._num_unique_motions
        M, J = int(num_motions), self.skeleton.num_joints
        if random_sample:
            idx = self.rng.choice(self._num_unique_motions, size=M, p=self._sampling_prob, replace=True)
        else:
            idx = np.remainder(np.arange(M) + start_idx, self._num_unique_motions)
        self._curr_motion_ids = idx
        self.curr_motion_keys = self._motion_data_keys[idx]
        self._sampling_batch_prob = self._sampling_prob[idx] / self._sampling_prob[idx].sum()

        poses, transs, nfs, fpss = [], [], [], []
        for f_, i in enumerate(idx):                             # f_: position in the loaded batch (shape_params[f_], like the reference)
            clip = self._motion_data_list[i]
            fps = float(clip.get("fps", 30))
            pose = np.asarray(clip["pose_aa"], np.float32)
            trans = np.asarray(clip["trans"] if "trans" in clip else clip["trans_orig"], np.float32)
            T = pose.shape[0]
            if self.max_length != -1 and T >= self.max_length:          # motion_lib_smpl.py:108-113
                s = int(self.rng.integers(0, T - self.max_length + 1))
                pose, trans = pose[s:s + self.max_length], trans[s:s + self.max_length]
            pose = pose.reshape(pose.shape[0], -1)
            if pose.shape[1] == 156 and J == 24:                        # SMPL-H pickles: body joints + zero hands (:118-119)
                pose = np.concatenate([pose[:, :66], np.zeros((pose.shape[0], 6), np.float32)], 1)
            if pose.shape[1] != 3 * J:
                raise ValueError(f"pose_aa has {pose.shape[1]} columns, the skeleton needs {3 * J}")
            pose = pose.reshape(-1, J, 3).copy()
            trans = trans.copy()
            if pose.shape[0] < 2:
                raise ValueError("a clip needs at least 2 frames")
            if self.randomrize_heading:                                 # motion_lib_smpl.py:128-134
                from scipy.spatial.transform import Rotation as sRot
                rot = sRot.from_euler("xyz", [0.0, 0.0, np.pi * (2 * self.rng.random() - 1.0)])
                pose[:, 0] = (rot * sRot.from_rotvec(pose[:, 0])).as_rotvec().astype(np.float32)
                trans = (trans @ rot.as_matrix().T.astype(np.float32)).astype(np.float32)
            if self.fix_height in (FixHeightMode.full_fix, FixHeightMode.ankle_fix):
                if self.mesh_parsers is not None:
                    gb = np.zeros(17, np.float32) if shape_params is None else np.asarray(shape_params[f_], np.float32)
                    trans, _ = self.fix_trans_height(pose, trans, gb, self.mesh_parsers, self.fix_height)
                else:
                    trans = np.asarray(self.height_fix(pose, trans), np.float32)
            poses.append(pose); transs.append(trans); nfs.append(pose.shape[0]); fpss.append(fps)

        nf = np.array(nfs, np.int64)
        self._motion_num_frames = nf
        self._motion_fps = np.array(fpss, self.dtype)
        self._motion_dt = (1.0 / self._motion_fps).astype(self.dtype)
        self._motion_lengths = (1.0 / self._motion_fps * (nf - 1)).astype(self.dtype)
        self._motion_aa = np.concatenate(poses).reshape(-1, 3 * J)
        self._motion_bodies = np.zeros((M, 17), self.dtype) if shape_params is None else np.stack(shape_params).astype(self.dtype)
        shifted = np.roll(nf, 1)
        shifted[0] = 0
        self.length_starts = shifted.cumsum(0)
        self.motion_ids = np.arange(M)
        self.num_bodies = self.num_joints = J
        self._num_motions = M
        F = int(nf.sum())
        if F >= 2 ** 31 // (J * 9):
            raise ValueError("too many frames for 32-bit indexing")
        dev = self.device

        def up(a, dt):
            return torch.as_tensor(np.ascontiguousarray(a, dtype=dt)).to(dev)

        off = np.broadcast_to(self.skeleton.offsets, (M, J, 3)) if offsets is None else np.asarray(offsets, np.float32).reshape(M, J, 3)
        self._d = dict(
            length_starts=up(self.length_starts, np.int32), motion_num_frames=up(nf, np.int32), motion_dt=up(self._motion_dt, np.float32),
            motion_lengths=up(self._motion_lengths, np.float32), frame_motion=up(np.repeat(np.arange(M), nf), np.int32),
            pose_aa=up(self._motion_aa, np.float32), trans=up(np.concatenate(transs), np.float32), offsets=up(off, np.float32))
        f32 = dict(dtype=torch.float32, device=dev)
        self.gts = torch.empty(F, J, 3, **f32); self.grs = torch.empty(F, J, 4, **f32); self.lrs = torch.empty(F, J, 4, **f32)
        self.gvs = torch.empty(F, J, 3, **f32); self.gavs = torch.empty(F, J, 3, **f32)
        self.dof_pos = torch.empty(F, J - 1, 3, **f32); self.dvs = torch.empty(F, J - 1, 3, **f32)
        self.qpos = torch.empty(F, 7 + 3 * (J - 1), **f32); self.qvel = torch.empty(F, 6 + 3 * (J - 1), **f32)
        self.grvs, self.gravs = self.gvs[:, 0], self.gavs[:, 0]
        self._d.update(gts=self.gts, grs=self.grs, lrs=self.lrs, gvs=self.gvs, gavs=self.gavs, dof_pos=self.dof_pos, dvs=self.dvs,
                       qpos=self.qpos, qvel=self.qvel)
        self.data = _cabi.MotionData(M, F, J, *[_ptr(self._d[n]) for n in _cabi.MOTION_DATA_ARRAYS])
        sk = self.skeleton
        self._sk_keep = (np.ascontiguousarray(sk.parents, np.int32), np.ascontiguousarray(sk.smpl_2_mujoco, np.int32))
        skel = _cabi.Skeleton(J, self._sk_keep[0].ctypes.data_as(C.c_void_p), self._sk_keep[1].ctypes.data_as(C.c_void_p))
        self._check(self._lib.ss_motion_cook(C.byref(skel), C.byref(self.data), int(self.filter_vel), self._stream()))
        if self.fix_height == FixHeightMode.geom_fix:
            self._geom_height_fix()
        self.motion_lengths_t, self.motion_num_frames_t = self._d["motion_lengths"], self._d["motion_num_frames"]
        cdf = np.cumsum(self._sampling_batch_prob)
        cdf[-1] = 1.0 + 1e-6                                        # rand < 1 always lands in a clip
        self.sampling_cdf = up(cdf, np.float32)
        self.load_count = getattr(self, "load_count", 0) + 1    # holders of device pointers into this library (fused imitation step) rebind
        if not silent:
            print(f"###### Sampling {M:d} motions:", idx[:5], self.curr_motion_keys[:5],
                  f"total length of {self.get_total_length():.3f}s and {F} frames.")
        return M

    def _geom_height_fix(self, frame_check=30):
        """fix_trans_height's full_fix (motion_lib_smpl.py:66-90: lower / raise every clip so that the lowest point of its first
        30 frames touches z = 0) with the model's collision geoms standing in for the SMPL mesh vertices.  A pure translation:
        applied to the cooked positions in place (velocities and rotations do not change), on the device."""
        gtype, gsize, gpos, gquat = self.skeleton.geoms
        dev, J = self.device, self.skeleton.num_joints
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
        q = self.grs                                                               # [F,J,4] wxyz body rotations
        w, x, y, z = q.unbind(-1)
        row2 = torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)   # third row of R_body
        gq = t(gquat)
        gw, gx, gy, gz = gq.unbind(-1)
        G = torch.stack([torch.stack([1 - 2 * (gy * gy + gz * gz), 2 * (gx * gy - gw * gz), 2 * (gx * gz + gw * gy)], -1),
                         torch.stack([2 * (gx * gy + gw * gz), 1 - 2 * (gx * gx + gz * gz), 2 * (gy * gz - gw * gx)], -1),
                         torch.stack([2 * (gx * gz - gw * gy), 2 * (gy * gz + gw * gx), 1 - 2 * (gx * gx + gy * gy)], -1)], -2)  # [J,3,3]
        zrow = torch.einsum("fjk,jkc->fjc", row2, G)                               # third row of R_body R_geom
        centre = self.gts[..., 2] + (row2 * t(gpos)).sum(-1)
        size = t(gsize)
        is_box = torch.as_tensor(np.asarray(gtype) == 0, device=dev)
        low_box = centre - (zrow.abs() * size).sum(-1)
        low_caps = centre - zrow[..., 2].abs() * size[:, 1] - size[:, 0]           # lower end sphere of the capsule
        low = torch.where(is_box, low_box, low_caps).min(dim=1).values             # [F] lowest geom point per frame
        fm = self._d["frame_motion"].long()
        t_in = torch.arange(low.shape[0], device=dev) - self._d["length_starts"].long()[fm]
        low = torch.where(t_in < frame_check, low, torch.full_like(low, float("inf")))
        diff = torch.full((self._num_motions,), float("inf"), device=dev).scatter_reduce(0, fm, low, reduce="amin")
        self.height_offsets = diff
        self.gts[..., 2] -= diff[fm][:, None]
        self.qpos[:, 2] -= diff[fm]

    @staticmethod
    def fix_trans_height(pose_aa, trans, curr_gender_betas, mesh_parsers, fix_height_mode, frame_check=30):
        """The reference's mesh height fix (motion_lib_smpl.py:67-92): pose the SMPL mesh of the clip's shape for the first 30
        frames (usually a calibration phase) and shift the whole clip so that its lowest vertex (full_fix) — or its lowest vertex
        not skinned to the toes / hands, minus 2.5 cm (ankle_fix) — touches z = 0.  Returns (trans, shift)."""
        if fix_height_mode == FixHeightMode.no_fix:
            return trans, 0.0
        gb = np.asarray(curr_gender_betas, np.float32)
        parser = mesh_parsers[str(int(gb[0]))]
        pose = torch.as_tensor(np.asarray(pose_aa, np.float32))[:frame_check]
        tr = torch.as_tensor(np.asarray(trans, np.float32))
        with torch.no_grad():
            verts, _ = parser.get_joints_verts(pose, torch.as_tensor(gb[1:])[None], tr[:frame_check])
            verts = torch.as_tensor(verts)
            if fix_height_mode == FixHeightMode.ankle_fix:
                tol = -0.025
                owner = torch.as_tensor(np.asarray(parser.lbs_weights)).argmax(dim=1)
                names = list(parser.joint_names)
                keep = torch.ones_like(owner, dtype=torch.bool)
                for n in ("L_Toe", "R_Toe", "R_Hand", "L_Hand"):
                    keep &= owner != names.index(n)
                verts = verts[:, keep]
            else:
                tol = 0.0
            diff = float((verts[..., -1].min(dim=-1).values - tol).min())
        out = np.asarray(trans, np.float32).copy()
        out[..., -1] -= diff
        return out, diff

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"libsmplsim_hip error {rc}: {self._lib.ss_last_error().decode()}")

    def _stream(self):
        from . import batch
        return batch._launch_stream(self.device)

    # ---- bookkeeping (motion_lib_base.py:210-276)
    def num_current_motions(self):
        return self._num_motions

    def num_all_motions(self):
        return self._num_unique_motions

    def get_total_length(self):
        return float(self._motion_lengths.sum())

    def get_termination_history(self):
        return {"termination_history": self._termination_history, "failed_keys": self.curr_failed_keys}

    def set_termination_history(self, termination_history):
        self._termination_history = termination_history["termination_history"]
        self.curr_failed_keys = termination_history["failed_keys"]
        self.update_sampling_prob(self._termination_history)

    def update_hard_sampling_weight(self, failed_keys):
        if len(failed_keys) > 0:
            keys = self._motion_data_keys.tolist()
            idx = [keys.index(k) for k in failed_keys]
            self._sampling_prob[:] = 0
            self._sampling_prob[idx] = 1 / len(idx)
        else:
            self._sampling_prob = np.ones(self._num_unique_motions) / self._num_unique_motions

    def update_soft_sampling_weight(self, failed_keys):
        if len(failed_keys) > 0:
            self.curr_failed_keys = failed_keys
            keys = self._motion_data_keys.tolist()
            idx = [keys.index(k) for k in failed_keys]
            self._termination_history[idx] += 1
            self.update_sampling_prob(self._termination_history)
        else:
            self._sampling_prob = np.ones(self._num_unique_motions) / self._num_unique_motions

    def update_sampling_prob(self, termination_history):
        if len(self._sampling_prob) == len(termination_history) and termination_history.sum() > 0:
            self._sampling_prob[:] = termination_history / termination_history.sum()
            self._termination_history = termination_history
            return True
        return False

    # ---- sampling (:277-310), on the device
    def sample_motions(self, n=1, generator=None):
        p = torch.as_tensor(self._sampling_batch_prob, dtype=torch.float32, device=self.device)
        return torch.multinomial(p, n, replacement=True, generator=generator).to(torch.int32)

    def sample_time(self, motion_ids, truncate_time=None, generator=None):
        phase = torch.rand(motion_ids.shape, device=self.device, generator=generator)
        length = self.motion_lengths_t[motion_ids.long()]
        if truncate_time is not None:
            assert truncate_time >= 0.0
            length = length - truncate_time
        return phase * length

    def get_motion_length(self, motion_ids=None):
        return self.motion_lengths_t if motion_ids is None else self.motion_lengths_t[motion_ids.long()]

    def get_motion_num_steps(self, motion_ids=None):
        steps = (self._motion_num_frames * 30 / self._motion_fps).astype(int)
        return steps if motion_ids is None else steps[np.asarray(motion_ids.cpu() if torch.is_tensor(motion_ids) else motion_ids)]

    # ---- lookup (:311-423)
    def resample(self, mask, motion_ids, start_times, truncate_time=0.0, generator=None, rand=None):
        """sample_motions + sample_time for the envs with mask != 0 (None = all), in place, one launch
        (motion_ids int32 [N], start_times float32 [N], device tensors).  rand: optional [N,2] uniform draws to use."""
        N = motion_ids.shape[0]
        if rand is None:
            rand = torch.rand(N, 2, device=self.device, generator=generator)
        self._keep_rs = (rand, mask)
        self._check(self._lib.ss_motion_resample(C.byref(self.data), _ptr(mask), _ptr(rand), _ptr(self.sampling_cdf), float(truncate_time), N,
                                                 _ptr(motion_ids), _ptr(start_times), self._stream()))

    def write_state(self, motion_ids, motion_times, offset, mask, qpos, qvel):
        """Reference-state init in place: the clip's qpos / qvel at (id, time) into the rows of the simulator's qpos / qvel
        tensors whose mask byte is set."""
        st = _cabi.MotionState(**{"qpos": _ptr(qpos).value, "qvel": _ptr(qvel).value})
        self._check(self._lib.ss_motion_state_at(C.byref(self.data), _ptr(motion_ids), _ptr(motion_times), _ptr(offset), _ptr(mask),
                                                 motion_ids.shape[0], 0, C.byref(st), self._stream()))

    def _lookup(self, motion_ids, motion_times, offset, intervaled, fields):
        ids = torch.as_tensor(motion_ids, device=self.device).to(torch.int32).contiguous()
        times = torch.as_tensor(motion_times, device=self.device).to(torch.float32).contiguous()
        off = None if offset is None else torch.as_tensor(offset, device=self.device).to(torch.float32).contiguous()
        N, J = ids.shape[0], self.skeleton.num_joints
        shapes = dict(root_pos=(N, 3), root_rot=(N, 4), dof_pos=(N, 3 * (J - 1)), root_vel=(N, 3), root_ang_vel=(N, 3),
                      dof_vel=(N, 3 * (J - 1)), rg_pos=(N, J, 3), rb_rot=(N, J, 4), body_vel=(N, J, 3), body_ang_vel=(N, J, 3),
                      qpos=(N, 7 + 3 * (J - 1)), qvel=(N, 6 + 3 * (J - 1)))
        out = {k: torch.empty(shapes[k], dtype=torch.float32, device=self.device) for k in fields}
        st = _cabi.MotionState(*[_ptr(out.get(k)) for k in _cabi.MOTION_STATE_FIELDS])
        self._keep = (ids, times, off, out)
        self._check(self._lib.ss_motion_state_at(C.byref(self.data), _ptr(ids), _ptr(times), _ptr(off), None, N, int(intervaled),
                                                 C.byref(st), self._stream()))
        return out, ids

    def _frame0(self, ids, times):
        """Global index of the frame at or before `times` (f0l of the reference), float32 like the kernel."""
        idl = ids.long()
        nf = self.motion_num_frames_t[idl]
        phase = (times / self.motion_lengths_t[idl]).clamp(0.0, 1.0)
        i0 = (phase * (nf - 1).to(torch.float32)).to(torch.int64)
        return torch.where(nf < 2, torch.zeros_like(i0), i0) + self._d["length_starts"][idl].long()

    def get_motion_state(self, motion_ids, motion_times, offset=None, with_qpos=False):
        fields = ["root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "rg_pos", "rb_rot", "body_vel", "body_ang_vel"]
        out, ids = self._lookup(motion_ids, motion_times, offset, False, fields + (["qpos", "qvel"] if with_qpos else []))
        out["motion_bodies"] = torch.as_tensor(self._motion_bodies, device=self.device)[ids.long()]
        out["motion_aa"] = self._d["pose_aa"][self._frame0(ids, self._keep[1])]      # raw clip pose of frame f0, as in the reference
        return out

    def get_motion_state_intervaled(self, motion_ids, motion_times, offset=None):
        out, ids = self._lookup(motion_ids, motion_times, offset, True, list(_cabi.MOTION_STATE_FIELDS))
        out["xpos"], out["xquat"] = out.pop("rg_pos"), out.pop("rb_rot")
        out["dof_pos"] = out["dof_pos"].view(ids.shape[0], -1, 3)
        out["motion_bodies"] = torch.as_tensor(self._motion_bodies, device=self.device)[ids.long()]
        return out

    def get_root_pos_smpl(self, motion_ids, motion_times):
        out, _ = self._lookup(motion_ids, motion_times, None, False, ["root_pos"])
        return out
