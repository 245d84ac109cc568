"""Motion-imitation rollouts on one GPU shard (BASELINE config 4).

The reference ships the pieces of this task (motion library, batched FK, the self observation that is "an exact
replica of PHC's", humanoid_env.py:636) but not the task itself; this env wires them the way PHC's HumanoidIm does:

  reset   reference-state init: sample a clip and a start time per env, write the clip's qpos/qvel at that time into
          the simulator state (ss_motion_state_at -> ss_reset with StateInit "External")
  step    ss_step (15 x (Stable-PD + mj_step), self observation)  ->  ss_kinematics (xpos, xmat)  ->
          ss_imitation_step (clip lookup at t and t + dt, task observation, tracking reward, early termination)
          obs = [self obs | task obs];  truncated when the clip ends;  finished envs are re-initialised in place.

All buffers are torch tensors on the shard's device; nothing leaves HBM between launches.
"""
import ctypes as C

import torch

from . import _cabi
from .batch import SMPLSimVecEnv, _check, _ptr
from ._lib import lib


class SMPLSimImitationVecEnv:
    def __init__(self, num_envs, motion_lib, model=None, device=0, self_obs_v=2, control_freq_inv=15, sim_timestep_inv=450,
                 termination_distance=0.25, reward_k=(100.0, 10.0, 0.1, 0.1), reward_w=(0.5, 0.3, 0.1, 0.1),
                 random_start=True, autoreset=True, seed=0, **env_kw):
        self.base = SMPLSimVecEnv(num_envs, model=model, device=device, task="HumanoidEnv", state_init="External",
                                  self_obs_v=self_obs_v, control_freq_inv=control_freq_inv, episode_length=2 ** 30,
                                  autoreset=False, fused_autoreset=False, seed=seed, **env_kw)
        b = self.base
        self.motion_lib = motion_lib
        if motion_lib.skeleton.num_joints != b.nbody:
            raise ValueError("motion library skeleton and simulated model have different body counts")
        if motion_lib.device != b.device:
            raise ValueError("motion library and env must live on the same device")
        self.num_envs, self.device, self.nbody = b.num_envs, b.device, b.nbody
        self.dt = control_freq_inv / float(sim_timestep_inv)
        self.cfg = _cabi.ImitationCfg(*reward_k, *reward_w, termination_distance, self.dt)
        self.random_start, self.autoreset = random_start, autoreset
        N, J, dev = self.num_envs, self.nbody, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.motion_ids = torch.zeros(N, dtype=torch.int32, device=dev)
        self.start_times = torch.zeros(N, **f32)
        self.motion_len = torch.zeros(N, **f32)
        self.times = torch.zeros(N, **f32)
        self.offset = torch.zeros(N, 3, **f32)
        self.xpos = torch.zeros(N, J, 3, **f32); self.xmat = torch.zeros(N, J, 9, **f32)
        self.task_obs = torch.zeros(N, 24 * J, **f32); self._task_obs_tmp = torch.zeros(N, 24 * J, **f32)
        self.rew_buf = torch.zeros(N, **f32); self._rew_tmp = torch.zeros(N, **f32)
        self.reward_parts = torch.zeros(N, 4, **f32); self._parts_tmp = torch.zeros(N, 4, **f32)
        self.terminated = torch.zeros(N, dtype=torch.uint8, device=dev); self._term_tmp = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.self_obs_size, self.task_obs_size = b.obs_size, 24 * J
        self.obs_size = self.self_obs_size + self.task_obs_size
        self.obs_buf = torch.zeros(N, self.obs_size, **f32)
        self.action_size = b.nu
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed) + 1)

    # ---- one imitation launch on the current simulator state
    def _imitation(self, task_obs, rew, parts, term):
        b = self.base
        _check(lib().ss_kinematics(b.handle, _ptr(self.xpos), _ptr(self.xmat), b._stream()))
        torch.add(self.start_times, b.cur_t.to(torch.float32), alpha=self.dt, out=self.times)
        _check(lib().ss_imitation_step(C.byref(self.motion_lib.data), C.byref(self.cfg), _ptr(self.motion_ids), _ptr(self.times),
                                       _ptr(self.offset), self.num_envs, _ptr(self.xpos), _ptr(self.xmat), _ptr(b.body_vel),
                                       _ptr(task_obs), _ptr(rew), _ptr(parts), _ptr(term), b._stream()))

    def _assemble(self):
        self.obs_buf[:, :self.self_obs_size] = self.base.obs_buf
        self.obs_buf[:, self.self_obs_size:] = self.task_obs

    def reset(self, mask=None, motion_ids=None, start_times=None):
        """Reference-state init of all envs (mask None) or those with mask != 0."""
        b, ml = self.base, self.motion_lib
        m = torch.ones(self.num_envs, dtype=torch.bool, device=self.device) if mask is None else mask.to(self.device).bool()
        ids = ml.sample_motions(self.num_envs, generator=self.gen) if motion_ids is None else torch.as_tensor(motion_ids, device=self.device).to(torch.int32)
        if start_times is not None:
            t0 = torch.as_tensor(start_times, device=self.device).to(torch.float32)
        elif self.random_start:
            t0 = ml.sample_time(ids, truncate_time=self.dt, generator=self.gen).clamp_(min=0.0)
        else:
            t0 = torch.zeros(self.num_envs, device=self.device)
        self.motion_ids.copy_(torch.where(m, ids, self.motion_ids))
        self.start_times.copy_(torch.where(m, t0, self.start_times))
        self.motion_len.copy_(ml.get_motion_length(self.motion_ids))
        out, _ = ml._lookup(self.motion_ids, self.start_times, self.offset, False, ["qpos", "qvel"])
        b.qpos.copy_(torch.where(m[:, None], out["qpos"], b.qpos))
        b.qvel.copy_(torch.where(m[:, None], out["qvel"], b.qvel))
        b.reset(mask=m)                                        # StateInit External: mj_forward + self observation on that state
        self._imitation(self._task_obs_tmp, self._rew_tmp, self._parts_tmp, self._term_tmp)
        self.task_obs.copy_(torch.where(m[:, None], self._task_obs_tmp, self.task_obs))
        self._assemble()
        return self.obs_buf, {"critic_state": self.obs_buf}

    def step(self, actions):
        b = self.base
        b.step(actions)
        self._imitation(self.task_obs, self.rew_buf, self.reward_parts, self.terminated)
        self._assemble()
        terminated = self.terminated.bool()
        truncated = (self.times + self.dt) >= self.motion_len            # no next reference frame left to track
        info = {"reward_parts": self.reward_parts}
        if self.autoreset:
            info["final_observation"] = self.obs_buf.clone()
            self.reset(mask=terminated | truncated)
        info["critic_state"] = self.obs_buf
        return self.obs_buf, self.rew_buf, terminated, truncated, info

    def reference_actions(self):
        """PD targets that replay the clip: the next frame's joint angles in action units (dof_pos / pi, the reference's own
        replay recipe in examples/motion_lib_test.py:67) — a policy-free tracking baseline."""
        t = self.start_times + (self.base.cur_t.to(torch.float32) + 1.0) * self.dt
        out, _ = self.motion_lib._lookup(self.motion_ids, t, None, False, ["dof_pos"])
        return (out["dof_pos"] / torch.pi).clamp_(-1.0, 1.0)

    def close(self):
        self.base.close()
