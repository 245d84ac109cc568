"""Motion-imitation rollouts on one GPU shard (BASELINE config 4).

The reference ships the pieces of this task (motion library, batched FK, the self observation that is "an exact
replica of PHC's", humanoid_env.py:636) but not the task itself; this env wires them the way PHC's HumanoidIm does:

  reset   reference-state init: sample a clip and a start time per env, write the clip's qpos/qvel at that time into
          the simulator state (ss_motion_state_at -> ss_reset with StateInit "External")
  step    ONE launch, ss_imitation_step_fused: ss_step (15 x (Stable-PD + mj_step), self observation, body frames) and, in the
          wavefront that stepped the env, the imitation task (clip lookup at t and t + dt, task observation written behind the
          self observation, tracking reward, early termination, truncation at the end of the clip) and the re-initialisation of
          finished envs (resample -> clip state into qpos / qvel -> reset forward -> observations).  (+ one launch for the
          longest-first hand-out order, + one torch.rand per 64 steps for the resampling draws.)
          fused=False keeps the sequence of separate launches the fused one is tested against: ss_step -> ss_imitation_step ->
          ss_motion_resample -> ss_motion_state_at (masked) -> ss_reset (masked) -> ss_imitation_step (masked, observation only).

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
                 random_start=True, autoreset=True, seed=0, fused=True, **env_kw):
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
        self.offset = torch.zeros(N, 3, **f32)
        self.xpos = torch.zeros(N, J, 3, **f32); self.xmat = torch.zeros(N, J, 9, **f32)
        self.self_obs_size, self.task_obs_size = b.obs_size, 24 * J
        self.obs_size = self.self_obs_size + self.task_obs_size
        self.obs_buf = torch.zeros(N, self.obs_size, **f32)          # [self obs | task obs]: the kernels write the task part in place
        self.task_obs = self.obs_buf[:, self.self_obs_size:]
        self.rew_buf = torch.zeros(N, **f32)
        self.reward_parts = torch.zeros(N, 4, **f32)
        self.terminated = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.reset_buf = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.action_size = self.nu = b.nu
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed) + 1)
        # every step / reset launch also writes the body frames of its last forward: no separate ss_kinematics launch
        _check(lib().ss_set_body_outputs(b.handle, _ptr(self.xpos), _ptr(self.xmat)))
        # resampling draws of the re-initialisations: a block of RAND_BLOCK steps per torch.rand launch
        self._rand_block, self._rand_i = None, self.RAND_BLOCK
        # (with body-body contacts the one-launch step is compiled for SMPL-sized single-shape models only)
        self.fused = bool(fused) and not (b.self_collision and (b.shape_id is not None or b.nv > 128))
        self.obs_final = torch.zeros(N, self.obs_size, **f32)
        self._bound = None
        self._bind()

    def _bind(self):
        """Hand the buffers of the fused step to the batch (again after the motion library loaded other clips: its arrays moved)."""
        ml = self.motion_lib
        if not self.fused or self._bound == ml.load_count:
            return
        io = _cabi.ImitationIO(C.pointer(ml.data), self.cfg, *[_ptr(t) for t in (self.motion_ids, self.start_times, self.offset,
                               ml.sampling_cdf)], self.dt, int(bool(self.random_start)), _ptr(self.obs_final), _ptr(self.obs_buf),
                               self.obs_size, _ptr(self.rew_buf), _ptr(self.reward_parts), _ptr(self.terminated), _ptr(self.truncated))
        _check(lib().ss_imitation_bind(self.base.handle, C.byref(io)))
        self._bound = ml.load_count

    RAND_BLOCK = 64

    def _draws(self):
        """[N,2] uniform draws for this step's re-initialisations (a view into a block drawn once per RAND_BLOCK steps)."""
        if self._rand_i >= self.RAND_BLOCK:
            self._rand_block = torch.rand(self.RAND_BLOCK, self.num_envs, 2, device=self.device, generator=self.gen)
            self._rand_i = 0
        self._rand_i += 1
        return self._rand_block[self._rand_i - 1]

    @property
    def times(self):
        """Clip time of every env (s)."""
        return self.start_times + self.base.cur_t.to(torch.float32) * self.dt

    @property
    def motion_len(self):
        return self.motion_lib.get_motion_length(self.motion_ids)

    # ---- one imitation launch on the current simulator state (envs with mask byte 0 are skipped)
    def _imitation(self, mask, rew, parts, term, trunc):
        b = self.base
        task_ptr = C.c_void_p(self.obs_buf.data_ptr() + 4 * self.self_obs_size)
        _check(lib().ss_imitation_step(C.byref(self.motion_lib.data), C.byref(self.cfg), _ptr(self.motion_ids), _ptr(self.start_times),
                                       _ptr(b.cur_t), _ptr(self.offset), _ptr(mask), self.num_envs, _ptr(self.xpos), _ptr(self.xmat),
                                       _ptr(b.body_vel), task_ptr, self.obs_size, _ptr(rew), _ptr(parts), _ptr(term), _ptr(trunc),
                                       b._stream()))

    def reset(self, mask=None, motion_ids=None, start_times=None, rand=None):
        """Reference-state init of all envs (mask None) or those with mask != 0: new clip + start time (unless given), the
        clip's qpos / qvel at that time written into the simulator state, mj_forward + observations — 4 launches, in place."""
        b, ml = self.base, self.motion_lib
        m = None if mask is None else mask.to(self.device).to(torch.uint8).contiguous()
        if motion_ids is None:
            ml.resample(m, self.motion_ids, self.start_times, truncate_time=self.dt, rand=self._draws() if rand is None else rand)
            if start_times is None and not self.random_start:
                self.start_times.zero_() if m is None else self.start_times.masked_fill_(m.bool(), 0.0)
        else:
            ids = torch.as_tensor(motion_ids, device=self.device).to(torch.int32)
            self.motion_ids.copy_(ids if m is None else torch.where(m.bool(), ids, self.motion_ids))
        if start_times is not None:
            t0 = torch.as_tensor(start_times, device=self.device).to(torch.float32)
            self.start_times.copy_(t0 if m is None else torch.where(m.bool(), t0, self.start_times))
        ml.write_state(self.motion_ids, self.start_times, self.offset, m, b.qpos, b.qvel)
        b.reset(mask=m)                                        # StateInit External: mj_forward + self observation on that state
        self._imitation(m, None, None, None, None)             # task observation of the re-initialised envs only
        self.obs_buf[:, :self.self_obs_size] = b.obs_buf
        return self.obs_buf, {"critic_state": self.obs_buf}

    def step(self, actions, _events=None):
        """_events: (start, end) torch.cuda.Event pair recorded around the step launch itself (bench.py's kernel time; fused only)."""
        b = self.base
        if self.fused:
            actions = actions.to(torch.float32).contiguous()
            assert actions.shape == (self.num_envs, self.nu) and actions.device == self.device
            rand = self._draws() if self.autoreset else None
            self._keep = (actions, rand)
            self._bind()
            if b.lpt_order:
                _check(lib().ss_schedule_longest_first(b.handle, b._stream()))
            if _events:
                _events[0].record()
            _check(lib().ss_imitation_step_fused(b.handle, _ptr(actions), _ptr(rand), b._stream()))
            if _events:
                _events[1].record()
            if b.lpt_order:
                _check(lib().ss_set_order(b.handle, None))
            # the flag bytes are 0 / 1: viewed, not converted (no launch)
            return self.obs_buf, self.rew_buf, self.terminated.view(torch.bool), self.truncated.view(torch.bool), \
                {"reward_parts": self.reward_parts, "final_observation": self.obs_final, "critic_state": self.obs_buf}
        b.step(actions)
        self._imitation(None, self.rew_buf, self.reward_parts, self.terminated, self.truncated)
        self.obs_buf[:, :self.self_obs_size] = b.obs_buf
        terminated, truncated = self.terminated.bool(), self.truncated.bool()
        info = {"reward_parts": self.reward_parts}
        if self.autoreset:
            self.obs_final.copy_(self.obs_buf)
            info["final_observation"] = self.obs_final
            torch.bitwise_or(self.terminated, self.truncated, out=self.reset_buf)
            self.reset(mask=self.reset_buf)
        info["critic_state"] = self.obs_buf
        return self.obs_buf, self.rew_buf, terminated, truncated, info

    def reference_actions(self):
        """PD targets that replay the clip: the next frame's joint angles in action units (dof_pos / pi, the reference's own
        replay recipe in examples/motion_lib_test.py:67) — a policy-free tracking baseline."""
        t = self.start_times + (self.base.cur_t.to(torch.float32) + 1.0) * self.dt
        out, _ = self.motion_lib._lookup(self.motion_ids, t, None, False, ["dof_pos"])
        return (out["dof_pos"] / torch.pi).clamp_(-1.0, 1.0)

    @torch.no_grad()
    def evaluate(self, policy=None, max_steps=None):
        """Score a policy the way the reference's evaluation scripts do (compute_metrics_lite, smpl_eval.py:58-94): env n
        replays clip n % M from its first frame until the clip ends or the early-termination rule fires; body positions of
        the simulated humanoid and of the clip are recorded on the device.  policy(obs) -> actions; None = PD clip replay.
        Returns success rate (clips tracked to their end) and the mean metrics over the frames before termination."""
        from . import metrics
        ml, N = self.motion_lib, self.num_envs
        auto, self.autoreset = self.autoreset, False
        ids = torch.arange(N, device=self.device, dtype=torch.int32) % ml.num_current_motions()
        obs, _ = self.reset(motion_ids=ids, start_times=torch.zeros(N, device=self.device))
        steps = int(max_steps or torch.ceil(self.motion_len.max() / self.dt).item())
        alive = torch.ones(N, dtype=torch.bool, device=self.device)
        finished = torch.zeros(N, dtype=torch.bool, device=self.device)
        pred, ref, valid = [], [], []
        for _ in range(steps):
            act = self.reference_actions() if policy is None else policy(obs)
            obs, _, term, trunc, _ = self.step(act)
            alive &= ~term
            valid.append(alive & ~finished)
            pred.append(self.xpos.clone())
            ref.append(ml._lookup(self.motion_ids, self.times, self.offset, False, ["rg_pos"])[0]["rg_pos"])
            finished |= trunc & alive
        self.autoreset = auto
        pred, ref, valid = torch.stack(pred), torch.stack(ref), torch.stack(valid)          # [T, N, ...]
        pp, gg = [], []
        for n in range(N):
            k = int(valid[:, n].sum().item())
            if k >= 3:                                           # the acceleration error needs three frames
                pp.append(pred[:k, n]); gg.append(ref[:k, n])
        out = {"success_rate": float(finished.float().mean().item()), "num_clips": N, "frames_scored": int(valid.sum().item())}
        if pp:
            m = metrics.compute_metrics_lite(pp, gg)
            out.update({k: float(v.mean().item()) for k, v in m.items()})
        return out

    def close(self):
        self.base.close()
