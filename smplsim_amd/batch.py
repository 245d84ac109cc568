"""Batched env on one GPU shard: torch tensors in, torch tensors out, one HIP launch per step.

Mirrors the tensor surface of the reference's Isaac path (`GymVectEnv`, reference
smpl_sim/envs/nv/gymwrapper.py:7-65: step(actions[N,nu]) -> obs[N,D], rew[N], terminated[N],
truncated[N], info with autoreset) on top of the MuJoCo-semantics stepper, keeping MuJoCo's
wxyz / qpos / qvel packing.  PyTorch is only the owner of device memory and
streams here; all arithmetic happens in libsmplsim_hip.so.
"""
import ctypes as C
import os
import warnings

import numpy as np
import torch

from . import _cabi
from ._lib import lib
from .gains import build_pd_tables
from .mjcf import compile_mjcf
from .mjcf_writer import default_xml_str

DEFAULT_CONTACT_BODIES = ("R_Ankle", "L_Ankle", "R_Toe", "L_Toe")


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"libsmplsim_hip error {rc}: {lib().ss_last_error().decode()}")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _shard_device(index):
    """The device the shard's tensors live on: always a ROCm GPU (there is no CPU path in this package)."""
    if not torch.cuda.is_available():
        raise RuntimeError("SMPLSimVecEnv needs a ROCm GPU (MI355X); there is no CPU fallback")
    return torch.device("cuda", int(index))


def _launch_stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_WARNED = False


def _warn_floor_only():
    """Once per process: the default of the batched env is the floor-contact path of BASELINE.json's north_star; the reference's
    MuJoCo model also collides the humanoid's bodies with each other."""
    global _WARNED
    if not _WARNED:
        _WARNED = True
        warnings.warn("SMPLSimVecEnv(self_collision=False): floor contacts and joint limits only — bodies of the humanoid pass through "
                      "each other, unlike mj_step on the reference MJCF (contype/conaffinity, smpl_humanoid.xml:5,24). Pass "
                      "self_collision=True for the reference's contact set (about 3x the step time under violent actions).", stacklevel=3)


class ShardModel:
    """Compiled model + device tables (ss_model) for one device."""

    def __init__(self, xml=None, humanoid="smpl_humanoid", device=0, contact_bodies=DEFAULT_CONTACT_BODIES,
                 control_mode="uhc_pd", clip_actions=True, pdp_scale=1.0, pdd_scale=1.0, sim_timestep_inv=450, tables=None, xmls=None, mcs=None, compiler="python", lazy=False):
        """compiler: "python" = smplsim_amd.mjcf + gains build the ss_model_desc; "native" = the library compiles the MJCF text and
        the gain tables itself (ss_model_create_from_mjcf — the entry a non-Python host uses; single shape, reference gain table).
        tables: optional (kp, kd, torque_lim, act_scale, act_offset) per actuator for models whose bodies are not in
        the reference's gain table (humanoid_env.py:62-84).
        xmls: a list of MJCF strings = body shapes of the same humanoid (cfg.robot.has_shape_variation): one model whose
        geometry tables have an entry per shape; the envs pick theirs through SMPLSimVecEnv(shape_id=...).
        mcs: the same as already compiled ModelConsts (smplsim_amd.robot.compile_tables / models_from_mesh: body shapes built from
        SMPL meshes by the reference's geometry rules, thousands per vectorised pass)."""
        if mcs is not None:
            self.mcs, self.xmls = list(mcs), []
            self.xml = None
        else:
            self.xmls = list(xmls) if xmls is not None else [xml if xml is not None else default_xml_str(humanoid)]
            self.xml = self.xmls[0]
            self.mcs = [compile_mjcf(x) for x in self.xmls]
        self.mc, self.num_shapes = self.mcs[0], len(self.mcs)
        rng = {n: self.mc.jnt_range[6 + i] for i, n in enumerate(self.mc.joint_names)}
        self.tables = tables if tables is not None else build_pd_tables(
            self.mc.actuator_names, lambda n: rng[n], clip_actions=clip_actions, control_mode=control_mode,
            pdp_scale=pdp_scale, pdd_scale=pdd_scale)
        self.device = int(device)
        if compiler == "native":
            if self.num_shapes != 1 or self.xml is None or tables is not None:
                raise ValueError("compiler='native' takes one MJCF text and the reference's gain table")
        elif compiler != "python":
            raise ValueError(f"unknown compiler {compiler!r}")
        self._args = dict(compiler=compiler, contact_bodies=tuple(contact_bodies), control_mode=control_mode, clip_actions=clip_actions,
                          pdp_scale=pdp_scale, pdd_scale=pdd_scale, sim_timestep_inv=sim_timestep_inv)
        self._handle, self._pid = None, None
        if not lazy:
            self.handle

    @property
    def handle(self):
        """The ss_model handle, created on first use IN THE PROCESS THAT USES IT (lazy=True: a model built before a fork — the
        reference's sampler workers, agents/agent.py:121-145 — gets its device tables in every worker's own HIP context)."""
        if self._handle is None or self._pid != os.getpid():
            a = self._args
            h = C.c_void_p()
            if a["compiler"] == "native":
                names = (C.c_char_p * len(a["contact_bodies"]))(*[n.encode() for n in a["contact_bodies"]])
                opt = _cabi.MjcfOptions(_cabi.CONTROL_MODES[a["control_mode"]], int(bool(a["clip_actions"])), a["pdp_scale"], a["pdd_scale"],
                                        1.0 / a["sim_timestep_inv"], len(a["contact_bodies"]), names)
                txt = self.xml.encode()
                _check(lib().ss_model_create_from_mjcf(txt, len(txt), C.byref(opt), self.device, C.byref(h)))
            else:
                descs, self._keep = (_cabi.ModelDesc * self.num_shapes)(), []
                for i, mc in enumerate(self.mcs):
                    descs[i], keep = _cabi.make_model_desc(mc, *self.tables, legal_bodies=a["contact_bodies"], timestep=1.0 / a["sim_timestep_inv"])
                    self._keep.append(keep)
                _check(lib().ss_model_create_shapes(descs, self.num_shapes, self.device, C.byref(h)))
            self._handle, self._pid = h, os.getpid()
        return self._handle

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h and getattr(self, "_pid", None) == os.getpid():
            try:
                lib().ss_model_destroy(h)
            except Exception:
                pass


class SMPLSimVecEnv:
    """N environments on one GPU.  All tensors live on `cuda:<device>` and are updated in place."""

    def __init__(self, num_envs, model=None, device=0, task="HumanoidEnv", state_init="Default", self_obs_v=1,
                 control_mode="uhc_pd", episode_length=300, control_freq_inv=15, root_height_obs=True,
                 power_scale=1.0, tar_speed=(0.0, 5.0), speed_change=(100, 200), tar_height=(0.5, 1.2),
                 height_change=(100, 200), recovery_steps=60, tar_dist_max=1.0, reach_body="R_Hand", newton_iters=0, fused_autoreset=True,
                 autoreset=True, seed=0, lpt_order=True, shape_id=None, self_collision=False, solver_tolerance=0.0,
                 **model_kw):
        self.device = _shard_device(device if model is None else model.device)   # raises before any table is built without a GPU
        self.model = model if model is not None else ShardModel(device=device, control_mode=control_mode, **model_kw)
        mc = self.model.mc
        self.num_envs = int(num_envs)
        self.nq, self.nv, self.nu, self.nbody = mc.nq, mc.nv, mc.nu, mc.nbody
        self.task_id = _cabi.TASKS[task] if isinstance(task, str) else int(task)
        if self.task_id == _cabi.TASK_REACH and isinstance(reach_body, str) and reach_body not in mc.body_names:
            raise ValueError(f"reach_body {reach_body!r} is not a body of this model")
        self.state_init = _cabi.STATE_INITS[state_init] if isinstance(state_init, str) else int(state_init)
        if self.state_init == _cabi.INIT_EXTERNAL and autoreset:
            raise ValueError("StateInit External keeps the state the caller wrote: pass autoreset=False and reset finished envs "
                             "yourself (write qpos / qvel, then reset(mask)), as SMPLSimImitationVecEnv does")
        self.cfg = _cabi.make_env_cfg(
            task=self.task_id, state_init=self.state_init, self_obs_v=self_obs_v,
            control_mode=_cabi.CONTROL_MODES[control_mode], episode_length=episode_length,
            control_freq_inv=control_freq_inv, root_height_obs=root_height_obs, power_scale=power_scale,
            tar_speed=tar_speed, speed_change=speed_change, tar_height=tar_height, height_change=height_change,
            recovery_steps=recovery_steps, newton_iters=newton_iters, tar_dist_max=tar_dist_max,
            reach_body=self._body_index(mc, reach_body), self_collision=self_collision, solver_tolerance=solver_tolerance)
        self.self_collision = bool(self_collision)
        if not self.self_collision:
            _warn_floor_only()
        N, dev = self.num_envs, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.qpos = torch.zeros(N, self.nq, **f32); self.qvel = torch.zeros(N, self.nv, **f32)
        self.qpos_prev = torch.zeros(N, self.nq, **f32); self.qvel_prev = torch.zeros(N, self.nv, **f32)
        self.qacc_warm = torch.zeros(N, self.nv, **f32); self.body_vel = torch.zeros(N, self.nbody, 6, **f32)
        self.touch = torch.zeros(N, 2, **i32); self.cur_t = torch.zeros(N, **i32)
        self.task_state = torch.zeros(N, 4, **f32); self.nwarn = torch.zeros(N, **i32)
        self.solver_iters = torch.zeros(N, **i32)
        # SimplePID controller state (control_mode simple_pid; lives as long as the env, like the reference's object)
        self.pid_integral = torch.zeros(N, self.nu, **f32); self.pid_last_error = torch.zeros(N, self.nu, **f32)
        self.pid_started = torch.zeros(N, **i32)
        self.self_contacts = torch.zeros(N, **i32)           # body-body contacts of every env at its last forward (self_collision)
        self.qpos[:, 3] = 1; self.qpos_prev[:, 3] = 1
        # per-env body shape (models built from several MJCFs): read by every launch, may be rewritten between launches
        self.shape_id = None
        if self.model.num_shapes > 1:
            sid = torch.arange(N, device=dev) % self.model.num_shapes if shape_id is None else torch.as_tensor(shape_id, device=dev)
            if sid.shape != (N,) or int(sid.min()) < 0 or int(sid.max()) >= self.model.num_shapes:
                raise ValueError("shape_id must hold one shape index in [0, num_shapes) per env")
            self.shape_id = sid.to(torch.int32).contiguous()
        elif shape_id is not None:
            raise ValueError("shape_id given but the model has a single shape (build it with ShardModel(xmls=[...]))")
        st = _cabi.State(N, *[_ptr(t) for t in (self.qpos, self.qvel, self.qpos_prev, self.qvel_prev, self.qacc_warm,
                                               self.body_vel, self.touch, self.cur_t, self.task_state, self.nwarn,
                                               self.solver_iters, self.pid_integral, self.pid_last_error,
                                               self.pid_started, self.shape_id, self.self_contacts)])
        self.handle = C.c_void_p()
        _check(lib().ss_batch_create(self.model.handle, C.byref(self.cfg), C.byref(st), C.byref(self.handle)))
        self.obs_size = lib().ss_obs_size(self.model.handle, C.byref(self.cfg))
        self.obs_buf = torch.zeros(N, self.obs_size, **f32)
        self.rew_buf = torch.zeros(N, **f32)
        self.terminated = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.reset_buf = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.obs_final = torch.zeros(N, self.obs_size, **f32)
        # Default and Fall resets of finished envs run inside the step launch (External has no reset state of its own)
        self._fused_autoreset = self.state_init in (_cabi.INIT_DEFAULT, _cabi.INIT_FALL) and fused_autoreset
        self._fall_buf = None
        if self._fused_autoreset and self.state_init == _cabi.INIT_FALL:
            self._fall_buf = torch.zeros(N, 3, self.nu, **f32)
            _check(lib().ss_set_fall_actions(self.handle, _ptr(self._fall_buf)))
        self.autoreset = autoreset
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))
        self.action_size = self.nu
        self.actuator_names = list(mc.actuator_names)
        self._keep = self._keep2 = None
        # longest-processing-time-first hand-out: env-step cost ~ Newton iterations, which are heavy-tailed and
        # autocorrelated from one control step to the next (scheduling only; results do not depend on it)
        self.lpt_order = bool(lpt_order) and N > 64
        self.order = torch.arange(N, dtype=torch.int32, device=dev)

    @staticmethod
    def _body_index(mc, body):
        if not isinstance(body, str):
            return int(body)
        if body in mc.body_names:
            return mc.body_names.index(body)
        return 0   # e.g. SMPL-X has no R_Hand body; only the reach task reads this field (checked there)

    # ---- random inputs the reference draws from np.random inside the env (targets, Fall actions)
    def _task_rand(self):
        if self.task_id == _cabi.TASK_BASE:
            return None
        return torch.rand(self.num_envs, 4, generator=self.gen, device=self.device)

    def _fall_actions(self):
        if self.state_init != _cabi.INIT_FALL:
            return None
        return torch.rand(self.num_envs, 3, self.nu, generator=self.gen, device=self.device)

    def _stream(self):
        return _launch_stream(self.device)

    def reset(self, mask=None, fall_actions=None, task_rand=None):
        """Reset all envs (mask None) or those with mask != 0.  Returns (obs, info)."""
        fa = fall_actions if fall_actions is not None else self._fall_actions()
        tr = task_rand if task_rand is not None else self._task_rand()
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        self._keep = (fa, tr, m)
        _check(lib().ss_reset(self.handle, _ptr(m), _ptr(fa), _ptr(tr), _ptr(self.obs_buf), self._stream()))
        return self.obs_buf, {"critic_state": self.obs_buf}

    def step(self, actions, task_rand=None, _events=None, task_rand2=None, _fall_drawn=False):
        """One control step of every env (+ masked autoreset).  `_events` = (start, end) torch.cuda.Event pair recorded
        around the step launch only (bench.py's per-launch kernel timing).  task_rand / task_rand2 (the step's and the in-launch
        reset's task draws) and _fall_drawn (self._fall_buf already holds this step's Fall draws) let a caller that splits one job
        into sub-batches hand every env the draws it would get in the unsplit batch (pipeline.PipelinedVecEnv)."""
        actions = actions.to(torch.float32).contiguous()
        assert actions.shape == (self.num_envs, self.nu) and actions.device == self.device
        tr = task_rand if task_rand is not None else self._task_rand()
        self._keep = (actions, tr)
        if self.lpt_order:                                     # longest-processing-time-first hand-out, computed on the device
            _check(lib().ss_schedule_longest_first(self.handle, self._stream()))
        if _events:
            _events[0].record()
        info = {}
        if self.autoreset and self._fused_autoreset:
            # one launch: step + Default reset of the envs whose episode ended (GymVectEnv semantics, reference
            # nv/gymwrapper.py:53-60): obs_final = the step's observation, obs_buf = what the policy acts on next
            tr2 = task_rand2 if task_rand2 is not None else self._task_rand()
            self._keep2 = (tr2,)
            if self._fall_buf is not None and not _fall_drawn:   # fresh draws for the envs that will be reset in this launch
                self._fall_buf.uniform_(0.0, 1.0, generator=self.gen)
            _check(lib().ss_step_autoreset(self.handle, _ptr(actions), _ptr(tr), _ptr(tr2), _ptr(self.obs_final), _ptr(self.obs_buf),
                                           _ptr(self.rew_buf), _ptr(self.terminated), _ptr(self.truncated), self._stream()))
            if _events:
                _events[1].record()
            if self.lpt_order:
                _check(lib().ss_set_order(self.handle, None))
            info["final_observation"] = self.obs_final
            info["critic_state"] = self.obs_buf
            return self.obs_buf, self.rew_buf, self.terminated.bool(), self.truncated.bool(), info
        _check(lib().ss_step(self.handle, _ptr(actions), _ptr(tr), _ptr(self.obs_buf), _ptr(self.rew_buf),
                             _ptr(self.terminated), _ptr(self.truncated), self._stream()))
        if _events:
            _events[1].record()
        if self.lpt_order:
            _check(lib().ss_set_order(self.handle, None))      # resets / diagnostics use the natural order
        if self.autoreset:
            # autoreset of finished envs: device-side mask, no host sync (the Fall reset is 45 mj_steps of work per env,
            # so it gets its own, load-balanced launch); the pre-reset observation is kept for the learner
            torch.bitwise_or(self.terminated, self.truncated, out=self.reset_buf)
            info["final_observation"] = self.obs_buf.clone()
            fa, tr2 = self._fall_actions(), self._task_rand()
            self._keep2 = (fa, tr2)
            _check(lib().ss_reset(self.handle, _ptr(self.reset_buf), _ptr(fa), _ptr(tr2), _ptr(self.obs_buf),
                                  self._stream()))
        info["critic_state"] = self.obs_buf
        return self.obs_buf, self.rew_buf, self.terminated.bool(), self.truncated.bool(), info

    def substep(self, actions, n):
        actions = actions.to(torch.float32).contiguous()
        self._keep = (actions,)
        _check(lib().ss_substep(self.handle, _ptr(actions), int(n), self._stream()))

    def kinematics(self):
        xpos = torch.zeros(self.num_envs, self.nbody, 3, device=self.device)
        xmat = torch.zeros(self.num_envs, self.nbody, 9, device=self.device)
        _check(lib().ss_kinematics(self.handle, _ptr(xpos), _ptr(xmat), self._stream()))
        return xpos, xmat

    def enable_power_usage(self):
        """HumanoidEnv.curr_power_usage (reference humanoid_env.py:443-451) for the batch: after every step `self.power_usage`
        [N, control_freq_inv, nv - 6] holds |qfrc_actuator * qvel| of every hinge dof per mj_step (ss_set_power_output; selects the
        kernel instantiation with the optional outputs)."""
        if getattr(self, "power_usage", None) is None:
            self.power_usage = torch.zeros(self.num_envs, int(self.cfg.control_freq_inv), self.nv - 6, device=self.device)
            _check(lib().ss_set_power_output(self.handle, _ptr(self.power_usage)))
        return self.power_usage

    def debug_forward(self, torques=None):
        """(M [N,nv,nv], qfrc_bias, qacc) of one mj_forward at the current state (parity triage)."""
        M = torch.zeros(self.num_envs, self.nv, self.nv, device=self.device)
        bias = torch.zeros(self.num_envs, self.nv, device=self.device)
        qacc = torch.zeros(self.num_envs, self.nv, device=self.device)
        tq = None if torques is None else torques.to(torch.float32).contiguous()
        _check(lib().ss_debug_forward(self.handle, _ptr(tq), _ptr(M), _ptr(bias), _ptr(qacc), self._stream()))
        return M, bias, qacc

    def debug_self_contacts(self):
        """Body-body contact records of every env's last forward pass (mjData.contact of the body pairs; parity triage):
        after every launch `self.self_records` [N, SS_MAX_SELF_CONTACTS, 24] holds body1 body2 | position 3 | normal 3 | first
        tangent 3 | 1/R | aref 4 | ... of the first `self.self_contacts[n]` contacts (ss_debug_self_contacts)."""
        if getattr(self, "self_records", None) is None:
            self.self_records = torch.zeros(self.num_envs, _cabi.SS_MAX_SELF_CONTACTS, 24, device=self.device)
            _check(lib().ss_debug_self_contacts(self.handle, _ptr(self.self_records)))
        return self.self_records

    def set_state(self, qpos, qvel, qpos_prev=None, qvel_prev=None, warm=None):
        """Teacher forcing / checkpoint restore.  *_prev default to the state itself (== after mj_forward)."""
        def as_t(x):
            if torch.is_tensor(x):
                return x.to(self.device, torch.float32)
            return torch.as_tensor(np.asarray(x), dtype=torch.float32, device=self.device)
        self.qpos.copy_(as_t(qpos)); self.qvel.copy_(as_t(qvel))
        self.qpos_prev.copy_(as_t(qpos if qpos_prev is None else qpos_prev))
        self.qvel_prev.copy_(as_t(qvel if qvel_prev is None else qvel_prev))
        if warm is not None:
            self.qacc_warm.copy_(as_t(warm))

    def launch_info(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().ss_launch_info(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return {"envs_per_workgroup": a.value, "lds_bytes": b.value, "vgprs": c.value}

    def close(self):
        h = getattr(self, "handle", None)
        if h:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            lib().ss_batch_destroy(h)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
