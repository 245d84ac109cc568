"""Body-shape variation across the envs of one GPU shard (the part that touches the hot path).

The reference gives every env process its own MJCF when `cfg.robot.has_shape_variation` is set (SMPL_Robot writes one per
betas draw; smpl_sim/envs/humanoid_env.py:205,218-247).  Two ways here:

  * per-env shapes in ONE launch (the default): `ShardModel(xmls=[...])` -> `ss_model_create_shapes`, the geometry tables
    (body offsets / inertias, contact candidates, inverse weights) have one entry per shape and every env reads those of
    `shape_id[env]`; `SMPLSimVecEnv(N, model, shape_id=...)`.  Any number of shapes, up to one per env; the shapes must
    differ in geometry only.  `ShapeVariedVecEnv(xmls, envs_per_shape)` is a convenience constructor for it.
  * shape groups on streams (`single_launch=False`): one compiled model + one `ss_batch` per shape, all groups stepped
    concurrently on their own HIP streams and joined at the end of the step — for variants that differ in more than
    geometry (gains, limits).  State, observation, reward and flag tensors are single [N, ...] tensors; every group's buffers
    are row ranges of them, so nothing is concatenated or copied per step.

Concurrency needs one hardware queue per group: ROCm maps streams onto GPU_MAX_HW_QUEUES (default 4) queues, so export
GPU_MAX_HW_QUEUES >= number of shapes + 1 before the process touches the GPU (measured, 4096 envs: 2 groups 2.15 ms per step
against 2.13 ms for one batch; 4 groups 2.64 ms, 8 groups 2.96 ms with 16 queues; 4.2 / 6.0 ms with the default 4).  Hundreds
of distinct shapes need a launch that indexes the model tables per workgroup.

Generating the per-shape MJCFs from SMPL betas needs the SMPL model files (not redistributable): callers pass the XML
strings (e.g. written by the reference's SMPL_Robot); `mjcf_writer.scaled_xml_str` makes synthetic variants for tests.
"""
import os
import warnings

import torch

from .batch import ShardModel, SMPLSimVecEnv


class ShapeVariedVecEnv:
    def __init__(self, xmls, envs_per_shape, device=0, seed=0, shape_params=None, single_launch=True, **env_kw):
        """xmls: one MJCF string per body shape (same tree / actuators, different geometry); envs_per_shape: int or list."""
        K = len(xmls)
        counts = [int(envs_per_shape)] * K if isinstance(envs_per_shape, int) else [int(c) for c in envs_per_shape]
        assert len(counts) == K and all(c > 0 for c in counts)
        self.single = None
        if single_launch:
            sid = torch.repeat_interleave(torch.arange(K), torch.tensor(counts))
            e = self.single = SMPLSimVecEnv(sum(counts), model=ShardModel(xmls=xmls, device=device), device=device, seed=seed, shape_id=sid if K > 1 else None, **env_kw)
            self.envs, self.models, self.num_envs, self.num_shapes = [e], [e.model], e.num_envs, K
            self.device, self.nq, self.nv, self.nu, self.nbody, self.obs_size, self.action_size = e.device, e.nq, e.nv, e.nu, e.nbody, e.obs_size, e.nu
            self.shape_id = e.shape_id
            self.obs_buf, self.rew_buf = e.obs_buf, e.rew_buf
            self.reset, self.step, self.state = e.reset, e.step, (lambda: (e.qpos, e.qvel))
            return
        self.models = [ShardModel(xml=x, device=device) for x in xmls]
        self.envs = [SMPLSimVecEnv(c, model=m, device=device, seed=seed + 1000 * g, **env_kw) for g, (c, m) in enumerate(zip(counts, self.models))]
        e0 = self.envs[0]
        if any((e.nq, e.nv, e.nu, e.obs_size) != (e0.nq, e0.nv, e0.nu, e0.obs_size) for e in self.envs):
            raise ValueError("all shapes must share the kinematic tree, actuators and observation layout")
        self.device, self.nq, self.nv, self.nu, self.nbody, self.obs_size = e0.device, e0.nq, e0.nv, e0.nu, e0.nbody, e0.obs_size
        self.num_envs, self.num_shapes = sum(counts), K
        self.starts = [0]
        for c in counts:
            self.starts.append(self.starts[-1] + c)
        self.shape_id = torch.repeat_interleave(torch.arange(K, device=self.device), torch.tensor(counts, device=self.device))
        self.shape_params = None if shape_params is None else torch.as_tensor(shape_params, dtype=torch.float32, device=self.device)[self.shape_id]
        N = self.num_envs
        f32 = dict(dtype=torch.float32, device=self.device)
        u8 = dict(dtype=torch.uint8, device=self.device)
        self.obs_buf = torch.zeros(N, self.obs_size, **f32); self.obs_final = torch.zeros(N, self.obs_size, **f32)
        self.rew_buf = torch.zeros(N, **f32)
        self.terminated = torch.zeros(N, **u8); self.truncated = torch.zeros(N, **u8); self.reset_buf = torch.zeros(N, **u8)
        self.qpos = torch.zeros(N, self.nq, **f32); self.qvel = torch.zeros(N, self.nv, **f32)
        for env, a, b in zip(self.envs, self.starts[:-1], self.starts[1:]):
            # outputs are passed to the C ABI by pointer on every call: re-point the group's buffers at its rows of the big ones
            env.obs_buf, env.obs_final, env.rew_buf = self.obs_buf[a:b], self.obs_final[a:b], self.rew_buf[a:b]
            env.terminated, env.truncated, env.reset_buf = self.terminated[a:b], self.truncated[a:b], self.reset_buf[a:b]
        if K + 1 > int(os.environ.get("GPU_MAX_HW_QUEUES", "4")):
            warnings.warn(f"{K} shape groups but GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '4 (default)')}: their launches will "
                          "share hardware queues and run one after the other; export GPU_MAX_HW_QUEUES >= shapes + 1 before starting")
        if K > 1:
            # concurrent launches share the GPU as full workgroups on 1/K of the CUs each (a spread small batch would claim
            # every CU's LDS and the launches would run one after the other)
            import ctypes as C
            from .batch import _check, lib
            cus = torch.cuda.get_device_properties(self.device).multi_processor_count
            for e in self.envs:
                _check(lib().ss_set_launch_geometry(e.handle, C.c_int32(e.launch_info()["envs_per_workgroup"]), C.c_int32(max(1, cus // K))))
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(K)]
        self.action_size = self.nu

    def _fan_out(self, fn):
        """Run fn(g, env) for every group on its stream, ordered after the caller's stream; the caller's stream then waits for all."""
        main = torch.cuda.current_stream(self.device)
        ready = main.record_event()
        outs = []
        for g, (env, s) in enumerate(zip(self.envs, self.streams)):
            s.wait_event(ready)
            with torch.cuda.stream(s):
                outs.append(fn(g, env))
            main.wait_event(s.record_event())
        return outs

    def reset(self):
        self._fan_out(lambda g, env: env.reset())
        return self.obs_buf, {"critic_state": self.obs_buf}

    def step(self, actions):
        actions = actions.to(torch.float32).contiguous()
        assert actions.shape == (self.num_envs, self.nu)
        infos = self._fan_out(lambda g, env: env.step(actions[self.starts[g]:self.starts[g + 1]])[4])
        info = {"critic_state": self.obs_buf}
        if all("final_observation" in i for i in infos):
            # fused autoreset: every group wrote its rows of obs_final in place; the two-launch path returns clones
            fin = [i["final_observation"] for i in infos]
            info["final_observation"] = self.obs_final if all(f.data_ptr() == self.obs_final[a:a + 1].data_ptr() for f, a in zip(fin, self.starts[:-1])) \
                else torch.cat(fin)
        return self.obs_buf, self.rew_buf, self.terminated.bool(), self.truncated.bool(), info

    def state(self):
        """(qpos, qvel) of all envs as [N, ...] tensors (gathered copies; each group owns its own state buffers)."""
        for env, a, b in zip(self.envs, self.starts[:-1], self.starts[1:]):
            self.qpos[a:b].copy_(env.qpos); self.qvel[a:b].copy_(env.qvel)
        return self.qpos, self.qvel

    def close(self):
        for e in self.envs:
            e.close()
