#!/usr/bin/env python3
"""bench.py — env-steps/sec of the fused SMPL env step on N MI355X (BASELINE.json metric).

Workload (BASELINE config 2): 4096 SMPL humanoids per GPU, flat ground, Stable-PD with fresh
uniform(-1,1) actions every control step, obs v1 + reward + reset flags computed in the step
launch, device-side autoreset.  One "step" = one control step of every env on every rank
(15 mj_steps each).  Weak scaling: independent shards, no collective in the data path.

    python bench.py --gpus 1 --steps 1000 --warmup 20
    python bench.py --gpus 8                       # starts its own 8 ranks (one per GPU), like the reference's one-call fan-out
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W     # the same job when the caller brings the ranks
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
WORKLOADS = {
    "smpl": "BASELINE config 2: {N} SMPL humanoids per GPU (24 bodies, nv=75), flat ground, Stable-PD, fresh uniform(-1,1) "
            "actions per control step, 15 mj_steps @450 Hz per step, obs v1 (289 f32) + reward + reset flags fused, "
            "device-side autoreset",
    "getup": "BASELINE config 3 shard: {N} SMPL humanoids, env=getup (obs 290, height reward, contact termination, 60-step "
             "recovery), StateInit.Fall (45 warm-up mj_steps per reset), uniform(-1,1) actions; the Fall reset of finished envs runs inside the step launch (ss_set_fall_actions + ss_step_autoreset)",
    "imitation": "BASELINE config 5 shard: {N} SMPL humanoids tracking motion clips (synthetic smooth clips in the AMASS pickle "
                 "format; no dataset in the image), reference-state init, PD replay of the clip as the policy, per step ONE launch "
                 "(ss_imitation_step_fused): ss_step (obs v2, body frames) + clip lookup at t and t+dt, 576-float task obs, PHC tracking "
                 "reward, early termination + in-place re-initialisation of finished envs (resample, clip state, reset forward, "
                 "observations); --unfused = the six-launch sequence it replaces",
    "smplx": "BASELINE config 4: {N} SMPL-X/H-layout humanoids (52 bodies, nv=159, nu=153), base env, obs v1 (625 f32), uniform(-1,1) actions",
}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_MEASURED_GBS = 6290.0       # the copy bandwidth measured on this part (same guide); roofline.frac_measured_peak
FP32_VECTOR_PEAK = 157.3e12     # MI355X_MICROARCH.md: FP32 vector peak (256 CUs x 4 SIMD-32 x 2 flop x 2.4 GHz, packed)
SCALING_NOTE = ("default = WEAK scaling: 4096 envs on every GPU, independent shards, value = sum over the ranks.  Read as STRONG scaling "
                "(--envs-total 4096: the same 4096 envs split over the ranks, 512 per GPU = 2 per CU at 8 GPUs) the metric is bounded by the "
                "step time of ONE env: a launch ends when its heaviest env ends, one env alone on a CU needs 1.19-1.24 ms for the ~100 Newton "
                "iterations of the heaviest env of a step (profiles/r04_lone_wave.txt) against 1.47 ms for the whole 4096-env launch on one "
                "GPU, so 8 GPUs can step the SAME 4096 envs at most ~1.25x faster than one (<= ~3.4 M env-steps/s), whereas the weak-scaling "
                "figure grows with the GPU count (no collective, no shared resource but the host).  Neither curve has been measured on hardware "
                "by the builder (one-GPU boxes only); the driver's --gpus N runs are weak scaling unless it passes --envs-total")
PARITY_PIN = ("none (mujoco absent): the oracle's mj_step restates MuJoCo's documented pipeline and is pinned to nothing; "
              "controllers / observations / rewards / gains are pinned to the reference's own code (tests/golden)")


class Gpu:
    """torch.cuda plumbing of this script in one place.  (The CPU test of the multi-process path, tests/test_multi_gpu_cpu.py,
    swaps it for host stand-ins and runs the same main() under gloo on the kernel emulator.)"""
    backend = "nccl"

    @staticmethod
    def device(index):
        import torch
        torch.cuda.set_device(index)
        return torch.device("cuda", index)

    @staticmethod
    def device_count():
        import torch
        return torch.cuda.device_count()

    @staticmethod
    def sync():
        import torch
        torch.cuda.synchronize()

    @staticmethod
    def event():
        import torch
        return torch.cuda.Event(enable_timing=True)

    @staticmethod
    def stream_ptr(dev):
        import ctypes as C
        import torch
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def flops_per_env_step(nb, nv, ncand, newton_iters_per_step, nsub=15):
    """Algorithmic float32 operations of one env-step as the kernel's formulation does them (FMA = 2), DESIGN.md §4 "Flop
    model": per mj_step one forward pass (kinematics, velocities, inertias, bias force), the contact candidates, one
    articulated-body solve for Stable PD and one per Newton iteration with its right-hand side and line search."""
    nn = nb + 1
    fk = 700 * nb + 12 * nv
    aba = 840 * nn + 500
    cons = 60 * ncand
    newton_other = 70 * nb + 10 * nv + 1600
    per_sub = fk + cons + aba + 8 * nv + 6 * nv
    return nsub * per_sub + newton_iters_per_step * (aba + newton_other) + 60 * nb


def pmc_summary(workload, n_envs, kernel_ms=None):
    """Limiter figures of the dominant kernel from the committed PMC passes of this same command (tools/gpu_prof.sh ->
    profiles/pmc_summary_<workload>.json); PMC counters cannot be read from inside the process."""
    path = os.path.join(ROOT, "profiles", f"pmc_summary_{workload}.json")
    if not os.path.exists(path):
        return {}
    j = json.load(open(path))
    if j.get("envs_per_gpu") != n_envs:
        return {}
    from smplsim_amd._lib import source_hash
    src = "profiles/" + os.path.basename(path) + " (" + j.get("profile", "?") + ")"
    if j.get("src_hash") != source_hash():
        # counters of another tree (kernel sources / headers / compiler flags changed since the passes were taken): not merged
        return {"pmc_stale": True, "pmc_source": src, "pmc_src_hash": j.get("src_hash"), "src_hash": source_hash()}
    keys = ("traffic", "valu_issue_frac", "lds_wait_frac", "lds_bank_conflict_frac", "wave_active_frac", "wave_slot_occupancy",
            "scratch_bytes_per_lane", "vgprs", "waves_per_cu")
    # the counters must describe the regime that was timed: a block whose own step launches (rocprofv3 kernel trace of the same passes)
    # lasted more than 10 % longer or shorter than this run's is of another regime (round 5: the getup block came from the first launches
    # after a fresh reset, without Fall resets inside them) and is not merged
    prof_ms = j.get("step_launch_avg_us") and j["step_launch_avg_us"] * 1e-3
    if kernel_ms and prof_ms and abs(prof_ms - kernel_ms) > 0.10 * kernel_ms:
        return {"pmc_stale": False, "pmc_regime_mismatch": True, "pmc_source": src, "pmc_step_launch_ms": prof_ms, "src_hash": j["src_hash"]}
    out = {k: j[k] for k in keys if k in j}
    out.update(pmc_stale=False, pmc_regime_mismatch=False, pmc_step_launch_ms=prof_ms, pmc_source=src, src_hash=j["src_hash"])
    return out


def algorithmic_bytes(nq, nv, nu, nobs):
    # read qpos,qvel,action + write qpos,qvel,obs,reward + 2 flag bytes
    return 4 * (2 * nq + 2 * nv + nu + nobs + 1) + 2


def usable_cores():
    """Cores this process may really use: sched affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def cpu_baseline(seconds=12.0):
    """The CPU oracle (float64 C restatement; 'port', NOT MuJoCo — MuJoCo is not installable here) on the
    host cores, same workload shape, bounded sample.  Built here, on the box it is timed on, with -O3 -march=native
    (BASELINE.md §3); the tests use the portable -O2 build."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    flags = O.use_native_build()
    from helpers import oracle_model
    cores = usable_cores()
    om = oracle_model()
    rs = np.random.default_rng(1234)
    # single-thread rate first (short), then all usable cores on a sample sized for ~`seconds`
    e1 = [O.OracleEnv(om)]
    e1[0].reset()
    t0 = time.perf_counter()
    O.batch_rollout(e1, rs.uniform(-1, 1, (40, 1, 69)), 1)
    rate1 = 40 / (time.perf_counter() - t0)
    nenv = cores * 4
    envs = [O.OracleEnv(om) for _ in range(nenv)]
    for e in envs:
        e.reset()
    steps2 = int(max(8, min(4000, seconds * rate1 * cores / nenv)))
    acts = rs.uniform(-1, 1, (steps2, nenv, 69))
    t0 = time.perf_counter()
    done = O.batch_rollout(envs, acts, cores)
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "single_thread_value": rate1,
            "build": flags,
            "sample": f"{nenv} envs x {steps2} control steps, uniform(-1,1) actions, float64 C oracle "
                      f"(oracle/oracle.c; NOT MuJoCo; dense Cholesky per Newton iteration), {cores} threads (cgroup/affinity-usable cores), {dt:.1f}s"}


def parity_probe(env, abuf, g, humanoid, n=64):
    """Live check of the timed configuration against the CPU oracle at MuJoCo's solver settings (the oracle as checker, after
    the timed region): one more control step of the batch; the envs with the most Newton iterations plus the first n / 2 envs
    are replayed by oracle/oracle.c (float64, mj_solPrimal's termination: tolerance 1e-8, 100 iterations) from the GPU's own
    pre-step state.  newton_cap_gap_frac = the fraction of the replayed samples (no episode end, no bad-state autoreset inside
    the step) whose GPU result is outside the stated per-step tolerance of the oracle's (tests/parity_tools.TOL_STEP; errors
    relative to max(1, |qvel|_max)) — float32 rounding on chaotic states included: tests/test_gpu_parity.py splits the causes."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_tools as P
    heavy = torch.argsort(env.solver_iters, descending=True)[:n // 2]
    idx = torch.unique(torch.cat([heavy, torch.arange(n // 2, device=heavy.device)]))
    pre = {k: getattr(env, k)[idx].double().cpu().numpy() for k in P.FIELDS}
    nw0 = env.nwarn[idx].clone()
    act = abuf.uniform_(-1.0, 1.0, generator=g)
    _, _, term, trunc, _ = env.step(act)
    Gpu.sync()
    alive = ~(term.bool() | trunc.bool())[idx].cpu().numpy()
    post = dict(qpos=env.qpos[idx].double().cpu().numpy(), qvel=env.qvel[idx].double().cpu().numpy())
    nw = (env.nwarn[idx] - nw0).cpu().numpy()
    orc = P.oracle_step(pre, act[idx].double().cpu().numpy(), humanoid, self_collision=bool(getattr(env, "self_collision", False)))
    ok = alive & (nw == 0) & (orc["nwarn"] == 0)
    e = P.rel_err(post, orc)
    within = P.within_tol(e)
    return {"samples": int(ok.sum()), "skipped_episode_end_or_bad_state_reset": int((~ok).sum()),
            "newton_cap_gap_frac": float((~within[ok]).mean()) if ok.any() else None,
            "max_rel_err_qpos_qvel": e[ok].max(axis=0).tolist() if ok.any() else None,
            "median_rel_err_qpos_qvel": np.median(e[ok], axis=0).tolist() if ok.any() else None,
            "stated_tolerance_qpos_qvel": P.TOL_STEP.tolist(),
            "oracle": "oracle/oracle.c float64 at MuJoCo's solver settings (tolerance 1e-8 x meaninertia x nv, 100 iterations); NOT MuJoCo itself"}


def reference_contact_set(args, rank, local_rank, dev, humanoid_model, workload_kw, steps=30, warmup=20):
    """The same batch with the reference MJCF's full contact set (body-body contacts on, smpl_humanoid.xml:5,24,231-242): a short
    run after the headline loop, so that the driver-run line carries the figure next to the floor-contact one."""
    import torch
    from smplsim_amd import shard
    from smplsim_amd.batch import SMPLSimVecEnv
    N = args.envs_per_gpu
    env = SMPLSimVecEnv(N, model=humanoid_model, autoreset=True, seed=shard.shard_seed(1234, rank), self_collision=True,
                        newton_iters=args.newton_iters, **workload_kw)
    g = torch.Generator(device=dev); g.manual_seed(shard.shard_seed(4321, rank))
    env.reset()
    abuf = torch.empty(N, env.nu, device=dev)
    for _ in range(warmup):
        env.step(abuf.uniform_(-1.0, 1.0, generator=g))
    from smplsim_amd._lib import lib
    from smplsim_amd.batch import _check, _ptr
    trunc = torch.zeros(N, dtype=torch.int32, device=dev)       # mj_steps whose body-body contact list exceeded the kernel's capacity
    _check(lib().ss_debug_self_truncation(env.handle, _ptr(trunc)))
    ev0 = [Gpu.event() for _ in range(steps)]; ev1 = [Gpu.event() for _ in range(steps)]
    Gpu.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(abuf.uniform_(-1.0, 1.0, generator=g), _events=(ev0[i], ev1[i]))
    Gpu.sync()
    el = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    bstep = algorithmic_bytes(env.nq, env.nv, env.nu, env.obs_size)
    return {"value": N * steps / el, "unit": "env-steps/s (this GPU)", "steps": steps, "ms_per_step": 1e3 * el / steps, "kernel_ms": kern_ms,
            "frac": N * bstep / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "envs_per_cu": env.launch_info()["envs_per_workgroup"],
            "mean_newton_iters_per_step": float(env.solver_iters.float().mean().item()),
            "envs_with_body_body_contact_frac": float((env.self_contacts > 0).float().mean().item()),
            "mean_body_body_contacts": float(env.self_contacts.float().mean().item()),
            "max_body_body_contacts_kept": int(env.self_contacts.max().item()),
            "truncated_mj_step_frac": float(trunc.sum().item()) / (N * steps * 15),
            "truncated_envs": int((trunc > 0).sum().item()),
            "truncation": "every body-body contact of the narrow phase is kept, like MuJoCo (one per lane of the env's wavefront; rounds 2-3 "
                          "kept the deepest 8 and cut 4.8 % of the mj_steps): what is left are mj_steps with more than 64 simultaneous "
                          "body-body contacts — humanoids folded into themselves a step or two before MuJoCo's bad-state reset",
            "roofline": dict({"frac": N * bstep / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_measured_peak": N * bstep / (kern_ms * 1e-3) / 1e9 / HBM_MEASURED_GBS,
                              "kernel_ms": kern_ms}, **pmc_summary("smpl_selfcollision", N, kern_ms)),
            "note": "self_collision=True: capsule-capsule / capsule-box / box-box between all non-excluded, non-adjacent body pairs, "
                    "as mj_step collides the reference MJCF"}


def synthetic_clips(num, frames, seed):
    """Smooth random clips in the reference's pickle format ({key: {pose_aa [T,72], trans [T,3], fps}}): a standing pose plus
    low-frequency joint oscillations and a slow walk of the root (there is no AMASS data in the image)."""
    rs = np.random.default_rng(seed)
    t = np.arange(frames)[:, None, None] / 30.0
    clips = {}
    for c in range(num):
        amp = rs.uniform(0.05, 0.35, size=(1, 24, 1)) * rs.uniform(0.2, 1.0, size=(1, 24, 3))
        amp[:, 0] *= 0.2
        pose = amp * np.sin(2 * np.pi * rs.uniform(0.2, 1.2, size=(1, 24, 3)) * t + rs.uniform(0, 2 * np.pi, size=(1, 24, 3)))
        # SMPL rest pose is y-up: the root rotation that stands the body up on the z-up floor (quaternion .5,.5,.5,.5)
        from scipy.spatial.transform import Rotation as sRot
        root = sRot.from_euler("z", rs.uniform(-np.pi, np.pi)) * sRot.from_quat([0.5, 0.5, 0.5, 0.5])
        pose[:, 0] = (root * sRot.from_rotvec(pose[:, 0])).as_rotvec()
        tt = t[:, 0, 0]
        trans = np.stack([0.3 * tt * np.cos(c), 0.3 * tt * np.sin(c), 0.94 + 0.01 * np.sin(2 * tt)], -1)
        clips[f"synthetic_{c:04d}"] = {"pose_aa": pose.reshape(frames, 72).astype(np.float32), "trans": trans.astype(np.float32), "fps": 30}
    return clips


def run_imitation(args, rank, local_rank, world, dist, dev):
    import ctypes as C
    import torch
    from smplsim_amd import shard
    from smplsim_amd._lib import lib
    from smplsim_amd.batch import ShardModel
    from smplsim_amd.imitation import SMPLSimImitationVecEnv
    from smplsim_amd.motion_lib import MotionLibSMPL, Skeleton
    N = args.envs_per_gpu
    model = ShardModel(device=local_rank)
    clips = synthetic_clips(args.clips, args.clip_frames, shard.shard_seed(77, rank))
    ml = MotionLibSMPL(clips, Skeleton.from_model_const(model.mc), device=local_rank, seed=shard.shard_seed(5, rank))
    ml.load_motions()                                           # warm-up (module load), then timed
    Gpu.sync()
    t0 = time.perf_counter()
    ml.load_motions()
    Gpu.sync()
    load_s = time.perf_counter() - t0
    F, J = ml.data.num_frames, 24
    env = SMPLSimImitationVecEnv(N, ml, model=model, device=local_rank, seed=shard.shard_seed(1234, rank), fused=not args.unfused)
    env.reset()

    def one_step():
        return env.step(env.reference_actions())

    for _ in range(args.warmup):
        one_step()
    ev0 = [Gpu.event() for _ in range(args.steps)]
    ev1 = [Gpu.event() for _ in range(args.steps)]
    shard.barrier(dist, world, dev)
    t0 = time.perf_counter()
    rew_sum, ended = torch.zeros((), device=dev), torch.zeros((), device=dev)
    rew_acc = torch.zeros(N, device=dev); end_acc = torch.zeros(N, dtype=torch.int32, device=dev)
    for i in range(args.steps):
        a = env.reference_actions()                              # the stand-in policy (one clip lookup launch + two elementwise ones)
        if env.fused:
            env.step(a, _events=(ev0[i], ev1[i]))                # events around the ONE launch of the step (ss_imitation_step_fused)
        else:
            ev0[i].record()
            env.step(a)                                          # --unfused: the whole launch sequence
            ev1[i].record()
        rew_acc.add_(env.rew_buf); end_acc.add_(env.terminated); end_acc.add_(env.truncated)
    rew_sum, ended = rew_acc.mean(), end_acc.sum()
    shard.barrier(dist, world, dev)
    elapsed = shard.max_over_ranks(dist, world, time.perf_counter() - t0, dev)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    # the row's own kernels, timed alone on this stream: imitation step (per launch) and the three cooking launches
    reps = 200
    e0, e1 = Gpu.event(), Gpu.event()
    st = Gpu.stream_ptr(dev)
    e0.record()
    for _ in range(reps):
        env._imitation(None, env.rew_buf, env.reward_parts, env.terminated, env.truncated)
    e1.record(); Gpu.sync()
    im_ms = e0.elapsed_time(e1) / reps
    from smplsim_amd import _cabi
    sk = _cabi.Skeleton(J, ml._sk_keep[0].ctypes.data_as(C.c_void_p), ml._sk_keep[1].ctypes.data_as(C.c_void_p))
    e0.record()
    for _ in range(20):
        lib().ss_motion_cook(C.byref(sk), C.byref(ml.data), 1, st)
    e1.record(); Gpu.sync()
    cook_ms = e0.elapsed_time(e1) / 20
    if rank == 0:
        # dominant kernel = the step launch itself (the IMIT instantiation of ss_env_kernel: physics + imitation task + re-initialisation of
        # finished envs).  Algorithmic bytes per env-step: the stepper's state in / out and the self observation (SURVEY 8d formula) plus the
        # imitation part: two clip lookups (t: reward / termination, t + dt: task observation) of two frames x 13 J floats each, the task
        # observation (24 J), reward + 4 parts + flag.  PMC figures from the committed passes of this command (tools/gpu_prof.sh).
        step_bytes = algorithmic_bytes(env.base.nq, env.base.nv, env.base.nu, env.self_obs_size)
        im_bytes = 4 * (2 * 2 * 13 * J + 24 * J + 5) + 1
        cook_bytes = 4 * (75 + 13 * J + 4 * J + 6 * J + 6 * (J - 1) + 76 + 75)  # raw clip in; gts,grs,lrs,gvs,gavs,dof_pos,dvs,qpos,qvel out
        bstep = step_bytes + im_bytes
        ach = N * bstep / (kern_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                # against the copy bandwidth measured on this part (profiles/README.md: 6.29 TB/s) instead of the data-sheet 8 TB/s
                "frac_measured_peak": ach / HBM_MEASURED_GBS, "measured_peak": HBM_MEASURED_GBS, "traffic": None,
                "traffic_unit": "bytes per step launch",
                "kernel": "ss_env_kernel<IMIT> via ss_imitation_step_fused" if env.fused else "launch sequence ss_step .. ss_imitation_step (--unfused)",
                "kernel_ms": kern_ms, "algorithmic_bytes_per_env_step": bstep,
                "algorithmic_bytes_split": {"stepper_state_and_self_obs": step_bytes, "clip_lookups_task_obs_reward": im_bytes},
                "waves_per_cu": env.base.launch_info()["envs_per_workgroup"],
                "imitation_task_alone": {"kernel": "ss_imitation_kernel<32> (the task part as its own launch, not part of the timed loop)",
                                         "kernel_ms": im_ms, "GB/s": N * (im_bytes + 4 * 18 * J) / (im_ms * 1e-3) / 1e9},
                "note": "the step launch is the physics of the env step (LDS-latency / VALU bound like the headline kernel, DESIGN.md); "
                        f"{N} envs are launched as ceil(N / CUs) envs per workgroup over all CUs"}
        roof.update(pmc_summary("imitation", N, kern_ms))
        out = {
            "metric": "env-steps/sec (whole node), motion-imitation rollout", "value": shard.whole_job_throughput(N * world * args.steps, elapsed),
            "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS["imitation"].format(N=N), "envs_per_gpu": N, "envs_total": N * world, "clips": ml.num_current_motions(), "frames": F,
                       "parallelism": f"independent shards x{world} (no collective)", "launch": env.base.launch_info(),
                       "mean_reward": float(rew_sum.item()) / args.steps, "episodes_ended": int(ended.item()),
                       "obs_finite": bool(torch.isfinite(env.obs_buf).all().item()), "parity_pin": PARITY_PIN,
                       "fused_step": bool(env.fused), "load_motions_s (upload + cook)": load_s,
                       "mean_newton_iters_per_step": float(env.base.solver_iters.float().mean().item()),
                       "cook": {"ms": cook_ms, "frames_per_s": F / (cook_ms * 1e-3), "GB/s": F * cook_bytes / (cook_ms * 1e-3) / 1e9,
                                "algorithmic_bytes_per_frame": cook_bytes}},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_motion()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline_motion():
    """The NumPy restatement of the reference's per-clip fk_batch (oracle/motion_oracle.py; 'port') on a bounded sample."""
    from oracle import motion_oracle as mo
    from smplsim_amd.mjcf import compile_mjcf
    from smplsim_amd.mjcf_writer import default_xml_str
    from smplsim_amd.motion_lib import Skeleton
    sk = Skeleton.from_model_const(compile_mjcf(default_xml_str("smpl_humanoid")))
    clips = list(synthetic_clips(8, 300, 1).values())
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 10.0:
        for c in clips:
            mo.cook(c["pose_aa"].reshape(-1, 24, 3), c["trans"], sk.offsets, sk.parents, sk.smpl_2_mujoco, 1 / 30, True)
        n += len(clips)
    dt = time.perf_counter() - t0
    return {"value": n * 300 / dt, "unit": "frames/s (clip cooking)", "cores": 1, "kind": "port",
            "sample": f"{n} clips x 300 frames through oracle/motion_oracle.cook (NumPy float64 restatement of fk_batch), {dt:.2f}s; "
                      "compare with config.cook.frames_per_s"}


def spawn_ranks(n, argv):
    """`bench.py --gpus N` called plainly (no RANK / WORLD_SIZE in the environment): start N ranks of THIS script, one per GPU, under
    torch.distributed.run on the loopback address and hand its exit status back — the one-call fan-out of the reference's
    examples/benchmark.py:78-81 (gym.vector.AsyncVectorEnv over worker processes).  The launcher sets RANK / LOCAL_RANK / WORLD_SIZE /
    LOCAL_WORLD_SIZE / MASTER_*; the ranks take the branch below this call."""
    import socket
    import subprocess
    have = Gpu.device_count()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} device(s) visible; refusing to run {n} ranks on fewer GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = os.path.abspath(sys.argv[0])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script, *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=None, help="default 4096 (imitation: 1024 = 8192 envs on 8 GPUs)")
    ap.add_argument("--envs-total", type=int, default=None,
                    help="STRONG scaling: this many envs in total, split evenly over the ranks (the metric's '4096-env rollout at 1/2/4/8 MI355X' "
                         "read literally: --envs-total 4096); default = weak scaling with --envs-per-gpu envs on every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1234, help="seed of the envs and of the action stream (A/B studies: the launch follows its heaviest env, "
                    "so two kernels with different rounding are only comparable over several trajectories)")
    ap.add_argument("--self-collision", action="store_true",
                    help="contacts between the humanoid's own bodies like mj_step on the reference MJCF (SURVEY 8f-4); default: floor "
                         "contacts and joint limits only")
    ap.add_argument("--newton-iters", type=int, default=0, help="mjOption.iterations: Newton iteration cap per mj_step (0 = MuJoCo's default, 100)")
    ap.add_argument("--no-reference-contact-set", action="store_true", help="skip the short self_collision=True run after the timed loop")
    ap.add_argument("--unfused", action="store_true", help="imitation: the separate launches instead of ss_imitation_step_fused")
    ap.add_argument("--clips", type=int, default=256, help="imitation: synthetic clips per shard")
    ap.add_argument("--clip-frames", type=int, default=300, help="imitation: frames per synthetic clip (30 fps)")
    ap.add_argument("--workload", default="smpl", choices=["smpl", "getup", "smplx", "imitation"],
                    help="smpl = BASELINE config 2 (the metric); getup = config 3 shard (Fall init, getup task); smplx = config 4; "
                         "imitation = config 5 shard (motion clips, tracking reward)")
    args = ap.parse_args(argv)

    import torch
    from smplsim_amd import shard
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, list(sys.argv[1:] if argv is None else argv)))
    rank, local_rank, world = shard.rank_info()
    if world != args.gpus:
        # never print a line whose n_gpus is not what was asked for
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a launcher with WORLD_SIZE={world}: start as many ranks as GPUs requested")
    host_cores = shard.pin_host_threads(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # one slice of the host cores per rank
    dist = shard.init_process_group(Gpu.backend, local_rank) if world > 1 else None
    dev = Gpu.device(local_rank)

    from smplsim_amd.batch import SMPLSimVecEnv
    args.scaling = "weak"
    if args.envs_total is not None:
        if args.envs_total % world:
            raise SystemExit(f"--envs-total {args.envs_total} is not a multiple of the {world} ranks")
        args.envs_per_gpu, args.scaling = args.envs_total // world, "strong"
    if args.envs_per_gpu is None:
        args.envs_per_gpu = 1024 if args.workload == "imitation" else ENVS_PER_GPU
    N = args.envs_per_gpu
    if args.workload == "imitation":
        return run_imitation(args, rank, local_rank, world, dist, dev)
    from smplsim_amd.batch import ShardModel
    humanoid = "smplx_humanoid" if args.workload == "smplx" else "smpl_humanoid"
    model = ShardModel(humanoid=humanoid, device=local_rank)
    workload_kw = dict(task="HumanoidGetup", state_init="Fall", self_obs_v=1) if args.workload == "getup" else \
        dict(task="HumanoidEnv", state_init="Default", self_obs_v=1)
    env = SMPLSimVecEnv(N, model=model, autoreset=True, seed=shard.shard_seed(args.seed, rank), self_collision=args.self_collision,
                        newton_iters=args.newton_iters, **workload_kw)
    g = torch.Generator(device=dev)
    g.manual_seed(shard.shard_seed(args.seed, rank))
    env.reset()

    abuf = torch.empty(N, env.nu, device=dev)                   # fresh uniform(-1, 1) actions per control step: one fill launch, in place

    def one_step():
        env.step(abuf.uniform_(-1.0, 1.0, generator=g))

    for _ in range(args.warmup):
        one_step()

    def barrier():
        shard.barrier(dist, world, dev)

    # per-launch duration of the dominant kernel (the fused step), HIP events on the launch stream
    ev0 = [Gpu.event() for _ in range(args.steps)]
    ev1 = [Gpu.event() for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # env.step = LPT hand-out order (argsort of last step's Newton counts) + the fused step launch + masked autoreset
        env.step(abuf.uniform_(-1.0, 1.0, generator=g), _events=(ev0[i], ev1[i]))
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(dist, world, elapsed, dev)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    finite = bool(torch.isfinite(env.obs_buf).all().item())
    nwarn = int(env.nwarn.sum().item())
    iters = float(env.solver_iters.float().mean().item())
    it_sorted = torch.sort(env.solver_iters.float()).values
    it_p50, it_p99, it_max = (float(it_sorted[int(q * (N - 1))].item()) for q in (0.5, 0.99, 1.0))

    # rank 0 only, after the timed region (the other ranks go straight to the final barrier of the process group: no lease time spent
    # on a figure that is reported for one GPU)
    refset = None
    if rank == 0 and not args.self_collision and not args.no_reference_contact_set:
        refset = reference_contact_set(args, rank, local_rank, dev, model, workload_kw)

    if rank == 0:
        total_envs = N * world
        value = shard.whole_job_throughput(total_envs * args.steps, elapsed)
        bstep = algorithmic_bytes(env.nq, env.nv, env.nu, env.obs_size)
        ach = N * bstep / (kern_ms * 1e-3) / 1e9
        launch = env.launch_info()
        flops = flops_per_env_step(env.nbody, env.nv, launch.get("contact_candidates", 4 * env.nbody), iters)
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                # against the copy bandwidth measured on this part (MI355X_MICROARCH.md: 6.29 TB/s) instead of the data-sheet 8 TB/s
                "frac_measured_peak": ach / HBM_MEASURED_GBS, "measured_peak": HBM_MEASURED_GBS,
                "traffic": None, "traffic_unit": "bytes per step launch",
                "kernel": "ss_env_kernel (MODE_STEP)", "kernel_ms": kern_ms,
                "algorithmic_bytes_per_env_step": bstep,
                # the real limiter beside the designated one: FP32 work of the formulation against the vector peak,
                # measured live; issue / LDS / residency figures from the committed PMC passes of this command
                "fp32_useful_frac": N * flops / (kern_ms * 1e-3) / FP32_VECTOR_PEAK, "flops_per_env_step_model": flops,
                "waves_per_cu": launch["envs_per_workgroup"],
                "note": "path is LDS-latency/VALU bound, not HBM bound (DESIGN.md §roofline)"}
        roof.update(pmc_summary(args.workload + ("_selfcollision" if args.self_collision else ""), N, kern_ms))
        out = {
            "metric": "env-steps/sec (whole node), 4096-env SMPL rollout at 1/2/4/8 MI355X", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload].format(N=N),
                       "envs_per_gpu": N, "envs_total": N * world, "parallelism": f"independent shards x{world} (no collective)",
                       "scaling_note": SCALING_NOTE,
                       "host_cores_pinned_per_rank": (len(host_cores) if host_cores else None),
                       "launch": launch,
                       "solver": {"iterations": args.newton_iters if args.newton_iters > 0 else 100, "tolerance": 1e-8,
                                  "rule": "mj_solPrimal's termination (improvement or gradient, scaled by 1 / (meaninertia nv), below the "
                                          "tolerance), MuJoCo's defaults; tests/test_gpu_parity.py holds the kernel at these settings to the "
                                          "oracle at the same"},
                       "newton_iters_cap": args.newton_iters if args.newton_iters > 0 else 100,
                       "mean_newton_iters_per_step": iters, "newton_iters_p50_p99_max": [it_p50, it_p99, it_max],
                       "bad_state_resets_total": nwarn, "self_collision": bool(getattr(env, "self_collision", False)),
                       "envs_with_body_body_contact_frac": float((env.self_contacts > 0).float().mean().item()),
                       "mean_body_body_contacts": float(env.self_contacts.float().mean().item()),
                       "obs_finite": finite, "parity_pin": PARITY_PIN},
            "roofline": roof,
        }
        if refset is not None:
            out["config"]["reference_contact_set"] = refset
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            probe = parity_probe(env, abuf, g, humanoid)
            out["config"]["parity_probe"] = probe
            out["config"]["newton_cap_gap_frac"] = probe["newton_cap_gap_frac"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
