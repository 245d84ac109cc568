"""Per-sample parity triage on the benchmark's own state distribution (test infrastructure).

One *sample* = one control step (15 mj_steps) of one env: the pre-step state the stepper holds in HBM
(qpos, qvel, the stale qpos_prev/qvel_prev the Stable-PD controller's M and C come from, the solver warm start),
the action, and the float32 post-step state some implementation of the kernel produced (the GPU, or the wavefront
emulator).  Every sample is replayed from exactly that pre-step state by

  oracle        oracle/oracle.c, float64, different formulation, at MuJoCo's solver settings (mj_solPrimal's termination:
                tolerance 1e-8 scaled by meaninertia * nv, 100 iterations) = what mj_step computes
  oracle conv   the same with the Newton iteration run to the rounding level
  f64           the kernel source instantiated in float64 on the emulator (tests/wave_emu, -DSS_F64) at the SHIPPED solver
                settings (ss_env_cfg defaults: MuJoCo's tolerance and iteration count, the kernel's own form of the test)
  f64 conv      the same with the tolerance test off (1e-30): Newton to convergence
  f64 perturbed f64 conv from inputs perturbed by half a float32 ulp (relative 2^-24, random signs): the response of the
                one-control-step map to float32-sized input noise = the sample's conditioning

which separates the three things a float32-vs-oracle difference can be made of:
  formulation   f64 conv vs oracle conv   (must be at the float64 rounding level times the conditioning)
  solver rule   f64 vs oracle             (the kernel's termination test against MuJoCo's, same arithmetic: must be within the
                                           stated per-step tolerance)
  precision     float32 vs f64            (must be at the float32 rounding level times the conditioning)
Errors are relative to the sample's velocity scale max(1, |qvel|_max) like in test_gpu_parity.py.
"""
import concurrent.futures as cf
import os

import numpy as np

from helpers import FEET, model_const, oracle_model, pd_tables
from oracle import oracle as O
from wave_emu import emu

EPS32, EPS64 = 2.0 ** -24, 2.0 ** -53
MAX_SELF = 0                              # the oracle keeps every body-body contact, like MuJoCo (and since round 4 the kernel)
FIELDS = ("qpos", "qvel", "qpos_prev", "qvel_prev", "qacc_warm")


def n_threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def _chunks(n, k):
    k = max(1, min(k, n))
    b = [n * i // k for i in range(k + 1)]
    return [(b[i], b[i + 1]) for i in range(k) if b[i + 1] > b[i]]


TOL_STEP = np.array([1e-5, 2e-3])        # stated per-control-step tolerance (qpos, qvel relative to the velocity scale)


def emu_step(pre, actions, f64, newton_iters=0, humanoid="smpl_humanoid", task="HumanoidEnv", task_state=None, cur_t=None,
             task_rand=None, **cfg):
    """One control step of len(actions) independent envs on the emulator from the given pre-step arrays (threaded).
    Returns dict(qpos, qvel, obs, reward, terminated, truncated, nwarn, iters)."""
    from smplsim_amd import _cabi
    n = len(actions)
    mc = model_const(humanoid)
    out = {}

    def run(lo_hi):
        lo, hi = lo_hi
        eb = emu.EmuBatch(mc, pd_tables(mc), hi - lo, legal_bodies=FEET, f64=f64, newton_iters=newton_iters,
                          task=_cabi.TASKS[task], **cfg)
        eb.set_state(pre["qpos"][lo:hi], pre["qvel"][lo:hi], pre["qpos_prev"][lo:hi], pre["qvel_prev"][lo:hi], pre["qacc_warm"][lo:hi])
        if task_state is not None:
            eb.task[:] = task_state[lo:hi]
        if cur_t is not None:
            eb.cur_t[:] = cur_t[lo:hi]
        obs, rew, term, trunc = eb.step(actions[lo:hi], None if task_rand is None else task_rand[lo:hi])
        return lo, hi, dict(qpos=eb.qpos.astype(np.float64), qvel=eb.qvel.astype(np.float64), obs=obs.astype(np.float64),
                            reward=rew.astype(np.float64), terminated=term, truncated=trunc, nwarn=eb.nwarn.copy(),
                            iters=eb.solver_iters.copy(), nself=eb.self_contacts.copy())

    with cf.ThreadPoolExecutor(n_threads()) as ex:
        for lo, hi, r in ex.map(run, _chunks(n, n_threads())):
            for k, v in r.items():
                out.setdefault(k, np.zeros((n,) + v.shape[1:], v.dtype))[lo:hi] = v
    return out


def oracle_step(pre, actions, humanoid="smpl_humanoid", self_collision=False, solver="mujoco", linesearch="exact"):
    """The same control step by the float64 oracle (base task: state only), threaded.  Also returns the Newton iterations of the
    control step (sum over its 15 mj_steps) and, with linesearch="mujoco", the line search's statistics per sample."""
    om = oracle_model(humanoid, self_collision=bool(self_collision), max_self_contacts=MAX_SELF if self_collision else 0, solver=solver,
                      linesearch=linesearch)
    n = len(actions)
    q, v, nw = np.zeros((n, om.nq)), np.zeros((n, om.nv)), np.zeros(n, np.int32)
    its, lss = np.zeros(n, np.int64), np.zeros((n, 3), np.int64)

    def run(lo_hi):
        for i in range(*lo_hi):
            d = O.OracleData(om)
            d.qpos = pre["qpos_prev"][i]; d.qvel = pre["qvel_prev"][i]; d.forward()      # stale M, C of the last mj_forward
            d.qpos = pre["qpos"][i]; d.qvel = pre["qvel"][i]; d.warm = pre["qacc_warm"][i]
            for _ in range(15):
                d.ctrl = d.spd_torque(actions[i]); d.step(); its[i] += d.solver_iter
            q[i], v[i], nw[i], lss[i] = d.qpos, d.qvel, d.nwarn, d.ls_stats

    with cf.ThreadPoolExecutor(n_threads()) as ex:
        list(ex.map(run, _chunks(n, n_threads())))
    return dict(qpos=q, qvel=v, nwarn=nw, iters=its, ls_stats=lss)


def rollout_samples_emu(n_envs, n_steps, seed, skip=8, humanoid="smpl_humanoid", amp=1.0, task="HumanoidEnv", state_init="Default",
                        **cfg):
    """Benchmark-distribution samples made on the float32 emulator: n_envs envs under fresh uniform(-amp, amp) actions with
    the vector env's autoreset (Fall resets draw their own actions); every (env, step >= skip) whose episode did not end in
    that step is a sample.  Returns (pre dict of [S, ...] float64 arrays, actions [S, nu], float32 post dict)."""
    from smplsim_amd import _cabi
    mc = model_const(humanoid)
    rs = np.random.default_rng(seed)
    acts = rs.uniform(-amp, amp, (n_steps, n_envs, mc.nu))
    trand = rs.uniform(size=(n_steps, 2, n_envs, 4))
    falls = rs.uniform(size=(n_steps + 1, n_envs, 3, mc.nu))
    pres, posts, A = [], [], []

    def run(lo_hi):
        lo, hi = lo_hi
        eb = emu.EmuBatch(mc, pd_tables(mc), hi - lo, legal_bodies=FEET, task=_cabi.TASKS[task], state_init=_cabi.STATE_INITS[state_init], **cfg)
        eb.reset(fall_actions=falls[n_steps, lo:hi], task_rand=trand[0, 1, lo:hi])
        recs = []
        for t in range(n_steps):
            pre = {k: getattr(eb, k).astype(np.float64) for k in FIELDS}
            pre["task"], pre["cur_t"], pre["task_rand"] = eb.task.astype(np.float64), eb.cur_t.copy(), trand[t, 0, lo:hi]
            nw0 = eb.nwarn.copy()
            obs, rew, term, trunc = eb.step(acts[t, lo:hi], trand[t, 0, lo:hi])
            keep = ~(term | trunc)
            if t >= skip and keep.any():
                recs.append(({k: v[keep] for k, v in pre.items()}, acts[t, lo:hi][keep],
                             dict(qpos=eb.qpos.astype(np.float64)[keep], qvel=eb.qvel.astype(np.float64)[keep], obs=obs.astype(np.float64)[keep],
                                  reward=rew.astype(np.float64)[keep], nwarn=(eb.nwarn - nw0)[keep], iters=eb.solver_iters.copy()[keep],
                                  nself=eb.self_contacts.copy()[keep])))
            if (~keep).any():
                eb.reset(mask=~keep, fall_actions=falls[t, lo:hi], task_rand=trand[t, 1, lo:hi])
        return recs

    with cf.ThreadPoolExecutor(n_threads()) as ex:
        for recs in ex.map(run, _chunks(n_envs, n_threads())):
            for pre, a, post in recs:
                pres.append(pre); A.append(a); posts.append(post)
    pre = {k: np.concatenate([p[k] for p in pres]) for k in pres[0]}
    post = {k: np.concatenate([p[k] for p in posts]) for k in posts[0]}
    return pre, np.concatenate(A), post


def perturb(pre, seed, rel=EPS32):
    """Inputs moved by `rel` times the magnitude of their field in that sample (random signs): one float32 rounding at the
    scale of the vector's largest entry — what a float32 sum over the vector carries — not of each entry."""
    rs = np.random.default_rng(seed)
    out = {}
    for k in FIELDS:
        x = pre[k]
        mag = np.maximum(np.abs(x).max(axis=1, keepdims=True), 1.0 if k.startswith("qpos") else 0.0)
        out[k] = x + rel * mag * rs.choice([-1.0, 1.0], size=x.shape)
    return out


def rel_err(a, b):
    """[S, 2]: max |dqpos|, max |dqvel| per sample, relative to the sample's velocity scale."""
    scale = np.maximum(1.0, np.abs(b["qvel"]).max(axis=1))
    return np.stack([np.abs(a["qpos"] - b["qpos"]).max(axis=1) / scale, np.abs(a["qvel"] - b["qvel"]).max(axis=1) / scale], 1)


def triage(pre, actions, post32, humanoid="smpl_humanoid", n_perturb=8, task="HumanoidEnv", **cfg):
    """Replays of every sample (module docstring).  Returns a dict of [S, 2] relative error arrays + bookkeeping."""
    kw = dict(humanoid=humanoid, task=task, task_state=pre.get("task"), cur_t=pre.get("cur_t"), task_rand=pre.get("task_rand"), **cfg)
    sc = cfg.get("self_collision", False)
    orc = oracle_step(pre, actions, humanoid, self_collision=sc)
    orc_conv = oracle_step(pre, actions, humanoid, self_collision=sc, solver="converged")
    f64 = emu_step(pre, actions, True, **kw)
    f64conv = emu_step(pre, actions, True, solver_tolerance=1e-30, **kw)
    cond = np.zeros((len(actions), 2))
    for s in range(n_perturb):
        p = emu_step(perturb(pre, 100 + s), actions, True, solver_tolerance=1e-30, **kw)
        cond = np.maximum(cond, rel_err(p, f64conv) / EPS32)        # output error per unit of relative input error
    # a sample is "reset" when any implementation hit MuJoCo's bad-state autoreset inside the step (|x| > 1e10): the state
    # was replaced by qpos0 then, which is a discontinuity of the map — compared only through the reset flags
    reset = (orc["nwarn"] > 0) | (orc_conv["nwarn"] > 0) | (f64["nwarn"] > 0) | (f64conv["nwarn"] > 0) | (post32["nwarn"] > 0)
    extra = {}
    if "obs" in post32:                                          # observation / reward of the float32 kernel vs its float64 twin
        vs = np.maximum(1.0, np.abs(f64["qvel"]).max(axis=1))
        extra = dict(obs=np.abs(post32["obs"] - f64["obs"]).max(axis=1) / vs, reward=np.abs(post32["reward"] - f64["reward"]), vscale=vs)
    return dict(**extra, formulation=rel_err(f64conv, orc_conv), precision=rel_err(post32, f64), solver_rule=rel_err(f64, orc),
                oracle_rule=rel_err(orc, orc_conv), f32_vs_oracle=rel_err(post32, orc), cond=cond, reset=reset,
                resets_agree=(orc["nwarn"] > 0) == (f64["nwarn"] > 0), iters=f64["iters"], iters32=post32.get("iters"), nself=f64["nself"],
                reset_oracle=orc["nwarn"] > 0, reset_f64=f64["nwarn"] > 0, reset_f32=post32["nwarn"] > 0)


def check_reset_rates(r, name=""):
    """MuJoCo's bad-state autoreset inside the step, per implementation: the float32 kernel's and the float64 kernel's rates must be
    the oracle's own rate on the same states within 20 % (or two samples, whichever is larger), and the float32 kernel must reset
    the SAME samples as the oracle up to the few that cross 1e10 one mj_step earlier or later under rounding (these states double
    per mj_step).  Returns the rates for the record."""
    n = len(r["reset_oracle"])
    no, n64, n32 = int(r["reset_oracle"].sum()), int(r["reset_f64"].sum()), int(r["reset_f32"].sum())
    slack = max(2, int(np.ceil(0.2 * no)))
    assert abs(n32 - no) <= slack and abs(n64 - no) <= slack, (name, no, n64, n32)
    flips = int((r["reset_oracle"] != r["reset_f32"]).sum())
    assert flips <= max(2, n // 50), (name, flips, n)
    return dict(samples=n, reset_rate_oracle=no / n, reset_rate_f64_kernel=n64 / n, reset_rate_f32_kernel=n32 / n, reset_flips_f32_vs_oracle=flips)


def within_tol(e, tol=TOL_STEP):
    return (e <= tol).all(axis=1)


def summarize(name, e, mask):
    e = e[mask]
    q = lambda p: np.quantile(e, p, axis=0)
    return f"{name:14s} n={len(e):4d}  median {q(0.5)}  p90 {q(0.9)}  p99 {q(0.99)}  max {e.max(axis=0)}"
