"""Per-sample parity on the benchmark's own state distribution, float32 kernel vs its float64 instantiation vs the oracle
(CPU: both kernel builds run on the wavefront emulator; the GPU twin is test_gpu_parity.py::test_per_sample_parity_*).

What is asserted PER SAMPLE (tests/parity_tools.py explains the replays), errors relative to max(1, |qvel|_max):
  formulation  float64 kernel vs oracle, both converged   <= max(1e-9, K cond 2^-53)   (the two formulations solve the same
               problem: measured max 1e-12 on the emulator's samples, 7e-10 on the GPU's stragglers)
  resets       MuJoCo bad-state autoresets (|x| > 1e10) happen in the same samples in oracle and float64 kernel
  solver rule  float64 kernel at the SHIPPED solver settings vs the oracle at MuJoCo's (mj_solPrimal's termination test,
               tolerance 1e-8, 100 iterations): within the stated per-step tolerance P.TOL_STEP on >= 99.5% of the samples, the
               others listed (the kernel's form of the test — DESIGN.md "solver termination" — against MuJoCo's own)
  precision    float32 kernel vs float64 kernel, both at the shipped settings   <= K * cond * 2^-24 with K = 8192, where cond is
               the sample's measured response to float32-sized input noise (floored at COND_FLOOR): the float32 error is a
               bounded multiple of rounding unit x conditioning; the float64 kernel's distance to the oracle obeys the same K with
               2^-53.  The fraction of samples whose float32 result is within P.TOL_STEP of the oracle at MuJoCo's settings is
               reported and bounded below
"""
import numpy as np
import pytest

import parity_tools as P

K_ROUND = 8192.0                          # float64 kernel vs oracle: error <= K cond 2^-53 (two formulations, float64 rounding)
COND_FLOOR = np.array([1.0, 50.0])       # qpos, qvel: the conditioning of a quiet sample (measured medians 0.7 / 60)
# float32 kernel vs float64 kernel, per sample: the stated per-step tolerance P.TOL_STEP holds for every sample whose measured
# conditioning is at most COND_REF (about the median of the benchmark's states) and grows in proportion to the conditioning
# beyond it:  error <= TOL_STEP * max(1, cond / COND_REF).  Round 6: COND_REF was (1, 80), 15-40 x looser than anything
# measured (worst precision / bound over the six configurations on the MI355X: 0.066 / 0.025).  From the per-sample records of the GPU
# run (gpurun_out/parity_samples_*.npz: precision and conditioning of every sample) the bound was moved to where the worst sample sits at
# 0.30 / 0.27 of it on the MI355X (getup, a sample of conditioning < 20 at 0.3 of TOL_STEP itself: the flat part cannot go lower without
# changing the stated tolerance; smpl for qvel) and at 0.38 / 0.89 on the emulator's samples: a float32 error 1.1 x (emulator) / 3 x (GPU)
# worse on the worst sample now fails.  profiles/r06_parity_measured.json.
COND_REF = np.array([20.0, 10000.0])


def f32_bound(cond):
    return P.TOL_STEP * np.maximum(1.0, cond / COND_REF)


def _check(r, label, min_samples, max_f32_outside=0.02):
    ok = ~r["reset"]
    lines = [f"[{label}] samples {len(ok)}, with a bad-state autoreset inside the step {int((~ok).sum())}"]
    for k in ("formulation", "solver_rule", "oracle_rule", "precision", "f32_vs_oracle", "cond"):
        lines.append(P.summarize(k, r[k], ok))
    rule_ok, f32_ok = P.within_tol(r["solver_rule"]), P.within_tol(r["f32_vs_oracle"])
    lines.append(f"solver rule within {P.TOL_STEP}: {rule_ok[ok].mean():.4f} of the samples; outside: {np.flatnonzero(ok & ~rule_ok).tolist()} {r['solver_rule'][ok & ~rule_ok].tolist()}")
    lines.append(f"float32 vs oracle (MuJoCo settings) within {P.TOL_STEP}: {f32_ok[ok].mean():.4f} of the samples; outside (sample, error, cond): "
                 f"{[(int(i), r['f32_vs_oracle'][i].tolist(), r['cond'][i].tolist()) for i in np.flatnonzero(ok & ~f32_ok)]}")
    if r.get("iters32") is not None:
        lines.append(f"Newton iterations per control step: float32 mean {r['iters32'].mean():.2f} max {r['iters32'].max()}, float64 kernel mean {r['iters'].mean():.2f} max {r['iters'].max()}")
    cond = np.maximum(r["cond"], COND_FLOOR)
    ratio32 = r["precision"] / (cond * P.EPS32)
    ratio64 = r["formulation"] / (cond * P.EPS64)
    lines.append(f"precision / (cond eps32): median {np.median(ratio32[ok], axis=0)} max {ratio32[ok].max(axis=0)}")
    lines.append(f"formulation / (cond eps64): median {np.median(ratio64[ok], axis=0)} max {ratio64[ok].max(axis=0)}")
    if "obs" in r:
        lines.append(f"obs (f32 vs f64 kernel) max {r['obs'][ok].max():.3e}  reward max {r['reward'][ok].max():.3e}")
    print("\n".join(lines))
    assert ok.sum() >= min_samples, ok.sum()
    # MuJoCo's bad-state autoreset (|qpos|, |qvel|, |qacc| > 1e10 or NaN): the oracle and the float64 kernel must take it on the
    # same samples — except that a trajectory which is blowing up crosses 1e10 one mj_step earlier or later depending on
    # rounding (these states double per step), so a sample in a few hundred may flip; such samples are excluded from the
    # comparisons above either way ("reset" = any implementation reset)
    assert (~r["resets_agree"]).sum() <= max(1, len(r["resets_agree"]) // 100), (~r["resets_agree"]).sum()
    assert (r["formulation"][ok] <= np.maximum(1e-9, K_ROUND * cond[ok] * P.EPS64)).all(), r["formulation"][ok].max(axis=0)
    med, p90 = np.median(r["precision"][ok], axis=0), np.quantile(r["precision"][ok], 0.9, axis=0)
    assert med[0] < 5e-7 and med[1] < 5e-5 and p90[0] < 5e-6 and p90[1] < 5e-4, (med, p90)
    assert (ratio64[ok] <= K_ROUND).all(), ratio64[ok].max(axis=0)
    bound32 = f32_bound(r["cond"])
    lines_ = f"precision / f32_bound(cond): max {(r['precision'][ok] / bound32[ok]).max(axis=0)}"
    print(lines_)
    assert (r["precision"][ok] <= bound32[ok]).all(), (r["precision"][ok] / bound32[ok]).max(axis=0)
    assert rule_ok[ok].mean() >= 0.995, (rule_ok[ok].mean(), r["solver_rule"][ok & ~rule_ok])
    # the float32 kernel within the stated tolerance of the oracle at MuJoCo's settings on all but 2 % of the samples (one at least:
    # the sets are small and heavy with stragglers); measured on the MI355X: 97.3 - 100 % on the six configurations
    assert (~f32_ok[ok]).sum() <= max(1, int(max_f32_outside * ok.sum())), (f32_ok[ok].mean(), np.flatnonzero(ok & ~f32_ok))
    rates = P.check_reset_rates(r, label)                      # bad-state autoresets: rate per implementation within 20 % of the oracle's
    print(f"bad-state reset rates: {rates}")
    return lines


def test_smpl_benchmark_distribution_per_sample():
    """BASELINE config 2's distribution: SMPL, Default init, uniform(-1,1) actions, samples from control step 8 on (thrown
    around, lying on the floor with many contacts, some diverging)."""
    pre, A, post = P.rollout_samples_emu(24, 20, seed=3, skip=8)
    _check(P.triage(pre, A, post, n_perturb=5), "smpl uniform(-1,1)", 250)


def test_getup_fall_distribution_per_sample():
    """BASELINE config 3's distribution: getup task, StateInit.Fall (45 warm-up mj_steps), uniform(-1,1) actions."""
    kw = dict(task="HumanoidGetup", state_init="Fall")
    pre, A, post = P.rollout_samples_emu(16, 10, seed=5, skip=0, **kw)
    r = P.triage(pre, A, post, task="HumanoidGetup", state_init=1, n_perturb=5)
    _check(r, "getup / Fall uniform(-1,1)", 120)
    ok = ~r["reset"]
    # the task observation / reward follow the state: their float32 error is bounded by the sample's state error
    worst = r["precision"].max(axis=1)
    # (measured on all six configurations: obs error / state error <= 1.0 for every sample whose state error is below 1 % of the velocity scale;
    #  a sample that is diverging — state error 6 % of the scale at a condition number of 9e5 — showed 6.6: the observation's rotations are
    #  not linear over such a distance.  Those samples get the looser factor.)
    lin = worst <= 1e-2
    assert (r["obs"][ok & lin] <= 4 * worst[ok & lin] + 1e-5).all()
    assert (r["obs"][ok & ~lin] <= 16 * worst[ok & ~lin]).all()
    assert (r["reward"][ok] <= 2 * r["precision"][ok, 0] * r["vscale"][ok] + 1e-6).all()


def test_smplx_benchmark_distribution_per_sample():
    """BASELINE config 4's distribution: the SMPL-X/H-layout humanoid (52 bodies, nv 159) under uniform(-1,1) actions."""
    pre, A, post = P.rollout_samples_emu(8, 14, seed=7, skip=6, humanoid="smplx_humanoid")
    _check(P.triage(pre, A, post, humanoid="smplx_humanoid", n_perturb=5), "smplx uniform(-1,1)", 50)


@pytest.mark.parametrize("obs_v", [1, 2])
def test_proprioception_is_yaw_invariant_on_the_emulator(obs_v):
    """The reference's commented heading-invariance check (humanoid_env.py:497-504) on the observation itself: a state and
    its copies turned about the vertical (and moved in the plane) observe the same vector.  GPU twin in test_gpu_parity.py."""
    from helpers import FEET, default_qpos, model_const, pd_tables
    from wave_emu import emu
    n = 6
    rs = np.random.default_rng(obs_v)
    q = np.tile(default_qpos(76), (n, 1)); v = np.zeros((n, 75))
    q[:, 7:] = rs.uniform(-0.6, 0.6, (1, 69)); v[:] = rs.normal(size=(1, 75))
    base = rs.normal(size=4); base /= np.linalg.norm(base)
    for i in range(n):
        th = 0.0 if i == 0 else rs.uniform(-np.pi, np.pi)
        w1, z1 = np.cos(th / 2), np.sin(th / 2)
        w2, x2, y2, z2 = base
        q[i, 3:7] = [w1 * w2 - z1 * z2, w1 * x2 - z1 * y2, w1 * y2 + z1 * x2, w1 * z2 + z1 * w2]
        c, s_ = np.cos(th), np.sin(th)
        v[i, 0], v[i, 1] = c * v[0, 0] - s_ * v[0, 1], s_ * v[0, 0] + c * v[0, 1]
        if i:
            q[i, :2] = rs.uniform(-5, 5, 2)
    mc = model_const()
    for f64, tol in ((False, 2e-5), (True, 1e-12)):
        eb = emu.EmuBatch(mc, pd_tables(mc), n, legal_bodies=FEET, f64=f64, state_init=2, self_obs_v=obs_v)
        eb.set_state(q, v)
        o = eb.reset()
        d = np.abs(o[1:] - o[:1]).max(axis=0)
        if obs_v == 1:
            # v1 feeds qvel[3:6] — the root's angular velocity in the BODY frame — through the inverse heading rotation as if it
            # were a world vector (humanoid_env.py:615-617): those two entries turn with the yaw in the reference too; the
            # restatement is pinned to the reference's numpy code, so this asserts the same quirk, not an invariance
            assert d[217] > 1e-3 or d[218] > 1e-3
            d[217:219] = 0.0
        assert d.max() < tol * max(1.0, np.abs(o[0]).max())
