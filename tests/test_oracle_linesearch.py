"""MuJoCo's line search in the oracle (OM_LS_MUJOCO; oracle.h MJ-(V9b)): PrimalSearch of mj_solPrimal restated —
bracketing + 1-D Newton steps, stopped at |slope| < tolerance * ls_tolerance * |direction| / scale or after ls_iterations evaluations —
next to the exact search the HIP kernel and the default oracle use.  The converged point is the same; what could differ are the
iterates, the iteration counts (what the benchmark's straggler tail is made of) and the point where `improvement < tolerance` ends
the iteration.  Measured here on the benchmark's own state distribution: nothing does (the stop rule asks for a slope of 1e-10 |direction|
meaninertia nv, and a 1-D Newton step on a piecewise quadratic is exact once it is on the minimiser's piece): the solver's output agrees to
1e-9 relative per control step and the iteration counts are identical — so the kernel needs no such option."""
import numpy as np
import pytest

import parity_tools as P
from helpers import oracle_model
from oracle import oracle as O


@pytest.mark.parametrize("selfcol", [False, True])
def test_mujoco_linesearch_gives_the_exact_searchs_result_on_the_benchmark_distribution(selfcol):
    pre, A, post = P.rollout_samples_emu(16, 18, seed=5, skip=6, **({"self_collision": True} if selfcol else {}))
    ex = P.oracle_step(pre, A, self_collision=selfcol)
    mj = P.oracle_step(pre, A, self_collision=selfcol, linesearch="mujoco")
    assert ((ex["nwarn"] > 0) == (mj["nwarn"] > 0)).all()                  # MuJoCo's bad-state resets on the same samples
    ok = ex["nwarn"] == 0
    assert ok.sum() >= 150
    e = P.rel_err(mj, ex)[ok]
    same = (ex["iters"][ok] == mj["iters"][ok]).mean()
    ls = mj["ls_stats"][ok].sum(axis=0)
    print(f"selfcol={selfcol}: {ok.sum()} samples, max relative difference (qpos, qvel) {e.max(axis=0)}, identical Newton counts on {same:.4f} "
          f"(mean {ex['iters'][ok].mean():.2f}, max {ex['iters'][ok].max()}), {ls[0] / ls[1]:.2f} evaluations per search, {ls[2]} of {ls[1]} searches out of ls_iterations")
    assert (e <= P.TOL_STEP * 1e-3).all(), e.max(axis=0)                  # 1e-8 / 2e-6: three orders inside the stated per-step tolerance
    assert same >= 0.99 and np.abs(ex["iters"][ok] - mj["iters"][ok]).max() <= 2
    assert ls[2] == 0 and ls[0] / ls[1] < 8                               # never out of ls_iterations (50); ~4 evaluations per search


def test_mujoco_linesearch_options_and_stop_rule():
    """ls_iterations caps the evaluations (a cap of 3 makes searches run out and the result drift); a loose ls_tolerance makes the
    search inexact — the Newton iteration then needs more iterations to the same solver tolerance, and still ends within it."""
    pre, A, post = P.rollout_samples_emu(8, 12, seed=2, skip=6)
    ex = P.oracle_step(pre, A)
    ok = ex["nwarn"] == 0
    import helpers
    def run(**kw):
        mc = helpers.model_const()
        kp, kd, tl, sc, of = helpers.pd_tables(mc)
        om = O.OracleModel(helpers.default_xml_str("smpl_humanoid"), kp, kd, tl, sc, of, legal_bodies=helpers.FEET, linesearch="mujoco", **kw)
        its, ls, q = [], np.zeros(3, np.int64), []
        for i in np.flatnonzero(ok)[:40]:
            d = O.OracleData(om)
            d.qpos = pre["qpos_prev"][i]; d.qvel = pre["qvel_prev"][i]; d.forward()
            d.qpos = pre["qpos"][i]; d.qvel = pre["qvel"][i]; d.warm = pre["qacc_warm"][i]
            n = 0
            for _ in range(15):
                d.ctrl = d.spd_torque(A[i]); d.step(); n += d.solver_iter
            its.append(n); ls += d.ls_stats; q.append(d.qvel)
        return np.asarray(its), ls, np.asarray(q)
    base_it, base_ls, base_q = run()
    cap_it, cap_ls, cap_q = run(ls_iterations=3)
    loose_it, loose_ls, loose_q = run(ls_tolerance=1e6)
    assert base_ls[2] == 0 and cap_ls[2] > 0                                # the cap bites
    assert loose_ls[0] / loose_ls[1] < base_ls[0] / base_ls[1]              # fewer evaluations per search ...
    assert loose_it.sum() >= base_it.sum()                                  # ... paid for by Newton iterations
    scale = np.maximum(1.0, np.abs(base_q).max(axis=1))
    assert (np.abs(loose_q - base_q).max(axis=1) / scale).max() < 2e-3      # and the control step still ends within the stated tolerance
