"""BASELINE config 1 — the reference's single-env gym surface against the oracle, step by step, with every step outside the
stated tolerance triaged (test infrastructure; shared by the GPU test and its emulator twin).

Every control step starts from the oracle's COMPLETE pre-step state: qpos, qvel, the state of the last forward pass (the stale
M / bias the Stable-PD controller reads belong to it: SURVEY 3.2; oracle field qpos_fwd / qvel_fwd) and the solver's warm start.
Rounds 3-4 forced qpos / qvel only and let the stepper keep its own stale state and warm start: one step with a large error then
leaked into the next two or three through those fields (round 4's record: 11 steps of 120 outside, in runs 9-11-12, 65-66, 72-73,
118-119; with the whole state forced the followers are gone and what is left are single, independent samples).

A step outside TOL_QPOS / TOL_QVEL / TOL_OBS is replayed like a sample of test_per_sample_parity (parity_tools.triage): float64
kernel and oracle, both converged (formulation) and at the shipped / MuJoCo's settings (solver rule), 8 input-perturbed float64
twins (the sample's conditioning), and the body-body contact list of the step's last forward pass is compared with the oracle's
(ss_debug_self_contacts).  An outlier is EXPLAINED when the two float64 legs are green and the float32 error is within
TOL_STEP * max(1, cond / COND_REF): its one-step map amplifies a float32 rounding of its input that much.  Anything else fails.
"""
import numpy as np

import parity_tools as P
from helpers import oracle_model
from oracle import oracle as O

TOL_QPOS, TOL_QVEL, TOL_OBS = 1e-5, 2e-3, 2e-3
RESET_SCALE = 1e4                                          # |qvel|_max from which "on its way to MuJoCo's bad-state reset" is an accepted cause
MAX_OUTSIDE = {"one_action": 3, "fresh_actions": 6}        # of 120 steps; measured 1 / 1-3 (profiles/r05_parity_measured.json)


def _pairs(b1, b2):
    return sorted((int(min(a, b)), int(max(a, b))) for a, b in zip(b1, b2))


def run(env, mode, steps=120, to_np=lambda t: np.asarray(t)):
    """env: smpl_sim.envs.tasks.HumanoidEnv of this package (device or emulator backed).  Returns the record dict."""
    from test_parity_f64 import COND_FLOOR, K_ROUND, f32_bound
    oenv = O.OracleEnv(oracle_model(self_collision=True))
    obs, info = env.reset(seed=54)
    assert obs.dtype == np.float32 and obs.shape == (289,) and info["critic_state"] is obs
    assert np.abs(obs - oenv.reset()).max() < 1e-6
    env.action_space.seed(0)
    action = env.action_space.sample()                        # benchmark.py:100 reuses one action for all reps
    vec = env._vec
    records = vec.debug_self_contacts()
    worst, with_self, resets, outside = np.zeros(3), 0, 0, []
    for i in range(steps):
        if mode == "fresh_actions":
            action = env.action_space.sample()
        d = oenv.data
        pre = dict(qpos=d.qpos[None], qvel=d.qvel[None], qpos_prev=d.qpos_fwd[None], qvel_prev=d.qvel_fwd[None], qacc_warm=d.warm[None])
        vec.set_state(pre["qpos"], pre["qvel"], pre["qpos_prev"], pre["qvel_prev"], pre["qacc_warm"])
        nw0, nwo0 = int(vec.nwarn[0]), d.nwarn
        obs, rew, term, trunc, info = env.step(action=action)
        o_ref, r, te, tu = oenv.step(action.astype(np.float64))
        assert isinstance(rew, float) and isinstance(term, bool) and isinstance(trunc, bool)
        assert (term, trunc) == (te, tu)
        d = oenv.data
        bad = int(vec.nwarn[0]) - nw0, d.nwarn - nwo0
        assert (bad[0] > 0) == (bad[1] > 0), (i, bad)          # MuJoCo's bad-state autoreset on the same steps
        if bad[0]:
            resets += 1
            continue
        with_self += d.nself > 0
        q32, v32 = to_np(vec.qpos)[0].astype(np.float64), to_np(vec.qvel)[0].astype(np.float64)
        scale = max(1.0, np.abs(d.qvel).max())
        e = np.array([np.abs(q32 - d.qpos).max(), np.abs(v32 - d.qvel).max(), np.abs(obs - o_ref).max()]) / scale
        if e[0] < TOL_QPOS and e[1] < TOL_QVEL and e[2] < TOL_OBS:
            worst = np.maximum(worst, e)
            continue
        # ---- triage of a step outside the tolerance
        n32 = int(vec.self_contacts[0])
        mine = _pairs(to_np(records)[0, :n32, 0], to_np(records)[0, :n32, 1])
        b1, b2 = d.con_body1, d.con_body
        theirs = _pairs(b1[b1 >= 0], b2[b1 >= 0])
        post32 = dict(qpos=q32[None], qvel=v32[None], nwarn=np.zeros(1, np.int32))
        t = P.triage(pre, action.astype(np.float64)[None], post32, self_collision=True)
        cond = np.maximum(t["cond"][0], COND_FLOOR)
        bound = f32_bound(t["cond"])[0]
        rec = dict(step=i, error=e.tolist(), velocity_scale=float(scale), body_body_contacts=int(d.nself), newton_iters_control_step=int(vec.solver_iters[0]),
                   cond=t["cond"][0].tolist(), precision=t["precision"][0].tolist(), precision_over_bound=(t["precision"][0] / bound).tolist(),
                   formulation=t["formulation"][0].tolist(), solver_rule=t["solver_rule"][0].tolist(),
                   contact_sets_equal=bool(mine == theirs), contacts_kernel=len(mine), contacts_oracle=len(theirs))
        green64 = bool((t["formulation"][0] <= np.maximum(1e-9, K_ROUND * cond * P.EPS64)).all() and P.within_tol(t["solver_rule"])[0])
        within = bool((t["precision"][0] <= bound).all())
        rec["cause"] = ("conditioning: the one-step map amplifies a float32 rounding of the input %.1e / %.1e times (qpos / qvel)" % tuple(t["cond"][0])
                        if green64 and within else "UNEXPLAINED")
        if not rec["contact_sets_equal"]:
            rec["cause"] += "; the body-body contact lists of the step's last forward pass differ (kernel %d, oracle %d: a pair within float32 rounding of the margin)" % (len(mine), len(theirs))
        if t["reset"][0]:
            # excused as a reset only when the state really is at the reset threshold for everybody: the float64 kernel replay resets
            # (t["reset"]) while kernel and oracle agreed on NOT resetting (asserted above) is a state within rounding of the check —
            # accept it only next to a velocity scale that says so, otherwise the step stays what the checks above made of it
            if scale > RESET_SCALE:
                rec["cause"] = ("a float64 replay takes MuJoCo's bad-state reset inside the step (state blowing up: velocity scale %.1e); was: " % scale) + rec["cause"]
            else:
                rec["cause"] = "UNEXPLAINED: a float64 replay resets at velocity scale %.1e while kernel and oracle do not; " % scale + rec["cause"]
        outside.append(rec)
        print("outside tolerance:", rec)
    return dict(steps=steps, qpos=worst[0], qvel=worst[1], obs=worst[2], steps_with_body_body_contact=int(with_self), bad_state_resets=resets,
                outside=outside)


def check(rec, mode):
    assert rec["steps_with_body_body_contact"] >= 30 and rec["bad_state_resets"] <= 12, rec
    for o in rec["outside"]:
        assert not o["cause"].startswith("UNEXPLAINED"), o
    assert len(rec["outside"]) <= MAX_OUTSIDE[mode], rec["outside"]
