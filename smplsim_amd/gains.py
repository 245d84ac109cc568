"""PD gains, torque limits and action scaling of the humanoid env.

Restates reference smpl_sim/envs/humanoid_env.py:62-84 (`GAINS["stablepd"]`: per-body
[kp, kd, _, torque_lim]) and `build_pd_action_scale` (:325-370).  The SMPL-X/H layout
has finger bodies that the reference's `stablepd` table lacks; they take the finger
rows of `GAINS_PHC` (reference smpl_sim/smpllib/skeleton_local.py:133-162: kp 100,
kd 10, limit 150), as SURVEY.md §8d config 4 prescribes.
"""
import numpy as np

# body -> (kp, kd, torque_lim); rows written once per gain class.
_CLASSES = {
    (800.0, 80.0, 1000.0): ["L_Hip", "L_Knee", "L_Ankle", "R_Hip", "R_Knee", "R_Ankle"],
    (500.0, 50.0, 500.0): ["L_Toe", "R_Toe"],
    (1000.0, 100.0, 500.0): ["Torso", "Spine", "Chest"],
    (500.0, 50.0, 250.0): ["Neck", "Head", "L_Elbow", "R_Elbow"],
    (500.0, 50.0, 1000.0): ["L_Thorax", "L_Shoulder", "R_Thorax", "R_Shoulder"],
    (300.0, 30.0, 250.0): ["L_Wrist", "L_Hand", "R_Wrist", "R_Hand"],
}
STABLEPD = {body: g for g, bodies in _CLASSES.items() for body in bodies}
_FINGER = (100.0, 10.0, 150.0)
_FINGER_PREFIXES = ("Index", "Middle", "Pinky", "Ring", "Thumb")


def body_gain(body: str):
    if body in STABLEPD:
        return STABLEPD[body]
    if "_" in body and body.split("_", 1)[1].rstrip("123") in _FINGER_PREFIXES:
        return _FINGER
    raise KeyError(f"no PD gain entry for body {body!r}")


def build_pd_tables(actuator_names, jnt_range_of, clip_actions=True, control_mode="uhc_pd",
                    pdp_scale=1.0, pdd_scale=1.0):
    """Per-actuator kp, kd, torque limit, action scale and offset.

    actuator_names: motor names (== joint names, `<body>_<x|y|z>`).
    jnt_range_of:   callable name -> (low, high) in radians.
    Mirrors humanoid_env.py:325-370 (scale = min(1.2*max|range|, pi), offset 0 when
    clip_actions; identity scaling otherwise) and setup_controller :312-323 (gains divided
    by pdp_scale / pdd_scale; `simple_pid` gets jkp/10, jkd/10; `torque` and `default` leave gains and
    torque limits at zero, exactly like the reference).
    """
    n = len(actuator_names)
    kp, kd, lim = np.zeros(n), np.zeros(n), np.zeros(n)
    lo, hi = np.zeros(n), np.zeros(n)
    for i, name in enumerate(actuator_names):
        low, high = jnt_range_of(name)
        s = min(1.2 * max(abs(low), abs(high)), np.pi)
        lo[i], hi[i] = -s, s
        if control_mode in ("pd", "uhc_pd", "simple_pid"):
            body = "_".join(name.split("_")[:-1])
            kp[i], kd[i], lim[i] = body_gain(body)
    if clip_actions:
        scale, offset = 0.5 * (hi - lo), 0.5 * (hi + lo)
    else:
        scale, offset = np.ones(n), np.zeros(n)
    if control_mode == "simple_pid":
        # setup_controller :318-319: SimplePID(jkp / 10, ones, jkd / 10, ...) — not divided by pdp/pdd_scale; ki = 1
        return kp / 10.0, kd / 10.0, lim, scale, offset
    return kp / pdp_scale, kd / pdd_scale, lim, scale, offset
