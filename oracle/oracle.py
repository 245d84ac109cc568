"""ctypes wrapper of the CPU oracle (oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; smplsim_amd/ never does.  The wrapper has its own minimal MJCF reader (it does not
use smplsim_amd.mjcf) so that the product's model compiler is checked against an
independent path: XML text -> primitives here -> mass/inertia/invweight0 in C.
"""
import ctypes as C
import os
import subprocess
import xml.etree.ElementTree as ET

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TASK_BASE, TASK_SPEED, TASK_GETUP, TASK_REACH = 0, 1, 2, 3
INIT_DEFAULT, INIT_FALL, INIT_EXTERNAL = 0, 1, 2

# field ids (oracle.h)
M_MASS, M_IPOS, M_IQUAT, M_INERTIA, M_GPOS, M_GQUAT, M_GSIZE, M_BODY_INVW, M_DOF_INVW, M_RANGE, M_MEANINERTIA = range(11)
SOLVER_CONVERGED, SOLVER_MUJOCO = 0, 1
(D_QPOS, D_QVEL, D_QACC, D_WARM, D_CTRL, D_M, D_BIAS, D_XPOS, D_XQUAT, D_LINVEL, D_ANGVEL, D_TOUCH, D_NCON,
 D_CON_POS, D_CON_DIST, D_CON_BODY, D_QACC_SMOOTH, D_NEFC, D_EFC_FORCE, D_SOLVER_ITER, D_ENERGY, D_XIPOS,
 D_QFRC_CONSTRAINT, D_CON_FRAME, D_CON_BODY1, D_NSELF, D_QPOS_FWD, D_QVEL_FWD, D_LS_STATS) = range(29)


class _Desc(C.Structure):
    _fields_ = [
        ("nbody", C.c_int), ("parent", C.c_void_p), ("body_pos", C.c_void_p), ("geom_type", C.c_void_p),
        ("geom_params", C.c_void_p), ("density", C.c_void_p), ("armature", C.c_void_p), ("range_deg", C.c_void_p),
        ("limited", C.c_void_p), ("nu", C.c_int), ("act_dof", C.c_void_p), ("kp", C.c_void_p), ("kd", C.c_void_p),
        ("torque_lim", C.c_void_p), ("act_scale", C.c_void_p), ("act_offset", C.c_void_p),
        ("legal_contact", C.c_void_p), ("timestep", C.c_double), ("gravity", C.c_double),
        ("solref", C.c_double * 2), ("solimp", C.c_double * 5), ("margin", C.c_double), ("mu", C.c_double),
        ("impratio", C.c_double),
        ("self_collision", C.c_int), ("contype", C.c_void_p), ("conaffinity", C.c_void_p), ("nexclude", C.c_int), ("exclude", C.c_void_p),
        ("max_self_contacts", C.c_int),
    ]


class _EnvCfg(C.Structure):
    _fields_ = [
        ("task", C.c_int), ("state_init", C.c_int), ("self_obs_v", C.c_int), ("control_mode", C.c_int),
        ("episode_length", C.c_int), ("control_freq_inv", C.c_int), ("root_height_obs", C.c_int),
        ("power_scale", C.c_double),
        ("tar_speed_min", C.c_double), ("tar_speed_max", C.c_double), ("speed_change_min", C.c_int),
        ("speed_change_max", C.c_int),
        ("tar_height_min", C.c_double), ("tar_height_max", C.c_double), ("height_change_min", C.c_int),
        ("height_change_max", C.c_int), ("recovery_steps", C.c_int), ("tar_dist_max", C.c_double), ("reach_body", C.c_int),
    ]


_NATIVE = False


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def use_native_build():
    """bench.py's cpu_baseline leg: compile the oracle for THIS host (-O3 -march=native) and use that build
    from now on.  Always rebuilt: the file must never travel to a machine with another CPU.  Returns the flags."""
    global _NATIVE, _LIB
    if _LIB is not None and not _NATIVE:
        raise RuntimeError("use_native_build() must come before the first use of the oracle in this process")
    subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "liboracle_native.so"])
    _NATIVE = True
    return subprocess.check_output(["make", "-s", "-C", _HERE, "print-native-flags"], text=True).strip()


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(os.path.join(_HERE, "liboracle_native.so") if _NATIVE else build())
        L.om_model_create.restype = C.c_void_p
        L.om_model_create.argtypes = [C.POINTER(_Desc)]
        L.om_data_create.restype = C.c_void_p
        L.om_data_create.argtypes = [C.c_void_p]
        L.om_env_create.restype = C.c_void_p
        L.om_env_create.argtypes = [C.c_void_p, C.POINTER(_EnvCfg)]
        L.om_env_data.restype = C.c_void_p
        L.om_env_data.argtypes = [C.c_void_p]
        for f in ("om_model_destroy", "om_data_destroy", "om_env_destroy", "om_kinematics", "om_forward", "om_step"):
            getattr(L, f).restype = None
        L.om_model_destroy.argtypes = [C.c_void_p]
        L.om_data_destroy.argtypes = [C.c_void_p]
        L.om_env_destroy.argtypes = [C.c_void_p]
        for f in ("om_kinematics", "om_forward", "om_step"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        L.om_model_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.om_model_set_solver.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
        L.om_model_set_solver.restype = None
        L.om_model_set_linesearch.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
        L.om_model_set_linesearch.restype = None
        L.om_get.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.om_set.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.om_spd_torque.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.om_spd_torque.restype = None
        L.om_ctrl_torque.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.om_ctrl_torque.restype = None
        L.om_set_pid_dt.argtypes = [C.c_void_p, C.c_double]
        L.om_set_pid_dt.restype = None
        L.om_env_obs_size.argtypes = [C.c_void_p]
        L.om_env_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.om_env_reset.restype = None
        L.om_env_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.om_env_step.restype = None
        L.om_env_obs.argtypes = [C.c_void_p, C.c_void_p]
        L.om_env_obs.restype = None
        L.om_quat_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.om_quat_op.restype = None
        L.om_env_get_task.argtypes = [C.c_void_p, C.c_void_p]
        L.om_env_set_task.argtypes = [C.c_void_p, C.c_void_p]
        L.om_obs_v1.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
        L.om_obs_v2.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
        L.om_narrow_phase.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.om_batch_rollout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.om_batch_rollout.restype = C.c_long
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def read_mjcf_primitives(xml):
    """Independent minimal MJCF reader: returns the raw numbers of the model."""
    root = ET.fromstring(xml)
    dflt = root.find("default")
    dj = dict(dflt.find("joint").attrib) if dflt is not None and dflt.find("joint") is not None else {}
    dg = dict(dflt.find("geom").attrib) if dflt is not None and dflt.find("geom") is not None else {}
    P = dict(names=[], parent=[], pos=[], gtype=[], gparams=[], density=[], armature=[0.0] * 6,
             range_deg=[[0.0, 0.0]] * 6, limited=[0] * 6, jnames=[], contype=[], conaffinity=[])
    f = lambda s: [float(x) for x in s.split()]

    def walk(el, par):
        for b in el.findall("body"):
            i = len(P["names"])
            P["names"].append(b.get("name")); P["parent"].append(par); P["pos"].append(f(b.get("pos", "0 0 0")))
            for j in b.findall("joint"):
                a = {**dj, **j.attrib}
                P["armature"].append(float(a.get("armature", 0)))
                P["range_deg"].append(f(a["range"]) if "range" in a else [0.0, 0.0])
                lim = a.get("limited", "auto")
                P["limited"].append(int(lim == "true" or (lim == "auto" and "range" in a)))
                P["jnames"].append(a["name"])
            g = {**dg, **b.find("geom").attrib}
            if g["type"] == "box":
                P["gtype"].append(0)
                P["gparams"].append(f(g.get("pos", "0 0 0")) + f(g["size"]) + f(g.get("quat", "1 0 0 0")))
            elif g["type"] == "capsule":
                P["gtype"].append(1)
                P["gparams"].append(f(g["fromto"]) + [f(g["size"])[0], 0.0, 0.0, 0.0])
            else:                                              # sphere
                P["gtype"].append(2)
                P["gparams"].append(f(g.get("pos", "0 0 0")) + [0.0, 0.0, 0.0] + [f(g["size"])[0], 0.0, 0.0, 0.0])
            P["density"].append(float(g.get("density", 1000)))
            P["contype"].append(int(g.get("contype", 1))); P["conaffinity"].append(int(g.get("conaffinity", 1)))
            walk(b, i)

    walk(root.find("worldbody"), -1)
    P["motors"] = [(m.get("name"), m.get("joint")) for m in root.find("actuator").findall("motor")]
    P["margin"] = float(dg.get("margin", 0))
    con = root.find("contact")
    P["exclude"] = [(P["names"].index(e.get("body1")), P["names"].index(e.get("body2"))) for e in con.findall("exclude")] if con is not None else []
    return P


class OracleModel:
    def __init__(self, xml, kp, kd, torque_lim, act_scale, act_offset, legal_bodies=(), timestep=1.0 / 450, self_collision=False,
                 max_self_contacts=0, solver="mujoco", tolerance=0.0, iterations=0, linesearch="exact", ls_tolerance=0.0, ls_iterations=0):
        """self_collision: body-body contacts per the MJCF's contype / conaffinity / excludes (False = floor only);
        max_self_contacts: keep only the deepest N of them (0 = all);
        solver: "mujoco" = mj_step's own termination of the Newton iteration (opt.tolerance 1e-8 scaled by meaninertia * nv,
        opt.iterations 100; `tolerance` / `iterations` override), "converged" = to the rounding level (parity triage);
        linesearch: "exact" (the kernel's, the default) or "mujoco" = mj_solPrimal's PrimalSearch restated (bracketing + 1-D Newton,
        opt.ls_tolerance 0.01, opt.ls_iterations 50; oracle.h MJ-(V9b))."""
        P = read_mjcf_primitives(xml)
        self.prim = P
        self.nbody = len(P["names"])
        self.nv = 6 + 3 * (self.nbody - 1)
        self.nq = self.nv + 1
        self.nu = len(P["motors"])
        self.body_names = P["names"]
        self._keep = dict(
            parent=np.array(P["parent"], dtype=np.int32), pos=np.array(P["pos"], dtype=np.float64),
            gtype=np.array(P["gtype"], dtype=np.int32), gparams=np.array(P["gparams"], dtype=np.float64),
            density=np.array(P["density"], dtype=np.float64), armature=np.array(P["armature"], dtype=np.float64),
            range_deg=np.array(P["range_deg"], dtype=np.float64), limited=np.array(P["limited"], dtype=np.int32),
            act_dof=np.array([6 + P["jnames"].index(j) for _, j in P["motors"]], dtype=np.int32),
            kp=np.ascontiguousarray(kp, dtype=np.float64), kd=np.ascontiguousarray(kd, dtype=np.float64),
            tl=np.ascontiguousarray(torque_lim, dtype=np.float64), sc=np.ascontiguousarray(act_scale, dtype=np.float64),
            of=np.ascontiguousarray(act_offset, dtype=np.float64),
            legal=np.array([int(n in legal_bodies) for n in P["names"]], dtype=np.int32),
            contype=np.array(P["contype"], dtype=np.int32), conaffinity=np.array(P["conaffinity"], dtype=np.int32),
            exclude=np.array(P["exclude"], dtype=np.int32).reshape(-1, 2),
        )
        k = self._keep
        d = _Desc(nbody=self.nbody, parent=_p(k["parent"]), body_pos=_p(k["pos"]), geom_type=_p(k["gtype"]),
                  geom_params=_p(k["gparams"]), density=_p(k["density"]), armature=_p(k["armature"]),
                  range_deg=_p(k["range_deg"]), limited=_p(k["limited"]), nu=self.nu, act_dof=_p(k["act_dof"]),
                  kp=_p(k["kp"]), kd=_p(k["kd"]), torque_lim=_p(k["tl"]), act_scale=_p(k["sc"]),
                  act_offset=_p(k["of"]), legal_contact=_p(k["legal"]), timestep=timestep, gravity=-9.81,
                  solref=(C.c_double * 2)(0.02, 1.0), solimp=(C.c_double * 5)(0.9, 0.95, 0.001, 0.5, 2.0),
                  margin=P["margin"], mu=1.0, impratio=1.0, self_collision=int(bool(self_collision)), contype=_p(k["contype"]),
                  conaffinity=_p(k["conaffinity"]), nexclude=len(P["exclude"]), exclude=_p(k["exclude"]) if len(P["exclude"]) else None,
                  max_self_contacts=int(max_self_contacts))
        self.h = lib().om_model_create(C.byref(d))
        if not self.h:
            raise RuntimeError("om_model_create failed")
        lib().om_model_set_solver(self.h, {"mujoco": SOLVER_MUJOCO, "converged": SOLVER_CONVERGED}[solver], float(tolerance), int(iterations))
        lib().om_model_set_linesearch(self.h, {"exact": 0, "mujoco": 1}[linesearch], float(ls_tolerance), int(ls_iterations))

    def get(self, field):
        out = np.zeros(16 * 64 + 8 * 200, dtype=np.float64)
        n = lib().om_model_get(self.h, field, _p(out))
        return out[:n].copy()

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            try:
                lib().om_model_destroy(self.h)
            except Exception:
                pass


class OracleData:
    """One mjData-like state; thin attribute access through om_get/om_set."""

    def __init__(self, model, handle=None):
        self.m = model
        self._own = handle is None
        self.h = lib().om_data_create(model.h) if handle is None else handle

    def get(self, field):
        nv = self.m.nv
        out = np.zeros(max(nv * nv, 4096), dtype=np.float64)
        n = lib().om_get(self.m.h, self.h, field, _p(out))
        if n < 0:
            raise KeyError(field)
        return out[:n].copy()

    def set(self, field, val):
        v = np.ascontiguousarray(val, dtype=np.float64)
        if lib().om_set(self.m.h, self.h, field, _p(v)) != 0:
            raise KeyError(field)

    qpos = property(lambda s: s.get(D_QPOS), lambda s, v: s.set(D_QPOS, v))
    qvel = property(lambda s: s.get(D_QVEL), lambda s, v: s.set(D_QVEL, v))
    ctrl = property(lambda s: s.get(D_CTRL), lambda s, v: s.set(D_CTRL, v))
    warm = property(lambda s: s.get(D_WARM), lambda s, v: s.set(D_WARM, v))
    qpos_fwd = property(lambda s: s.get(D_QPOS_FWD))      # the state of the last forward pass: the stale M / bias the
    qvel_fwd = property(lambda s: s.get(D_QVEL_FWD))      # Stable-PD controller reads belong to it
    qacc = property(lambda s: s.get(D_QACC))
    bias = property(lambda s: s.get(D_BIAS), lambda s, v: s.set(D_BIAS, v))
    M = property(lambda s: s.get(D_M).reshape(s.m.nv, s.m.nv), lambda s, v: s.set(D_M, v))
    xpos = property(lambda s: s.get(D_XPOS).reshape(-1, 3))
    xipos = property(lambda s: s.get(D_XIPOS).reshape(-1, 3))
    xquat = property(lambda s: s.get(D_XQUAT).reshape(-1, 4))
    linvel = property(lambda s: s.get(D_LINVEL).reshape(-1, 3))
    angvel = property(lambda s: s.get(D_ANGVEL).reshape(-1, 3))
    touch = property(lambda s: s.get(D_TOUCH))
    ncon = property(lambda s: int(s.get(D_NCON)[0]))
    con_body = property(lambda s: s.get(D_CON_BODY).astype(int))
    con_body1 = property(lambda s: s.get(D_CON_BODY1).astype(int))          # -1 = floor
    con_pos = property(lambda s: s.get(D_CON_POS).reshape(-1, 3))
    con_dist = property(lambda s: s.get(D_CON_DIST))
    con_frame = property(lambda s: s.get(D_CON_FRAME).reshape(-1, 3, 3))
    nself = property(lambda s: int(s.get(D_NSELF)[0]))                      # contacts between two bodies
    solver_iter = property(lambda s: int(s.get(D_SOLVER_ITER)[0]))
    nwarn = property(lambda s: int(s.get(D_SOLVER_ITER)[1]))
    ls_stats = property(lambda s: s.get(D_LS_STATS).astype(np.int64))   # [evaluations, searches, searches out of ls_iterations] (linesearch='mujoco')
     # mj_checkPos/Vel/Acc autoresets so far

    def kinematics(self):
        lib().om_kinematics(self.m.h, self.h)

    def forward(self):
        lib().om_forward(self.m.h, self.h)

    def step(self):
        lib().om_step(self.m.h, self.h)

    def spd_torque(self, action):
        a = np.ascontiguousarray(action, dtype=np.float64)
        tau = np.zeros(self.m.nu)
        lib().om_spd_torque(self.m.h, self.h, _p(a), _p(tau))
        return tau

    def set_pid_dt(self, dt):
        lib().om_set_pid_dt(self.h, float(dt))

    def ctrl_torque(self, action, mode=0, power_scale=1.0):
        a = np.ascontiguousarray(action, dtype=np.float64)
        tau = np.zeros(self.m.nu)
        lib().om_ctrl_torque(self.m.h, self.h, mode, power_scale, _p(a), _p(tau))
        return tau

    def __del__(self):
        if self._own and getattr(self, "h", None):
            try:
                lib().om_data_destroy(self.h)
            except Exception:
                pass


def narrow_phase(kind, g1, g2, margin=0.001):
    """Pair function on raw geometry.  kind: "cc" | "cb" | "bb"; capsule = (centre, axis, radius, half length), box = (centre,
    rotation 3x3 with the box axes as columns, half sizes).  Returns a list of (pos, normal geom1->geom2, dist)."""
    flat = lambda g: np.concatenate([np.ravel(np.asarray(x, dtype=np.float64)) for x in g])
    inp = np.ascontiguousarray(np.concatenate([flat(g1), flat(g2), [margin]]))
    out = np.zeros(1 + 7 * 8)
    n = lib().om_narrow_phase({"cc": 0, "cb": 1, "bb": 2}[kind], _p(inp), _p(out))
    return [(out[1 + 7 * i:4 + 7 * i].copy(), out[4 + 7 * i:7 + 7 * i].copy(), float(out[7 + 7 * i])) for i in range(n)]


def _rand4(u):
    u = np.asarray(u, dtype=np.float64).ravel()
    return np.ascontiguousarray(np.concatenate([u, np.zeros(4)])[:4])


class OracleEnv:
    def __init__(self, model, task=TASK_BASE, state_init=INIT_DEFAULT, self_obs_v=1, control_mode=0,
                 episode_length=300, control_freq_inv=15, root_height_obs=True, power_scale=1.0,
                 tar_speed=(0.0, 5.0), speed_change=(100, 200), tar_height=(0.5, 1.2), height_change=(100, 200),
                 recovery_steps=60, tar_dist_max=1.0, reach_body=0):
        self.m = model
        cfg = _EnvCfg(task, state_init, self_obs_v, control_mode, episode_length, control_freq_inv,
                      int(root_height_obs), power_scale, tar_speed[0], tar_speed[1], speed_change[0],
                      speed_change[1], tar_height[0], tar_height[1], height_change[0], height_change[1],
                      recovery_steps, tar_dist_max, reach_body)
        self.h = lib().om_env_create(model.h, C.byref(cfg))
        self.data = OracleData(model, handle=lib().om_env_data(self.h))
        self.obs_size = lib().om_env_obs_size(self.h)

    def reset(self, fall_actions=None, task_rand=None):
        obs = np.zeros(self.obs_size, dtype=np.float32)
        fa = None if fall_actions is None else np.ascontiguousarray(fall_actions, dtype=np.float64)
        tr = None if task_rand is None else _rand4(task_rand)
        lib().om_env_reset(self.h, None if fa is None else _p(fa), None if tr is None else _p(tr), _p(obs))
        return obs

    def step(self, action, task_rand=None):
        obs = np.zeros(self.obs_size, dtype=np.float32)
        a = np.ascontiguousarray(action, dtype=np.float64)
        tr = None if task_rand is None else _rand4(task_rand)
        rew = C.c_double(); te = C.c_int(); tu = C.c_int()
        lib().om_env_step(self.h, _p(a), None if tr is None else _p(tr), _p(obs), C.byref(rew), C.byref(te), C.byref(tu))
        return obs, rew.value, bool(te.value), bool(tu.value)

    def obs(self):
        obs = np.zeros(self.obs_size, dtype=np.float32)
        lib().om_env_obs(self.h, _p(obs))
        return obs

    def get_task(self):
        o = np.zeros(9)
        lib().om_env_get_task(self.h, _p(o))
        return o

    def set_task(self, v):
        v = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64), np.zeros(9)])[:9], dtype=np.float64)
        lib().om_env_set_task(self.h, _p(v))

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().om_env_destroy(self.h)
            except Exception:
                pass


def batch_rollout(envs, actions, nthreads):
    """actions: [steps, nenv, nu] float64.  Returns env-steps executed."""
    arr = (C.c_void_p * len(envs))(*[e.h for e in envs])
    a = np.ascontiguousarray(actions, dtype=np.float64)
    return lib().om_batch_rollout(arr, len(envs), a.shape[0], _p(a), nthreads)


def obs_v1(qpos, qvel, xpos, xquat, root_height_obs=True):
    nb = xpos.shape[0]
    out = np.zeros((1 if root_height_obs else 0) + 3 * (nb - 1) * 2 + 6 * nb + 6, dtype=np.float32)
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (qpos, qvel, xpos, xquat)]
    lib().om_obs_v1(nb, *[_p(x) for x in a], int(root_height_obs), _p(out))
    return out


def obs_v2(xpos, xquat, linvel, angvel, root_height_obs=True):
    nb = xpos.shape[0]
    out = np.zeros((1 if root_height_obs else 0) + 3 * (nb - 1) + 12 * nb, dtype=np.float32)
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (xpos, xquat, linvel, angvel)]
    lib().om_obs_v2(nb, *[_p(x) for x in a], int(root_height_obs), _p(out))
    return out


def quat_op(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.zeros(4) if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = np.zeros(4)
    lib().om_quat_op(op, _p(a), _p(b), _p(out))
    return out[:3] if op == 1 else out
