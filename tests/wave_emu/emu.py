"""numpy front-end to the wavefront-emulator build of the kernel (tests/wave_emu/libss_emu.so).

UNIT-TEST INFRASTRUCTURE ONLY: it runs the kernel source on the CPU (64 fibers = 64 lanes) so the
float32 kernel logic can be checked against the oracle in the GPU-less build container.  It binds
the same C ABI as the product library, with host (numpy) buffers instead of device pointers.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

from smplsim_amd import _cabi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
_LOCK = threading.Lock()              # parity_tools steps batches from a thread pool: one `make`, one load


def lib(f64=False):
    """f64=True: the float64 instantiation of the same kernel source (-DSS_F64, Newton run to convergence): every array of
    the C ABI declared float* is then a float64 array."""
    name = "libss_emu64.so" if f64 else "libss_emu.so"
    with _LOCK:
        if name not in _LIBS:
            subprocess.check_call(["make", "-s", "-C", _HERE, name])
            _LIBS[name] = _cabi.bind(C.CDLL(os.path.join(_HERE, name)))
    return _LIBS[name]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _rand4(tr, n, ft=np.float32):
    if tr is None:
        return None
    t = np.zeros((n, 4), ft)
    a = np.asarray(tr, ft).reshape(n, -1)
    t[:, :a.shape[1]] = a
    return t


class EmuBatch:
    def __init__(self, mc, tables, num_envs, legal_bodies=(), timestep=1.0 / 450, shape_mcs=None, shape_id=None, f64=False, mjcf_text=None, mjcf_options=None, **cfg):
        """shape_mcs: list of ModelConst = body shapes of one humanoid (ss_model_create_shapes), shape_id [N] picks per env.
        f64: run the float64 instantiation of the kernel (state / action / observation arrays are float64 then)."""
        L = self.L = lib(f64)
        ft = self.ft = np.float64 if f64 else np.float32
        self.mc = mc
        self.model = C.c_void_p()
        self.shape_id = None
        if mjcf_text is not None:      # the library's own MJCF compiler + gain tables (mc is used for sizes only)
            txt = mjcf_text.encode()
            self._chk(L.ss_model_create_from_mjcf(txt, len(txt), None if mjcf_options is None else C.byref(mjcf_options), 0, C.byref(self.model)))
        elif shape_mcs is None:
            desc, self._keep = _cabi.make_model_desc(mc, *tables, legal_bodies=legal_bodies, timestep=timestep)
            self._chk(L.ss_model_create(C.byref(desc), 0, C.byref(self.model)))
        else:
            descs, self._keep = (_cabi.ModelDesc * len(shape_mcs))(), []
            for i, m in enumerate(shape_mcs):
                descs[i], keep = _cabi.make_model_desc(m, *tables, legal_bodies=legal_bodies, timestep=timestep)
                self._keep.append(keep)
            self._chk(L.ss_model_create_shapes(descs, len(shape_mcs), 0, C.byref(self.model)))
            self.shape_id = None if shape_id is None else np.ascontiguousarray(shape_id, np.int32)
        self.cfg = _cabi.make_env_cfg(**cfg)
        N, nq, nv, nb = num_envs, mc.nq, mc.nv, mc.nbody
        self.N = N
        self.qpos = np.zeros((N, nq), ft); self.qvel = np.zeros((N, nv), ft)
        self.qpos_prev = np.zeros((N, nq), ft); self.qvel_prev = np.zeros((N, nv), ft)
        self.qacc_warm = np.zeros((N, nv), ft); self.body_vel = np.zeros((N, nb, 6), ft)
        self.touch = np.zeros((N, 2), np.int32); self.cur_t = np.zeros(N, np.int32)
        self.task = np.zeros((N, 4), ft); self.nwarn = np.zeros(N, np.int32)
        self.solver_iters = np.zeros(N, np.int32)
        self.pid_integral = np.zeros((N, mc.nu), ft); self.pid_last_error = np.zeros((N, mc.nu), ft)
        self.pid_started = np.zeros(N, np.int32)
        self.self_contacts = np.zeros(N, np.int32)
        self.qpos[:, 3] = 1; self.qpos_prev[:, 3] = 1
        st = _cabi.State(N, *[_p(x) for x in (self.qpos, self.qvel, self.qpos_prev, self.qvel_prev, self.qacc_warm,
                                             self.body_vel, self.touch, self.cur_t, self.task, self.nwarn,
                                             self.solver_iters, self.pid_integral, self.pid_last_error,
                                             self.pid_started, self.shape_id, self.self_contacts)])
        self.batch = C.c_void_p()
        self._chk(L.ss_batch_create(self.model, C.byref(self.cfg), C.byref(st), C.byref(self.batch)))
        self.obs_size = L.ss_obs_size(self.model, C.byref(self.cfg))
        self.obs = np.zeros((N, self.obs_size), ft)
        self.reward = np.zeros(N, ft)
        self.terminated = np.zeros(N, np.uint8); self.truncated = np.zeros(N, np.uint8)

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(f"ss error {rc}: {self.L.ss_last_error().decode()}")

    def set_state(self, qpos, qvel, qpos_prev=None, qvel_prev=None, warm=None):
        self.qpos[:] = qpos; self.qvel[:] = qvel
        self.qpos_prev[:] = qpos if qpos_prev is None else qpos_prev
        self.qvel_prev[:] = qvel if qvel_prev is None else qvel_prev
        if warm is not None:
            self.qacc_warm[:] = warm

    def reset(self, mask=None, fall_actions=None, task_rand=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        fa = None if fall_actions is None else np.ascontiguousarray(fall_actions, self.ft)
        tr = _rand4(task_rand, self.N, self.ft)
        self._chk(self.L.ss_reset(self.batch, _p(m), _p(fa), _p(tr), _p(self.obs), None))
        return self.obs.copy()

    def step(self, actions, task_rand=None):
        a = np.ascontiguousarray(actions, self.ft)
        tr = _rand4(task_rand, self.N, self.ft)
        self._chk(self.L.ss_step(self.batch, _p(a), _p(tr), _p(self.obs), _p(self.reward), _p(self.terminated),
                                _p(self.truncated), None))
        return self.obs.copy(), self.reward.copy(), self.terminated.copy().astype(bool), self.truncated.copy().astype(bool)

    def step_autoreset(self, actions, task_rand=None, reset_task_rand=None):
        a = np.ascontiguousarray(actions, self.ft)
        tr, tr2 = _rand4(task_rand, self.N, self.ft), _rand4(reset_task_rand, self.N, self.ft)
        self.obs_next = np.zeros_like(self.obs)
        self._chk(self.L.ss_step_autoreset(self.batch, _p(a), _p(tr), _p(tr2), _p(self.obs), _p(self.obs_next), _p(self.reward),
                                          _p(self.terminated), _p(self.truncated), None))
        return self.obs.copy(), self.obs_next.copy(), self.reward.copy(), self.terminated.copy().astype(bool), self.truncated.copy().astype(bool)

    def substep(self, actions, n):
        a = np.ascontiguousarray(actions, self.ft)
        self._chk(self.L.ss_substep(self.batch, _p(a), n, None))

    def set_body_outputs(self):
        self.xpos_out = np.zeros((self.N, self.mc.nbody, 3), self.ft); self.xmat_out = np.zeros((self.N, self.mc.nbody, 9), self.ft)
        self._chk(self.L.ss_set_body_outputs(self.batch, _p(self.xpos_out), _p(self.xmat_out)))

    def set_power_output(self):
        self.power = np.zeros((self.N, int(self.cfg.control_freq_inv), self.mc.nv - 6), self.ft)
        self._chk(self.L.ss_set_power_output(self.batch, _p(self.power)))
        return self.power

    def kinematics(self):
        xpos = np.zeros((self.N, self.mc.nbody, 3), self.ft); xmat = np.zeros((self.N, self.mc.nbody, 9), self.ft)
        self._chk(self.L.ss_kinematics(self.batch, _p(xpos), _p(xmat), None))
        return xpos, xmat

    def debug_self_contacts(self):
        """Turn on the record dump of the body-body contacts: self.self_records [N, SS_MAX_SELF_CONTACTS = 64, 24] after every
        launch (one record per lane of the env's wavefront; the first self.self_contacts[n] are valid)."""
        from smplsim_amd import _cabi
        self.self_records = np.zeros((self.N, _cabi.SS_MAX_SELF_CONTACTS, 24), self.ft)
        self._chk(self.L.ss_debug_self_contacts(self.batch, _p(self.self_records)))
        return self.self_records

    def debug_forward(self, torques=None):
        nv = self.mc.nv
        M = np.zeros((self.N, nv, nv), self.ft)
        bias = np.zeros((self.N, nv), self.ft); qacc = np.zeros((self.N, nv), self.ft)
        tq = None if torques is None else np.ascontiguousarray(torques, self.ft)
        self._chk(self.L.ss_debug_forward(self.batch, _p(tq), _p(M), _p(bias), _p(qacc), None))
        return M, bias, qacc
