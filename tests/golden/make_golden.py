#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python code on seeded inputs.

Run in the build container only (needs /root/reference; the GPU box has no copy):
    python tests/golden/make_golden.py
It writes tests/golden/reference_vectors.npz, which is committed.  The reference imports
`mujoco`, `gymnasium`, `lxml`, ... at module level; none is installed here, so they are
stubbed just far enough for the import to succeed — every function *called* below is the
reference's unmodified NumPy/SciPy code:

  * StablePDController.control                  smpl_sim/envs/controllers.py:116-190
      (mujoco.mj_fullM is stubbed to hand over a seeded dense SPD matrix)
  * compute_humanoid_self_obs_v1 / _v2          smpl_sim/envs/humanoid_env.py:565-687
  * HumanoidEnv.build_pd_action_scale           smpl_sim/envs/humanoid_env.py:325-370
  * forward_reward / compute_speed_observations smpl_sim/envs/tasks/humanoid_speed.py:9-46
  * height_reward                               smpl_sim/envs/tasks/humanoid_getup.py:9-18
  * quaternion helpers                          smpl_sim/utils/np_transform_utils.py

MuJoCo's mj_step itself cannot be run here (no wheel, no source) — the physics stays
"parity unpinned" (see oracle/oracle.h).
"""
import os
import sys
import types
from unittest import mock

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    gym = types.ModuleType("gymnasium")

    class Env:  # minimal base for `class BaseEnv(gym.Env)`
        def reset(self, seed=None, options=None):
            pass

    gym.Env = Env
    gym.spaces = mock.MagicMock()
    sys.modules["gymnasium"] = gym
    mj = types.ModuleType("mujoco")
    mj.MjModel = object
    mj.MjData = object
    mj.viewer = mock.MagicMock()

    def mj_fullM(model, dst, qM):          # hand the seeded dense matrix over
        dst[:] = model._dense_M

    mj.mj_fullM = mj_fullM
    sys.modules["mujoco"] = mj
    sys.modules["mujoco.viewer"] = mj.viewer
    for name in ("imageio", "lxml", "lxml.etree", "smplx", "stl", "stl.mesh", "mujoco_py", "joblib", "easydict",
                 "open3d", "cv2", "tqdm", "numpy_stl", "scipy.spatial.qhull", "trimesh", "wandb"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = mock.MagicMock()


class _AutoMock:
    """meta-path finder: any module that is not installed imports as a MagicMock."""

    def find_spec(self, name, path=None, target=None):
        import importlib.machinery
        if name.split(".")[0] in ("smpl_sim", "numpy", "scipy", "torch"):
            return None
        return importlib.machinery.ModuleSpec(name, self)

    def create_module(self, spec):
        m = mock.MagicMock()
        m.__path__ = []
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


def _shell_packages():
    """Create the reference's packages WITHOUT running their __init__.py (which import the
    dm_control-era env and other dead code); submodules then import normally by path."""
    for pkg in ("smpl_sim", "smpl_sim.envs", "smpl_sim.envs.tasks", "smpl_sim.utils", "smpl_sim.smpllib"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, *pkg.split("."))]
        sys.modules[pkg] = m


def main():
    sys.dont_write_bytecode = True          # never write __pycache__ into /root/reference
    sys.path.insert(0, REF)
    _stub_modules()
    _shell_packages()
    sys.meta_path.append(_AutoMock())
    import smpl_sim.envs.controllers as ctrls
    import smpl_sim.utils.np_transform_utils as npt
    try:
        import smpl_sim.envs.humanoid_env as henv
    except Exception as e:  # heavy optional deps of smpl_local_robot: stub and retry
        for name in ("smpl_sim.smpllib.smpl_local_robot", "smpl_sim.smpllib.smpl_xml_addons",
                     "smpl_sim.smpllib.motion_lib_base"):
            sys.modules[name] = mock.MagicMock()
        import smpl_sim.envs.humanoid_env as henv
    import smpl_sim.envs.tasks.humanoid_speed as hspeed
    import smpl_sim.envs.tasks.humanoid_getup as hgetup

    rs = np.random.default_rng(20240925)
    out = {}
    nb, nv, nu = 24, 75, 69

    # ---- actuator tables (needs the fixture's actuator names and ranges)
    import xml.etree.ElementTree as ET
    root = ET.parse(os.path.join(REF, "smpl_sim/data/assets/mjcf/smpl_humanoid.xml")).getroot()
    rng = {j.get("name"): np.deg2rad([float(x) for x in j.get("range").split()]) for j in root.iter("joint") if j.get("range")}
    act_names = [m.get("name") for m in root.find("actuator").findall("motor")]
    fake = types.SimpleNamespace(dof_size=nu, actuator_names=act_names, control_mode="uhc_pd", clip_actions=True,
                                 mj_model=types.SimpleNamespace(joint=lambda n: types.SimpleNamespace(range=rng[n])))
    henv.HumanoidEnv.build_pd_action_scale(fake)
    out["act_kp"], out["act_kd"], out["act_lim"] = fake.jkp, fake.jkd, fake.torque_lim
    out["act_scale"], out["act_offset"] = fake._pd_action_scale, fake._pd_action_offset
    out["act_names"] = np.array(act_names)

    # ---- Stable PD
    ncase = 6
    spd = dict(M=[], C=[], qpos=[], qvel=[], action=[], torque=[])
    ctrl = ctrls.StablePDController(fake._pd_action_scale, fake._pd_action_offset, nv, fake.torque_lim, fake.jkp, fake.jkd)
    for c in range(ncase):
        A = rs.normal(size=(nv, nv)) * 0.3
        M = A @ A.T / nv + np.diag(rs.uniform(0.02, 3.0, nv))
        C = rs.normal(size=nv) * (30.0 if c < 4 else 300.0)
        qpos = np.concatenate([rs.normal(size=3), rs.normal(size=4), rs.uniform(-2, 2, nv - 6)])
        qvel = rs.normal(size=nv) * (2.0 if c < 3 else 20.0)
        action = rs.uniform(-1, 1, nu)
        model = types.SimpleNamespace(opt=types.SimpleNamespace(timestep=1.0 / 450), nv=nv, _dense_M=M.copy())
        data = types.SimpleNamespace(qpos=qpos.copy(), qvel=qvel.copy(), qM=None, qfrc_bias=C.copy())
        tq = ctrl.control(action, model, data)
        for k, v in zip(spd, (M, C, qpos, qvel, action, tq)):
            spd[k].append(v)
    for k, v in spd.items():
        out["spd_" + k] = np.array(v)

    # ---- observations
    ob = dict(qpos=[], qvel=[], xpos=[], xquat=[], linvel=[], angvel=[], v1=[], v2=[])
    for c in range(8):
        qpos = np.concatenate([rs.normal(size=3), rs.normal(size=4), rs.uniform(-2, 2, nv - 6)])
        qvel = rs.normal(size=nv) * 3
        xpos = rs.normal(size=(nb, 3)); xpos[:, 2] = np.abs(xpos[:, 2])
        xquat = rs.normal(size=(nb, 4)); xquat /= np.linalg.norm(xquat, axis=1, keepdims=True)
        if c == 0:
            xquat[0] = [0.5, 0.5, 0.5, 0.5]
        linvel, angvel = rs.normal(size=(nb, 3)) * 2, rs.normal(size=(nb, 3)) * 4
        o1 = henv.compute_humanoid_self_obs_v1(qpos[None], qvel[None], xpos[None], xquat[None], False, True, "smpl")
        o2 = henv.compute_humanoid_self_obs_v2(xpos[None], xquat[None], linvel[None], angvel[None], False, True, "smpl")
        v1 = np.concatenate([v.ravel() for v in o1.values()], axis=0, dtype=np.float32)
        v2 = np.concatenate([v.ravel() for v in o2.values()], axis=0, dtype=np.float32)
        for k, v in zip(ob, (qpos, qvel, xpos, xquat, linvel, angvel, v1, v2)):
            ob[k].append(v)
    for k, v in ob.items():
        out["obs_" + k] = np.array(v)

    # ---- task functions
    tar_speed = rs.uniform(0, 5, 6)
    root_pos, prev = rs.normal(size=(6, 1, 3)), rs.normal(size=(6, 1, 3))
    out["spd_rew_tar"] = tar_speed; out["spd_rew_pos"] = root_pos; out["spd_rew_prev"] = prev
    out["spd_rew"] = np.array([hspeed.forward_reward(tar_speed[i], root_pos[i], prev[i], 15.0 / 450)[0] for i in range(6)])
    rq = rs.normal(size=(6, 4)); rq /= np.linalg.norm(rq, axis=1, keepdims=True)
    so = [hspeed.compute_speed_observations(rq[i:i + 1], tar_speed[i], False, "smpl") for i in range(6)]
    out["spd_obs_quat"] = rq
    out["spd_obs"] = np.array([np.concatenate([v.ravel() for v in o.values()]) for o in so])
    th = rs.uniform(0.5, 1.2, (6, 1, 1))
    out["getup_tar"] = th; out["getup_pos"] = root_pos
    out["getup_rew"] = np.array([hgetup.height_reward(th[i], root_pos[i])[0] for i in range(6)])

    # ---- quaternion helpers
    qa = rs.normal(size=(16, 4)); qa /= np.linalg.norm(qa, axis=1, keepdims=True)
    qb = rs.normal(size=(16, 4)); qb /= np.linalg.norm(qb, axis=1, keepdims=True)
    v = rs.normal(size=(16, 3))
    out["q_a"], out["q_b"], out["q_v"] = qa, qb, v
    out["q_mul"] = npt.quat_mul(qa, qb)
    out["q_rot"] = npt.quat_rotate(qa, v)
    out["q_head_inv"] = npt.calc_heading_quat_inv(qa)
    out["q_tan_norm"] = npt.quat_to_tan_norm(qa)
    out["q_remove_base"] = np.array([npt.remove_base_rot(qa[i:i + 1])[0] for i in range(16)])

    path = os.path.join(HERE, "reference_vectors.npz")
    # ---- the other controllers selectable by control_mode (humanoid_env.py:312-323), separate RNG stream so that the
    # vectors above stay as they were: PIDController (`pd`, zero integral gain), SimpleTorqueController (`torque`, with
    # the stablepd limits instead of the zeros the env leaves there) and the stateful SimplePID (`simple_pid`)
    rs2 = np.random.default_rng(777)
    pdc = ctrls.PIDController(fake._pd_action_scale, fake._pd_action_offset, fake.torque_lim, fake.jkp, fake.jkd, np.zeros_like(fake.jkd))
    tqc = ctrls.SimpleTorqueController(0.7 * fake.torque_lim, fake.torque_lim)
    pid = ctrls.SimplePID(fake.jkp / 10, np.ones_like(fake.jkp), fake.jkd / 10, (1.0 / 450) * 15, fake.torque_lim,
                          fake._pd_action_scale, fake._pd_action_offset)
    cq, cv, ca, cpd, ctq, cpid = [], [], [], [], [], []
    q = rs2.uniform(-0.5, 0.5, nu)
    a = rs2.uniform(-1, 1, nu)
    model = types.SimpleNamespace(opt=types.SimpleNamespace(timestep=1.0 / 450))
    for t in range(16):
        if t % 5 == 4:
            a = rs2.uniform(-1, 1, nu)
        qd = rs2.normal(size=nu) * 2.0
        data = types.SimpleNamespace(qpos=np.concatenate([np.zeros(7), q]), qvel=np.concatenate([np.zeros(6), qd]))
        cq.append(q.copy()); cv.append(qd.copy()); ca.append(a.copy())
        cpd.append(pdc.control(a, model, data)); ctq.append(tqc.control(a, model, data)); cpid.append(pid.control(a, model, data))
        q = q + rs2.normal(size=nu) * 0.05
    out["ctl_q"], out["ctl_qd"], out["ctl_action"] = np.array(cq), np.array(cv), np.array(ca)
    out["ctl_pd"], out["ctl_torque"], out["ctl_simple_pid"] = np.array(cpd), np.array(ctq), np.array(cpid)
    out["ctl_power_scale"] = np.array(0.7)

    # ---- learning side: GAE, Gaussian policy / value nets, running norm, PPO surrogate, action rescale
    import torch
    import smpl_sim.learning.learning_utils as lu
    from smpl_sim.learning.policy_gaussian import PolicyGaussian
    from smpl_sim.learning.critic import Value
    from smpl_sim.learning.mlp import MLP
    from smpl_sim.agents.agent_ppo import AgentPPO
    rs3 = np.random.default_rng(4242)
    torch.manual_seed(4242)
    nS = 300
    rew = torch.tensor(rs3.normal(size=(nS, 1)), dtype=torch.float64)
    val = torch.tensor(rs3.normal(size=(nS, 1)), dtype=torch.float64)
    done = rs3.uniform(size=(nS, 1)) < 0.06
    dead = done & (rs3.uniform(size=(nS, 1)) < 0.5)
    done[-1] = True
    nd, ndead = torch.tensor(1.0 - done), torch.tensor(1.0 - dead)
    adv, ret = lu.estimate_advantages(rew.clone(), nd.clone(), ndead.clone(), val.clone(), 0.99, 0.95)
    out["gae_rewards"], out["gae_values"], out["gae_not_done"], out["gae_not_dead"] = rew.numpy(), val.numpy(), nd.numpy(), ndead.numpy()
    out["gae_adv"], out["gae_ret"] = adv.numpy(), ret.numpy()
    lcfg = types.SimpleNamespace(learning=types.SimpleNamespace(mlp=types.SimpleNamespace(units=[16, 12], activation="silu"),
                                                                fix_std=False, log_std=-1.0))
    pol = PolicyGaussian(lcfg, action_dim=5, state_dim=11).double()
    vnet = Value(MLP(11, [16, 12], "silu")).double()
    xs = [torch.tensor(rs3.normal(size=(40, 11)) * (1 + i) + i, dtype=torch.float64) for i in range(3)]
    pol.train()
    for x in xs:                                              # three training-mode passes update the running norm
        pol(x)
    out["rn_inputs"] = np.stack([x.numpy() for x in xs])
    out["rn_n"], out["rn_mean"], out["rn_var"], out["rn_std"] = (pol.norm.n.numpy(), pol.norm.mean.numpy(), pol.norm.var.numpy(), pol.norm.std.numpy())
    pol.eval()
    xq = torch.tensor(rs3.normal(size=(7, 11)) * 4, dtype=torch.float64)
    aq = torch.tensor(rs3.normal(size=(7, 5)), dtype=torch.float64)
    with torch.no_grad():
        dist = pol(xq)
        out["pol_x"], out["pol_a"] = xq.numpy(), aq.numpy()
        out["pol_mean"], out["pol_logp"] = dist.loc.numpy(), pol.get_log_prob(xq, aq).numpy()
        out["pol_norm_out"] = pol.norm(xq).numpy()
        out["val_out"] = vnet(xq).numpy()
    for k, v in pol.state_dict().items():
        out["polsd_" + k] = v.numpy()
    for k, v in vnet.state_dict().items():
        out["valsd_" + k] = v.numpy()
    fixed = torch.tensor(out["pol_logp"]) + torch.tensor(rs3.normal(size=(7, 1)) * 0.3)
    advq = torch.tensor(rs3.normal(size=(7, 1)), dtype=torch.float64)
    fake_agent = types.SimpleNamespace(policy_net=pol, clip_epsilon=0.2)
    with torch.no_grad():
        out["ppo_loss"] = AgentPPO.ppo_loss(fake_agent, xq, aq, advq, fixed, torch.arange(7)).numpy()
    out["ppo_fixed"], out["ppo_adv"] = fixed.numpy(), advq.numpy()
    lo, hi, act = rs3.uniform(-2, -1, 5), rs3.uniform(1, 3, 5), rs3.uniform(-1, 1, (4, 5))
    out["resc_low"], out["resc_high"], out["resc_in"], out["resc_out"] = lo, hi, act, lu.rescale_actions(lo, hi, act)

    np.savez_compressed(path, **out)
    print("wrote", path, {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
