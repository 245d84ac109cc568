"""Attribute-style config objects (OmegaConf/hydra are not dependencies of this package).

`HumanoidEnv(cfg)` accepts any object with the reference's attribute layout (`cfg.env.*`,
`cfg.robot.*`, `cfg.headless`, `.get(key, default)`; reference smpl_sim/data/cfg/**), e.g. an
OmegaConf DictConfig, or the `AttrDict` below built from a plain dict / YAML text.
"""
import copy


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in {**(d or {}), **kw}.items():
            self[k] = AttrDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


# reference smpl_sim/data/cfg/env/{base_env,speed,getup}.yaml + robot/smpl_humanoid.yaml
_ENV_COMMON = dict(episode_length=300, sim_timestep_inv=450, control_frequency_inv=15, power_scale=10,
                   root_height_obs=True, enable_early_termination=True, self_obs_v=1, kp_scale=1.0, kd_scale=1.0,
                   cycle_motion=False, power_reward=True, clip_actions=True, control_mode="uhc_pd", render_mode=None,
                   camera="side", state_init="Default", pdp_scale=1, pdd_scale=1, pdi_scale=1)
ENV_CFGS = {
    "HumanoidEnv": dict(_ENV_COMMON, task="HumanoidEnv", contact_bodies=[]),
    "HumanoidSpeed": dict(_ENV_COMMON, task="HumanoidSpeed", tar_speed_min=0.0, tar_speed_max=5.0,
                          speed_change_steps_min=100, speed_change_steps_max=200,
                          contact_bodies=["R_Ankle", "L_Ankle", "R_Toe", "L_Toe"]),
    "HumanoidGetup": dict(_ENV_COMMON, task="HumanoidGetup", state_init="Fall", recovery_steps=60, tar_height_min=0.5,
                          tar_height_max=1.2, height_change_steps_min=100, height_change_steps_max=200,
                          contact_bodies=["R_Ankle", "L_Ankle", "R_Toe", "L_Toe"]),
    "HumanoidReach": dict(_ENV_COMMON, task="HumanoidReach", contact_bodies=["R_Ankle", "L_Ankle", "R_Toe", "L_Toe"],
                          reach_body_name="R_Hand", tar_dist_max=1, tar_height_min=0.2, tar_height_max=2.0,
                          tar_change_steps_min=50, tar_change_steps_max=100),
}
ROBOT_CFG = dict(humanoid_type="smpl", has_upright_start=False, has_shape_obs=False, has_weight_obs=False,
                 has_shape_variation=False, has_mesh=False, replace_feet=True, has_jt_limit=False,
                 height_fix_mode="full", big_ankle=True, remove_toe=False, real_weight_porpotion_capsules=True,
                 real_weight_porpotion_boxes=True, real_weight=True, box_body=True, smpl_data_dir="data/smpl",
                 create_vel_sensors=False)


def default_cfg(task="HumanoidEnv", **env_overrides):
    env = copy.deepcopy(ENV_CFGS[task])
    env.update(env_overrides)
    return AttrDict(env=env, robot=copy.deepcopy(ROBOT_CFG), headless=True)
