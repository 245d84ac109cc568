from .humanoid_env import HumanoidEnv, HumanoidGetup, HumanoidSpeed, HumanoidTask, SMPLSimGymVecEnv  # noqa: F401
