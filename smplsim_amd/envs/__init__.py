from .humanoid_env import (HumanoidEnv, HumanoidGetup, HumanoidReach, HumanoidSpeed, HumanoidTask,  # noqa: F401
                           SMPLSimGymVecEnv)
