from smplsim_amd.envs import HumanoidEnv, HumanoidGetup, HumanoidReach, HumanoidSpeed  # noqa: F401
