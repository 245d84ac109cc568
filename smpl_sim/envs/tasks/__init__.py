from smplsim_amd.envs import HumanoidEnv, HumanoidGetup, HumanoidSpeed  # noqa: F401
