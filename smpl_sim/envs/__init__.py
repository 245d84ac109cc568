from smplsim_amd.envs import HumanoidEnv, HumanoidGetup, HumanoidSpeed, HumanoidTask  # noqa: F401
