from smplsim_amd.envs.humanoid_env import HumanoidEnv  # noqa: F401
from smplsim_amd.gains import STABLEPD as _S

GAINS = {"stablepd": {k: [v[0], v[1], 1, v[2]] for k, v in _S.items()}}
