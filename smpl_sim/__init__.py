"""Import-path shim: `smpl_sim.envs` resolves to the HIP-backed envs of smplsim_amd.

The reference's callers import `smpl_sim.envs.tasks.HumanoidEnv` etc. (reference
examples/benchmark.py:72, smpl_sim/agents/agent_humanoid.py:92); with this repository on the
path those imports get the MI355X stepper, everything else of the reference is out of scope.
"""
# Let the rest of the reference resolve behind this shim: with this repository BEFORE a SMPLSim checkout on sys.path,
# `smpl_sim.envs` / `smpl_sim.smpllib.motion_lib_*` are the MI355X versions and every other submodule (`smpl_sim.learning`,
# `smpl_sim.agents`, `smpl_sim.utils`, ...) is found in the checkout — the reference's PPO loop then runs on this stepper unchanged.
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
__version__ = "0.0.1"
