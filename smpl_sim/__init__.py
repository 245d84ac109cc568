"""Import-path shim: `smpl_sim.envs` resolves to the HIP-backed envs of smplsim_amd.

The reference's callers import `smpl_sim.envs.tasks.HumanoidEnv` etc. (reference
examples/benchmark.py:72, smpl_sim/agents/agent_humanoid.py:92); with this repository on the
path those imports get the MI355X stepper, everything else of the reference is out of scope.
"""
__version__ = "0.0.1"
