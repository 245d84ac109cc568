"""Import-path shim for `from smpl_sim.smpllib.motion_lib_base import FixHeightMode` (reference humanoid_env.py:21)."""
from smplsim_amd.motion_lib import FixHeightMode, MotionLibSMPL as MotionLibBase  # noqa: F401
