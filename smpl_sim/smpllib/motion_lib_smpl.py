"""Import-path shim: `smpl_sim.smpllib.motion_lib_smpl.MotionLibSMPL` (reference smpl_sim/smpllib/motion_lib_smpl.py:50)
resolves to the device-side motion library of smplsim_amd."""
from smplsim_amd.motion_lib import FixHeightMode, MotionLibSMPL, Skeleton  # noqa: F401
