"""smplsim_amd — MI355X-native batched SMPL-humanoid env stepper (hot path of SMPLSim)."""
__version__ = "0.1.0"
