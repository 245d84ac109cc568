"""On-device learning stack for the batched stepper: policy / value networks with the reference's
checkpoint layout, running observation normalisation, device-side GAE."""
from .networks import MLP, PolicyGaussian, RunningNorm, Value  # noqa: F401
from .gae import estimate_advantages_columns, normalize_advantages  # noqa: F401
