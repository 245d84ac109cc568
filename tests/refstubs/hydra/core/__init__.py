from . import hydra_config  # noqa: F401
