from .ppo import AgentPPO, PPOConfig  # noqa: F401
