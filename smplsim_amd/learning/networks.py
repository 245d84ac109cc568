"""Policy and value networks of the PPO agent, state_dict-compatible with the reference's checkpoints.

Parameter and buffer names follow smpl_sim/learning (so a `Humanoid_*.pth` written by either side loads in the other):
  policy: norm.{n,mean,var,std}, net.affine_layers.<i>.{weight,bias}, action_mean.{weight,bias}, action_log_std
  value : net.affine_layers.<i>.{weight,bias}, value_head.{weight,bias}
Semantics restated from policy_gaussian.py:14-41, mlp.py:36-60, critic.py:5-18, running_norm.py:5-42,
distributions.py:6-29.  Everything is batched over the env dimension and stays on the env's device.
"""
import math

import torch
from torch import nn

_ACTIVATIONS = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid,
                "gelu": nn.functional.gelu, "silu": nn.functional.silu}
_LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


class MLP(nn.Module):
    """Stack of Linear layers, the activation after every layer (also the last one)."""

    def __init__(self, input_dim, hidden_dims=(128, 128), activation="tanh"):
        super().__init__()
        if activation not in _ACTIVATIONS:
            raise ValueError(f"unknown activation {activation!r}")
        self._act = _ACTIVATIONS[activation]
        self.activation_name = activation
        dims = [int(input_dim), *[int(h) for h in hidden_dims]]
        self.out_dim = dims[-1]
        self.affine_layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for layer in self.affine_layers:
            x = self._act(layer(x))
        return x


class RunningNorm(nn.Module):
    """y = clip((x - mean) / (std + 1e-8), +-clip) with running batch statistics (updated in train mode only)."""

    def __init__(self, dim, demean=True, destd=True, clip=5.0):
        super().__init__()
        self.dim, self.demean, self.destd, self.clip = dim, demean, destd, clip
        self.register_buffer("n", torch.tensor(0, dtype=torch.long))
        self.register_buffer("mean", torch.zeros(dim))
        self.register_buffer("var", torch.zeros(dim))
        self.register_buffer("std", torch.zeros(dim))

    @torch.no_grad()
    def update(self, x):
        """Merge the (biased) statistics of a batch [m, dim] into the running ones (parallel-variance formula)."""
        m = x.shape[0]
        batch_var, batch_mean = torch.var_mean(x, dim=0, unbiased=False)
        w_old = self.n.to(x.dtype) / (self.n + m).to(x.dtype)
        w_new = 1.0 - w_old
        shift = batch_mean - self.mean
        self.var.copy_(w_old * self.var + w_new * batch_var + w_old * w_new * shift * shift)
        self.mean.copy_(w_old * self.mean + w_new * batch_mean)
        self.std.copy_(self.var.sqrt())
        self.n += m

    def forward(self, x):
        if self.training:
            self.update(x)
        y = x
        if self.demean:
            y = y - self.mean
        if self.destd:
            y = y / (self.std + 1e-8)
        if self.clip:
            y = y.clamp(-self.clip, self.clip)
        # identity until the first update (reference running_norm.py) — selected on the device: reading `n` on the host
        # would synchronise the stream once per policy forward, i.e. once per env step of the sampler
        return torch.where(self.n > 0, y, x)


class PolicyGaussian(nn.Module):
    """Diagonal Gaussian policy: mean = Linear(MLP(norm(obs))), state-independent log-std (fixed or learned)."""

    type = "gaussian"

    def __init__(self, state_dim, action_dim, hidden=(2048, 1536, 1024, 1024, 512, 512), activation="silu",
                 log_std=-2.5, fix_std=True):
        super().__init__()
        self.norm = RunningNorm(state_dim)
        self.net = MLP(state_dim, hidden, activation)
        self.action_mean = nn.Linear(self.net.out_dim, action_dim)
        with torch.no_grad():
            self.action_mean.weight.mul_(0.1)
            self.action_mean.bias.zero_()
        self.action_log_std = nn.Parameter(torch.full((1, action_dim), float(log_std)), requires_grad=not fix_std)

    def mean_and_log_std(self, obs):
        mean = self.action_mean(self.net(self.norm(obs)))
        return mean, self.action_log_std.expand_as(mean)

    def select_action(self, obs, mean_action=False, generator=None):
        mean, log_std = self.mean_and_log_std(obs)
        if mean_action:
            return mean
        noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
        return mean + log_std.exp() * noise

    def get_log_prob(self, obs, action):
        """log N(action; mean, std) summed over the action dimensions -> [B, 1]."""
        mean, log_std = self.mean_and_log_std(obs)
        z = (action - mean) * torch.exp(-log_std)
        return (-0.5 * z * z - log_std - _LOG_SQRT_2PI).sum(dim=1, keepdim=True)

    def get_kl(self, obs):
        """KL(old || new) with old = detached current distribution (zero value, non-zero gradient; TRPO-style)."""
        mean, log_std = self.mean_and_log_std(obs)
        mean0, log_std0 = mean.detach(), log_std.detach()
        kl = log_std - log_std0 + (torch.exp(2 * log_std0) + (mean0 - mean) ** 2) / (2.0 * torch.exp(2 * log_std)) - 0.5
        return kl.sum(dim=1, keepdim=True)


class Value(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net
        self.value_head = nn.Linear(net.out_dim, 1)
        with torch.no_grad():
            self.value_head.weight.mul_(0.1)
            self.value_head.bias.zero_()

    def forward(self, x):
        return self.value_head(self.net(x))
