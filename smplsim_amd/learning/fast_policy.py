"""Policy inference of the sampler on the matrix cores (include/smplsim_mlp.h).

PolicyGaussian.select_action (reference policy_gaussian.py:25-32 -> MLP.forward, mlp.py:52-60, behind RunningNorm in eval mode,
running_norm.py:33-42) for a whole batch of envs per control step: observation clamp + normalisation -> bf16, then one fused
launch per Linear layer (bf16 operands, fp32 accumulation, bias + activation in the epilogue, bf16 activations between layers),
fp32 action mean out; the Gaussian noise is added in fp32.  Nine launches per step instead of ~40 torch ones, and the GEMMs run at
MFMA rate.  The weights are snapshots (bf16 copies, padded to the kernels' tile sizes): call refresh() after an optimiser step.

This is an inference path only — no autograd; the PPO update evaluates the fp32 networks as before.
"""
import ctypes as C
import math

import torch

from .. import _cabi
from .._lib import lib
from ..batch import _check, _launch_stream, _ptr


def _pad_to(n, m):
    return (n + m - 1) // m * m


class FusedPolicyInference:
    def __init__(self, policy, clip_obs_range=(-5.0, 5.0), max_batch=0):
        self.policy = policy
        self.device = next(policy.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("FusedPolicyInference needs the policy on a GPU (there is no CPU path)")
        act = getattr(policy.net, "activation_name", None)
        if act is None:
            raise ValueError("the policy's MLP does not record its activation name")
        if act not in _cabi.ACTIVATIONS:
            raise ValueError(f"activation {act!r} has no fused epilogue (silu, tanh, relu)")
        self.act = _cabi.ACTIVATIONS[act]
        self.clip = (float(clip_obs_range[0]), float(clip_obs_range[1])) if clip_obs_range else (-3.0e38, 3.0e38)
        self.layers = list(policy.net.affine_layers) + [policy.action_mean]
        self.state_dim = self.layers[0].in_features
        self.action_dim = self.layers[-1].out_features
        self.kpad = [_pad_to(l.in_features, 32) for l in self.layers]
        self.w, self.b = [], []
        self._bufs = {}
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Take bf16 snapshots of the current weights (rows = output features, K zero-padded to a multiple of 32)."""
        self.w, self.b = [], []
        for l, kp in zip(self.layers, self.kpad):
            w = torch.zeros(l.out_features, kp, dtype=torch.bfloat16, device=self.device)
            w[:, :l.in_features] = l.weight.detach().to(torch.bfloat16)
            self.w.append(w.contiguous())
            self.b.append(l.bias.detach().to(torch.float32).contiguous())

    def _buffers(self, M, slot=0):
        """Scratch of one forward pass in flight; `slot` = which of several concurrent passes (sub-batches on their own streams)."""
        if slot not in self._bufs or self._bufs[slot][0] != M:
            width = max(max(l.out_features for l in self.layers[:-1]), self.kpad[0])
            bf = dict(dtype=torch.bfloat16, device=self.device)
            # flat ping-pong buffers: a layer's output [M, N] is the next layer's operand [M, K = N], row stride = its own width
            self._bufs[slot] = (M, torch.zeros(M * width, **bf), torch.zeros(M * width, **bf), torch.zeros(M, self.action_dim, dtype=torch.float32, device=self.device))
        return self._bufs[slot][1:]

    @torch.no_grad()
    def mean(self, obs, slot=0):
        """Action means [M, action_dim] (fp32) of a batch of raw observations [M, state_dim] (fp32, any row stride)."""
        assert obs.dtype == torch.float32 and obs.dim() == 2 and obs.shape[1] == self.state_dim and obs.stride(1) == 1
        M = obs.shape[0]
        a, b, out = self._buffers(M, slot)
        L, st = lib(), _launch_stream(self.device)
        nm = self.policy.norm
        _check(L.ss_obs_to_bf16(_ptr(obs), M, self.state_dim, obs.stride(0), _ptr(nm.mean), _ptr(nm.std), _ptr(nm.n), self.clip[0], self.clip[1],
                                float(nm.clip) if nm.clip else 3.0e38, _ptr(a), self.kpad[0], st))
        x, y, ldx = a, b, self.kpad[0]
        for i, l in enumerate(self.layers):
            last = i == len(self.layers) - 1
            dst = out if last else y
            ldy = self.action_dim if last else l.out_features
            # a hidden layer's output is the next layer's [M, K] operand: its row stride must be that layer's padded K
            if not last:
                assert self.kpad[i + 1] == l.out_features, "hidden widths must be multiples of 32"
            _check(L.ss_linear_bf16(_ptr(x), _ptr(self.w[i]), _ptr(self.b[i]), _ptr(dst), M, l.out_features, ldx, ldy,
                                    _cabi.ACTIVATIONS["none"] if last else self.act, int(last), st))
            x, y, ldx = y, x, l.out_features
        return out

    @torch.no_grad()
    def sample_into(self, mean, noise, action_out, action_env, clip, logp_out=None):
        """One launch for the Gaussian head (ss_gaussian_sample): action_out [M, nu] = mean + exp(log_std) * noise (a row slice of the
        rollout's action tensor), action_env [M, nu] = its copy clipped to `clip` = (lo, hi) (what the env is stepped with), logp_out
        [M, 1] = the draw's log-density under this behaviour policy.  Replaces eight elementwise torch launches per control step."""
        M = mean.shape[0]
        assert mean.is_contiguous() and noise.is_contiguous() and action_out.stride(1) == 1 and action_env.stride(1) == 1
        assert logp_out is None or logp_out.is_contiguous()
        _check(lib().ss_gaussian_sample(_ptr(mean), _ptr(noise), _ptr(self.policy.action_log_std), M, self.action_dim, _ptr(action_out),
                                        action_out.stride(0), _ptr(action_env), action_env.stride(0), float(clip[0]), float(clip[1]),
                                        _ptr(logp_out), _launch_stream(self.device)))

    @torch.no_grad()
    def select_action(self, obs, mean_action=False, generator=None, return_log_prob=False):
        """return_log_prob: also the log-density of the drawn action under THIS (bf16) behaviour policy, [M, 1] — what the PPO ratio's
        denominator must be when the sampler runs here while the update evaluates the fp32 network (normal_log_density of the
        reference's PolicyGaussian.get_log_prob, learning_utils.py: sum over the action dimensions)."""
        mean = self.mean(obs)
        if mean_action:
            return (mean, None) if return_log_prob else mean
        noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
        log_std = self.policy.action_log_std
        act = torch.addcmul(mean, log_std.exp(), noise)
        if not return_log_prob:
            return act
        logp = (-0.5 * noise.pow(2) - 0.5 * math.log(2.0 * math.pi) - log_std).sum(1, keepdim=True)
        return act, logp
