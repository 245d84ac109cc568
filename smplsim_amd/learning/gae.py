"""Device-side generalised advantage estimation over a time-major rollout [T, N] (C ABI ss_gae).

Replaces the Python loop of the reference's estimate_advantages (learning_utils.py:198-218), which moves the batch to
the CPU and walks it element by element.  The recursion itself is unchanged; it runs per env column, so the rollout
never leaves the GPU.
"""
import torch

from .._lib import lib
from ..batch import _check, _ptr


def estimate_advantages_columns(rewards, not_done, not_dead, values, gamma, tau, bootstrap=None):
    """rewards, not_done, not_dead, values: float32 [T, N] on one device; bootstrap: [N] value of the observation that
    follows the last step (None = 0, the reference's behaviour on episode-complete batches).
    Returns (advantages, returns) [T, N], NOT normalised."""
    T, N = rewards.shape
    args = [t.to(torch.float32).contiguous() for t in (rewards, not_done, not_dead, values)]
    boot = None if bootstrap is None else bootstrap.to(torch.float32).contiguous()
    adv, ret = torch.empty_like(args[0]), torch.empty_like(args[0])
    stream = torch.cuda.current_stream(rewards.device).cuda_stream if rewards.is_cuda else None
    _check(lib().ss_gae(*[_ptr(a) for a in args], _ptr(boot), int(T), int(N), float(gamma), float(tau), _ptr(adv), _ptr(ret), stream))
    return adv, ret


def normalize_advantages(adv, mask=None):
    """(A - mean) / std with the unbiased std, over the whole batch like the reference (learning_utils.py:214)."""
    a = adv if mask is None else adv[mask]
    return (adv - a.mean()) / a.std()
