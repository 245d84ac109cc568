"""CPU restatement (float64 numpy, plain loops) of the learning-side arithmetic next to the env path: test
infrastructure, like the rest of oracle/.  Pinned by golden vectors generated from the reference's own modules
(tests/golden/make_golden.py).

  gae_flat            smpl_sim/learning/learning_utils.py:198-218   estimate_advantages
  gaussian_log_prob   smpl_sim/learning/distributions.py:6-26       DiagGaussian.log_prob (summed over action dims)
  running_norm_update smpl_sim/learning/running_norm.py:22-29
  ppo_surrogate       smpl_sim/agents/agent_ppo.py:91-99            ppo_loss
  rescale_actions     smpl_sim/learning/learning_utils.py:224-227
"""
import numpy as np


def gae_flat(rewards, not_done, not_dead, values, gamma, tau, bootstrap=0.0, normalize=True):
    r, nd, ndead, v = (np.asarray(x, np.float64).reshape(-1) for x in (rewards, not_done, not_dead, values))
    adv = np.zeros_like(r)
    next_v, next_a = float(bootstrap), 0.0
    for i in range(len(r) - 1, -1, -1):
        delta = r[i] + gamma * next_v * ndead[i] - v[i]
        adv[i] = delta + gamma * tau * next_a * nd[i]
        next_v, next_a = v[i], adv[i]
    ret = v + adv
    if normalize:
        adv = (adv - adv.mean()) / adv.std(ddof=1)
    return adv, ret


def gae_columns(rewards, not_done, not_dead, values, gamma, tau, bootstrap=None):
    T, N = np.shape(rewards)
    adv, ret = np.zeros((T, N)), np.zeros((T, N))
    for n in range(N):
        b = 0.0 if bootstrap is None else float(bootstrap[n])
        adv[:, n], ret[:, n] = gae_flat(np.asarray(rewards)[:, n], np.asarray(not_done)[:, n], np.asarray(not_dead)[:, n],
                                        np.asarray(values)[:, n], gamma, tau, bootstrap=b, normalize=False)
    return adv, ret


def gaussian_log_prob(mean, log_std, action):
    mean, log_std, action = (np.asarray(x, np.float64) for x in (mean, log_std, action))
    z = (action - mean) / np.exp(log_std)
    return (-0.5 * z * z - log_std - 0.5 * np.log(2 * np.pi)).sum(axis=1, keepdims=True)


def running_norm_update(n, mean, var, x):
    x = np.asarray(x, np.float64)
    m = x.shape[0]
    bm, bv = x.mean(axis=0), x.var(axis=0)
    w = n / (n + m)
    var = w * var + (1 - w) * bv + w * (1 - w) * (bm - mean) ** 2
    mean = w * mean + (1 - w) * bm
    return n + m, mean, var


def ppo_surrogate(log_probs, fixed_log_probs, advantages, clip_epsilon):
    ratio = np.exp(np.asarray(log_probs, np.float64) - np.asarray(fixed_log_probs, np.float64))
    a = np.asarray(advantages, np.float64)
    return -np.minimum(ratio * a, np.clip(ratio, 1 - clip_epsilon, 1 + clip_epsilon) * a).mean()


def rescale_actions(low, high, action):
    d, m = (high - low) / 2.0, (high + low) / 2.0
    return action * d + m
