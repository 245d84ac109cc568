"""On-device PPO agent for the batched stepper.

What it replaces in the reference: Agent.sample (agents/agent.py:64-145: up to 36 CPU worker processes, one env each,
pickled Memory objects), the Python GAE loop (learning_utils.py:198-218) and AgentPPO.update_policy
(agents/agent_ppo.py:20-99) — keeping their semantics (sampling from the Gaussian policy with eval-mode normalisation,
obs clipping to clip_obs_range, actions clipped to the action space before they reach the env while the unclipped
sample is what is stored, full-batch epochs of one critic step + one clipped-surrogate step, grad-norm clipping,
the checkpoint dictionary layout of agent_humanoid.py:115-140).

What is different by construction: N envs advance in lock step for a fixed horizon T instead of every worker finishing
whole episodes, so a rollout column can end mid-episode; its last step bootstraps with V(next observation)
(`bootstrap=False` reproduces the reference's zero).  Everything — observations, actions, rewards, flags, advantages —
stays in HBM; the only host synchronisation per epoch is the logging read-back.
"""
import math
from dataclasses import dataclass, field

import torch


from ..learning.gae import estimate_advantages_columns, normalize_advantages
from ..learning.networks import MLP, PolicyGaussian, Value


@dataclass
class PPOConfig:
    """Defaults = smpl_sim/data/cfg/learning/simple_mlp.yaml."""
    gamma: float = 0.99
    tau: float = 0.95
    clip_epsilon: float = 0.2
    opt_num_epochs: int = 10
    value_opt_niter: int = 1
    policy_lr: float = 5e-5
    value_lr: float = 3e-4
    policy_weightdecay: float = 0.0
    value_weightdecay: float = 0.0
    policy_grad_clip: float = 25.0
    hidden: tuple = (2048, 1536, 1024, 1024, 512, 512)
    activation: str = "silu"
    log_std: float = -2.5
    fix_std: bool = True
    clip_obs: bool = True
    clip_obs_range: tuple = (-5.0, 5.0)
    clip_actions: bool = True
    min_batch_size: int = 51200
    bootstrap: bool = True
    amp_bf16: bool = False      # run the update's network passes under bf16 autocast (MFMA rate); off = the reference's fp32
    mfma_inference: bool = False   # sampler: policy forward by the library's fused bf16 MFMA kernels (learning/fast_policy.py)
    mfma_update: bool = False      # update: the networks' forward AND backward passes on the library's own GEMM (learning/fused_train.py), bf16 operands
    extra: dict = field(default_factory=dict)


class AgentPPO:
    def __init__(self, env, cfg=None, seed=0):
        self.env, self.cfg = env, cfg or PPOConfig()
        c = self.cfg
        self.device = env.device
        torch.manual_seed(seed)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.state_dim, self.action_dim = env.obs_size, env.nu
        self.policy_net = PolicyGaussian(self.state_dim, self.action_dim, c.hidden, c.activation, c.log_std, c.fix_std).to(self.device)
        self.value_net = Value(MLP(self.state_dim, c.hidden, c.activation)).to(self.device)
        # (with mfma_update: torch's single-kernel Adam — the same update rule; the default multi-tensor form is 8 launches per step, 3 ms of an 80 ms update)
        fused = dict(fused=True) if c.mfma_update and self.device.type == "cuda" else {}
        self.optimizer_policy = torch.optim.Adam(self.policy_net.parameters(), lr=c.policy_lr, eps=1e-8, weight_decay=c.policy_weightdecay, **fused)
        self.optimizer_value = torch.optim.Adam(self.value_net.parameters(), lr=c.value_lr, eps=1e-8, weight_decay=c.value_weightdecay, **fused)
        self.epoch, self.num_steps = 0, 0
        self.horizon = max(1, -(-c.min_batch_size // env.num_envs))
        self._obs = None
        self.fast_policy = None
        if c.mfma_inference:
            from ..learning.fast_policy import FusedPolicyInference
            self.fast_policy = FusedPolicyInference(self.policy_net, c.clip_obs_range if c.clip_obs else None)
        self.fused_policy = self.fused_value = None
        if c.mfma_update:
            from ..learning.fused_train import FusedMLPTrain
            self.fused_policy = FusedMLPTrain(self.policy_net.net.affine_layers, self.policy_net.action_mean, c.activation)
            self.fused_value = FusedMLPTrain(self.value_net.net.affine_layers, self.value_net.value_head, c.activation)

    # ------------------------------------------------------------------ sampling
    def _prep_obs(self, obs):
        lo, hi = self.cfg.clip_obs_range
        return obs.clamp(lo, hi) if self.cfg.clip_obs else obs

    def _prep_actions(self, actions):
        # rescale_actions(low, high, clip(a, low, high)) with the env's action space [-1, 1] is the clip itself
        return actions.clamp(-1.0, 1.0) if self.cfg.clip_actions else actions

    # One control step of the sampler for a set of env rows (all of them, or one sub-batch of a pipeline): observation -> action ->
    # env step -> rollout rows.  Per step: one clamp (straight into the rollout's state row), the policy forward, ONE launch for the
    # Gaussian head on the bf16 path (ss_gaussian_sample: draw, clipped copy, log-density), the env's launches, three row stores; the
    # episode flags stay boolean rows until the rollout is complete (round 5: ~12 launches per step instead of ~25 next to the policy's nine).
    def _sample_step(self, t, r, obs, noise, buf, mean_action, slot, step_fn):
        c = self.cfg
        state = buf["states"][t, r]
        if c.clip_obs:
            torch.clamp(obs, c.clip_obs_range[0], c.clip_obs_range[1], out=state)
        else:
            state.copy_(obs)
        act_row = buf["actions"][t, r]
        lo, hi = (-1.0, 1.0) if c.clip_actions else (-3.0e38, 3.0e38)
        if self.fast_policy is not None:
            mean = self.fast_policy.mean(state, slot=slot)
            if mean_action:
                act_row.copy_(mean)
                a_env = self._prep_actions(mean)
            else:
                a_env = buf["a_env"][r]
                self.fast_policy.sample_into(mean, noise, act_row, a_env, (lo, hi), buf["logp"][t, r] if buf["logp"] is not None else None)
        else:
            with self._autocast():
                mean = self._f32(self.policy_net.select_action(state, True))
            a = mean if mean_action else mean + self.policy_net.action_log_std.exp() * noise
            act_row.copy_(a)
            a_env = self._prep_actions(a)
        obs, rew, died, timed_out, _ = step_fn(a_env)
        buf["rewards"][t, r] = rew
        buf["dead"][t, r] = died                               # (boolean rows: one launch each; turned into the float masks once, in _rollout_out)
        torch.logical_or(died, timed_out, out=buf["done"][t, r])
        return obs

    def _rollout_buffers(self, T, N, mean_action):
        f = dict(device=self.device, dtype=torch.float32)
        # bf16 sampler: the behaviour policy's own log-densities go into the batch (the update's fp32 network would give the PPO
        # ratio a denominator from a slightly different policy: a mean error of 1e-2 at sigma 0.08 moves log-probs noticeably)
        return dict(states=torch.empty(T, N, self.state_dim, **f), actions=torch.empty(T, N, self.action_dim, **f),
                    rewards=torch.empty(T, N, **f), dead=torch.empty(T, N, dtype=torch.bool, device=self.device),
                    done=torch.empty(T, N, dtype=torch.bool, device=self.device), a_env=torch.empty(N, self.action_dim, **f),
                    logp=torch.empty(T, N, 1, **f) if (self.fast_policy is not None and not mean_action) else None)

    def _rollout_out(self, buf, last_obs_rows, mean_action):
        T, N = buf["rewards"].shape
        f = dict(device=self.device, dtype=torch.float32)
        last = torch.empty(N, self.state_dim, **f)
        for r, o in last_obs_rows:
            last[r] = self._prep_obs(o)                          # post-autoreset observation = first state of the next episode
        self.num_steps += T * N
        out = dict(states=buf["states"], actions=buf["actions"], rewards=buf["rewards"], not_done=(~buf["done"]).to(torch.float32),
                   not_dead=(~buf["dead"]).to(torch.float32), exps=torch.full((T, N), 0.0 if mean_action else 1.0, **f), last_state=last)
        if buf["logp"] is not None:
            out["log_probs"] = buf["logp"]
        return out

    @torch.no_grad()
    def sample(self, horizon=None, mean_action=False):
        """Roll all envs `horizon` control steps forward.  Returns time-major tensors on the env's device."""
        env, T, N = self.env, horizon or self.horizon, self.env.num_envs
        self.policy_net.eval()
        if self._obs is None:
            self._obs, _ = env.reset()
        buf = self._rollout_buffers(T, N, mean_action)
        obs, rows = self._obs, slice(0, N)
        for t in range(T):
            noise = None if mean_action else torch.randn(N, self.action_dim, generator=self.gen, device=self.device, dtype=torch.float32)
            obs = self._sample_step(t, rows, obs, noise, buf, mean_action, 0, env.step)
        self._obs = obs
        return self._rollout_out(buf, [(rows, obs)], mean_action)

    @torch.no_grad()
    def sample_pipelined(self, pipe, horizon=None, mean_action=False):
        """sample() over a pipeline.PipelinedVecEnv of the same job: per control step, sub-batch g's policy forward, action sampling and
        step launch are issued on stream g, so the GEMM tiles of one sub-batch run on the CUs that the tail of another sub-batch's step
        launch has already left.  Same draws as sample(): the Gaussian noise of a step is drawn for all N envs
        from self.gen and sliced, the env's inputs come from the pipeline's master generator — every env gets what it gets in the
        single batch, the returned tensors are bit-identical to sample()'s (GPU test; the bf16 policy's row results do not depend
        on the number of rows: the K order of the MFMA kernels is the same for every tile width).  Measured (profiles/r05_sampler.txt):
        no gain over the serial order at G = 2 and a loss at G >= 4 — a sub-batch's launch is as long as its heaviest env's chain
        whatever its size, two step kernels cannot share a CU's LDS, and the host issues G times the launches."""
        T, N, G = horizon or self.horizon, pipe.num_envs, pipe.sub_batches
        self.policy_net.eval()
        if getattr(self, "_pipe_obs", None) is None:
            self._pipe_obs = pipe.reset()
        buf = self._rollout_buffers(T, N, mean_action)
        from ..pipeline import _record_event
        cuda = self.device.type == "cuda"
        obs = list(self._pipe_obs)
        for t in range(T):
            noise = None if mean_action else torch.randn(N, self.action_dim, generator=self.gen, device=self.device, dtype=torch.float32)
            pipe.draw_step_inputs()
            ev = _record_event(self.device)                      # (after the buffers' allocation, this step's noise and the env draws)
            for g in range(G):
                r, s = pipe.rows(g), pipe.streams[g]
                with pipe.stream(g):
                    s.wait_event(ev)
                    if noise is not None and cuda:
                        noise.record_stream(s)
                    obs[g] = self._sample_step(t, r, obs[g], None if noise is None else noise[r], buf, mean_action, g,
                                               lambda a, g=g: pipe.step_async(g, a))
        if cuda:
            cur = torch.cuda.current_stream(self.device)
            for s in pipe.streams:                                # the caller's stream continues after every sub-batch's last step
                cur.wait_stream(s)
        self._pipe_obs = obs
        return self._rollout_out(buf, [(pipe.rows(g), obs[g]) for g in range(G)], mean_action)

    # ------------------------------------------------------------------ update
    def _autocast(self):
        on = self.cfg.amp_bf16 and self.device.type == "cuda"
        return torch.autocast(device_type=self.device.type, dtype=torch.bfloat16, enabled=on)

    def _f32(self, x):
        return x.float() if x.dtype == torch.bfloat16 else x

    def _policy_log_prob(self, states, actions):
        """PolicyGaussian.get_log_prob with the network passes on the library's GEMM (mfma_update): RunningNorm (train mode: its statistics
        follow the passes as in the torch path), mean = head(MLP(.)), the log-density in fp32 torch."""
        p = self.policy_net
        mean = self.fused_policy(p.norm(states))
        log_std = p.action_log_std.expand_as(mean)
        z = (actions - mean) * torch.exp(-log_std)
        terms = -0.5 * z * z - log_std - 0.5 * math.log(2.0 * math.pi)
        # the row sums as a matrix-vector product: torch's reduction over the 69 columns of a [53 248, 69] tensor (and its backward) was 180 us a call
        return torch.matmul(terms, torch.ones(terms.shape[1], 1, dtype=terms.dtype, device=terms.device))

    def ppo_loss(self, states, actions, advantages, fixed_log_probs):
        if getattr(self, "fused_policy", None) is not None:
            log_probs = self._policy_log_prob(states, actions)
        else:
            with self._autocast():
                log_probs = self._f32(self.policy_net.get_log_prob(states, actions))
        ratio = torch.exp(log_probs - fixed_log_probs)
        clipped = ratio.clamp(1.0 - self.cfg.clip_epsilon, 1.0 + self.cfg.clip_epsilon)
        return -torch.minimum(ratio * advantages, clipped * advantages).mean()

    def update_value(self, critic_states, returns):
        for _ in range(self.cfg.value_opt_niter):
            if getattr(self, "fused_value", None) is not None:
                pred = self.fused_value(critic_states)
            else:
                with self._autocast():
                    pred = self._f32(self.value_net(critic_states))
            loss = (pred - returns).pow(2).mean()
            self.optimizer_value.zero_grad(set_to_none=True)
            loss.backward()
            self.optimizer_value.step()
        return loss.detach()

    def update_params(self, batch):
        c = self.cfg
        T, N = batch["rewards"].shape
        states = batch["states"].reshape(T * N, -1)
        self.policy_net.eval(); self.value_net.eval()
        with torch.no_grad(), self._autocast():
            # the critic's passes for GAE: on the library's GEMM with mfma_update, under the update's bf16 autocast with amp_bf16 (round 6: they ran
            # in fp32 whatever the update's precision was: 0.8 TFLOP at the fp32 rate, 8 ms of a 116 ms update), fp32 otherwise (the reference)
            vf = self.fused_value if getattr(self, "fused_value", None) is not None else self.value_net
            values = self._f32(vf(states)).reshape(T, N)
            boot = self._f32(vf(batch["last_state"])).reshape(N) if c.bootstrap else None
        adv, ret = estimate_advantages_columns(batch["rewards"], batch["not_done"], batch["not_dead"], values, c.gamma, c.tau, boot)
        adv = normalize_advantages(adv).reshape(T * N, 1)
        ret = ret.reshape(T * N, 1)
        actions = batch["actions"].reshape(T * N, -1)
        ind = batch["exps"].reshape(-1).nonzero(as_tuple=False).squeeze(1)
        if "log_probs" in batch:                               # sampled by the bf16 inference path: its own log-densities
            fixed_log_probs = batch["log_probs"].reshape(T * N, 1)
        else:
            with torch.no_grad(), self._autocast():
                fixed_log_probs = self._f32(self.policy_net.get_log_prob(states, actions))
        self.policy_net.train(); self.value_net.train()        # RunningNorm statistics follow the training passes
        s_i, a_i, adv_i, flp_i = states[ind], actions[ind], adv[ind], fixed_log_probs[ind]
        info = {}
        for _ in range(c.opt_num_epochs):
            info["value_loss"] = self.update_value(states, ret)
            loss = self.ppo_loss(s_i, a_i, adv_i, flp_i)
            self.optimizer_policy.zero_grad(set_to_none=True)
            loss.backward()
            if c.policy_grad_clip is not None:
                torch.nn.utils.clip_grad_norm_(self.policy_net.parameters(), c.policy_grad_clip)
            self.optimizer_policy.step()
            info["surr_loss"] = loss.detach()
        if self.fast_policy is not None:
            self.fast_policy.refresh()                          # bf16 snapshots of the updated weights for the sampler
        self.epoch += 1
        info["mean_reward"] = batch["rewards"].mean()
        info["episodes_ended"] = (1.0 - batch["not_done"]).sum()
        return info

    def optimize_policy(self, epochs=1):
        log = []
        for _ in range(epochs):
            log.append({k: float(v) for k, v in self.update_params(self.sample()).items()})
        return log

    # ------------------------------------------------------------------ checkpoints (agent_humanoid.py:115-140 layout)
    def get_full_state_weights(self):
        return {"policy": self.policy_net.state_dict(), "value": self.value_net.state_dict(), "epoch": self.epoch,
                "optimizer_policy": self.optimizer_policy.state_dict(), "optimizer_value": self.optimizer_value.state_dict(),
                "frame": self.num_steps}

    def set_full_state_weights(self, state):
        self.policy_net.load_state_dict(state["policy"]); self.value_net.load_state_dict(state["value"])
        if self.fast_policy is not None:
            self.fast_policy.refresh()
        self.epoch = state["epoch"]
        self.optimizer_value.load_state_dict(state["optimizer_value"])
        self.optimizer_policy.load_state_dict(state["optimizer_policy"])
        self.num_steps = state.get("frame", 0)
