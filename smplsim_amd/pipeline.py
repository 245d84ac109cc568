"""Pipelined sub-batches on one GPU: G SMPLSimVecEnv shards of ONE N-env job, each on its own HIP stream.

A step launch ends when its slowest env ends (heavy-tailed Newton iteration counts), so a single batch leaves wave slots idle
during the tail of every launch (wave-slot occupancy 0.55 on the headline).  With G sub-batches stepped round-robin — sub-batch
g + 1 is launched while the tail of sub-batch g is still running, and the policy GEMMs / action sampling of one sub-batch run on
the CUs the others' tails have freed — the tails are filled (+12 % env-steps/s on the stepper alone at G = 4,
profiles/r01v_straggler_tail.txt; the sampler's figures: profiles/r05_sampler.txt).

What the caller sees is the SAME N-env job as one SMPLSimVecEnv(N, seed=s): env i of the job is env i - g n of sub-batch g = i // n,
and every random input the env draws (task targets, Fall-reset actions) comes from ONE generator seeded like the single batch's, drawn
for all N envs per step and handed to the sub-batches as row slices — so every env sees the bit-identical inputs in either form, and
since an env's step depends on nothing but its own state and inputs, the rollouts are bit-identical to the single batch's
(tests/test_host_flows_emu.py, tests/test_gpu_parity.py).  Only WHEN an env's result becomes available differs: a sub-batch's rows are
ready when its own launch ends, not when the slowest env of all N does.  bench.py's headline keeps the one-batch form (the vector env's
barrier over all envs, examples/benchmark.py:78-116); samplers that act per sub-batch use this one (agents/ppo.py sample_pipelined).
"""
import torch

from .batch import SMPLSimVecEnv, _cabi


def _make_stream(device):
    return torch.cuda.Stream(device=device)


def _stream_ctx(stream):
    return torch.cuda.stream(stream)


def _record_event(device):
    """An event on the CURRENT stream of `device` (the master draws happen there); the sub-batch streams wait for it."""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    return ev


class PipelinedVecEnv:
    def __init__(self, num_envs, sub_batches=4, device=0, seed=0, model=None, shape_id=None, **env_kw):
        assert num_envs % sub_batches == 0
        self.num_envs, self.sub_batches = num_envs, sub_batches
        n = self.n = num_envs // sub_batches
        # body shapes belong to the JOB's env ids: env i has shape i % num_shapes by default (SMPLSimVecEnv's rule on N envs), so
        # sub-batch g gets rows g n .. (g + 1) n of the job's table, not a table that restarts at 0 with every sub-batch
        num_shapes = model.num_shapes if model is not None else 1
        if shape_id is None and num_shapes > 1:
            shape_id = torch.arange(num_envs) % num_shapes
        if shape_id is not None:
            shape_id = torch.as_tensor(shape_id)
            if shape_id.shape != (num_envs,):
                raise ValueError("shape_id must hold one shape index per env of the whole job")

        def sid(g):
            return None if shape_id is None else shape_id[g * n:(g + 1) * n]

        first = SMPLSimVecEnv(n, device=device, seed=seed, model=model, shape_id=sid(0), **env_kw)
        # one model (one set of device tables) for all sub-batches
        self.envs = [first] + [SMPLSimVecEnv(n, device=device, seed=seed, model=first.model, shape_id=sid(g), **env_kw) for g in range(1, sub_batches)]
        if not (first.autoreset and first._fused_autoreset):
            raise ValueError("PipelinedVecEnv steps with the in-launch autoreset (StateInit Default / Fall, autoreset=True)")
        self.device, self.nu, self.obs_size = first.device, first.nu, first.obs_size
        self.task_id, self.state_init = first.task_id, first.state_init
        self.streams = [_make_stream(self.device) for _ in range(sub_batches)]
        self.gen = torch.Generator(device=self.device)          # the single batch's generator: same seed, same draws, same order
        self.gen.manual_seed(int(seed))
        self._draws, self._consumed = None, set()

    def rows(self, g):
        return slice(g * self.n, (g + 1) * self.n)

    # ---- the random inputs of one call for all N envs, in the order SMPLSimVecEnv draws them
    def _rand4(self):
        return None if self.task_id == _cabi.TASK_BASE else torch.rand(self.num_envs, 4, generator=self.gen, device=self.device)

    def _fall(self):
        return None if self.state_init != _cabi.INIT_FALL else torch.rand(self.num_envs, 3, self.nu, generator=self.gen, device=self.device)

    def reset(self):
        """All envs (SMPLSimVecEnv.reset's draws: Fall actions, then task targets).  Returns the sub-batches' observation buffers."""
        fa, tr = self._fall(), self._rand4()
        ev = _record_event(self.device)
        out = []
        for g, (env, s) in enumerate(zip(self.envs, self.streams)):
            r = self.rows(g)
            with _stream_ctx(s):
                s.wait_event(ev)
                out.append(env.reset(fall_actions=None if fa is None else fa[r], task_rand=None if tr is None else tr[r])[0])
        return out

    def draw_step_inputs(self):
        """One control step's draws for all N envs on the current stream — SMPLSimVecEnv.step's order: task_rand, then (fused
        autoreset) the reset's task_rand and fresh Fall actions.  Call once per step before the step_async calls of that step."""
        tr = self._rand4()
        tr2, fa = self._rand4(), None
        if self.envs[0]._fall_buf is not None:
            fa = torch.empty(self.num_envs, 3, self.nu, device=self.device).uniform_(0.0, 1.0, generator=self.gen)
        if self._draws is not None:
            missing = sorted(set(range(self.sub_batches)) - self._consumed)
            raise RuntimeError(f"draw_step_inputs: sub-batches {missing} have not stepped on the previous draws (every sub-batch steps once per "
                               "control step; a new draw now would desynchronise the generator from the single batch's sequence)")
        self._draws, self._consumed = (tr, tr2, fa, _record_event(self.device)), set()

    def step_async(self, g, actions, wait=None):
        """Enqueue one control step of sub-batch g on its stream (after `wait`, an event on another stream, if given); returns the
        sub-batch's (obs, rew, term, trunc, info) tensors, valid once its stream has reached this point (torch ops issued under
        `stream(g)` are ordered after it)."""
        if self._draws is None:
            self.draw_step_inputs()
        if g in self._consumed:
            raise RuntimeError(f"step_async: sub-batch {g} stepped twice on one control step's draws (the others have not: "
                               f"{sorted(set(range(self.sub_batches)) - self._consumed)})")
        tr, tr2, fa, ev = self._draws
        r, env, s = self.rows(g), self.envs[g], self.streams[g]
        with _stream_ctx(s):
            s.wait_event(ev)
            if wait is not None:
                s.wait_event(wait)
            for t in (tr, tr2, fa):                               # drawn on the master stream, read on this one: the allocator must not
                if t is not None and t.is_cuda:                   # hand the block out again before this stream is done with it
                    t.record_stream(s)
            if fa is not None:
                env._fall_buf.copy_(fa[r])
            out = env.step(actions, None if tr is None else tr[r], task_rand2=None if tr2 is None else tr2[r], _fall_drawn=fa is not None)
        self._consumed.add(g)                                     # any order: the draws go when every sub-batch has used its rows
        if len(self._consumed) == self.sub_batches:
            self._draws, self._consumed = None, set()
        return out

    def stream(self, g):
        return _stream_ctx(self.streams[g])

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        for e in self.envs:
            e.close()
