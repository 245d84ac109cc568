"""The reference's PPO loop, executed unchanged over this package's env (north_star: "the PPO loop in smpl_sim/learning ... drop in
unchanged"; VERDICT r4 missing #2).  Loaded from /root/reference (never copied) behind the `smpl_sim` import-path shim:

  * `smpl_sim/run.py::main` itself — hydra-composed cfg (config.yaml + env=speed + robot + learning), `agent_dict[...]` ->
    `AgentHumanoid(cfg, dtype, device)` (agents/agent_humanoid.py:33-75: `eval(cfg.env.task)(cfg)`, PolicyGaussian / Value / optimizers /
    logger / seed) and `optimize_policy()` (:181-213) for two epochs: `Agent.sample` with num_threads=2 (agents/agent.py:121-145: one
    forked worker process + the parent, queue hand-over, TrajBatch, LoggerRL.merge), `AgentPG.update_params` (agent_pg.py:42-62: GAE)
    and `AgentPPO.update_policy` (agent_ppo.py:20-83), `save_curr`, `log_train` -> wandb.log.
  * the same agent object driven by hand, with the memory / logger / batch shapes asserted.

wandb / hydra / omegaconf / gymnasium are not in this image: tests/refstubs holds test-side stand-ins (joblib and tqdm are installed).
The reference tree exists in the build container only, so this runs on the wavefront emulator; smpl_sim.envs.tasks.HumanoidSpeed is
smplsim_amd's env (kernel source on the emulator), everything under smpl_sim.agents / smpl_sim.learning / smpl_sim.utils is the reference's."""
import importlib
import importlib.util
import os
import sys

import numpy as np
import pytest

REF_ROOT = "/root/reference"
RUN_PY = os.path.join(REF_ROOT, "smpl_sim", "run.py")
CFG_DIR = os.path.join(REF_ROOT, "smpl_sim", "data", "cfg")
STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refstubs")
OVERRIDES = ["env=speed", "num_threads=2", "learning.min_batch_size=48", "learning.max_epoch=2", "learning.mlp.units=[64,32]",
             "learning.opt_num_epochs=2", "learning.save_curr_frequency=1", "exp_name=shim_check"]

pytestmark = pytest.mark.skipif(not os.path.exists(RUN_PY), reason="the reference tree is only present in the build container")


@pytest.fixture()
def reference_on_path(emu_backend, monkeypatch, tmp_path):
    """This repository first (the shim), the reference checkout behind it, stand-ins for the absent third-party modules last."""
    sys.dont_write_bytecode = True                              # the checkout is read-only: no __pycache__ in it
    import smpl_sim                                             # the shim: extends its __path__ over the checkout behind it
    for name in ("wandb", "hydra", "omegaconf", "gymnasium"):
        if importlib.util.find_spec(name) is None and STUBS not in sys.path:
            monkeypatch.syspath_prepend(STUBS)
    monkeypatch.syspath_prepend(REF_ROOT) if REF_ROOT not in sys.path else None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.remove(root); sys.path.insert(0, root)             # shim before the checkout
    ref_pkg = os.path.join(REF_ROOT, "smpl_sim")
    if ref_pkg not in smpl_sim.__path__:
        smpl_sim.__path__.append(ref_pkg)                       # what pkgutil.extend_path does at import when the checkout is on sys.path
    monkeypatch.chdir(tmp_path)
    yield tmp_path
    for m in [m for m in sys.modules if m.startswith(("smpl_sim.agents", "smpl_sim.learning", "smpl_sim.utils.flags", "reference_run"))]:
        del sys.modules[m]


def test_reference_run_py_trains_two_epochs_through_the_shim(reference_on_path, monkeypatch, capsys):
    import torch
    out = reference_on_path / "out"
    monkeypatch.setattr(sys, "argv", ["run.py", "--config-path", CFG_DIR, f"hydra.run.dir={out}"] + OVERRIDES)
    spec = importlib.util.spec_from_file_location("reference_run", RUN_PY)
    run = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(run)
    run.main()                                                  # smpl_sim/run.py:34-83, unchanged
    import smplsim_amd.envs as E
    from smpl_sim.agents import agent_dict                      # the reference's
    assert agent_dict["agent_humanoid"].__module__ == "smpl_sim.agents.agent_humanoid"
    assert sys.modules["smpl_sim.agents.agent_humanoid"].__file__.startswith(REF_ROOT)
    assert sys.modules["smpl_sim.envs.tasks"].HumanoidSpeed is E.HumanoidSpeed
    log = open(out / "log.txt").read()
    assert log.count("Ep: ") == 2 and "State_dim: 292" in log, log
    ck = torch.load(out / "Humanoid.pth", weights_only=False)
    assert ck["epoch"] == 2 and ck["frame"] >= 48 and {"policy", "value", "optimizer_policy", "optimizer_value"} <= set(ck)
    assert ck["policy"]["action_mean.weight"].shape == (69, 32) and ck["policy"]["norm.mean"].shape == (292,)
    import wandb
    assert len(wandb.logged) == 2
    for step, data in wandb.logged:
        assert np.isfinite(data["avg_episode_reward"]) and data["eps_len"] >= 1 and np.isfinite(data["avg_rwd"])
    assert "training done!" in capsys.readouterr().out


def test_reference_sample_worker_and_update_policy_shapes(reference_on_path):
    import hydra
    import torch
    from smpl_sim.agents.agent_humanoid import AgentHumanoid     # the reference's class
    cfg, _ = hydra.compose(CFG_DIR, "config", OVERRIDES + ["num_threads=1", "no_log=True", f"output_dir={reference_on_path}/o2"])
    torch.set_default_dtype(torch.float32)
    agent = AgentHumanoid(cfg, torch.float32, torch.device("cpu"), training=True, checkpoint_epoch=0)
    import smplsim_amd.envs as E
    assert isinstance(agent.env, E.HumanoidSpeed) and agent.state_dim == 292 and agent.action_dim == 69
    with torch.no_grad():
        memory, logger = agent.sample_worker(0, None, 40)         # agents/agent.py:64-109
    n = len(memory)
    assert n >= 40 and logger.num_steps == n and logger.num_episodes >= 1 and abs(logger.avg_episode_len * logger.num_episodes - n) < 1e-9
    assert set(logger.info_dict) >= {"critic_state"} and len(logger.info_dict["critic_state"]) == n
    batch = agent.traj_cls([memory])
    assert batch.states.shape == (n, 292) and batch.critic_states.shape == (n, 292) and batch.next_states.shape == (n, 292)
    assert batch.actions.shape == (n, 69) and batch.rewards.shape == (n,) and batch.not_done.shape == (n,) and batch.exps.shape == (n,)
    assert np.isfinite(batch.states).all() and np.abs(batch.states).max() <= 5.0 + 1e-6       # clip_obs [-5, 5] (agent.py:147-151)
    assert (batch.rewards > 0).all() and (batch.rewards <= 1).all()                           # forward_reward is an exp(-...)
    assert batch.not_done.sum() == n - logger.num_episodes                                    # one terminal transition per episode
    before = [p.detach().clone() for p in agent.policy_net.parameters()] + [p.detach().clone() for p in agent.value_net.parameters()]
    agent.update_params(batch)                                   # agent_pg.py:42-62 -> agent_ppo.py:20-83
    after = list(agent.policy_net.parameters()) + list(agent.value_net.parameters())
    assert any(not torch.equal(a, b) for a, b in zip(after, before)) and all(torch.isfinite(a).all() for a in after)
    # sampling through the forking path by hand: the batch of two processes is the concatenation of both memories
    agent.num_threads = 2
    tb, lg = agent.sample(40)                                    # agents/agent.py:121-145
    assert tb.states.shape[0] == lg.num_steps >= 40 and lg.num_episodes >= 2
    agent.env.close()
