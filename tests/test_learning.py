"""Learning side next to the env path: networks, running norm, PPO surrogate and GAE against golden
vectors generated from the reference's own smpl_sim.learning / agents modules, and the C-ABI GAE against the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import golden
from oracle import ppo_oracle as PO
from smplsim_amd.learning.networks import MLP, PolicyGaussian, Value


def _load(mod, g, prefix):
    sd = {k[len(prefix):]: torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}
    mod.load_state_dict(sd)                                   # same parameter / buffer names as the reference's checkpoints
    return mod


def test_oracle_gae_matches_reference_estimate_advantages():
    g = golden()
    adv, ret = PO.gae_flat(g["gae_rewards"], g["gae_not_done"], g["gae_not_dead"], g["gae_values"], 0.99, 0.95)
    # the reference allocates its work tensors with type(rewards)(n, 1), i.e. float32 whatever the input precision
    assert g["gae_adv"].dtype == np.float32
    assert np.allclose(adv, g["gae_adv"].ravel(), rtol=0, atol=2e-6)
    assert np.allclose(ret, g["gae_ret"].ravel(), rtol=0, atol=2e-6)


def test_networks_load_reference_state_dicts_and_match():
    g = golden()
    pol = _load(PolicyGaussian(11, 5, (16, 12), "silu", log_std=-1.0, fix_std=False).double(), g, "polsd_").eval()
    val = _load(Value(MLP(11, (16, 12), "silu")).double(), g, "valsd_").eval()
    x, a = torch.tensor(g["pol_x"]), torch.tensor(g["pol_a"])
    with torch.no_grad():
        mean, log_std = pol.mean_and_log_std(x)
        assert torch.allclose(pol.norm(x), torch.tensor(g["pol_norm_out"]), rtol=1e-12, atol=1e-12)
        assert torch.allclose(mean, torch.tensor(g["pol_mean"]), rtol=1e-10, atol=1e-12)
        assert torch.allclose(pol.get_log_prob(x, a), torch.tensor(g["pol_logp"]), rtol=1e-10, atol=1e-10)
        assert torch.allclose(val(x), torch.tensor(g["val_out"]), rtol=1e-10, atol=1e-12)
        assert np.allclose(PO.gaussian_log_prob(mean.numpy(), log_std.numpy(), a.numpy()), g["pol_logp"], rtol=1e-10)
        assert float(pol.get_kl(x).abs().max()) < 1e-12       # KL(old || new) is zero at the current parameters


def test_running_norm_statistics_match_reference():
    g = golden()
    pol = PolicyGaussian(11, 5, (16, 12), "silu").double().train()
    n, mean, var = 0, np.zeros(11), np.zeros(11)
    for x in g["rn_inputs"]:
        pol.norm(torch.tensor(x))
        n, mean, var = PO.running_norm_update(n, mean, var, x)
    for got, ref in ((pol.norm.mean.numpy(), g["rn_mean"]), (pol.norm.var.numpy(), g["rn_var"]), (pol.norm.std.numpy(), g["rn_std"]),
                     (mean, g["rn_mean"]), (var, g["rn_var"])):
        assert np.allclose(got, ref, rtol=1e-12, atol=1e-12)
    assert int(pol.norm.n) == int(g["rn_n"]) == n


def test_ppo_surrogate_and_action_rescale_match_reference():
    from smplsim_amd.agents.ppo import AgentPPO, PPOConfig
    g = golden()
    pol = _load(PolicyGaussian(11, 5, (16, 12), "silu", log_std=-1.0, fix_std=False).double(), g, "polsd_").eval()
    fake = AgentPPO.__new__(AgentPPO)                         # no env needed for the loss itself
    fake.policy_net, fake.cfg, fake.device = pol, PPOConfig(), torch.device("cpu")
    x, a = torch.tensor(g["pol_x"]), torch.tensor(g["pol_a"])
    with torch.no_grad():
        loss = AgentPPO.ppo_loss(fake, x, a, torch.tensor(g["ppo_adv"]), torch.tensor(g["ppo_fixed"]))
    assert np.isclose(float(loss), float(g["ppo_loss"]), rtol=1e-10)
    assert np.isclose(PO.ppo_surrogate(g["pol_logp"], g["ppo_fixed"], g["ppo_adv"], 0.2), float(g["ppo_loss"]), rtol=1e-10)
    assert np.allclose(PO.rescale_actions(g["resc_low"], g["resc_high"], g["resc_in"]), g["resc_out"], rtol=1e-14)
    # the env's action space is [-1, 1], where rescale(clip(a)) is the clip itself (what AgentPPO._prep_actions does)
    assert np.allclose(PO.rescale_actions(-np.ones(5), np.ones(5), g["resc_in"]), g["resc_in"])


def _gae_case(T, N, seed):
    rs = np.random.default_rng(seed)
    rew, val = rs.normal(size=(T, N)).astype(np.float32), rs.normal(size=(T, N)).astype(np.float32)
    done = rs.uniform(size=(T, N)) < 0.1
    dead = done & (rs.uniform(size=(T, N)) < 0.5)
    boot = rs.normal(size=N).astype(np.float32)
    return rew, (1.0 - done).astype(np.float32), (1.0 - dead).astype(np.float32), val, boot


def test_c_abi_gae_on_the_emulator_library_matches_oracle():
    from wave_emu import emu
    rew, nd, ndead, val, boot = _gae_case(37, 19, 5)
    adv, ret = np.zeros_like(rew), np.zeros_like(rew)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    L = emu.lib()
    assert L.ss_gae(p(rew), p(nd), p(ndead), p(val), p(boot), 37, 19, C.c_float(0.99), C.c_float(0.95), p(adv), p(ret), None) == 0
    a_ref, r_ref = PO.gae_columns(rew, nd, ndead, val, 0.99, 0.95, boot)
    assert np.abs(adv - a_ref).max() < 1e-4 and np.abs(ret - r_ref).max() < 1e-4
    assert L.ss_gae(p(rew), p(nd), p(ndead), p(val), None, 37, 19, C.c_float(0.99), C.c_float(0.95), p(adv), p(ret), None) == 0
    a_ref, _ = PO.gae_columns(rew, nd, ndead, val, 0.99, 0.95, None)
    assert np.abs(adv - a_ref).max() < 1e-4
    assert L.ss_gae(p(rew), p(nd), p(ndead), p(val), None, 0, 19, C.c_float(0.99), C.c_float(0.95), p(adv), p(ret), None) != 0


@pytest.mark.gpu
def test_gae_on_gpu_matches_oracle_and_flat_reference_layout():
    from smplsim_amd.learning.gae import estimate_advantages_columns, normalize_advantages
    dev = torch.device("cuda", 0)
    rew, nd, ndead, val, boot = _gae_case(25, 4096, 9)
    t = lambda x: torch.tensor(x, device=dev)
    adv, ret = estimate_advantages_columns(t(rew), t(nd), t(ndead), t(val), 0.99, 0.95, t(boot))
    a_ref, r_ref = PO.gae_columns(rew[:, :64], nd[:, :64], ndead[:, :64], val[:, :64], 0.99, 0.95, boot[:64])
    assert np.abs(adv.cpu().numpy()[:, :64] - a_ref).max() < 1e-4 and np.abs(ret.cpu().numpy()[:, :64] - r_ref).max() < 1e-4
    # one column without bootstrap == the reference's flat batch; golden vector of estimate_advantages itself
    g = golden()
    col = lambda k: torch.tensor(g[k].astype(np.float32), device=dev).reshape(-1, 1)
    adv, ret = estimate_advantages_columns(col("gae_rewards"), col("gae_not_done"), col("gae_not_dead"), col("gae_values"), 0.99, 0.95)
    assert np.abs(normalize_advantages(adv).cpu().numpy() - g["gae_adv"]).max() < 1e-4
    assert np.abs(ret.cpu().numpy() - g["gae_ret"]).max() < 1e-4


@pytest.mark.gpu
def test_ppo_agent_runs_on_device_and_checkpoints_round_trip(tmp_path):
    from smplsim_amd.agents.ppo import AgentPPO, PPOConfig
    from smplsim_amd.batch import SMPLSimVecEnv
    env = SMPLSimVecEnv(256, task="HumanoidSpeed", autoreset=True, seed=3)
    cfg = PPOConfig(hidden=(256, 128), min_batch_size=256 * 8, opt_num_epochs=3)
    agent = AgentPPO(env, cfg, seed=1)
    batch = agent.sample()
    assert batch["states"].shape == (8, 256, env.obs_size) and batch["states"].is_cuda
    assert float(batch["states"].abs().max()) <= 5.0 and torch.isfinite(batch["rewards"]).all()
    w0 = agent.policy_net.action_mean.weight.clone()
    info = agent.update_params(batch)
    assert all(np.isfinite(float(v)) for v in info.values())
    assert not torch.equal(w0, agent.policy_net.action_mean.weight) and int(agent.policy_net.norm.n) == 3 * 8 * 256
    path = tmp_path / "Humanoid.pth"
    torch.save(agent.get_full_state_weights(), path)
    other = AgentPPO(env, cfg, seed=2)
    other.set_full_state_weights(torch.load(path, map_location=env.device))
    x = batch["states"][0]
    other.policy_net.eval(); agent.policy_net.eval()
    with torch.no_grad():
        assert torch.equal(other.policy_net.select_action(x, True), agent.policy_net.select_action(x, True))
    assert set(torch.load(path, map_location="cpu")) == {"policy", "value", "epoch", "optimizer_policy", "optimizer_value", "frame"}


def test_fused_train_has_no_cpu_path():
    """learning.fused_train (the update's passes on the library's GEMM) refuses layers that are not on a GPU: like the rest of the package
    it has no CPU fallback (the oracle and torch are the checkers, tests/test_gpu_parity.py::test_fused_mlp_train_gradients_match_autograd)."""
    import pytest
    from smplsim_amd.learning.fused_train import FusedMLPTrain
    from smplsim_amd.learning.networks import MLP
    net = MLP(10, (64, 64), "silu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        FusedMLPTrain(net.affine_layers, torch.nn.Linear(64, 3), "silu")
    with pytest.raises(ValueError, match="fused epilogue"):
        FusedMLPTrain(net.affine_layers, torch.nn.Linear(64, 3), "gelu")


def test_fused_train_work_tensors_are_reused_only_between_finished_passes():
    """learning.fused_train._Buffers (host logic, no kernel): a pass's work tensors are kept between calls, initialised once (the ones row behind a
    transposed activation), keyed by shape; a pass that starts while the previous one's backward has not run gets fresh tensors."""
    from smplsim_amd.learning.fused_train import _Buffers
    b = _Buffers()
    calls = []

    def init(t):
        calls.append(1)
        t[2, :3] = 1.0
    t1 = b.get("ht", (4, 8), torch.float32, "cpu", False, init)
    t1[0, 0] = 5.0
    t2 = b.get("ht", (4, 8), torch.float32, "cpu", False, init)
    assert t2 is t1 and len(calls) == 1 and float(t2[2, :3].sum()) == 3.0 and float(t2[0, 0]) == 5.0     # kept, initialised once
    t3 = b.get("ht", (4, 16), torch.float32, "cpu", False, init)
    assert t3 is not t1 and tuple(t3.shape) == (4, 16) and len(calls) == 2                                 # another shape: another tensor
    f = b.get("ht", (4, 8), torch.float32, "cpu", True, init)
    assert f is not t1 and float(f[0, 0]) == 0.0 and float(f[2, :3].sum()) == 3.0 and len(calls) == 3       # fresh: zeroed and initialised
    assert b.busy is False
