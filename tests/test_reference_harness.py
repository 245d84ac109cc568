"""The reference's own harness, executed unchanged: `make_env_mj` and `evaluate_env` of /root/reference/examples/benchmark.py
(:68-116; BASELINE config 1) are loaded from the reference tree and run through the `smpl_sim` import-path shim — `eval(cfg.env.task)(cfg)`
builds smplsim_amd's HumanoidEnv from the harness's own YAML text, `evaluate_env` times reset(seed=54) / step(action=...) on it, and
the vector form goes through gym.vector.AsyncVectorEnv.  gymnasium / omegaconf / mujoco are not in this image: tests/refstubs holds
minimal stand-ins (used only when the real packages are missing).  The reference tree exists in the build container only, so this
runs on the wavefront emulator (no GPU here); tests/test_gpu_parity.py::test_benchmark_harness_shape_runs is its restatement on the
GPU box, where /root/reference does not exist."""
import importlib.util
import os
import sys

import numpy as np
import pytest

REF = "/root/reference/examples/benchmark.py"
STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refstubs")


def _load_harness():
    sys.dont_write_bytecode = True                              # the checkout is read-only: no __pycache__ in it
    added = []
    for name in ("gymnasium", "omegaconf", "mujoco"):
        if importlib.util.find_spec(name) is None and STUBS not in sys.path:
            sys.path.insert(0, STUBS); added.append(STUBS)
    spec = importlib.util.spec_from_file_location("reference_benchmark", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                                # module level: imports, YAML_CONFIG_STR, the function definitions
    return mod


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree is only present in the build container")
def test_reference_benchmark_functions_run_unchanged_through_the_shim(emu_backend, capsys):
    bm = _load_harness()
    env = bm.make_env_mj(bm.YAML_CONFIG_STR, 1)                  # benchmark.py:68-85: OmegaConf.create + eval(cfg.env.task)(cfg)
    import smplsim_amd.envs as E
    assert isinstance(env, E.HumanoidEnv) and env.self_collision and env.contact_bodies == ["R_Ankle", "L_Ankle", "R_Toe", "L_Toe"]
    m1 = bm.evaluate_env(env, time_format="ms", reps=3)         # benchmark.py:97-116: reset(seed=54) x reps, step(action=one sample) x reps
    assert set(m1) == {"reset/avg_time", "step/avg_time", "step/sps"} and all(np.isfinite(v) and v > 0 for v in m1.values())
    obs, rew, term, trunc, info = env.step(action=env.action_space.sample())
    assert obs.shape == (289,) and np.isfinite(obs).all()
    env.close()
    venv = bm.make_env_mj(bm.YAML_CONFIG_STR, 3)                 # the vector form: gym.vector.AsyncVectorEnv over three thunks
    assert venv.num_envs == 3
    mv = bm.evaluate_env(venv, time_format="ms", reps=2)
    assert mv["step/sps"] > 0
    venv.close()
    with capsys.disabled():
        print(f"\nreference harness on the emulator (CPU; a functional record, not a speed): single {m1}  vector(3) {mv}")
