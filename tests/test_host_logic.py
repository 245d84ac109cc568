"""CPU-side tests (-m "not gpu"): host logic, the C-ABI library's exported symbols (no compute calls without
a GPU), loud failure without a GPU, config/space shims.  The multi-process (gloo, world_size 2) shard path: test_multi_gpu_cpu.py."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import torch  # noqa: F401  (always before the library: one HIP runtime per process)
    from smplsim_amd import _cabi, _lib
    path = _lib.build()
    lib = ctypes.CDLL(path)
    header = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include"))))
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)        # prose in comments mentions reference functions
    declared = set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", header))
    declared -= {"ss_status"}
    assert declared == set(_cabi.EXPORTS) | set(_cabi.MLP_EXPORTS), declared ^ (set(_cabi.EXPORTS) | set(_cabi.MLP_EXPORTS))
    for name in declared:
        assert getattr(lib, name) is not None
    lib.ss_last_error.restype = ctypes.c_char_p
    assert lib.ss_last_error() is not None


def test_struct_layouts_match_the_header():
    """ctypes mirrors must have the same field order/count as the C structs."""
    from smplsim_amd import _cabi
    header = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include"))))
    for cname, ctype in (("ss_model_desc", _cabi.ModelDesc), ("ss_env_cfg", _cabi.EnvCfg), ("ss_state", _cabi.State),
                         ("ss_skeleton", _cabi.Skeleton), ("ss_motion_data", _cabi.MotionData),
                         ("ss_motion_state", _cabi.MotionState), ("ss_imitation_cfg", _cabi.ImitationCfg)):
        end = header.index("} %s;" % cname)
        body = header[header.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.split(",")
            for i, p in enumerate(parts):
                nm = re.sub(r"\[.*?\]", "", p.strip().split()[-1]).lstrip("*")
                names.append(nm)
        assert names == [f[0] for f in ctype._fields_], (cname, names)


def test_no_gpu_no_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from smplsim_amd.batch import SMPLSimVecEnv
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SMPLSimVecEnv(4)


def test_product_never_imports_the_oracle_or_the_emulator():
    pkg = os.path.join(ROOT, "smplsim_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests|wave_emu)\b", src, re.M), f
                assert "liboracle" not in src and "libss_emu" not in src, f


def test_config_and_spaces_shims():
    from smplsim_amd.config import AttrDict, default_cfg
    from smplsim_amd.spaces import Box
    cfg = default_cfg("HumanoidGetup", episode_length=50)
    assert cfg.env.state_init == "Fall" and cfg.env.episode_length == 50 and cfg.robot.get("remove_toe", False) is False
    assert AttrDict({"a": {"b": 1}}).a.b == 1
    b = Box(-np.ones(69), np.ones(69), dtype=np.float32)
    x = b.sample()
    assert x.shape == (69,) and x.dtype == np.float32 and b.contains(x)


def test_shard_ranges_partition_the_batch():
    from smplsim_amd.shard import shard_range, shard_seed
    for total, world in ((32768, 8), (4096, 1), (10, 4), (8192, 8)):
        seen = []
        for r in range(world):
            lo, hi = shard_range(total, world, r)
            seen.extend(range(lo, hi))
        assert seen == list(range(total))
    assert shard_seed(1234, 3) == 1237


def test_import_shim_lets_the_rest_of_the_reference_resolve_behind_it():
    """With this repo before a SMPLSim checkout on sys.path, smpl_sim.envs is ours and smpl_sim.learning is the reference's."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "smpl_sim", "learning")):
        pytest.skip("no reference checkout in this environment")
    code = ("import smpl_sim.envs.tasks as t, smpl_sim.learning.mlp as m, smpl_sim.smpllib.motion_lib_smpl as ml, sys;"
            "print(t.HumanoidEnv.__module__, m.__file__, ml.MotionLibSMPL.__module__)")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + ref, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-2000:]
    mod, mlp_file, ml_mod = out.stdout.split()
    assert mod.startswith("smplsim_amd.") and mlp_file.startswith(ref) and ml_mod == "smplsim_amd.motion_lib"


def test_mlp_entry_points_validate_their_arguments():
    """include/smplsim_mlp.h: the argument checks run before any launch, so they can be exercised without a GPU."""
    import torch  # noqa: F401
    from smplsim_amd import _cabi, _lib
    _lib.build()
    lib = _cabi.bind_mlp(ctypes.CDLL(_lib.LIB_PATH))
    lib.ss_last_error.restype = ctypes.c_char_p
    one = ctypes.c_void_p(16)                                       # never dereferenced: every call below fails its checks first
    assert lib.ss_linear_bf16(None, one, None, one, 4, 4, 32, 4, 0, 0, None) == -1 and b"null" in lib.ss_last_error()
    assert lib.ss_linear_bf16(one, one, None, one, 4, 4, 33, 4, 0, 0, None) == -1 and b"multiple of 32" in lib.ss_last_error()
    assert lib.ss_linear_bf16(one, one, None, one, 4, 8, 32, 4, 0, 0, None) == -1                 # ldy < N
    assert lib.ss_linear_bf16(one, one, None, one, 4, 4, 32, 4, 9, 0, None) == -1 and b"activation" in lib.ss_last_error()
    assert lib.ss_obs_to_bf16(one, 4, 289, 289, None, None, None, -5.0, 5.0, 5.0, one, 300, None) == -1      # kpad not a multiple of 32
    assert lib.ss_obs_to_bf16(one, 4, 289, 100, None, None, None, -5.0, 5.0, 5.0, one, 320, None) == -1      # row stride < dim
    # the update's entry points (round 6)
    assert lib.ss_linear_bf16_train(one, one, None, None, None, None, None, 64, 64, 64, 64, 0, 0, 0, None) == -1 and b"null" in lib.ss_last_error()
    assert lib.ss_linear_bf16_train(one, one, None, None, one, None, None, 64, 64, 96, 64, 0, 0, 0, None) == -1 and b"multiple of 64" in lib.ss_last_error()
    assert lib.ss_linear_bf16_train(one, one, None, one, one, None, None, 64, 64, 64, 64, 0, 0, 1, None) == -1 and b"accumulating" in lib.ss_last_error()
    assert lib.ss_linear_bf16_dx(one, one, None, one, one, 4096, 512, 256, 512, None) == -1 and b"null" in lib.ss_last_error()
    assert lib.ss_linear_bf16_dx(one, one, one, one, one, 1000, 512, 256, 512, None) == -1 and b"256 x 256 kernel only" in lib.ss_last_error()   # too few rows
    assert lib.ss_linear_bf16_dx(one, one, one, one, one, 4096, 512, 192, 512, None) == -1                                                      # K not a multiple of 128
    assert lib.ss_wgrad_bf16(one, None, one, 1024, 64, 64, 64, 64, 64, None) == -1 and b"null" in lib.ss_last_error()
    assert lib.ss_wgrad_bf16(one, one, one, 1000, 64, 64, 64, 64, 64, None) == -1 and b"multiple of 128" in lib.ss_last_error()
    assert lib.ss_wgrad_bf16(one, one, one, 1024, 60, 64, 64, 64, 64, None) == -1                                # n_out not a multiple of 8
    assert lib.ss_wgrad_bf16(one, one, one, 1024, 64, 64, 32, 64, 64, None) == -1                                # ldz < n_out
