"""Oracle vs REAL MuJoCo vectors — runs only when tools/dump_mujoco_golden.py has been executed somewhere with a
`mujoco` wheel and its output committed as tests/golden/mujoco_vectors.npz (absent in round 1: no wheel, no network;
the physics part of the oracle is therefore "parity unpinned", oracle/oracle.h)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, oracle_model
from oracle import oracle as O

PATH = os.path.join(GOLDEN, "mujoco_vectors.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="no MuJoCo golden vectors (mujoco not installable here)")


def test_oracle_matches_mujoco_forward_and_rollout():
    g = np.load(PATH)
    om = oracle_model()
    assert np.allclose(om.get(O.M_MASS), g["mass"], rtol=1e-9)
    assert np.allclose(om.get(O.M_BODY_INVW).reshape(-1, 2), g["body_invweight0"], rtol=1e-6)
    d = O.OracleData(om)
    for i in range(len(g["qpos"])):
        d.qpos = g["qpos"][i]; d.qvel = g["qvel"][i]; d.ctrl = g["ctrl"][i]; d.warm = np.zeros(om.nv)
        d.forward()
        assert np.allclose(d.M, g["qM"][i], atol=1e-9)
        assert np.allclose(d.bias, g["bias"][i], atol=1e-7)
        assert np.allclose(d.xpos, g["xpos"][i], atol=1e-10)
        assert d.ncon == int(g["ncon"][i])
        assert np.allclose(d.qacc, g["qacc"][i], rtol=1e-5, atol=1e-5)
        a = g["roll_action"][i]
        for _ in range(15):
            d.ctrl = d.spd_torque(a); d.step()
        assert np.allclose(d.qpos, g["roll_qpos"][i], atol=1e-6) and np.allclose(d.qvel, g["roll_qvel"][i], atol=1e-4)
