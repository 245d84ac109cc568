"""Shared test helpers: build oracle models/envs for the shipped fixtures."""
import functools
import os

import numpy as np

from oracle import oracle as O
from smplsim_amd.gains import build_pd_tables
from smplsim_amd.mjcf import compile_mjcf
from smplsim_amd.mjcf_writer import default_xml_str

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FEET = ("L_Ankle", "R_Ankle", "L_Toe", "R_Toe")


@functools.lru_cache(maxsize=None)
def model_const(name="smpl_humanoid"):
    return compile_mjcf(default_xml_str(name))


def pd_tables(mc, **kw):
    rng = {n: mc.jnt_range[6 + i] for i, n in enumerate(mc.joint_names)}
    return build_pd_tables(mc.actuator_names, lambda n: rng[n], **kw)


@functools.lru_cache(maxsize=None)
def oracle_model(name="smpl_humanoid", timestep=1.0 / 450, control_mode="uhc_pd", self_collision=False, max_self_contacts=0,
                 solver="mujoco", linesearch="exact"):
    """solver: "mujoco" = mj_step's termination of the Newton iteration (tolerance 1e-8, 100 iterations: what the product is
    compared with), "converged" = to rounding level (the formulation leg of the parity triage)."""
    mc = model_const(name)
    kp, kd, tl, sc, of = pd_tables(mc, control_mode=control_mode)
    return O.OracleModel(default_xml_str(name), kp, kd, tl, sc, of, legal_bodies=FEET, timestep=timestep,
                         self_collision=self_collision, max_self_contacts=max_self_contacts, solver=solver, linesearch=linesearch)


def golden():
    return np.load(os.path.join(GOLDEN, "reference_vectors.npz"))


def default_qpos(nq):
    q = np.zeros(nq)
    q[2] = 0.94
    q[3:7] = 0.5
    return q
