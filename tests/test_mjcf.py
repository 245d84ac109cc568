"""Model compiler: known answers from SURVEY.md §7/§8c and agreement with the oracle's
independent compile (XML -> primitives in oracle.py -> mass/inertia/invweight0 in C)."""
import dataclasses

import numpy as np
import pytest

from helpers import model_const, oracle_model
from oracle import oracle as O
from smplsim_amd.mjcf import MjcfError, compile_mjcf
from smplsim_amd.mjcf_writer import default_xml_str


def test_smpl_fixture_known_answers():
    mc = model_const()
    assert (mc.nbody, mc.nq, mc.nv, mc.nu) == (24, 76, 75, 69)
    assert abs(mc.total_mass - 71.805) < 1e-3
    # tree parents == reference torch_smpl_humanoid_batch.py:44
    assert mc.body_parent.tolist() == [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 17, 11, 19, 20, 21, 22]
    assert mc.body_names[:5] == ["Pelvis", "L_Hip", "L_Knee", "L_Ankle", "L_Toe"]
    assert mc.actuator_names[0] == "L_Hip_x" and mc.actuator_dof.tolist() == list(range(6, 75))
    assert np.all(mc.dof_armature[:6] == 0) and np.all(mc.dof_armature[6:] == 0.01)
    assert np.allclose(mc.jnt_range[6], [-np.pi, np.pi])
    assert mc.geom_margin == 0.001 and mc.friction == 1.0 and mc.has_vel_sensors


def test_smplx_fixture_known_answers():
    mc = model_const("smplx_humanoid")
    assert (mc.nbody, mc.nv, mc.nu) == (52, 159, 153)
    assert abs(mc.total_mass - 73.568) < 1e-3


def test_capsule_inertia_formula_against_quadrature():
    from smplsim_amd.mjcf import GEOM_CAPSULE, geom_mass_inertia
    r, hl = 0.06, 0.115
    m, inert = geom_mass_inertia(GEOM_CAPSULE, np.array([r, hl, 0]), 1000.0)
    n = 200
    xs = (np.arange(n) + 0.5) / n * 2 * r - r
    zs = (np.arange(2 * n) + 0.5) / (2 * n) * 2 * (hl + r) - (hl + r)
    X, Y, Z = np.meshgrid(xs, xs, zs, indexing="ij")
    zc = np.clip(Z, -hl, hl)
    inside = X ** 2 + Y ** 2 + (Z - zc) ** 2 <= r ** 2
    dv = (2 * r / n) ** 2 * (2 * (hl + r) / (2 * n)) * 1000.0
    assert abs(inside.sum() * dv - m) / m < 5e-3
    assert abs(((Y ** 2 + Z ** 2) * inside).sum() * dv - inert[0]) / inert[0] < 5e-3
    assert abs(((X ** 2 + Y ** 2) * inside).sum() * dv - inert[2]) / inert[2] < 1e-2


@pytest.mark.parametrize("name", ["smpl_humanoid", "smplx_humanoid"])
def test_compiler_matches_oracle_compile(name):
    mc, om = model_const(name), oracle_model(name)
    assert np.allclose(om.get(O.M_MASS), mc.body_mass, rtol=1e-13)
    assert np.allclose(om.get(O.M_INERTIA).reshape(-1, 3), mc.body_inertia, rtol=1e-12)
    assert np.allclose(om.get(O.M_IPOS).reshape(-1, 3), mc.body_ipos, atol=1e-15)
    assert np.allclose(om.get(O.M_IQUAT).reshape(-1, 4), mc.body_iquat, atol=1e-14)
    assert np.allclose(om.get(O.M_GSIZE).reshape(-1, 3), mc.geom_size, atol=1e-15)
    assert np.allclose(om.get(O.M_BODY_INVW).reshape(-1, 2), mc.body_invweight0, rtol=1e-9)
    assert np.allclose(om.get(O.M_DOF_INVW), mc.dof_invweight0, rtol=1e-9)
    assert np.allclose(om.get(O.M_RANGE).reshape(-1, 2)[6:], mc.jnt_range[6:], rtol=1e-14)


def test_rejects_unsupported_models():
    xml = default_xml_str()
    with pytest.raises(MjcfError):
        compile_mjcf(xml.replace('type="hinge"', 'type="slide"', 1))
    with pytest.raises(MjcfError):
        compile_mjcf("<mujoco><worldbody/></mujoco>")
    with pytest.raises(MjcfError):
        compile_mjcf("not xml")


@pytest.mark.refonly
@pytest.mark.parametrize("ref,name", [("smpl_sim/data/assets/mjcf/smpl_humanoid.xml", "smpl_humanoid"),
                                      ("smpl_humanoid.xml", "smplx_humanoid")])
def test_shipped_table_equals_reference_xml(ref, name):
    a = compile_mjcf(open("/root/reference/" + ref).read())
    b = model_const(name)
    for f in dataclasses.fields(a):
        x, y = getattr(a, f.name), getattr(b, f.name)
        assert np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y, f.name
