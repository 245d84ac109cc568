"""Body-body contacts of the oracle: the static pair filters of the reference MJCF, known answers of the
three pair functions, and the physics of a two-body contact — constraint optimality (KKT) and Newton's third law (a
self-contact is an internal force: it must not change the total momentum rate of a free-floating body).
"Parity unpinned" like the rest of mj_step (MuJoCo is absent): these are analytic / physical checks of the restatement."""
import numpy as np
import pytest

from helpers import default_qpos, model_const, oracle_model
from oracle import oracle as O

I3 = np.eye(3)


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return I3 + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def test_pair_filters_of_the_reference_model():
    """276 body pairs - 23 parent-child - 10 <exclude> = 243 (all contype & conaffinity combinations collide:
    smpl_humanoid.xml:5,24), SURVEY §8a-4."""
    om = oracle_model(self_collision=True)
    d = O.OracleData(om)
    assert d.get(O.D_NSELF)[1] == 243
    assert O.OracleData(oracle_model(self_collision=False)).get(O.D_NSELF)[1] == 0
    assert O.OracleData(oracle_model("smplx_humanoid", self_collision=True)).get(O.D_NSELF)[1] == 52 * 51 // 2 - 51 - 10


def test_capsule_capsule_known_answers():
    # crossed at right angles, axes 0.15 apart along z: closest points at the centres
    c = O.narrow_phase("cc", ([0, 0, 0], [1, 0, 0], 0.05, 0.2), ([0, 0, 0.15], [0, 1, 0], 0.06, 0.3), margin=0.05)
    assert len(c) == 1
    pos, n, dist = c[0]
    assert abs(dist - 0.04) < 1e-12 and np.allclose(n, [0, 0, 1]) and np.allclose(pos, [0, 0, 0.05 + 0.02])
    # beyond the margin: nothing
    assert O.narrow_phase("cc", ([0, 0, 0], [1, 0, 0], 0.05, 0.2), ([0, 0, 0.15], [0, 1, 0], 0.06, 0.3), margin=0.001) == []
    # end against side: capsule 2 pointing at capsule 1's flank
    c = O.narrow_phase("cc", ([0, 0, 0], [1, 0, 0], 0.05, 0.2), ([0.1, 0.4, 0], [0, 1, 0], 0.05, 0.3))
    assert len(c) == 1 and abs(c[0][2] - 0.0) < 1e-12 and np.allclose(c[0][1], [0, 1, 0])
    # parallel, overlapping over [-0.1, 0.2] in x: two contacts at the ends of the overlap
    c = O.narrow_phase("cc", ([0, 0, 0], [1, 0, 0], 0.05, 0.2), ([0.2, 0, 0.09], [1, 0, 0], 0.05, 0.3))
    assert len(c) == 2 and all(abs(x[2] + 0.01) < 1e-12 for x in c) and sorted(round(x[0][0], 6) for x in c) == [-0.1, 0.2]


def test_capsule_box_known_answers():
    box = ([0, 0, 0], I3, [0.1, 0.2, 0.05])
    # capsule lying flat on the top face, inside the rectangle: the middle and the far end, both 5 mm deep
    c = O.narrow_phase("cb", ([0.02, 0, 0.085], [1, 0, 0], 0.04, 0.05), box)
    assert len(c) == 2 and all(abs(x[2] + 0.005) < 1e-12 and np.allclose(x[1], [0, 0, -1]) for x in c)
    assert sorted(round(x[0][0], 6) for x in c) == [-0.03, 0.02]
    # capsule standing on the top face: one contact below its lower end (the upper end is out of range)
    c = O.narrow_phase("cb", ([0.03, 0.1, 0.05 + 0.04 + 0.1 - 0.002], [0, 0, 1], 0.04, 0.1), box)
    assert len(c) == 1 and abs(c[0][2] + 0.002) < 1e-12 and np.allclose(c[0][0], [0.03, 0.1, 0.049])
    # tilted capsule crossing above a box edge: the minimiser is interior, the contact normal is the edge-to-axis direction
    ax = np.array([0.0, 1.0, 1.0]) / np.sqrt(2)
    c = O.narrow_phase("cb", ([0.15, 0, 0.1], ax, 0.06, 0.4), box, margin=0.01)
    assert len(c) == 1
    pos, n, dist = c[0]
    # brute force: distance from the segment to the box
    ts = np.linspace(-0.4, 0.4, 400001)
    pts = np.array([0.15, 0, 0.1]) + ts[:, None] * ax
    dd = np.linalg.norm(np.maximum(np.abs(pts) - np.array([0.1, 0.2, 0.05]), 0), axis=1)
    assert abs(dist - (dd.min() - 0.06)) < 1e-9


def test_box_box_known_answers():
    a = ([0, 0, 0], I3, [0.1, 0.1, 0.05])
    # small box resting 2 mm deep on the big one: its four bottom vertices
    c = O.narrow_phase("bb", a, ([0.02, 0.01, 0.05 + 0.03 - 0.002], I3, [0.03, 0.04, 0.03]))
    assert len(c) == 4 and all(abs(x[2] + 0.002) < 1e-12 and np.allclose(x[1], [0, 0, 1]) for x in c)
    # the same with the roles swapped (big on small): the small box's top vertices, normal still first -> second
    c = O.narrow_phase("bb", ([0.02, 0.01, 0.05 + 0.03 - 0.002], I3, [0.03, 0.04, 0.03]), a)
    assert len(c) == 4 and all(np.allclose(x[1], [0, 0, -1]) for x in c)
    # turned 45 degrees about z and overhanging: only the vertices over the face make contacts
    c = O.narrow_phase("bb", a, ([0.1, 0, 0.05 + 0.03 - 0.001], _rot([0, 0, 1], np.pi / 4), [0.05, 0.05, 0.03]))
    assert 1 <= len(c) <= 4 and all(x[0][0] <= 0.1 + 1e-9 for x in c)
    # edge against edge: two long bars crossed at right angles, both rolled 45 degrees about their long axes so that an
    # edge of each faces the other
    R1 = _rot([1, 0, 0], np.pi / 4)
    R2 = _rot([0, 1, 0], np.pi / 4)
    h = 0.05 * np.sqrt(2)
    c = O.narrow_phase("bb", ([0, 0, 0], R1, [0.3, 0.05, 0.05]), ([0, 0, 2 * h - 0.003], R2, [0.05, 0.3, 0.05]))
    assert len(c) == 1 and abs(c[0][2] + 0.003) < 1e-9 and np.allclose(c[0][1], [0, 0, 1], atol=1e-9) and np.allclose(c[0][0], [0, 0, h - 0.0015], atol=1e-9)
    # separated
    assert O.narrow_phase("bb", a, ([0.3, 0, 0], I3, [0.1, 0.1, 0.05])) == []


def test_box_box_face_manifold_of_crossed_boxes():
    """Two bars crossed in a '+', the upper one 1 cm into the lower one: no vertex of either box is inside the other, the
    contact manifold is the overlap rectangle of the two faces (mjc_BoxBox / ODE clip the incident face against the reference
    face's side planes and return its 4 corners).  Also yawed, where the clipped polygon's corners are edge-edge crossings."""
    lower = ([0, 0, 0], I3, [0.3, 0.05, 0.05])
    for yaw in (0.0, 0.3, -1.1):
        Rz = _rot([0, 0, 1], yaw)
        upper = ([0, 0, 0.1 - 0.01], Rz, [0.05, 0.3, 0.05])
        c = O.narrow_phase("bb", lower, upper)
        assert len(c) == 4, (yaw, c)
        assert all(abs(x[2] + 0.01) < 1e-12 and np.allclose(x[1], [0, 0, 1]) and abs(x[0][2] - 0.045) < 1e-12 for x in c)
        # the corners are the crossings of the lower bar's long edges (y = +-0.05) with the upper bar's (its local x = +-0.05)
        P2 = np.array([x[0][:2] for x in c])
        assert np.allclose(np.abs(P2[:, 1]), 0.05, atol=1e-12)
        loc = P2 @ Rz[:2, :2]                                    # in the upper bar's frame
        assert np.allclose(np.abs(loc[:, 0]), 0.05, atol=1e-12)
        assert abs(P2.mean(axis=0)).max() < 1e-12
        # roles swapped: same points, opposite normal
        c2 = O.narrow_phase("bb", upper, lower)
        assert len(c2) == 4 and all(np.allclose(x[1], [0, 0, -1]) for x in c2)
        assert np.allclose(sorted(map(tuple, np.round(P2, 9))), sorted(tuple(np.round(x[0][:2], 9)) for x in c2))
    # a box tilted onto a face: only the corner region below the margin survives the depth test
    c = O.narrow_phase("bb", ([0, 0, 0], I3, [0.2, 0.2, 0.05]), ([0, 0, 0.05 + 0.0705], _rot([1, 1, 0], 0.6), [0.05, 0.05, 0.05]))
    assert 1 <= len(c) <= 2 and all(x[2] < 0.001 for x in c)


def _self_contact_state(rs, om):
    """Arms folded into the torso / legs crossed: random large joint angles until the model reports body-body contacts."""
    d = O.OracleData(om)
    for _ in range(200):
        q = default_qpos(76); q[2] = 5.0
        q[7:] = rs.uniform(-1.5, 1.5, 69)
        d.qpos = q; d.qvel = rs.normal(size=75) * 0.5; d.ctrl = rs.normal(size=69) * 5; d.forward()
        if d.nself >= 2:
            return d, q
    raise AssertionError("no self-contact state found")


def test_two_body_contacts_satisfy_kkt_and_are_internal_forces():
    om_on, om_off = oracle_model(self_collision=True), oracle_model(self_collision=False)
    mc = model_const()
    rs = np.random.default_rng(2)
    mass = O.OracleModel.get(om_on, O.M_MASS)
    for trial in range(6):
        d, q = _self_contact_state(rs, om_on)
        assert d.ncon == d.nself and (d.con_body1 >= 0).all()        # in the air: body-body contacts only
        b1, b2 = d.con_body1, d.con_body
        for a, b in zip(b1, b2):
            names = (mc.body_names[a], mc.body_names[b])
            assert mc.body_parent[a] != b and mc.body_parent[b] != a and names not in mc.excludes and names[::-1] not in mc.excludes
        assert (d.con_dist < 0.001 + 1e-12).all()
        # KKT of the convex problem
        M, acc, a_s, fc, f = d.M, d.qacc, d.get(O.D_QACC_SMOOTH), d.get(O.D_QFRC_CONSTRAINT), d.get(O.D_EFC_FORCE)
        assert np.abs(M @ (acc - a_s) - fc).max() < 1e-7 * (1 + np.abs(fc).max()) and (f >= 0).all() and f.max() > 0
        # Newton's third law: the contact forces between bodies do not act on the free joint's translation as a net force —
        # total momentum rate = total weight, with or without them
        assert np.abs(fc[:3]).max() < 1e-8 * (1 + np.abs(fc).max())
        # ... and they push the bodies apart: the relative normal acceleration of a penetrating contact rises
        d0 = O.OracleData(om_off); d0.qpos = q; d0.qvel = d.qvel; d0.ctrl = d.ctrl; d0.forward()
        assert np.abs(d0.qacc - acc).max() > 1e-3
