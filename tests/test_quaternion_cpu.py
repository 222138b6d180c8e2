"""The reference's own unit tests (the whole of /root/reference/tests/quaternion.py:35-99: five checks of
``hive.geometric.Quaternion`` against scipy's ``Rotation``), run against ``hive_amd.geometric.Quaternion``."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from hive_amd.geometric import Quaternion

AXES = Rotation.from_euler('xyz', [[90, 0, 0], [0, 90, 0], [0, 0, 90]], degrees=True)


def to_scipy(q: Quaternion) -> Rotation:
    return Rotation.from_quat(np.asarray(q.values.T))


def from_scipy(r: Rotation) -> Quaternion:
    return Quaternion(torch.tensor(r.as_quat().T))


def test_normalise():
    np.testing.assert_allclose(AXES.as_rotvec(), to_scipy(from_scipy(AXES).normalise()).as_rotvec())
    scaled = Quaternion(from_scipy(AXES).values * 3.0)
    np.testing.assert_allclose(np.linalg.norm(np.asarray(scaled.normalise().values), axis=0), 1.0)


def test_conjugate():
    np.testing.assert_allclose(AXES.inv().as_rotvec(), to_scipy(from_scipy(AXES).conjugate()).as_rotvec())


def test_multiplying_by_conjugate_gives_identity():
    q = from_scipy(Rotation.from_euler('xyz', [[90, 0, 0]], degrees=True))
    np.testing.assert_allclose(np.array([[0.], [0.], [0.], [1.]]), np.asarray((q * q.conjugate()).values), atol=1e-15)


def test_multiplication():
    r2 = Rotation.from_euler('xyz', [[45, 0, 0], [0, 45, 0], [0, 0, 45]], degrees=True)
    np.testing.assert_allclose((AXES * r2).as_rotvec(), to_scipy(from_scipy(AXES) * from_scipy(r2)).as_rotvec())


@pytest.mark.parametrize("v", [[[1, 0, 0], [0, 1, 0], [0, 0, 1]], [[0, 1, 0], [0, 0, 1], [1, 0, 0]], [[0, 0, 1], [1, 0, 0], [0, 1, 0]]])
def test_rotating_vector(v):
    v = np.array(v)
    np.testing.assert_allclose(AXES.apply(v.T), np.asarray(from_scipy(AXES).apply(torch.tensor(v, dtype=torch.float64))).T, atol=1e-15)


def test_shape_and_type_errors():
    with pytest.raises(ValueError):
        Quaternion(torch.zeros(3, 2))
    with pytest.raises(TypeError):
        from_scipy(AXES) * 2.0
