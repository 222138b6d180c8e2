import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def small_sequence():
    """8 frames of the synthetic room at 120 x 160 (seeded)."""
    from hive_amd import synthetic
    return synthetic.make_sequence(num_frames=8, height=120, width=160, yaw_step_deg=45.0)


@pytest.fixture(scope="session")
def fusable_sequence():
    """10 frames of the synthetic room at 120 x 160, 9 degrees apart: hive_tsdf_integrate_batch sweeps them in groups of
    4 + 4 + 2 (a group grows while the optical axis stays within 36 degrees of its first frame's)."""
    from hive_amd import synthetic
    return synthetic.make_sequence(num_frames=10, height=120, width=160, yaw_step_deg=9.0, seed=77)


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a HIP device")
    from hive_amd import _lib
    return _lib.default_context(0)
