import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def extr_rig(oracle):
    from smartedgesensor3dhumanpose_amd import rig as R
    r = R.extr_rig_16cam()
    r.set_tensors(*oracle.rig_tensors(r))
    return r


@pytest.fixture(scope="session")
def demo_params():
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    return pm.Params(pm.POSE_SIMPLE, 0, 0.045)  # pose_triangulate_demo.launch:4-5,12-14,26-27
