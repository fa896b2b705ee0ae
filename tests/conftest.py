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
    """The reference's 16-camera rig (EXTR) with the PRODUCT's tensors (se3d_rig_from_transforms, host code): what bench.py and a node built
    on the library run with.  tests/test_abi.py asserts that they equal the oracle's restatement of S3D:1187-1211 bit for bit."""
    import numpy as np
    from smartedgesensor3dhumanpose_amd import rig as R
    from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors
    r = R.extr_rig_16cam()
    Rt, F = rig_tensors(r)
    Rt_o, F_o = oracle.rig_tensors(r)
    assert np.array_equal(Rt, Rt_o) and np.array_equal(F.view(np.uint32), F_o.view(np.uint32))
    r.set_tensors(Rt, F)
    return r


@pytest.fixture(scope="session")
def demo_params():
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    return pm.Params(pm.POSE_SIMPLE, 0, 0.045)  # pose_triangulate_demo.launch:4-5,12-14,26-27


class DeviceBuffer:
    """Raw HIP device allocation for the device-pack tests (ctypes on the already loaded HIP runtime): the GPU test-suite never
    initialises torch.cuda, whose first use on a fresh box can page in its GPU code objects for minutes."""

    def __init__(self, nbytes):
        import ctypes as C
        from smartedgesensor3dhumanpose_amd import _lib
        _lib.lib()
        self._C, self.nbytes = C, int(nbytes)
        # the HIP runtime the product library is already running on (torch's bundled copy when torch is installed): open THAT file —
        # dlopen("libamdhip64.so") could pick another copy of the runtime (system ROCm) that does not match the loaded HSA runtime
        loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]
        self._hip = C.CDLL(loaded[0] if loaded else "libamdhip64.so")
        p = C.c_void_p()
        assert self._hip.hipMalloc(C.byref(p), C.c_size_t(self.nbytes)) == 0
        self.ptr = p.value
        assert self._hip.hipMemset(C.c_void_p(self.ptr), 0, C.c_size_t(self.nbytes)) == 0

    def data_ptr(self):
        return self.ptr

    def numpy(self, dtype):
        import numpy as np
        out = np.zeros(self.nbytes, np.uint8)
        assert self._hip.hipMemcpy(out.ctypes.data_as(self._C.c_void_p), self._C.c_void_p(self.ptr), self._C.c_size_t(self.nbytes), 2) == 0  # D2H
        return out.view(dtype)

    def upload(self, arr):
        import numpy as np
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        assert a.nbytes <= self.nbytes
        assert self._hip.hipMemcpy(self._C.c_void_p(self.ptr), a.ctypes.data_as(self._C.c_void_p), self._C.c_size_t(a.nbytes), 1) == 0  # H2D

    def free(self):
        if self.ptr:
            self._hip.hipFree(self._C.c_void_p(self.ptr))
            self.ptr = None


@pytest.fixture
def device_buffer():
    made = []

    def make(nbytes):
        made.append(DeviceBuffer(nbytes))
        return made[-1]

    yield make
    for b in made:
        b.free()
