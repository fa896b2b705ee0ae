"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/se3d.h declares,
and the POD layouts agree between the header (compiled with gcc), numpy and ctypes.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "se3d.h")


@pytest.fixture(scope="module")
def product_lib():
    from smartedgesensor3dhumanpose_amd import _lib
    _lib.build()
    return _lib


def test_every_declared_symbol_is_exported(product_lib):
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = sorted(set(re.findall(r"\b(se3d_[a-z0-9_]+)\s*\(", txt)))
    assert len(declared) >= 20
    L = product_lib.lib()
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(product_lib.EXPORTS) == declared


def test_struct_layouts_match_header():
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "se3d.h"
int main(void){
 printf("%zu %zu %zu %zu ", sizeof(se3d_keypoint2d), sizeof(se3d_person2d), sizeof(se3d_keypoint3d), sizeof(se3d_person3d));
 printf("%zu %zu %zu %zu %zu %zu ", sizeof(se3d_rig), sizeof(se3d_params), sizeof(se3d_limits), sizeof(se3d_frame_in), sizeof(se3d_frame_out), sizeof(se3d_fused_out));
 printf("%zu %zu %zu %zu\n", offsetof(se3d_person2d, keypoints), offsetof(se3d_keypoint3d, cov), offsetof(se3d_person3d, keypoints), offsetof(se3d_frame_out, persons));
 return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        vals = list(map(int, subprocess.check_output([os.path.join(d, "t")]).split()))
    assert vals[:4] == [pm.keypoint2d_dt.itemsize, pm.person2d_dt.itemsize, pm.keypoint3d_dt.itemsize, pm.person3d_dt.itemsize]
    assert vals[4:10] == [C.sizeof(pm.Rig), C.sizeof(pm.Params), C.sizeof(pm.Limits), C.sizeof(pm.FrameIn), C.sizeof(pm.FrameOut), C.sizeof(pm.FusedOut)]
    assert vals[10] == pm.person2d_dt.fields["keypoints"][1]
    assert vals[11] == pm.keypoint3d_dt.fields["cov"][1]
    assert vals[12] == pm.person3d_dt.fields["keypoints"][1]
    assert vals[13] == pm.FrameOut.persons.offset


def test_create_fails_loudly_without_gpu(product_lib):
    """The product path has no CPU fallback: without a HIP device se3d_create returns NULL and the wrapper raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from smartedgesensor3dhumanpose_amd import rig as R
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Se3dError, Skeleton3D, rig_tensors
    rig = R.extr_rig_16cam()
    rig.set_tensors(*rig_tensors(rig))  # host-only entry works without a GPU
    with pytest.raises(Se3dError):
        Skeleton3D(rig, max_frames=1, max_det_per_cam=8)


def test_product_rig_tensors_match_oracle(product_lib, oracle):
    """se3d_rig_from_transforms (product) vs the oracle's restatement of S3D:1187-1204: both form P^+ by the reference's SVD pseudo-inverse
    with its tolerance (S3D:236-240) and the product (e_x P_j) P_i^+ left to right (S3D:1201) by the same operation sequence in fp64, so the
    float F tables must be IDENTICAL — F feeds a strict `<` against max_epi_dist (S3D:620).  The EXTR-rig GPU parity tests and bench.py run on
    the product's tensors (tests/conftest.py)."""
    from smartedgesensor3dhumanpose_amd import rig as R
    from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors
    rigs = [R.extr_rig_16cam()] + [R.elliptical_rig(n, seed=s) for n, s in ((24, 3), (64, 1), (7, 11), (2, 5), (256, 0))]
    for rig in rigs:
        Rt_p, F_p = rig_tensors(rig)
        Rt_o, F_o = oracle.rig_tensors(rig)
        assert np.array_equal(Rt_p, Rt_o)
        assert np.array_equal(F_p.view(np.uint32), F_o.view(np.uint32)), (rig.n_cam, int((F_p != F_o).sum()))
    # and the matrices are fundamental matrices: x_j^T F_ij x_i = 0 for the projections of any 3-D point (fp64 check of the float table)
    rig = rigs[0]
    Rt, F = rig_tensors(rig)
    rng = np.random.default_rng(0)
    X = np.concatenate([rng.uniform(-3, 3, (50, 3)), np.ones((50, 1))], 1)
    P = rig.T[:, :3, :]
    x = np.einsum("cij,nj->cni", P, X)
    x = x / np.linalg.norm(x, axis=-1, keepdims=True)   # homogeneous image points, unit length
    k = 0
    for i in range(rig.n_cam):
        for j in range(i + 1, rig.n_cam):
            Fk = F[k].reshape(3, 3).astype(np.float64)
            r = np.einsum("ni,ij,nj->n", x[j], Fk, x[i])
            assert np.abs(r).max() < 1e-6 * np.linalg.norm(Fk), (i, j, np.abs(r).max())
            k += 1


def test_device_bytes_and_bench_memory_plan(product_lib):
    """se3d_device_bytes walks se3d_create's allocation list on the host: the pair slots dominate a 256cam128p context (the whole table would
    be 16 frames x 32 640 camera pairs x 128^2 doubles = 68.5 GB); bench.py's plan fails fast with the figures when the contexts do not fit."""
    import bench
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Se3dError, device_bytes
    b = device_bytes(256, 16, 128, 256)
    # round 3: large rigs keep two camera-column chunks of the pair table per frame (16 chunks at 256 cameras: 2 x 2 296 of the 32 640 pairs)
    whole = 16 * (256 * 255 // 2) * 128 * 128 * 8
    assert whole / 16 < b < whole / 5      # two chunks of ~1 / 16 of the pairs each, plus everything else
    assert whole < device_bytes(256, 16, 128, 256, options=dict(pair_chunks=-1)) < whole + 4.5e9   # the whole table (+ 2 GB of second-pass workspaces since round 4)
    assert device_bytes(256, 16, 128, 256, options=dict(pair_chunks=16)) == b
    with pytest.raises(Se3dError):
        device_bytes(256, 16, 128, 256, options=dict(pair_chunks=1))   # an option out of range: rejected like unsupported sizes
    assert device_bytes(16, 2, 8, 12) < device_bytes(16, 4, 8, 12) < 10e6
    with pytest.raises(Se3dError):
        device_bytes(1, 1, 8, 12)          # fewer than two cameras: se3d_create rejects it, so does the plan
    plan = bench.memory_plan("256cam128p", 4, 16, exchange=True)
    assert plan["total_bytes"] == 4 * b + plan["exchange_bytes"] and plan["total_bytes"] < 288e9
    # the bench's own preset for north_star's configuration, with the exchange buffers of an eight-rank run: fits one MI355X (288 GiB = 309 GB) with room to spare
    w = bench.WORKLOADS["256cam128p"]
    os.environ["WORLD_SIZE"] = "8"
    try:
        preset = bench.memory_plan("256cam128p", w["contexts"], w["seqs"] * bench.T_STEPS, exchange=True, free_bytes=int(288e9))
    finally:
        del os.environ["WORLD_SIZE"]
    assert preset["contexts"] == w["contexts"] and preset["total_bytes"] < 260e9 and preset["exchange_bytes"] < 5e9
    with pytest.raises(SystemExit) as e:
        bench.memory_plan("256cam128p", 4, 512, free_bytes=int(288e9))
    assert "needs" in str(e.value) and "GB" in str(e.value) and "--contexts" in str(e.value)


def test_create_failure_reason_is_retrievable(product_lib):
    """se3d_last_error(NULL): why the calling thread's last se3d_create returned NULL (round 2 only printed it)."""
    import ctypes as C
    from smartedgesensor3dhumanpose_amd import person_msgs as pm, rig as R
    from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors
    L = product_lib.lib()
    rig = R.extr_rig_16cam()
    rig.set_tensors(*rig_tensors(rig))
    rc = pm.Rig(rig.n_cam, pm.fptr(rig.Rt), pm.fptr(rig.K), pm.fptr(rig.F))
    prm = pm.Params(pm.POSE_SIMPLE, 0, 0.045)
    lim = pm.Limits(1, 8, 4, 4)       # max_hyp < max_det_per_cam: rejected before any device call
    L.se3d_create.restype = C.c_void_p
    L.se3d_last_error.restype = C.c_char_p
    L.se3d_last_error.argtypes = [C.c_void_p]
    assert not L.se3d_create(C.byref(rc), C.byref(prm), C.byref(lim), 0)
    assert b"unsupported sizes" in L.se3d_last_error(None)


def test_header_is_plain_c99(tmp_path):
    """The boundary is a C-ABI: include/se3d.h must compile as C (not only as C++) without warnings, and the POD mirrors of person_msgs
    must have the sizes the Python / C++ sides assume."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "se3d.h"\nint main(void) { return (sizeof(se3d_person3d) == 1688 && sizeof(se3d_person2d) == 428 && sizeof(se3d_keypoint3d) == 80) ? 0 : 1; }\n')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0
