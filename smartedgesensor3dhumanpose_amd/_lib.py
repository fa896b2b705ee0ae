"""Loader of the product library libse3d_hip.so (C-ABI of include/se3d.h).  No fallback: a missing library is an error."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SE3D_LIB_OVERRIDE") or os.path.join(_HERE, "libse3d_hip.so")  # override: ablation/diagnostic builds only

# every entry point declared in include/se3d.h (checked against the header in tests/test_abi.py)
EXPORTS = [
    "se3d_rig_from_transforms", "se3d_create", "se3d_create_ex", "se3d_set_options", "se3d_get_options", "se3d_destroy", "se3d_last_error", "se3d_device_bytes", "se3d_device_bytes_ex", "se3d_version", "se3d_abi_version",
    "se3d_triangulate", "se3d_triangulate_batch", "se3d_stage_frames", "se3d_run_stage1", "se3d_sync", "se3d_fetch_stage1",
    "se3d_refine_reset", "se3d_refine_state_bytes", "se3d_refine_save", "se3d_refine_load", "se3d_refine", "se3d_run_stage2", "se3d_fetch_stage2",
    "se3d_hungarian", "se3d_get_pair_costs", "se3d_selftest_exact_math", "se3d_selftest_shared_quotient", "se3d_get_sweep_stats", "se3d_selftest_dlt32", "se3d_get_exact_list", "se3d_get_pair_chunks", "se3d_get_fusion_constants", "se3d_pack_stage1_device", "se3d_pack_stage2_device",
    "se3d_profile_enable", "se3d_profile_reset", "se3d_profile_get",
    "se3d_set_reprojection_cameras", "se3d_reproject", "se3d_reproject_resident", "se3d_reproject_resident_batch", "se3d_fetch_reprojection",
    "se3d_run_stage2_walk", "se3d_fetch_stage2_walk", "se3d_pack_stage2_walk_device", "se3d_stream_handle", "se3d_pack_stage2_walk_device_slots", "se3d_pack_stage2_walk_device_dense",
    "se3d_triangulate_assoc", "se3d_stage_assoc", "se3d_run_dlt", "se3d_fetch_dlt", "se3d_stream_submit", "se3d_stream_collect", "se3d_copy_stream_probe",
    "se3d_run_stage1_phases", "se3d_set_shard", "se3d_shard_bytes", "se3d_run_stage1_shard_pre", "se3d_run_stage1_shard_post", "se3d_run_stage2_shard_pre", "se3d_run_stage2_shard_post",
]


def source_hash():
    """Fingerprint of the kernel sources (csrc/*.hip, *.h, include/se3d.h): 16 hex digits.  tools/pmc_sq.py stamps the counter summaries it writes with
    it and bench.py uses a committed summary only when it matches (instruction counts are properties of the code they were collected on)."""
    import hashlib
    src_dir = os.path.join(_HERE, "csrc")
    if not os.path.isdir(src_dir):
        return None   # an installed package without its sources: no fingerprint (callers treat None as "does not match")
    files = sorted(os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".h"))) + [os.path.join(_HERE, "..", "include", "se3d.h")]
    h = hashlib.sha256()
    for f in files:
        if os.path.exists(f):   # (the header lives outside the package directory)
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "se3d.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", src_dir] + (["-B"] if force else []), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def _preload_host_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so.7; the system ROCm has one with the same SONAME.  Whichever is loaded first
    serves both, and torch does not find its GPUs on the system runtime.  When torch is installed, load its runtime first so that
    this library and torch (bench.py / shard.py: RCCL exchange, device buffers) can be used in one process in any import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libse3d_hip.so is missing (%s): build it with `make -C smartedgesensor3dhumanpose_amd/csrc` "
                               "or __graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
        _preload_host_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.se3d_create.restype = C.c_void_p
        L.se3d_create_ex.restype = C.c_void_p
        L.se3d_destroy.argtypes = [C.c_void_p]
        L.se3d_last_error.restype = C.c_char_p
        L.se3d_last_error.argtypes = [C.c_void_p]
        L.se3d_version.restype = C.c_char_p
        L.se3d_abi_version.restype = C.c_int32
        from .person_msgs import ABI_VERSION   # the struct layouts this mirror was written against (include/se3d.h: SE3D_ABI_VERSION)
        if L.se3d_abi_version() != ABI_VERSION:
            raise RuntimeError("libse3d_hip.so has struct layout version %d, the Python mirror %d (include/se3d.h: SE3D_ABI_VERSION): rebuild the library"
                               % (L.se3d_abi_version(), ABI_VERSION))
        L.se3d_stream_handle.restype = C.c_void_p
        L.se3d_stream_handle.argtypes = [C.c_void_p]
        for name in ("se3d_triangulate", "se3d_triangulate_batch", "se3d_stage_frames", "se3d_run_stage1", "se3d_sync", "se3d_fetch_stage1",
                     "se3d_refine_reset", "se3d_refine_state_bytes", "se3d_refine_save", "se3d_refine_load", "se3d_refine", "se3d_run_stage2", "se3d_fetch_stage2", "se3d_pack_stage1_device",
                     "se3d_pack_stage2_device", "se3d_hungarian", "se3d_get_pair_costs", "se3d_selftest_exact_math", "se3d_selftest_shared_quotient", "se3d_get_sweep_stats", "se3d_selftest_dlt32", "se3d_get_exact_list", "se3d_get_pair_chunks", "se3d_get_fusion_constants", "se3d_profile_enable", "se3d_profile_reset", "se3d_profile_get",
                     "se3d_set_reprojection_cameras", "se3d_reproject", "se3d_reproject_resident", "se3d_reproject_resident_batch", "se3d_fetch_reprojection",
                     "se3d_run_stage2_walk", "se3d_fetch_stage2_walk", "se3d_pack_stage2_walk_device", "se3d_pack_stage2_walk_device_slots", "se3d_pack_stage2_walk_device_dense",
                     "se3d_triangulate_assoc", "se3d_stage_assoc", "se3d_run_dlt", "se3d_fetch_dlt", "se3d_stream_submit", "se3d_stream_collect",
                     "se3d_run_stage1_phases", "se3d_set_shard", "se3d_run_stage1_shard_pre", "se3d_run_stage1_shard_post", "se3d_run_stage2_shard_pre", "se3d_run_stage2_shard_post"):
            getattr(L, name).restype = C.c_int
        L.se3d_copy_stream_probe.restype = C.c_int
        L.se3d_shard_bytes.restype = C.c_int64
        L.se3d_device_bytes.restype = C.c_int64
        L.se3d_device_bytes_ex.restype = C.c_int64
        L.se3d_set_options.restype = C.c_int
        L.se3d_get_options.restype = C.c_int
        L.se3d_refine_state_bytes.restype = C.c_int64
        L.se3d_refine_state_bytes.argtypes = [C.c_void_p]
        L.se3d_refine_save.restype = C.c_int
        L.se3d_refine_load.restype = C.c_int
        _lib = L
    return _lib
