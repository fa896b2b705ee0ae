import sys, os
sys.path.insert(0, os.getcwd())
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
from smartedgesensor3dhumanpose_amd import rig as R
import ctypes as C
hip = C.CDLL("libamdhip64.so")
n = C.c_int(0)
print("count before", hip.hipGetDeviceCount(C.byref(n)), n.value)
node = Skeleton3D(R.extr_rig_16cam(), max_frames=1, max_det_per_cam=8, max_hyp=8)
print("count after create", hip.hipGetDeviceCount(C.byref(n)), n.value, "lasterr", hip.hipGetLastError())
import torch
print(torch.cuda.device_count())
try:
    x = torch.zeros(4, device="cuda"); print("ok", x.sum().item())
except Exception as e:
    print("FAIL", e)
    print("HIP_VISIBLE_DEVICES", os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"), os.environ.get("CUDA_VISIBLE_DEVICES"))
