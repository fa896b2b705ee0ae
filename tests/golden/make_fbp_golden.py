#!/usr/bin/env python3
"""Golden values for row K (constants): FUSION_BODY_PARTS joint indices and vel_sigmas, read from the REFERENCE's own
skeleton_3d/include/skeleton_3d/fusion_body_parts.h compiled in place into oracle/_ref/libfbp_ref.so (oracle/Makefile target `ref`,
shim oracle/fbp_ref_shim.cpp).  Run in the build container only (the reference does not travel); the JSON it writes is data."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle_py as O

NAMES = ["Nose", "Neck", "RShoulder", "RElbow", "RWrist", "LShoulder", "LElbow", "LWrist", "MidHip", "RHip", "RKnee", "RAnkle", "LHip", "LKnee",
         "LAnkle", "REye", "LEye", "REar", "LEar", "Head", "Belly", "NUM_KEYPOINTS"]
idx, vel = O.ref_fusion_constants()
out = {"source": "skeleton_3d/include/skeleton_3d/fusion_body_parts.h of the reference, compiled in place (oracle/_ref/libfbp_ref.so)",
       "indices": dict(zip(NAMES, idx)), "vel_sigmas": vel}
json.dump(out, open(os.path.join(os.path.dirname(__file__), "fusion_body_parts.json"), "w"), indent=1)
print(out)
