#!/usr/bin/env python3
"""Golden values for row K (constants and tables): the named scalar constants (`g_*`, MAX_DIST) of the reference's skeleton_3d and pose_prior
nodes and the per-pose-method tables (EdgeTPU_BodyParts_Simple / _H36M: kpParent, limbLength, limbLSigma, shoulderDist / shoulderSigma;
g_kp2kpFusion_idx_*), read out of the reference's sources by name.  The nodes themselves need ROS / Eigen / GTSAM and cannot be compiled
here, so the values are parsed, not executed; the JSON holds numbers keyed by the reference's names, nothing else.  Build container only."""
import json
import os
import re

S3D = "/root/reference/skeleton_3d/src/skeleton_3d_triang_mult_node.cpp"
PP = "/root/reference/pose_prior/src/pose_prior_mult_node.cpp"
FBP = json.load(open(os.path.join(os.path.dirname(__file__), "fusion_body_parts.json")))["indices"]
NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def scalars(path):
    out = {}
    for line in open(path):
        m = re.match(r"^(?:static\s+)?(?:const\s+|constexpr\s+)?(?:double|float|int)\s+(g_\w+|MAX_DIST)\s*=\s*(" + NUM + r")f?\s*;", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


def struct_tables(path, struct):
    src = open(path).read()
    body = src[src.index("struct " + struct):]
    body = body[:body.index("\n};")]
    out = {}
    for m in re.finditer(r"static\s+constexpr\s+(?:int|double)\s+(\w+)\[\d*\]\s*=\s*\{([^}]*)\}", body):
        out[m.group(1)] = [float(x) for x in re.findall(NUM, m.group(2))]
    for m in re.finditer(r"static\s+constexpr\s+double\s+((?:\w+\s*=\s*" + NUM + r"\s*,?\s*)+);", body):
        for n, v in re.findall(r"(\w+)\s*=\s*(" + NUM + ")", m.group(1)):
            out[n] = float(v)
    return out


def kp2fusion(path, name):
    src = open(path).read()
    body = src[src.index(name):]
    body = body[body.index("{") + 1:body.index("}")]
    return [FBP[x] for x in re.findall(r"FUSION_BODY_PARTS::(\w+)", body)]


out = {"source": "named constants and tables parsed from skeleton_3d_triang_mult_node.cpp / pose_prior_mult_node.cpp of the reference",
       "skeleton_3d": scalars(S3D), "pose_prior": scalars(PP),
       "simple": struct_tables(S3D, "EdgeTPU_BodyParts_Simple"), "h36m": struct_tables(S3D, "EdgeTPU_BodyParts_H36M")}


def bone_tables(path):
    """addBinaryFactors (PP:384-481): per branch (heights normalised / absolute) the ordered list of range factors
    [k1, k2, length, sigma coefficient, [joints that must be measured], [joints that must NOT be measured]]."""
    src = open(path).read()
    body = src[src.index("void addBinaryFactors("):]
    body = body[:body.index("\n}\n")]
    norm, absolute = body.split("else{", 1)
    out = {}
    for key, part in (("normalized", norm), ("absolute", absolute)):
        rows = []
        for m in re.finditer(r"if\(([^\n]*)\)\s*\n\s*graph\.emplace_shared<RangeFactor<Point3>>\(\s*(\d+)\s*,\s*(\d+)\s*,\s*(" + NUM + r")\s*,\s*noiseModel::Isotropic::Sigma\(1,\s*(" + NUM + ")", part):
            need = [FBP[x] for x in re.findall(r"(?<!!)joints_measured\[FUSION_BODY_PARTS::(\w+)\]", m.group(1))]
            deny = [FBP[x] for x in re.findall(r"!joints_measured\[FUSION_BODY_PARTS::(\w+)\]", m.group(1))]
            rows.append([int(m.group(2)), int(m.group(3)), float(m.group(4)), float(m.group(5)), need, deny])
        out[key] = rows
    return out


out["bones"] = bone_tables(PP)
out["simple"]["kp2fusion"] = kp2fusion(S3D, "g_kp2kpFusion_idx_simple[17]")
out["h36m"]["kp2fusion"] = kp2fusion(S3D, "g_kp2kpFusion_idx_h36m[17]")
json.dump(out, open(os.path.join(os.path.dirname(__file__), "pose_constants.json"), "w"), indent=1)
print(json.dumps(out)[:1500])
