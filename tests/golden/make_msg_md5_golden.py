#!/usr/bin/env python3
"""Golden values for the person_msgs wire format: the ROS message md5 sums of the REFERENCE's own
person_msgs/msg/*.msg files, computed here with genmsg's rule (comments and blank lines dropped, "type name" per field, a field of message
type replaced by that type's md5 with the array suffix dropped).  An md5 pins field names, types and order; the JSON holds the sums only.
Run in the build container (the reference does not travel)."""
import hashlib
import json
import os

REF = "/root/reference/person_msgs/msg"
KNOWN = {"std_msgs/Header": "2176decaecbce78abc3b96ef049fabed", "geometry_msgs/Point": "4a842b65f413084dc2b10fb484ea7f17",
         "geometry_msgs/Vector3": "4a842b65f413084dc2b10fb484ea7f17", "geometry_msgs/Quaternion": "a779879fadf0160734f906b8c19c7004",
         "geometry_msgs/Pose": "e45d45a5a1ce597b249e23fb30fc871f"}   # published sums of the ROS common messages
BUILTIN = {"bool", "int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64", "string", "time", "duration", "char", "byte"}


def md5_of(name):
    if name in KNOWN:
        return KNOWN[name]
    lines = []
    for raw in open(os.path.join(REF, name.split("/")[1] + ".msg")):
        line = raw.split("#")[0].strip()
        if not line:
            continue
        ftype, fname = line.split()[:2]
        base = ftype.split("[")[0]
        if base in BUILTIN:
            lines.append("%s %s" % (ftype, fname))
        else:
            lines.append("%s %s" % (md5_of(base if "/" in base else "person_msgs/" + base), fname))
    return hashlib.md5("\n".join(lines).encode()).hexdigest()


out = {"source": "person_msgs/msg/*.msg of the reference, genmsg md5 rule", "md5": {}}
for f in sorted(x for x in os.listdir(REF) if x.endswith(".msg")):
    t = "person_msgs/" + f[:-4]
    out["md5"][t] = md5_of(t)
json.dump(out, open(os.path.join(os.path.dirname(__file__), "person_msgs_md5.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
