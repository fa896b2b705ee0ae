#!/usr/bin/env python3
"""Golden values of the demo configuration (config 1 of the benchmark): the arguments / parameters of the reference's
pose_prior/launch/pose_triangulate_demo.launch (DEMO:2-28).  Run in the build container; the JSON holds the values only."""
import json
import os
import xml.etree.ElementTree as ET

root = ET.parse("/root/reference/pose_prior/launch/pose_triangulate_demo.launch").getroot()
args = {a.get("name"): a.get("default") for a in root.iter("arg") if a.get("default") is not None}
methods = sorted({p.get("value") for p in root.iter("param") if p.get("name") == "pose_method"})
out = {"source": "pose_prior/launch/pose_triangulate_demo.launch of the reference",
       "norm_height": args["norm_height"] == "true", "max_epi_dist": float(args["max_epi_dist"]), "pose_methods": methods,
       "n_cameras": len(args["cameras"].strip("[]").split(","))}
json.dump(out, open(os.path.join(os.path.dirname(__file__), "demo_params.json"), "w"), indent=1)
print(out)
