"""MI355X-native hot path of AIS-Bonn/SmartEdgeSensor3DHumanPose: skeleton_3d association + triangulation and
pose_prior refinement behind a C-ABI (include/se3d.h).  See DESIGN.md."""
from . import person_msgs, rig, scene  # noqa: F401

__version__ = "0.1.0"
