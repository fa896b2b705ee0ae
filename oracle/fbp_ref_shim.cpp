// C shim (ours) around the REFERENCE's fusion_body_parts.h (skeleton_3d/include/skeleton_3d/fusion_body_parts.h), compiled in place by
// oracle/Makefile: exports the joint indices and the vel_sigmas table the hot path uses (FBP:3-24, FBP:33; stage-2 track cost PP:84-101).
// Test infrastructure only: generates tests/golden/fusion_body_parts.json and cross-checks the oracle's restated constants.
#include <skeleton_3d/fusion_body_parts.h>

extern "C" int fbp_ref_num_keypoints() { return FUSION_BODY_PARTS::NUM_KEYPOINTS; }
extern "C" void fbp_ref_indices(int *out22) {  // order: the enum's declaration order, NUM_KEYPOINTS last
  const int v[22] = {FUSION_BODY_PARTS::Nose, FUSION_BODY_PARTS::Neck, FUSION_BODY_PARTS::RShoulder, FUSION_BODY_PARTS::RElbow, FUSION_BODY_PARTS::RWrist,
                     FUSION_BODY_PARTS::LShoulder, FUSION_BODY_PARTS::LElbow, FUSION_BODY_PARTS::LWrist, FUSION_BODY_PARTS::MidHip, FUSION_BODY_PARTS::RHip,
                     FUSION_BODY_PARTS::RKnee, FUSION_BODY_PARTS::RAnkle, FUSION_BODY_PARTS::LHip, FUSION_BODY_PARTS::LKnee, FUSION_BODY_PARTS::LAnkle,
                     FUSION_BODY_PARTS::REye, FUSION_BODY_PARTS::LEye, FUSION_BODY_PARTS::REar, FUSION_BODY_PARTS::LEar, FUSION_BODY_PARTS::Head,
                     FUSION_BODY_PARTS::Belly, FUSION_BODY_PARTS::NUM_KEYPOINTS};
  for (int i = 0; i < 22; ++i) out22[i] = v[i];
}
extern "C" void fbp_ref_vel_sigmas(double *out21) {
  for (int i = 0; i < FUSION_BODY_PARTS::NUM_KEYPOINTS; ++i) out21[i] = FUSION_BODY_PARTS::vel_sigmas[i];
}
