// se3d_stage2.hip — pose_prior refinement on gfx950 (placeholder until the LM kernels land in this round).
#include "se3d_device.h"
struct se3d_ctx;
extern "C" {
int se3d_refine_reset(se3d_ctx *) { return SE3D_ERR_STATE; }
int se3d_refine(se3d_ctx *, int32_t, int32_t, const se3d_person3d *, int32_t, const float *, double, se3d_fused_out *) { return SE3D_ERR_STATE; }
int se3d_run_stage2(se3d_ctx *, int32_t, const double *) { return SE3D_ERR_STATE; }
int se3d_fetch_stage2(se3d_ctx *, int32_t, se3d_fused_out *) { return SE3D_ERR_STATE; }
int se3d_pack_stage2_device(se3d_ctx *, void *, void *) { return SE3D_ERR_STATE; }
}
