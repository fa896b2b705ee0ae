// se3d_stream.hip — device side of the streaming ingest path (se3d_stream_submit / se3d_stream_collect): the caller's ragged Person2D
// payload (S3D:1069: one synchronised set of Person2DList messages per frame) crosses PCIe WITHOUT padding, is expanded into the
// resident [frame][camera][slot] layout on the device, and only the PUBLISHED persons of the stage-2 walk travel back.
#include "se3d_device.h"

// Ragged -> resident: rag holds the persons of every frame back to back in (frame, camera, detection) order; off[f * (N + 1) + c] is
// the absolute index of the first person of (frame f, camera c), off[f * (N + 1) + N] the end of the frame.  One workgroup per
// (camera, frame); a person is 107 dwords.
__global__ __launch_bounds__(128) void se3d_k_stream_scatter(Se3dDev D, const se3d_person2d *__restrict__ rag, const int32_t *__restrict__ off) {
  const int c = blockIdx.x, f = blockIdx.y, N = D.d.N, P = D.d.P;
  const int o0 = off[f * (N + 1) + c], n = min(off[f * (N + 1) + c + 1] - o0, P);
  if (threadIdx.x == 0) const_cast<int32_t *>(D.nraw)[f * N + c] = n;
  constexpr int W = (int)(sizeof(se3d_person2d) / 4);
  const uint32_t *src = reinterpret_cast<const uint32_t *>(rag + o0);
  uint32_t *dst = reinterpret_cast<uint32_t *>(const_cast<se3d_person2d *>(D.raw) + ((size_t)(f * N + c)) * P);
  for (int i = threadIdx.x; i < n * W; i += 128) dst[i] = src[i];
}

// Published persons of a walk, compacted: exclusive prefix of the per-frame-slot counts (one workgroup), then one wavefront per frame
// slot copies its persons (fused and predicted) to their place in the dense arrays.
// mirror (optional): a second copy of the offsets — the slot's pinned HOST buffer when the persons are written there directly (the gather kernel
// reads the device copy: reads over PCIe are round trips)
__global__ __launch_bounds__(1024) void se3d_k_stream_scan(const int32_t *__restrict__ n_out, int32_t *__restrict__ out_off, int32_t *__restrict__ mirror, int n_frames, int cap) {
  __shared__ int part[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n_frames; base += 1024) {
    const int i = base + tid;
    const int v = i < n_frames ? min(n_out[i], cap) : 0;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    int offw = s_carry;
    for (int w = 0; w < wave; ++w) offw += part[w];
    if (i < n_frames) { out_off[i] = offw + inc - v; if (mirror) mirror[i] = offw + inc - v; }
    __syncthreads();
    if (tid == 1023) s_carry = offw + inc;
    __syncthreads();
  }
  if (tid == 0) { out_off[n_frames] = s_carry; if (mirror) mirror[n_frames] = s_carry; }
}

__global__ __launch_bounds__(64) void se3d_k_stream_gather(const se3d_person3d *__restrict__ fused, const se3d_person3d *__restrict__ pred, const int32_t *__restrict__ out_off,
                                                           int H, se3d_person3d *__restrict__ dense_fused, se3d_person3d *__restrict__ dense_pred) {
  const int f = blockIdx.x, lane = threadIdx.x;
  const int o0 = out_off[f], n = out_off[f + 1] - o0;
  constexpr int W = (int)(sizeof(se3d_person3d) / 4);
  const uint32_t *sf = reinterpret_cast<const uint32_t *>(fused + (size_t)f * H), *sp = reinterpret_cast<const uint32_t *>(pred + (size_t)f * H);
  uint32_t *df = reinterpret_cast<uint32_t *>(dense_fused + o0), *dp = reinterpret_cast<uint32_t *>(dense_pred + o0);
  for (int i = lane; i < n * W; i += 64) { df[i] = sf[i]; dp[i] = sp[i]; }
}

extern "C" void se3d_launch_stream_scatter(const Se3dDev &D, const se3d_person2d *rag, const int32_t *off, hipStream_t st) {
  hipLaunchKernelGGL(se3d_k_stream_scatter, dim3(D.d.N, D.d.B), dim3(128), 0, st, D, rag, off);
}
extern "C" void se3d_launch_stream_compact(const se3d_person3d *fused, const se3d_person3d *pred, const int32_t *n_out, int n_frames, int H, int32_t *out_off,
                                           int32_t *out_off_mirror, se3d_person3d *dense_fused, se3d_person3d *dense_pred, hipStream_t st) {
  hipLaunchKernelGGL(se3d_k_stream_scan, dim3(1), dim3(1024), 0, st, n_out, out_off, out_off_mirror, n_frames, H);
  hipLaunchKernelGGL(se3d_k_stream_gather, dim3(n_frames), dim3(64), 0, st, fused, pred, out_off, H, dense_fused, dense_pred);
}

// Dense pack of the published fused persons for the skeleton exchange (se3d_pack_stage2_walk_device_dense): exclusive prefix of the
// per-frame counts clipped at `capacity` persons — off[f] .. off[f + 1] is frame slot f's range in the dense array, off[n_frames] the number
// of persons written, off[n_frames + 1] the number that did not fit (0 in a correctly sized exchange) — then one wavefront per frame slot
// copies its persons.  An all-gather of padded [frame][slot] blocks moves 8 slots for 6 persons; this moves the persons.
__global__ __launch_bounds__(1024) void se3d_k_pack_scan(const int32_t *__restrict__ n_out, int32_t *__restrict__ off, int n_frames, int H, int capacity) {
  __shared__ int part[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n_frames; base += 1024) {
    const int i = base + tid;
    const int v = i < n_frames ? min(n_out[i], H) : 0;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    int offw = s_carry;
    for (int w = 0; w < wave; ++w) offw += part[w];
    if (i < n_frames) off[i] = min(offw + inc - v, capacity);
    __syncthreads();
    if (tid == 1023) s_carry = offw + inc;
    __syncthreads();
  }
  if (tid == 0) { off[n_frames] = min(s_carry, capacity); off[n_frames + 1] = s_carry - min(s_carry, capacity); }
}
__global__ __launch_bounds__(64) void se3d_k_pack_gather(const se3d_person3d *__restrict__ fused, const int32_t *__restrict__ off, int H, se3d_person3d *__restrict__ dense) {
  const int f = blockIdx.x, lane = threadIdx.x;
  const int o0 = off[f], n = off[f + 1] - o0;
  constexpr int W = (int)(sizeof(se3d_person3d) / 4);
  const uint32_t *sf = reinterpret_cast<const uint32_t *>(fused + (size_t)f * H);
  uint32_t *df = reinterpret_cast<uint32_t *>(dense + o0);
  for (int i = lane; i < n * W; i += 64) df[i] = sf[i];
}
extern "C" void se3d_launch_pack_dense(const se3d_person3d *fused, const int32_t *n_out, int n_frames, int H, int capacity, int32_t *off, se3d_person3d *dense, hipStream_t st) {
  hipLaunchKernelGGL(se3d_k_pack_scan, dim3(1), dim3(1024), 0, st, n_out, off, n_frames, H, capacity);
  hipLaunchKernelGGL(se3d_k_pack_gather, dim3(n_frames), dim3(64), 0, st, fused, off, H, dense);
}
