// Exhaustive check of cheaper correctly-rounded float32 square-root sequences against sqrtf over every float in [2^-60, 2^60] (the range of
// se3d_sqrt_core).  hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -o /tmp/sqrt_candidates tools/microbench/sqrt_candidates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float cand_a(float x) {   // rsq, two multiplies, one Newton correction: 5 instructions
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y, h = 0.5f * y;
  const float r = __builtin_fmaf(-s0, s0, x);
  return __builtin_fmaf(r, h, s0);
}
__device__ __forceinline__ float cand_b(float x) {   // the same with a second correction: 7 instructions
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y, h = 0.5f * y;
  const float r = __builtin_fmaf(-s0, s0, x);
  const float s1 = __builtin_fmaf(r, h, s0);
  const float r1 = __builtin_fmaf(-s1, s1, x);
  return __builtin_fmaf(r1, h, s1);
}
__device__ __forceinline__ float cand_c(float x) {   // refine h first (Goldschmidt style): 7 instructions
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y, h0 = 0.5f * y;
  const float e = __builtin_fmaf(-s0, h0, 0.5f);
  const float s1 = __builtin_fmaf(s0, e, s0), h1 = __builtin_fmaf(h0, e, h0);
  const float r = __builtin_fmaf(-s1, s1, x);
  return __builtin_fmaf(r, h1, s1);
}
__device__ __forceinline__ float cand_d(float x) {   // hardware sqrt + one Newton correction with h = 0.5 * rcp: two transcendental instructions
  const float s0 = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rcpf(s0);
  const float r = __builtin_fmaf(-s0, s0, x);
  return __builtin_fmaf(r, h, s0);
}
// the rows kernel's shared reciprocal RN(1 / root) from the same v_rsq_f32 (the division refinement seeded with it, a = 1)
__device__ __forceinline__ float rcp_seeded(float b, float r0) {
  const float e0 = __builtin_fmaf(-b, r0, 1.f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const float q0 = r1;
  const float e1 = __builtin_fmaf(-b, q0, 1.f);
  const float q1 = __builtin_fmaf(e1, r1, q0);
  const float e2 = __builtin_fmaf(-b, q1, 1.f);
  return __builtin_fmaf(e2, r1, q1);
}
__global__ void k(unsigned long long *out) {
  const unsigned long long tid = (unsigned long long)blockIdx.x * 256 + threadIdx.x, nthr = (unsigned long long)gridDim.x * 256;
  unsigned long long b[4] = {0, 0, 0, 0};
  for (unsigned long long u = 0x21800000ull + tid; u <= 0x5d800000ull; u += nthr) {
    const float x = __uint_as_float((unsigned)u);
    const unsigned ref = __float_as_uint(sqrtf(x));
    b[0] += __float_as_uint(cand_a(x)) != ref;
    b[1] += __float_as_uint(cand_b(x)) != ref;
    b[2] += __float_as_uint(cand_c(x)) != ref;
    b[3] += __float_as_uint(cand_d(x)) != ref;
    {   // ONE Newton step on the rsq seed: is it already RN(1 / root)?  (then the quotient is Markstein's three instructions behind it)
      const float root = __uint_as_float(ref), y = __builtin_amdgcn_rsqf(x);
      const float r1 = __builtin_fmaf(__builtin_fmaf(-root, y, 1.f), y, y);
      if (__float_as_uint(r1) != __float_as_uint(1.0f / root)) {
        atomicAdd(&out[20], 1ull);
        if ((ref & 0x7fffffu) != 0x7fffffu) atomicAdd(&out[21], 1ull);
      }
    }
    const float root = __uint_as_float(ref), rc = rcp_seeded(root, __builtin_amdgcn_rsqf(x));
    if (__float_as_uint(rc) != __float_as_uint(1.0f / root)) {
      const unsigned long long at = atomicAdd(&out[4], 1ull);
      if (at < 6) { out[5 + 2 * at] = u; out[6 + 2 * at] = ((unsigned long long)__float_as_uint(rc) << 32) | __float_as_uint(1.0f / root); }
    }
  }
  for (int i = 0; i < 4; ++i) if (b[i]) atomicAdd(&out[i], b[i]);
}
int main() {
  unsigned long long *d, h[24] = {0};
  (void)hipMalloc(&d, sizeof(h)); (void)hipMemset(d, 0, sizeof(h));
  hipLaunchKernelGGL(k, dim3(256 * 64), dim3(256), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("floats checked: %llu\nmismatches vs sqrtf: a (rsq + 1 correction) %llu, b (rsq + 2 corrections) %llu, c (refined h) %llu, d (sqrt + rcp correction) %llu\n",
         0x5d800000ull - 0x21800000ull + 1, h[0], h[1], h[2], h[3]);
  printf("one Newton step on the rsq seed vs RN(1 / root): %llu mismatches, of them with a root whose significand is not all ones: %llu\n", h[20], h[21]);
  printf("reciprocal of the root from the same rsq (a = 1): %llu mismatches\n", h[4]);
  for (int i = 0; i < 6 && i < (int)h[4]; ++i) printf("   x = 0x%08llx: got 0x%08llx, 1 / root = 0x%08llx\n", h[5 + 2 * i], h[6 + 2 * i] >> 32, h[6 + 2 * i] & 0xffffffffull);
  return 0;
}
