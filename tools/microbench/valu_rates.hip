// Issue-rate microbenchmark for the instruction mix of the pair-cost kernels (gfx950): cycles per wave64 instruction per SIMD at W
// wavefronts per SIMD, for the instruction forms the inner loop uses.  hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(64) void k_rate(float *out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  float b = seed * 0.5f + 1.f, c = seed * 0.25f + 0.5f;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // v_fmac_f32 (VOP2), 8 independent chains
      REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                        "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 1) {  // v_fma_f32 (VOP3) with neg / abs modifiers, 3 VGPR sources
      REP8(asm volatile("v_fma_f32 %0, -%8, %9, |%0|\n v_fma_f32 %1, -%8, %9, |%1|\n v_fma_f32 %2, -%8, %9, |%2|\n v_fma_f32 %3, -%8, %9, |%3|\n"
                        "v_fma_f32 %4, -%8, %9, |%4|\n v_fma_f32 %5, -%8, %9, |%5|\n v_fma_f32 %6, -%8, %9, |%6|\n v_fma_f32 %7, -%8, %9, |%7|\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 2) {  // v_mul_f32 + v_add_f32 (VOP2)
      REP8(asm volatile("v_mul_f32 %0, %8, %0\n v_add_f32 %1, %9, %1\n v_mul_f32 %2, %8, %2\n v_add_f32 %3, %9, %3\n"
                        "v_mul_f32 %4, %8, %4\n v_add_f32 %5, %9, %5\n v_mul_f32 %6, %8, %6\n v_add_f32 %7, %9, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 3) {  // v_add_f64, 4 independent chains (x2 to make 8 instructions)
      REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                        "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)b));)
    } else if (KIND == 4) {  // v_cvt_f64_f32
      REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                        "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                        : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if (KIND == 5) {  // the pair-joint item: 6 mul, 5 add, 2 mul(abs), 4 fmac, 4 fma(neg/abs), cvt, add_f64 = 23 instructions, dependent as in the kernel
      REP8(asm volatile(
          "v_mul_f32 %0, %8, %9\n v_mul_f32 %1, %8, %9\n v_add_f32 %1, %1, %9\n v_add_f32 %0, %0, %1\n"      // n1
          "v_mul_f32 %2, %8, %9\n v_mul_f32 %3, %8, %9\n v_add_f32 %3, %3, %9\n v_add_f32 %2, %2, %3\n"      // n2
          "v_mul_f32 %4, %8, |%0|\n v_fma_f32 %5, -%9, %4, |%0|\n v_fmac_f32 %4, %5, %8\n v_fma_f32 %5, -%9, %4, |%0|\n v_fmac_f32 %4, %5, %8\n"  // d1
          "v_mul_f32 %6, %8, |%2|\n v_fma_f32 %7, -%9, %6, |%2|\n v_fmac_f32 %6, %7, %8\n v_fma_f32 %7, -%9, %6, |%2|\n v_fmac_f32 %6, %7, %8\n"  // d2
          "v_add_f32 %4, %4, %6\n v_mul_f32 %5, %8, %9\n v_mul_f32 %4, %4, %5\n"
          "v_cvt_f64_f32 %10, %4\n v_add_f64 %11, %11, %10\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(d1), "v"(d0));)
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
}

// The pair-joint item with its det_a operand (32 B) read from LDS by two ds_read_b128 per item, half-wave uniform addresses as in
// se3d_k_pairs_rows, one item of prefetch distance.  NREAD = 0 / 1 / 2 reads per item.
template <int NREAD>
__global__ __launch_bounds__(256) void k_item_lds(float *out, int iters, float seed) {
  __shared__ float4 lds[2 * 17 * 48];
  for (int i = threadIdx.x; i < 2 * 17 * 48; i += 256) lds[i] = make_float4(seed + i, seed, 1.f, 2.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  float b = seed * 0.5f + 1.f, c = seed * 0.25f + 0.5f;
  double d0 = a0, d1 = a1;
  const float4 *p = lds + 2 * (wave * 12 + half);
  float4 u0 = p[0], u1 = p[1];
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float4 n0 = u0, n1 = u1;
      if (NREAD >= 1) n0 = p[2 * 48 * ((k + 1) & 7)];
      if (NREAD >= 2) n1 = p[2 * 48 * ((k + 1) & 7) + 1];
      asm volatile(
          "v_mul_f32 %0, %8, %12\n v_mul_f32 %1, %8, %13\n v_add_f32 %1, %1, %9\n v_add_f32 %0, %0, %1\n"
          "v_mul_f32 %2, %8, %14\n v_mul_f32 %3, %8, %15\n v_add_f32 %3, %3, %9\n v_add_f32 %2, %2, %3\n"
          "v_mul_f32 %4, %8, |%0|\n v_fma_f32 %5, -%9, %4, |%0|\n v_fmac_f32 %4, %5, %8\n v_fma_f32 %5, -%9, %4, |%0|\n v_fmac_f32 %4, %5, %8\n"
          "v_mul_f32 %6, %8, |%2|\n v_fma_f32 %7, -%9, %6, |%2|\n v_fmac_f32 %6, %7, %8\n v_fma_f32 %7, -%9, %6, |%2|\n v_fmac_f32 %6, %7, %8\n"
          "v_add_f32 %4, %4, %6\n v_mul_f32 %5, %8, %9\n v_mul_f32 %4, %4, %5\n"
          "v_cvt_f64_f32 %10, %4\n v_add_f64 %11, %11, %10\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b), "+v"(c), "+v"(d1), "+v"(d0)
          : "v"(u0.x), "v"(u0.y), "v"(u1.x), "v"(u1.y)
          : "memory");
      u0 = n0; u1 = n1;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1) + u0.z + u1.w;
}

template <int NREAD>
void run_lds(int wgs_per_cu) {
  const int blocks = 256 * wgs_per_cu, iters = 2000;
  float *out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_item_lds<NREAD>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_item_lds<NREAD>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double items_per_simd = (double)wgs_per_cu * iters * 8;
  printf("item + %d ds_read_b128 per item, %d waves/SIMD: %.3f ms, %.2f ns per item per SIMD (%.3f ns per VALU instruction)\n", NREAD, wgs_per_cu, ms,
         ms * 1e6 / items_per_simd, ms * 1e6 / items_per_simd / 23.0);
  hipFree(out);
}

template <int KIND>
double run(int waves_per_simd, int instr_per_iter, const char *name) {
  const int n_cu = 256, blocks = n_cu * 4 * waves_per_simd, iters = 4000;
  float *out;
  hipMalloc(&out, sizeof(float) * blocks * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)waves_per_simd * iters * instr_per_iter;
  const double ns_per_instr = ms * 1e6 / instr_per_simd;
  printf("%-28s waves/SIMD %d: %.3f ms, %.3f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz (%.2f at 2.0)\n", name, waves_per_simd, ms, ns_per_instr,
         ns_per_instr * 2.4, ns_per_instr * 2.0);
  hipFree(out);
  return ns_per_instr;
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>(w, 64, "v_fmac_f32 (VOP2)");
    run<1>(w, 64, "v_fma_f32 neg/abs (VOP3)");
    run<2>(w, 64, "v_mul_f32 / v_add_f32");
    run<3>(w, 64, "v_add_f64");
    run<4>(w, 64, "v_cvt_f64_f32");
    run<5>(w, 8 * 23, "pair-joint item (23 instr)");
  }
  for (int w : {2, 3, 4}) { run_lds<0>(w); run_lds<1>(w); run_lds<2>(w); }
  return 0;
}
