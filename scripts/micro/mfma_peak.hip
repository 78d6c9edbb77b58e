// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate and shader clock under MFMA load (hipcc --offload-arch=gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, long long* clk) {
  f32x4 acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
  float* out; long long* clk;
  hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&clk, 16);
  for (int wgs : {256, 2048}) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<wgs, 512>>>(out, 10, clk);
    hipEventRecord(e0); k<<<wgs, 512>>>(out, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double fl = 2.0 * 16 * 16 * 4 * 32.0 * iters * 8 * wgs;
    printf("wgs %d: %.3f ms  %.1f TFLOP/s   clock64 %lld wall(100MHz) %lld -> clock64 rate %.1f MHz; cycles/mfma/SIMD %.2f\n", wgs, ms,
           fl / ms / 1e9, h[0], h[1], (double)h[0] / h[1] * 100.0, (double)h[0] / (32.0 * iters * 2));
  }
  return 0;
}
