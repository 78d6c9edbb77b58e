// Microbenchmark: the fp32 matrix pipe under its power limit — v_mfma_f32_16x16x4_f32 back to back with (0) zero operands, (1) one
// constant operand pair per lane, (2) 8 x 4 different random operand registers per lane cycling between consecutive MFMAs (what a real
// GEMM feeds it).  Same instruction stream in all three; only the data differ.  hipcc --offload-arch=gfx950 -O3 mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f; }
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, long long* clk) {
  f32x4 acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  float a[8], b[4];
  for (int i = 0; i < 8; ++i) a[i] = MODE == 0 ? 0.f : (MODE == 1 ? 0.37f : rnd(s));
  for (int i = 0; i < 4; ++i) b[i] = MODE == 0 ? 0.f : (MODE == 1 ? -0.61f : rnd(s));
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 7], b[(i >> 3) & 3], acc[i], 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  float t = 0.f;
  for (int i = 0; i < 32; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = t;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int MODE>
void run(float* out, long long* clk, const char* name) {
  const int wgs = 256, iters = 40000;    // ~0.6 s per launch: long enough for the power controller to settle
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<wgs, 512>>>(out, 1000, clk);
  hipEventRecord(e0); k<MODE><<<wgs, 512>>>(out, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double fl = 2.0 * 16 * 16 * 4 * 32.0 * iters * 8 * wgs;
  printf("%-22s %.1f ms  %.1f TFLOP/s  shader clock %.0f MHz  cycles/MFMA/SIMD %.2f\n", name, ms, fl / ms / 1e9, (double)h[0] / h[1] * 100.0,
         (double)h[0] / (32.0 * iters * 2));
}
int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(out, clk, "zero operands");
    run<1>(out, clk, "constant operands");
    run<2>(out, clk, "random operands");
  }
  return 0;
}
