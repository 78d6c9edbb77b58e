// Microbenchmark: how many of a wave's OWN VALU / LDS instructions fit in the shadow of its v_mfma_f32_16x16x4_f32 stream?
// One loop iteration = 8 MFMAs, each followed by K v_add_f32 (and optionally one ds_read_b32 per MFMA); 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma_coissue.hip -o mfma_coissue && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int LDS, int NW>
__global__ __launch_bounds__(64 * NW) void k(float* out, int iters, long long* clk) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 64 * NW) smem[i] = i;
  __syncthreads();
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = tid * 1e-3f, b = 2.f;
  float x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned la = (tid & 63) * 4;
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int r = 0; r < K; ++r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(i + r) & 7]) : "v"(b));
      if (LDS) asm volatile("ds_read_b32 %0, %1" : "=v"(l[i]) : "v"(la));
    }
    if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long c1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i] + l[i];
  out[blockIdx.x * 64 * NW + tid] = s;
  if (tid == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}
template <int K, int LDS, int NW>
void run(float* out, long long* clk) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)k<K, LDS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<K, LDS, NW><<<256, 64 * NW, 100 * 1024>>>(out, 10, clk);
  k<K, LDS, NW><<<256, 64 * NW, 100 * 1024>>>(out, iters, clk);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("waves/SIMD %d  VALU per MFMA %d  ds_read per MFMA %d: %.1f cycles per MFMA of this wave (pipe needs %d)\n", NW / 4, K, LDS,
         (double)h / (8.0 * iters), 32 * NW / 4);
}
int main() {
  float* out; long long* clk;
  hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&clk, 16);
  run<0, 0, 4>(out, clk); run<1, 0, 4>(out, clk); run<2, 0, 4>(out, clk); run<4, 0, 4>(out, clk); run<6, 0, 4>(out, clk); run<7, 0, 4>(out, clk); run<8, 0, 4>(out, clk);
  run<0, 1, 4>(out, clk); run<2, 1, 4>(out, clk);
  run<0, 0, 8>(out, clk); run<1, 0, 8>(out, clk); run<2, 0, 8>(out, clk); run<4, 0, 8>(out, clk); run<6, 0, 8>(out, clk); run<8, 0, 8>(out, clk);
  run<0, 1, 8>(out, clk); run<2, 1, 8>(out, clk);
  return 0;
}
