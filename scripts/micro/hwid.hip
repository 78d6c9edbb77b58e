// Which SIMD / wave slot does wave i of a 512-thread workgroup land on?  (decides which two waves of a k_wino workgroup share
// a SIMD, i.e. which pairs must run their transform / multiply phases in opposite order)
//   hipcc --offload-arch=gfx950 -O2 hwid.hip -o hwid && ./hwid
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(unsigned* out) {
  extern __shared__ float smem[];
  smem[threadIdx.x] = 1.f;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
}
int main() {
  unsigned* d; const int nb = 1024;
  hipMalloc(&d, nb * 8 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(nb), dim3(512), 108 * 1024, 0, d);
  static unsigned h[nb * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int hist[8][4] = {};
  for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
  for (int w = 0; w < 8; ++w) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int b = 0; b < 4; ++b) { printf("wg %d:", b); for (int w = 0; w < 8; ++w) printf("  w%d simd%u slot%u cu%u", w, (h[b*8+w] >> 4) & 3, h[b*8+w] & 15, (h[b*8+w] >> 8) & 15); printf("\n"); }
  return 0;
}
