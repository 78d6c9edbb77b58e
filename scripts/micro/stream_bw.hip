// HBM streaming microbench (gfx950): what do 1R+1W / 2R+1W / write-only / read-only float4 streams reach, and do
// non-temporal stores / loads change it?   hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_bw.hip -o scripts/micro/stream_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE, int ITERS>
__global__ __launch_bounds__(256) void k(f4* __restrict__ o, const f4* __restrict__ a, const f4* __restrict__ b, size_t n4) {
  const size_t base = (size_t)blockIdx.x * 256 * ITERS + threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const size_t i = base + (size_t)it * 256;
    if (i < n4) {
      if (MODE == 0) o[i] = a[i] * 1.5f;                                             // 1R + 1W
      if (MODE == 1) __builtin_nontemporal_store(a[i] * 1.5f, &o[i]);                // 1R + 1W, nt store
      if (MODE == 2) __builtin_nontemporal_store(__builtin_nontemporal_load(&a[i]) * 1.5f, &o[i]);   // nt load + nt store
      if (MODE == 3) o[i] = a[i] * b[i];                                             // 2R + 1W
      if (MODE == 4) __builtin_nontemporal_store(a[i] * b[i], &o[i]);                // 2R + 1W nt store
      if (MODE == 5) o[i] = (f4){1.f, 2.f, 3.f, 4.f};                                // write only
      if (MODE == 6) __builtin_nontemporal_store((f4){1.f, 2.f, 3.f, 4.f}, &o[i]);   // write only nt
      if (MODE == 7) acc += a[i];                                                    // read only
    }
  }
  if (MODE == 7 && acc.x == 12345.678f) o[0] = acc;
}
template <int MODE, int ITERS>
static void run(const char* name, f4* o, f4* a, f4* b, size_t n4, double bytes_per_elem) {
  const unsigned nb = (unsigned)((n4 + 256 * ITERS - 1) / (256 * ITERS));
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, ITERS>), dim3(nb), dim3(256), 0, 0, o, a, b, n4);
  hipEventRecord(s);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE, ITERS>), dim3(nb), dim3(256), 0, 0, o, a, b, n4);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= 10;
  printf("%-34s iters %d: %8.1f us  %7.1f GB/s\n", name, ITERS, ms * 1e3, n4 * 16.0 * bytes_per_elem / 16.0 / (ms * 1e-3) / 1e9);
}
int main() {
  const size_t n4 = (size_t)512 * 1024 * 1024 / 16;   // 512 MB per tensor
  f4 *o, *a, *b;
  hipMalloc(&o, n4 * 16); hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16);
  hipMemset(a, 0, n4 * 16); hipMemset(b, 0, n4 * 16);
  run<0, 4>("1R+1W", o, a, b, n4, 32); run<0, 8>("1R+1W", o, a, b, n4, 32); run<0, 1>("1R+1W", o, a, b, n4, 32);
  run<1, 4>("1R+1W nt-store", o, a, b, n4, 32); run<1, 8>("1R+1W nt-store", o, a, b, n4, 32);
  run<2, 4>("1R+1W nt-load nt-store", o, a, b, n4, 32);
  run<3, 4>("2R+1W", o, a, b, n4, 48); run<4, 4>("2R+1W nt-store", o, a, b, n4, 48);
  run<5, 4>("write only", o, a, b, n4, 16); run<6, 4>("write only nt", o, a, b, n4, 16);
  run<7, 4>("read only", o, a, b, n4, 16); run<7, 8>("read only", o, a, b, n4, 16);
  return 0;
}
