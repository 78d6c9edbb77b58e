// Microbenchmark: how much does a co-resident wave's VALU / LDS / VMEM activity slow a wave streaming v_mfma_f32_16x16x4_f32 on
// the same SIMD?  512-thread workgroups, 1 per CU; waves 0-3 (slot 0 of each SIMD) run MFMAs, waves 4-7 (slot 1) run `mode`.
//   hipcc --offload-arch=gfx950 -O3 mfma_mix.hip -o mfma_mix && ./mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int MREAD>
__global__ __launch_bounds__(512) void k(float* out, const float4* gsrc, int iters, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 16384; i += 512) smem[i] = i * 1e-3f;
  __syncthreads();
  if (wave < 4) {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = tid * 1e-3f;
    const float2* vb = reinterpret_cast<const float2*>(smem + wave * 2048 + lane * 2);
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float2 b = make_float2(1.f, 2.f);
        if (MREAD) b = vb[s * 64 + (it & 3) * 256];      // one ds_read_b64 per 8 MFMAs, consumed at once (worst case: no lookahead)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[s * 8 + 2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.x, acc[s * 8 + 2 * i], 0, 0, 0);
          acc[s * 8 + 2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.y, acc[s * 8 + 2 * i + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    long long c1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
  } else {
    float x0 = tid, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
    const int n = iters * 8;     // long enough to outlast the MFMA waves in every mode
    if (MODE == 0) return;
    if (MODE == 1) {
      f32x4 acc[32];
      for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, acc[i], 0, 0, 0);
      for (int i = 0; i < 32; ++i) x2 += acc[i][0];
    }
    if (MODE == 2) {   // VALU stream, 8 independent chains
      for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { x0 = x0 * 1.0001f + x1; x1 = x1 * 0.999f + x2; x2 += x3; x3 -= x4; x4 = x4 * 1.01f + x5; x5 += x6; x6 -= x7; x7 += x0; }
      }
    }
    if (MODE == 3 || MODE == 4) {   // LDS reads: conflict-free / 4-way conflicts
      const float* p = smem + 8192 + (MODE == 3 ? lane : lane * 4 % 64 + 32 * (lane / 16));
      for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x0 += p[r * 64 + (it & 7) * 512];
      }
    }
    if (MODE == 5) {   // LDS writes
      float* p = smem + 8192 + tid - 256;
      for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) p[r * 256 + (it & 3) * 2048] = x0 + r;
        x0 += 1.f;
      }
    }
    if (MODE == 6) {   // 16-byte global loads from an L2-resident buffer, consumed immediately
      const float4* g = gsrc + (blockIdx.x & 63) * 4096 + (tid - 256);
      for (int it = 0; it < n / 4; ++it) {
        float4 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = g[r * 256 + (it & 1) * 2048];
#pragma unroll
        for (int r = 0; r < 8; ++r) x0 += v[r].x + v[r].w;
      }
    }
    if (MODE == 7) {   // transform-like mix: 8 ds_read2 + 32 VALU + 16 ds_write per round
      const float* p = smem + 8192 + lane * 2 + 3;
      float* qv = smem + 12288 + (tid - 256);
      for (int it = 0; it < n / 2; ++it) {
        float d[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = p[(r >> 2) * 40 + (r & 3) + (it & 3) * 256];
        float t[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) { t[c] = d[c] - d[8 + c]; t[4 + c] = d[4 + c] + d[8 + c]; t[8 + c] = d[8 + c] - d[4 + c]; t[12 + c] = d[4 + c] - d[12 + c]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          qv[(4 * r + 0) * 256] = t[4 * r] - t[4 * r + 2]; qv[(4 * r + 1) * 256] = t[4 * r + 1] + t[4 * r + 2];
          qv[(4 * r + 2) * 256] = t[4 * r + 2] - t[4 * r + 1]; qv[(4 * r + 3) * 256] = t[4 * r + 1] - t[4 * r + 3];
        }
      }
    }
    out[blockIdx.x * 512 + tid] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
}

template <int MODE, int MREAD>
void run(const char* name, float* out, const float4* g, long long* clk) {
  const int iters = 4000, wgs = 256;
  hipFuncSetAttribute((const void*)k<MODE, MREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<MODE, MREAD><<<wgs, 512, 100 * 1024>>>(out, g, 10, clk);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE, MREAD><<<wgs, 512, 100 * 1024>>>(out, g, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("%-44s lds-read-in-mfma-loop %d: cycles per MFMA %.2f (ideal 32)   kernel %.3f ms\n", name, MREAD, (double)h / (32.0 * iters), ms);
}
int main() {
  float* out; long long* clk; float4* g;
  hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&clk, 16); hipMalloc(&g, 64 * 4096 * 16 * 2); hipMemset(g, 0, 64 * 4096 * 16 * 2);
  run<0, 0>("other wave: idle", out, g, clk);        run<0, 1>("other wave: idle", out, g, clk);
  run<1, 0>("other wave: MFMA (cycles are for 2 streams)", out, g, clk);
  run<2, 0>("other wave: VALU stream", out, g, clk);   run<2, 1>("other wave: VALU stream", out, g, clk);
  run<3, 0>("other wave: ds_read_b32", out, g, clk);   run<3, 1>("other wave: ds_read_b32", out, g, clk);
  run<4, 0>("other wave: ds_read_b32 4-way conflict", out, g, clk); run<4, 1>("other wave: ds_read_b32 4-way conflict", out, g, clk);
  run<5, 0>("other wave: ds_write_b32", out, g, clk);  run<5, 1>("other wave: ds_write_b32", out, g, clk);
  run<6, 0>("other wave: global_load_dwordx4 (L2)", out, g, clk); run<6, 1>("other wave: global_load_dwordx4 (L2)", out, g, clk);
  run<7, 0>("other wave: transform-like mix", out, g, clk); run<7, 1>("other wave: transform-like mix", out, g, clk);
  return 0;
}
