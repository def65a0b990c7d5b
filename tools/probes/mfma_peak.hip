// mfma_peak.hip -- ceiling probe: back-to-back exact-fp32 MFMAs with register operands only (tuning aid, not product).
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o rl-x_amd/build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE>   // SHAPE 0: 32x32x2, 1: 16x16x4
__global__ __launch_bounds__(256) void k_peak(float* out, int iters, float a0, float b0) {
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  if (SHAPE == 0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

template <int NACC, int SHAPE>
void run(float* out, int wgs_per_cu, int iters) {
  const int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_peak<NACC, SHAPE>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1e-3f);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_peak<NACC, SHAPE>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1e-3f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 5.0 * grid * 4.0 * iters * 16.0 * NACC * (SHAPE == 0 ? 4096.0 : 2048.0);
  printf("shape %s nacc %d wgs/cu %d iters %d: %.1f us/launch  %.1f TFLOP/s\n", SHAPE == 0 ? "32x32x2" : "16x16x4", NACC, wgs_per_cu,
         iters, ms * 1e3 / 5, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int iters : {64, 2048}) {
    run<4, 0>(out, 1, iters); run<4, 0>(out, 2, iters); run<1, 0>(out, 1, iters); run<2, 0>(out, 1, iters);
    run<4, 1>(out, 1, iters); run<8, 1>(out, 2, iters); run<1, 1>(out, 1, iters);
  }
  return 0;
}
