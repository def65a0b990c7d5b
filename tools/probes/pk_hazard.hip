// pk_hazard.hip -- stand-alone attempt at the co-residency hazard of DESIGN.md section 4: a "victim" kernel whose packed-f32
// VALU results (v_pk_mul_f32 / v_pk_fma_f32) are consumed by the next instruction, checked lane by lane against the same
// arithmetic in scalar instructions, while an "aggressor" kernel streams v_mfma_f32_32x32x16_bf16 on a second stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/pk_hazard.hip -o rl-x_amd/build/pk_hazard && rl-x_amd/build/pk_hazard
// Prints the number of lanes whose packed result differed from the scalar one, with the aggressor off and on.
// RESULT (MI355X, round 2): 0 / 0 -- this miniature (packed chain fed from ordinary VGPRs) does NOT reproduce the hazard; in
// k_l1fwd_mfma the packed operands are MFMA accumulators (two per register pair) and the kernel keeps four waves per SIMD.  Kept
// as the starting point for a vendor report; the working reproducer is the library-level one below.
// (The library-level reproducer is: RLX_REPRO_PACKED_F32=1 python rl-x_amd/build.py --force; python tools/probes/l1fwd_victim.py 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// victim: the LayerNorm + ELU tail of k_l1fwd_mfma in miniature -- (z - mean) * rstd * gamma + beta on register pairs
// (packed), then a scalar consumer of the HIGH half; the same chain in scalar form next to it; 512 threads, ~45 KB of LDS
// like the real kernel so that the occupancy and the co-residency with the aggressor are comparable
__global__ __launch_bounds__(512, 2) void k_victim(const float* __restrict__ in, unsigned long long* __restrict__ bad,
                                                   float* __restrict__ sink, int iters) {
  extern __shared__ float smem[];
  const int t = threadIdx.x;
  smem[t] = in[t];
  __syncthreads();
  float acc_p = 0.f, acc_s = 0.f;
  unsigned long long nbad = 0;
  float z0 = in[(blockIdx.x * 512 + t) & 4095], z1 = in[(blockIdx.x * 512 + t + 7) & 4095];
  const float g0 = 1.0f + 0.01f * (t & 15), g1 = 0.9f + 0.02f * (t & 7), b0 = 0.1f, b1 = -0.2f;
  for (int it = 0; it < iters; ++it) {
    const float mean = smem[(t + it) & 511] * 0.01f;
    const float var = fabsf(smem[(t + 3 * it) & 511]) + 0.5f;
    const float rs = __frsqrt_rn(var);
    // packed chain
    f2 z = {z0, z1};
    f2 m2 = {mean, mean}, r2 = {rs, rs}, gg = {g0, g1}, bb = {b0, b1};
    f2 y = (z - m2) * r2;
    y = y * gg + bb;
    const float hp = fminf(0.f, y[1]);                 // consumer of the high half right behind the packed fma
    const float ep = __expf(hp) - 1.0f;
    const float op1 = y[1] > 0.f ? y[1] : ep;
    const float op0 = y[0] > 0.f ? y[0] : __expf(fminf(0.f, y[0])) - 1.0f;
    // scalar chain (asm barriers keep the compiler from merging it with the packed one)
    float s0 = z0, s1 = z1;
    asm volatile("" : "+v"(s0), "+v"(s1));
    float y0 = (s0 - mean) * rs, y1 = (s1 - mean) * rs;
    asm volatile("" : "+v"(y0), "+v"(y1));
    y0 = y0 * g0 + b0;
    y1 = y1 * g1 + b1;
    asm volatile("" : "+v"(y0), "+v"(y1));
    const float os1 = y1 > 0.f ? y1 : __expf(fminf(0.f, y1)) - 1.0f;
    const float os0 = y0 > 0.f ? y0 : __expf(fminf(0.f, y0)) - 1.0f;
    // (the two chains may round differently in the last bit; the hazard produced errors of 1e-3 .. 1e-1)
    if (fabsf(op1 - os1) > 1e-4f * (1.0f + fabsf(os1))) ++nbad;
    if (fabsf(op0 - os0) > 1e-4f * (1.0f + fabsf(os0))) ++nbad;
    acc_p += op0 + op1;
    acc_s += os0 + os1;
    z0 = z0 * 0.999f + 0.001f * os1;
    z1 = z1 * 0.998f - 0.001f * os0;
  }
  if (nbad) atomicAdd(bad, nbad);
  sink[blockIdx.x * 512 + t] = acc_p - acc_s;
}

// aggressor: back-to-back bf16 MFMAs on four accumulators (VGPR form), 256 threads, 48 KB of static LDS like k_gemm_bx
__global__ __launch_bounds__(256, 2) void k_aggressor(const float* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ char lds[49152];
  lds[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  s8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x * 3 + i + lds[i]); b[i] = (short)(threadIdx.x * 5 - i); }
  f32x16 acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q], 0, 0, 0);
  }
  float s = 0.f;
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s + in[0];
}

int main() {
  float *din, *dsink, *dout;
  unsigned long long* dbad;
  CK(hipMalloc(&din, 4096 * 4));
  CK(hipMalloc(&dsink, 1024 * 512 * 4));
  CK(hipMalloc(&dout, 1024 * 256 * 4));
  CK(hipMalloc(&dbad, 8));
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 32768.0f - 1.0f;
  CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)k_victim, hipFuncAttributeMaxDynamicSharedMemorySize, 46 * 1024));
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0));
  CK(hipStreamCreate(&s1));
  for (int with = 0; with < 2; ++with) {
    unsigned long long total = 0;
    for (int rep = 0; rep < 20; ++rep) {
      CK(hipMemsetAsync(dbad, 0, 8, s0));
      CK(hipStreamSynchronize(s0));
      if (with) hipLaunchKernelGGL(k_aggressor, dim3(512), dim3(256), 0, s1, din, dout, 20000);
      for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_victim, dim3(512), dim3(512), 45 * 1024, s0, din, dbad, dsink, 2000);
      CK(hipDeviceSynchronize());
      unsigned long long b = 0;
      CK(hipMemcpy(&b, dbad, 8, hipMemcpyDeviceToHost));
      total += b;
    }
    printf("aggressor %s: %llu lanes with a packed result different from the scalar one (20 x 4 victim launches)\n",
           with ? "ON " : "off", total);
  }
  return 0;
}
