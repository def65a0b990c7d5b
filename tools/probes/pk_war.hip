// Stand-alone reproducer for docs/PACKED_F32_HAZARD.md (MI355X / gfx950, ROCm 7.2).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast tools/probes/pk_war.hip -o /tmp/pk_war && /tmp/pk_war          # vectorizers on
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -fno-slp-vectorize -fno-vectorize tools/probes/pk_war.hip -o /tmp/pk_war_nv && /tmp/pk_war_nv
//
// Victim: the dZ loop of sac.hip's k_head_bwd, verbatim (thread <-> hidden column, W and the transposed d_out tile in LDS, sixteen
// fmaf chains per thread).  With the SLP vectorizer on, hipcc pairs the chains into v_pk_fma_f32, software-pipelines the loop over
// two output columns and lets the second column's ds_read_b128 land in the registers that are the src2 (addend) operands of the
// packed FMAs issued just before (a write-after-read the hardware is expected to order).  Aggressor: a kernel that only streams
// v_mfma_f32_32x32x16_f16, launched on a second stream so that its waves share SIMDs with the victim's.
// The victim's output is compared with its own output from a run WITHOUT the aggressor: any difference is the fault.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_victim(const float* __restrict__ H, const float* __restrict__ W,
                                                const float* __restrict__ d_out, float* __restrict__ out, int M, int K, int OD) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Dt = smem;               // [OD][16]
  float* Ws = smem + 16 * OD;     // [K][OD]
  const int t = threadIdx.x;
  const long r0 = (long)blockIdx.x * 16;
  const bool kv = t < K;
  float h[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) h[r] = (kv && r0 + r < M) ? H[(r0 + r) * K + t] : 0.f;
  for (int i = t; i < K * OD; i += 256) Ws[i] = W[i];
  for (int i = t; i < 16 * OD; i += 256) {
    const int r = i / OD, a = i - r * OD;
    Dt[a * 16 + r] = (r0 + r < M) ? d_out[r0 * OD + i] : 0.f;
  }
  __syncthreads();
  if (kv) {
    float dz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[r] = 0.f;
    for (int a = 0; a < OD; ++a) {
      const float w = Ws[t * OD + a];
      const float4* d4 = reinterpret_cast<const float4*>(Dt + a * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 d = d4[q];
        dz[4 * q + 0] = fmaf(d.x, w, dz[4 * q + 0]);
        dz[4 * q + 1] = fmaf(d.y, w, dz[4 * q + 1]);
        dz[4 * q + 2] = fmaf(d.z, w, dz[4 * q + 2]);
        dz[4 * q + 3] = fmaf(d.w, w, dz[4 * q + 3]);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r0 + r < M) out[(r0 + r) * K + t] = dz[r] * (h[r] > 0.f ? 1.f : 0.f);   // ReLU' from the output, as in the failing test
  }
}

__global__ __launch_bounds__(256) void k_aggressor(float* __restrict__ sink, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int M = 4096, K = 256, OD = 34, reps = argc > 1 ? atoi(argv[1]) : 200;
  const int agg_grid = argc > 2 ? atoi(argv[2]) : 1024, agg_iters = argc > 3 ? atoi(argv[3]) : 400;   // 1024 x 4 waves: half of the wave slots stay free for the victim
  std::vector<float> hH((size_t)M * K), hW((size_t)K * OD), hD((size_t)M * OD);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hH) v = rnd();
  for (auto& v : hW) v = 0.1f * rnd();
  for (auto& v : hD) v = rnd();
  float *H, *W, *D, *out, *sink;
  CHECK(hipMalloc(&H, hH.size() * 4)); CHECK(hipMalloc(&W, hW.size() * 4)); CHECK(hipMalloc(&D, hD.size() * 4));
  CHECK(hipMalloc(&out, hH.size() * 4)); CHECK(hipMalloc(&sink, 4096 * 256 * 4));
  CHECK(hipMemcpy(H, hH.data(), hH.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(D, hD.data(), hD.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const size_t lds = (size_t)(16 * OD + K * OD) * sizeof(float);
  std::vector<float> ref(hH.size()), got(hH.size());
  hipLaunchKernelGGL(k_victim, dim3(M / 16), dim3(256), lds, s1, H, W, D, out, M, K, OD);
  CHECK(hipStreamSynchronize(s1));
  CHECK(hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost));
  long bad_runs = 0, bad_elems = 0, bad_hi_lanes = 0;
  for (int mode = 0; mode < 2; ++mode) {       // 0: victim alone (control), 1: next to the MFMA stream
    bad_runs = bad_elems = bad_hi_lanes = 0;
    for (int rep = 0; rep < reps; ++rep) {
      CHECK(hipMemsetAsync(out, 0, ref.size() * 4, s1));
      if (mode) hipLaunchKernelGGL(k_aggressor, dim3(agg_grid), dim3(256), 0, s2, sink, agg_iters);
      for (int v = 0; v < 8; ++v) hipLaunchKernelGGL(k_victim, dim3(M / 16), dim3(256), lds, s1, H, W, D, out, M, K, OD);
      CHECK(hipStreamSynchronize(s1));
      CHECK(hipStreamSynchronize(s2));
      CHECK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
      long n = 0;
      for (size_t i = 0; i < got.size(); ++i)
        if (got[i] != ref[i]) { ++n; if ((i % K) % 64 >= 48) ++bad_hi_lanes; }
      bad_elems += n;
      bad_runs += n != 0;
    }
    printf("%s: %ld of %d repetitions differ from the solo run (%ld elements, %ld of them in lanes 48-63 of their wave)\n",
           mode ? "next to v_mfma_f32_32x32x16_f16 waves" : "victim alone", bad_runs, reps, bad_elems, bad_hi_lanes);
  }
  return bad_runs ? 1 : 0;
}
