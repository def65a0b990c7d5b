// bx_probe.hip -- accuracy + speed probe of the split-fp16 forward GEMM (gemm_bx.h) against fp64 / sequential fp32.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -I rl-x_amd/csrc -I include tools/probes/bx_probe.hip -o rl-x_amd/build/bx_probe
#include "gemm_bx.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
using namespace rlx;

#ifndef VARIANT
#define VARIANT 0
#endif
// VARIANT 0: the product main loop (bx_kloop, hand pipelined); 1: the plain loop (staging pass in front of the MFMA block)
__global__ __launch_bounds__(G_THREADS, 2) void k_fwd_bx(const float* __restrict__ A, const u32x4* __restrict__ Wf,
                                                         const float* __restrict__ bias, float* __restrict__ C,
                                                         int64_t M, int N, int K, int lda, int ntn) {
  __shared__ __attribute__((aligned(16))) char lds[2 * X_OPER];
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(tile / ntn) * G_BM;
  const int n0 = (tile % ntn) * G_BN;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
  const int a_r = t >> 3, a_c = (t & 7) * 4;
  const int NT = ntn * 4, nt0 = n0 / 32 + wn * 2;
  f32x16 acc[2][2];
  zero_acc(acc);
  const int nk = K / X_BK;
  const float* ap = A + (m0 + a_r) * lda + a_c;
  auto load = [&](int kt, float4 (&r)[4]) {
    const int kk = (kt < nk ? kt : nk - 1) * X_BK;
#pragma unroll
    for (int p = 0; p < 4; ++p) r[p] = *reinterpret_cast<const float4*>(ap + (int64_t)(32 * p) * lda + kk);
  };
#if VARIANT == 0
  bx_kloop<2>(lds, Wf, nk, NT, nt0, wm, lane, a_r, a_c, load, acc);
#else
  float4 ra[4];
  u32x4 fb0[2][3], fb1[2][3], fa0[2][3], fa1[2][3];
  load(0, ra);
  bx_load_b(Wf, 0, NT, nt0, lane, fb0);
#pragma unroll
  for (int p = 0; p < 4; ++p) bx_stage_k4(lds, a_r + 32 * p, a_c, ra[p]);
  load(1, ra);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = lds + (kt & 1) * X_OPER;
    char* nxt = lds + ((kt + 1) & 1) * X_OPER;
    bx_load_frag<2>(cur, wm * 64, lane, 0, fa0);
    bx_load_b(Wf, 2 * kt + 1, NT, nt0, lane, fb1);
    if (kt + 1 < nk) {
#pragma unroll
      for (int p = 0; p < 4; ++p) bx_stage_k4(nxt, a_r + 32 * p, a_c, ra[p]);
      if (kt + 2 < nk) load(kt + 2, ra);
    }
    bx_mma<2>(fa0, fb0, acc);
    bx_load_frag<2>(cur, wm * 64, lane, 1, fa1);
    if (kt + 1 < nk) bx_load_b(Wf, 2 * kt + 2, NT, nt0, lane, fb0);
    bx_mma<2>(fa1, fb1, acc);
    __syncthreads();
  }
#endif
  float bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bv[j] = bias[n0 + acc_col(wn, j, lane)];
  float* cb = C + (m0 + wm * 64 + 4 * (lane >> 5)) * N + n0 + wn * 64 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) cb[(i * 32 + (r & 3) + 8 * (r >> 2)) * N + j * 32] = acc[i][j][r] + bv[j];
}

// minimal image builder for the probe (the product one lives in gemm_bx.hip)
__global__ __launch_bounds__(256) void k_probe_wfrag(const float* __restrict__ W, int K, int N, u32x4* __restrict__ out) {
  const int NT = N / 32, KB = K / 16;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= KB * NT * 64) return;
  const int lane = idx & 63, blk = idx >> 6, nt = blk % NT, kb = blk / NT;
  const int j = nt * 32 + (lane & 31), k0 = kb * 16 + 8 * (lane >> 5);
  u32x4 pl[3];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t p0, p1, p2;
    bx_split2(W[(int64_t)(k0 + 2 * e) * N + j], W[(int64_t)(k0 + 2 * e + 1) * N + j], p0, p1, p2);
    pl[0][e] = p0; pl[1][e] = p1; pl[2][e] = p2;
  }
  u32x4* o = out + ((int64_t)blk * 3) * 64 + lane;
  o[0] = pl[0]; o[64] = pl[1]; o[128] = pl[2];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static float frand(uint64_t& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((s >> 40) & 0xffffff) / 8388608.0f - 1.0f;   // [-1, 1)
}

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 32768;
  const int N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 512;
  std::vector<float> hA((size_t)M * K), hW((size_t)K * N), hb(N), hC((size_t)M * N);
  uint64_t s = 12345;
  for (auto& v : hA) v = tanhf(2.f * frand(s));
  for (auto& v : hW) v = 0.08f * frand(s);
  for (auto& v : hb) v = 0.1f * frand(s);
  float *dA, *dW, *db, *dC;
  u32x4* dWf;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dW, hW.size() * 4));
  CK(hipMalloc(&db, hb.size() * 4));
  CK(hipMalloc(&dC, hC.size() * 4));
  CK(hipMalloc(&dWf, (size_t)(K / 16) * (N / 32) * 3 * 64 * 16));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  const int ntn = N / G_BN, grid = (int)(M / G_BM) * ntn;
  const int prep_threads = (K / 16) * (N / 32) * 64;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_probe_wfrag, dim3((prep_threads + 255) / 256), dim3(256), 0, 0, dW, K, N, dWf);
    hipLaunchKernelGGL(k_fwd_bx, dim3(grid), dim3(G_THREADS), 0, 0, dA, dWf, db, dC, M, N, K, K, ntn);
  }
  CK(hipDeviceSynchronize());
  const int reps = 50;
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < reps; ++rep)
    hipLaunchKernelGGL(k_fwd_bx, dim3(grid), dim3(G_THREADS), 0, 0, dA, dWf, db, dC, M, N, K, K, ntn);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < reps; ++rep)
    hipLaunchKernelGGL(k_probe_wfrag, dim3((prep_threads + 255) / 256), dim3(256), 0, 0, dW, K, N, dWf);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms2;
  CK(hipEventElapsedTime(&ms2, e0, e1));
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  // accuracy on a sample of rows: fp64 reference, sequential-fmaf fp32 for the error scale
  double max_err = 0, max_err32 = 0, sum2 = 0, sum2_32 = 0, maxc = 0;
  int64_t cnt = 0;
  for (int64_t r = 0; r < M; r += (r < 256 ? 1 : 509)) {
    for (int j = 0; j < N; ++j) {
      double ref = hb[j];
      float f32 = 0.f;
      for (int k = 0; k < K; ++k) {
        ref += (double)hA[r * K + k] * (double)hW[(size_t)k * N + j];
        f32 = fmaf(hA[r * K + k], hW[(size_t)k * N + j], f32);
      }
      f32 += hb[j];
      const double e = fabs((double)hC[r * N + j] - ref), e32 = fabs((double)f32 - ref);
      max_err = fmax(max_err, e);
      max_err32 = fmax(max_err32, e32);
      sum2 += e * e;
      sum2_32 += e32 * e32;
      maxc = fmax(maxc, fabs(ref));
      ++cnt;
    }
  }
  printf("bx variant=%d products=%d M=%lld N=%d K=%d: %.1f us/launch  (%.1f fp32-equivalent TFLOP/s), wfrag prep %.1f us\n", VARIANT, RLX_BX_PRODUCTS,
         (long long)M, N, K, 1e3 * ms / reps, 2.0 * M * N * K / (ms / reps * 1e-3) / 1e12, 1e3 * ms2 / reps);
  printf("  |C|max %.3f  split-fp16: max abs err %.3e rms %.3e   sequential fp32 fmaf: max %.3e rms %.3e  (%lld samples)\n", maxc,
         max_err, sqrt(sum2 / cnt), max_err32, sqrt(sum2_32 / cnt), (long long)cnt);
  return 0;
}
