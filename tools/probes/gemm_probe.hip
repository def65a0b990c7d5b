// gemm_probe.hip -- ablation probe for the fp32-MFMA forward GEMM tile loop (tuning aid, not product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rl-x_amd/csrc tools/probes/gemm_probe.hip rl-x_amd/build/core.o -o /tmp/gemm_probe
#include "gemm.h"
#include <cstdio>
#include <vector>
using namespace rlx;

// VARIANT 0: full; 1: no in-loop global loads; 2: + no LDS writes; 3: + no barriers (MFMA only)
template <int VARIANT>
__global__ __launch_bounds__(G_THREADS) void k_probe(const float* __restrict__ A, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ C,
                                                     int64_t M, int N, int K, int ntn) {
  __shared__ __attribute__((aligned(16))) float As[G_LDS_A];
  __shared__ __attribute__((aligned(16))) float Bs[G_LDS_B];
  const int tile = blockIdx.x;
  const int64_t m0 = (int64_t)(tile / ntn) * G_BM;
  const int n0 = (tile % ntn) * G_BN;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
  const int a_r = t >> 3, a_c = (t & 7) * 4;
  const int b_r = t >> 5, b_c = (t & 31) * 4;
  f32x16 acc[2][2];
  zero_acc(acc);
  float4 ra[4], rb[4];
  const int nk = (K + G_BK - 1) / G_BK;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    ra[p] = ld4(A, m0 + a_r + 32 * p, a_c, M, K, K);
    rb[p] = ld4(W, b_r + 8 * p, n0 + b_c, K, N, N);
  }
  for (int kt = 0; kt < nk; ++kt) {
    if (VARIANT < 2 || kt == 0) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float* d = As + (a_r + 32 * p) * G_SA_ROW + a_c;
        d[0] = ra[p].x; d[1] = ra[p].y; d[2] = ra[p].z; d[3] = ra[p].w;
        *reinterpret_cast<float4*>(Bs + (b_r + 8 * p) * G_SB + b_c) = rb[p];
      }
    }
    if (VARIANT < 3 || kt == 0) __syncthreads();
    if (VARIANT < 1 && kt + 1 < nk) {
      const int k0 = (kt + 1) * G_BK;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        ra[p] = ld4(A, m0 + a_r + 32 * p, k0 + a_c, M, K, K);
        rb[p] = ld4(W, k0 + b_r + 8 * p, n0 + b_c, K, N, N);
      }
    }
    mma_ktile<G_SA_ROW, 1>(As, Bs, acc, wm, wn, lane);
    if (VARIANT < 3) __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + acc_col(wn, j, lane);
    const float bv = bias[col];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + acc_row(wm, i, r, lane);
        if (row < M) C[row * N + col] = acc[i][j][r] + bv;
      }
  }
}

template <int V>
void run(const float* A, const float* W, const float* b, float* C, int64_t M, int N, int K) {
  const int ntn = N / G_BN;
  const int grid = (int)(M / G_BM) * ntn;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_probe<V>, dim3(grid), dim3(G_THREADS), 0, 0, A, W, b, C, M, N, K, ntn);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_probe<V>, dim3(grid), dim3(G_THREADS), 0, 0, A, W, b, C, M, N, K, ntn);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20;
  printf("variant %d  M=%lld N=%d K=%d: %8.1f us  %7.1f TFLOP/s\n", V, (long long)M, N, K, us, 2.0 * M * N * K / us / 1e6);
}

int main() {
  const int64_t M = 32768;
  float *A, *W, *b, *C;
  hipMalloc(&A, M * 512 * 4); hipMalloc(&W, 512 * 256 * 4); hipMalloc(&b, 1024); hipMalloc(&C, M * 256 * 4);
  std::vector<float> h(M * 512);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(A, h.data(), M * 512 * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), 512 * 256 * 4, hipMemcpyHostToDevice);
  hipMemset(b, 0, 1024);
  for (int shape = 0; shape < 2; ++shape) {
    const int N = shape ? 128 : 256, K = shape ? 256 : 512;
    run<0>(A, W, b, C, M, N, K);
    run<1>(A, W, b, C, M, N, K);
    run<2>(A, W, b, C, M, N, K);
    run<3>(A, W, b, C, M, N, K);
  }
  return 0;
}
