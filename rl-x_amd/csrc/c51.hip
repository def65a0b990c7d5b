// c51.hip -- the distributional ("C51", categorical) critic step of FastSAC: projected target distributions, double-Q selection,
// cross-entropy loss and its gradient w.r.t. the critics' logits, in one launch + one small reduction.
// Reference: rl_x/algorithms/fastsac/pytorch/fastsac.py:144-213 (closure critic_and_entropy_loss_fn); CPU twin: oracle/c51.py.
//
// One wave per sample (four samples per workgroup), lane <-> atom (two atoms per lane up to 128 atoms).  The projection is the
// reference's two index_add_ passes made deterministic: output bin i walks the source atoms j in ascending order and adds
// (u_j - b_j) p_j where l_j == i, then (b_j - l_j) p_j where u_j == i -- the order a sequential index_add_ produces, with no atomics.
// The l / u / weight tables are shared by the two target networks (they depend on the transition only).
#include "common.h"

namespace rlx {

constexpr int C51_MAXA = 128;

__device__ __forceinline__ float c51_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float c51_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// softmax / log-softmax of one row held two atoms per lane (x0: atom lane, x1: atom lane + 64; absent atoms = -inf)
__device__ __forceinline__ void c51_softmax2(float x0, float x1, float& p0, float& p1, float& lse) {
  const float m = c51_wave_max(fmaxf(x0, x1));
  const float e0 = expf(x0 - m), e1 = expf(x1 - m);
  const float s = c51_wave_sum(e0 + e1);
  p0 = e0 / s;
  p1 = e1 / s;
  lse = m + logf(s);
}

__global__ __launch_bounds__(256) void k_c51_loss(const float* __restrict__ q1, const float* __restrict__ q2,
                                                  const float* __restrict__ q1n, const float* __restrict__ q2n,
                                                  const float* __restrict__ rew, const float* __restrict__ dones,
                                                  const float* __restrict__ truncs, const float* __restrict__ nsteps,
                                                  const float* __restrict__ nlogp, const float* __restrict__ log_alpha,
                                                  float* __restrict__ dq1, float* __restrict__ dq2, float* __restrict__ loss_b,
                                                  float* __restrict__ v1_b, int64_t B, int NA, float gamma, float v_min, float v_max,
                                                  int clipped) {
  __shared__ float s_p[4][2][C51_MAXA], s_w[4][2][C51_MAXA];
  __shared__ int s_lu[4][2][C51_MAXA];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  if (b >= B) return;                                   // (whole waves leave: no workgroup barrier below)
  const float ninf = -INFINITY;
  const int j0 = lane, j1 = lane + 64;
  const bool h0 = j0 < NA, h1 = j1 < NA;
  // ---- next-state distributions of the two target networks
  float p10, p11, p20, p21, lse;
  c51_softmax2(h0 ? q1n[b * NA + j0] : ninf, h1 ? q1n[b * NA + j1] : ninf, p10, p11, lse);
  c51_softmax2(h0 ? q2n[b * NA + j0] : ninf, h1 ? q2n[b * NA + j1] : ninf, p20, p21, lse);
  // ---- where every atom of the support lands (fastsac.py:147-162)
  const float delta_z = (v_max - v_min) / (float)(NA - 1);
  const float bootstrap = 1.0f - dones[b] * (1.0f - truncs[b]);
  const float discount = powf(gamma, nsteps[b]) * bootstrap;
  const float r = rew[b] - discount * expf(log_alpha[0]) * nlogp[b];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int j = hh ? j1 : j0;
    if (j < NA) {
      // torch.linspace(v_min, v_max, NA): start + j * step for the first half, end - (NA - 1 - j) * step for the second
      const float step = (v_max - v_min) / (float)(NA - 1);
      const float z = j < NA / 2 ? v_min + step * (float)j : v_max - step * (float)(NA - 1 - j);
      const float tz = fminf(fmaxf(r + discount * z, v_min), v_max);
      const float bb = (tz - v_min) / delta_z;
      int l = (int)floorf(bb), u = (int)ceilf(bb);
      const bool is_int = l == u;
      const bool lm = is_int && l > 0, um = is_int && l == 0;
      if (lm) l -= 1;
      if (um) u += 1;
      s_lu[w][0][j] = l;
      s_lu[w][1][j] = u;
      s_w[w][0][j] = (float)u - bb;
      s_w[w][1][j] = bb - (float)l;
      s_p[w][0][j] = hh ? p11 : p10;
      s_p[w][1][j] = hh ? p21 : p20;
    }
  }
  // (same-wave LDS writes are visible to the wave's later reads: no barrier)
  // ---- projection: bin i collects in the order of a sequential index_add_
  float t1[2] = {0.f, 0.f}, t2[2] = {0.f, 0.f};
  for (int pass = 0; pass < 2; ++pass)
    for (int j = 0; j < NA; ++j) {
      const int tgt = s_lu[w][pass][j];
      const float wj = s_w[w][pass][j];
      const float a1 = s_p[w][0][j] * wj, a2 = s_p[w][1][j] * wj;
      if (tgt == j0) { t1[0] += a1; t2[0] += a2; }
      if (tgt == j1) { t1[1] += a1; t2[1] += a2; }
    }
  // ---- expectations and the double-Q selection (fastsac.py:192-201)
  const float step = (v_max - v_min) / (float)(NA - 1);
  const float z0 = j0 < NA / 2 ? v_min + step * (float)j0 : v_max - step * (float)(NA - 1 - j0);
  const float z1 = j1 < NA / 2 ? v_min + step * (float)j1 : v_max - step * (float)(NA - 1 - j1);
  const float v1 = c51_wave_sum((h0 ? t1[0] * z0 : 0.f) + (h1 ? t1[1] * z1 : 0.f));
  const float v2 = c51_wave_sum((h0 ? t2[0] * z0 : 0.f) + (h1 ? t2[1] * z1 : 0.f));
  float ta[2] = {t1[0], t1[1]}, tb[2] = {t2[0], t2[1]};
  if (clipped) {
    const bool first = v1 < v2;
    ta[0] = tb[0] = first ? t1[0] : t2[0];
    ta[1] = tb[1] = first ? t1[1] : t2[1];
  }
  // ---- cross-entropy against the online critics' logits and its gradient (fastsac.py:203-209)
  float loss = 0.f;
  const float invB = 1.0f / (float)B;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float* q = k ? q2 : q1;
    float* dq = k ? dq2 : dq1;
    const float* t = k ? tb : ta;
    const float x0 = h0 ? q[b * NA + j0] : ninf, x1 = h1 ? q[b * NA + j1] : ninf;
    float s0, s1, lz;
    c51_softmax2(x0, x1, s0, s1, lz);
    const float tsum = c51_wave_sum((h0 ? t[0] : 0.f) + (h1 ? t[1] : 0.f));
    loss += c51_wave_sum((h0 ? -t[0] * (x0 - lz) : 0.f) + (h1 ? -t[1] * (x1 - lz) : 0.f));
    if (h0) dq[b * NA + j0] = (s0 * tsum - t[0]) * invB;
    if (h1) dq[b * NA + j1] = (s1 * tsum - t[1]) * invB;
  }
  if (lane == 0) {
    loss_b[b] = loss;
    v1_b[b] = v1;
  }
}

// out = {mean_b loss_b (= q1_loss + q2_loss), min_b v1, max_b v1, 0}; one workgroup, fixed order
__global__ __launch_bounds__(256) void k_c51_finish(const float* __restrict__ loss_b, const float* __restrict__ v1_b, int64_t B,
                                                    float* __restrict__ out) {
  __shared__ double s_s[256];
  __shared__ float s_mn[256], s_mx[256];
  double s = 0.0;
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < B; i += 256) {
    s += (double)loss_b[i];
    mn = fminf(mn, v1_b[i]);
    mx = fmaxf(mx, v1_b[i]);
  }
  s_s[threadIdx.x] = s; s_mn[threadIdx.x] = mn; s_mx[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s_s[threadIdx.x] += s_s[threadIdx.x + o];
      s_mn[threadIdx.x] = fminf(s_mn[threadIdx.x], s_mn[threadIdx.x + o]);
      s_mx[threadIdx.x] = fmaxf(s_mx[threadIdx.x], s_mx[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)(s_s[0] / (double)B);
    out[1] = s_mn[0];
    out[2] = s_mx[0];
    out[3] = 0.f;
  }
}

}  // namespace rlx

extern "C" {

int rlx_c51_critic_loss_f32(rlx_ctx* ctx, const float* q1_logits, const float* q2_logits, const float* q1_next_logits,
                            const float* q2_next_logits, const float* rewards, const float* dones, const float* truncations,
                            const float* effective_n_steps, const float* next_log_probs, const float* log_alpha, int64_t B,
                            int nr_atoms, float gamma, float v_min, float v_max, int clipped_double_q, float* d_q1_logits,
                            float* d_q2_logits, float* out4, void* stream) {
  RLX_REQUIRE(ctx && q1_logits && q2_logits && q1_next_logits && q2_next_logits && rewards && dones && truncations &&
                  effective_n_steps && next_log_probs && log_alpha && d_q1_logits && d_q2_logits && out4,
              RLX_EINVAL, "rlx_c51_critic_loss_f32: NULL pointer");
  RLX_REQUIRE(B > 0 && nr_atoms >= 2 && nr_atoms <= rlx::C51_MAXA && v_max > v_min, RLX_EINVAL,
              "rlx_c51_critic_loss_f32: need B > 0, 2 <= nr_atoms <= 128, v_max > v_min");
  hipStream_t st = (hipStream_t)stream;
  float* tmp = (float*)rlx::scratch(ctx, rlx::SL_STAGE, (size_t)2 * B * sizeof(float));
  if (!tmp) return RLX_ENOMEM;
  hipLaunchKernelGGL(rlx::k_c51_loss, dim3(rlx::div_up(B, 4)), dim3(256), 0, st, q1_logits, q2_logits, q1_next_logits, q2_next_logits,
                     rewards, dones, truncations, effective_n_steps, next_log_probs, log_alpha, d_q1_logits, d_q2_logits, tmp, tmp + B,
                     B, nr_atoms, gamma, v_min, v_max, clipped_double_q);
  RLX_LAUNCH_CHECK();
  hipLaunchKernelGGL(rlx::k_c51_finish, dim3(1), dim3(256), 0, st, tmp, tmp + B, B, out4);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

}  // extern "C"
