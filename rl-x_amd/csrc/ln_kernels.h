// ln_kernels.h -- row-wise LayerNorm + activation, forward and backward (flax.linen.LayerNorm, eps 1e-6, "fast variance"):
// the post-cell latent and the first torso layer of the recurrent policy (ppo_lstm/flax_full_jit/policy.py:80-106) and the
// LayerNorm after a WIDE first Dense layer of the feed-forward nets (sac/flax_full_jit/policy.py:31-34, critic.py:24-27;
// the narrow-input first layers have their own fused kernels in mlp.hip / l1fused.hip).
#pragma once
#include "common.h"

namespace rlx {

// ---------------------------------------------------------------------------------------
// LayerNorm + activation over [M, D], D % 64 == 0, D <= 512.  One wave per row.
//   fwd: Y = act(LN(Z) * g + b)                       (Z kept for the backward)
//   bwd: dY (in place) -> dZ; per-block partial dg, db -> partials[grid][2*D]
// ---------------------------------------------------------------------------------------
// NJ: 64-column groups a lane holds (D <= 64 * NJ); eps: 1e-6 (flax.linen.LayerNorm) or 1e-5 (torch.nn.LayerNorm).
// SiLU has no derivative in terms of its output: its backward uses the recomputed pre-activation.
// ACT >= 0: the activation as a compile-time constant (the kernels below take the run-time switch ONCE per launch: with it inside
// the unrolled per-element loops every element is its own branchy dependent chain -- rollout.hip, RO_ACT_SWITCH, has the numbers)
template <int ACT>
__device__ __forceinline__ float ln_act_fwd(float v, int act) {
  if (ACT == RLX_ACT_SILU) return silu_fwd(v);
  if (ACT >= 0) return act_fwd_t<(ACT >= 0 && ACT != RLX_ACT_SILU) ? ACT : RLX_ACT_NONE>(v);
  return act_fwd(v, act);
}
template <bool BWD, int NJMAX = 8, int ACT = -1>
__device__ __forceinline__ void ln_act_body(const float* __restrict__ Z, float* __restrict__ Y /*fwd out; bwd: dY -> dZ*/,
                                            const float* __restrict__ g, const float* __restrict__ be,
                                            float* __restrict__ partials, int64_t M, int D, int act, float eps = 1e-6f) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // bwd: [4][2*D]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int NJ = D >> 6;
  float gam[NJMAX], bet[NJMAX], dg[NJMAX], db[NJMAX];
#pragma unroll
  for (int j = 0; j < NJMAX; ++j) {
    gam[j] = j < NJ ? g[lane + 64 * j] : 0.f;
    bet[j] = j < NJ ? be[lane + 64 * j] : 0.f;
    dg[j] = db[j] = 0.f;
  }
  const float invD = 1.0f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < M; row += (int64_t)gridDim.x * 4) {
    float z[NJMAX], dy[NJMAX];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < NJMAX; ++j) {
      z[j] = j < NJ ? Z[row * D + lane + 64 * j] : 0.f;
      if (BWD) dy[j] = j < NJ ? Y[row * D + lane + 64 * j] : 0.f;
      s += z[j];
      ss += z[j] * z[j];
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    const float mean = s * invD;
    const float rstd = rsqrtf(fmaxf(0.f, ss * invD - mean * mean) + eps);
    if (!BWD) {
#pragma unroll
      for (int j = 0; j < NJMAX; ++j)
        if (j < NJ) Y[row * D + lane + 64 * j] = ln_act_fwd<ACT>((z[j] - mean) * rstd * gam[j] + bet[j], act);
    } else {
      float m1 = 0.f, m2 = 0.f, xh[NJMAX], dxh[NJMAX];
#pragma unroll
      for (int j = 0; j < NJMAX; ++j) {
        xh[j] = (z[j] - mean) * rstd;
        const float yv = xh[j] * gam[j] + bet[j];
        const int act_c = ACT >= 0 ? ACT : act;
        const float ag = act_c == RLX_ACT_SILU ? silu_grad(yv) : act_grad_from_out(ln_act_fwd<ACT>(yv, act), act_c);
        const float d = (j < NJ) ? dy[j] * ag : 0.f;
        dg[j] += d * xh[j];
        db[j] += d;
        dxh[j] = d * gam[j];
        m1 += dxh[j];
        m2 += dxh[j] * xh[j];
      }
      m1 = wave_sum(m1) * invD;
      m2 = wave_sum(m2) * invD;
#pragma unroll
      for (int j = 0; j < NJMAX; ++j)
        if (j < NJ) Y[row * D + lane + 64 * j] = rstd * (dxh[j] - m1 - xh[j] * m2);
    }
  }
  if (BWD) {
#pragma unroll
    for (int j = 0; j < NJMAX; ++j)
      if (j < NJ) {
        smem[w * 2 * D + lane + 64 * j] = dg[j];
        smem[w * 2 * D + D + lane + 64 * j] = db[j];
      }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256)
      partials[(int64_t)blockIdx.x * 2 * D + i] = (smem[i] + smem[2 * D + i]) + (smem[4 * D + i] + smem[6 * D + i]);
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void k_ln_act(const float* __restrict__ Z, float* __restrict__ Y, const float* __restrict__ g,
                                                const float* __restrict__ be, float* __restrict__ partials, int64_t M, int D,
                                                int act) {
  if (act == RLX_ACT_ELU) ln_act_body<BWD, 8, RLX_ACT_ELU>(Z, Y, g, be, partials, M, D, act);
  else if (act == RLX_ACT_TANH) ln_act_body<BWD, 8, RLX_ACT_TANH>(Z, Y, g, be, partials, M, D, act);
  else ln_act_body<BWD>(Z, Y, g, be, partials, M, D, act);
}

// D <= 768, LayerNorm eps as an argument (FastSAC's torch.nn.LayerNorm + SiLU blocks, fastsac.hip)
template <bool BWD>
__global__ __launch_bounds__(256) void k_ln_act_wide(const float* __restrict__ Z, float* __restrict__ Y, const float* __restrict__ g,
                                                     const float* __restrict__ be, float* __restrict__ partials, int64_t M, int D,
                                                     int act, float eps) {
  if (act == RLX_ACT_SILU) ln_act_body<BWD, 12, RLX_ACT_SILU>(Z, Y, g, be, partials, M, D, act, eps);
  else if (act == RLX_ACT_ELU) ln_act_body<BWD, 12, RLX_ACT_ELU>(Z, Y, g, be, partials, M, D, act, eps);
  else ln_act_body<BWD, 12>(Z, Y, g, be, partials, M, D, act, eps);
}

// two nets of the same shape in one launch (grid.y == 2): blockIdx.y == 1 takes {Z, Y, g, partials} from tw; its LayerNorm bias
// sits at the same distance from its scale as the first net's (same parameter layout)
template <bool BWD>
__global__ __launch_bounds__(256) void k_ln_act_twin(const float* __restrict__ Z, float* __restrict__ Y,
                                                     const float* __restrict__ g, const float* __restrict__ be,
                                                     float* __restrict__ partials, int64_t M, int D, int act, Twin tw) {
  if (blockIdx.y) {
    const float* g1 = static_cast<const float*>(tw.p[2]);
    be = g1 + (be - g);
    g = g1;
    Z = static_cast<const float*>(tw.p[0]);
    Y = const_cast<float*>(static_cast<const float*>(tw.p[1]));
    partials = const_cast<float*>(static_cast<const float*>(tw.p[3]));
  }
  if (act == RLX_ACT_ELU) ln_act_body<BWD, 8, RLX_ACT_ELU>(Z, Y, g, be, partials, M, D, act);
  else ln_act_body<BWD>(Z, Y, g, be, partials, M, D, act);
}

}  // namespace rlx
