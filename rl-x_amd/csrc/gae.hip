// gae.hip -- `calculate_gae_advantages` (rl_x/algorithms/ppo/flax/ppo.py:122-135,
// full-jit twin rl_x/algorithms/ppo/flax_full_jit/ppo.py:161-175), minus the critic
// forward on next_states (that is rlx_mlp_fwd_f32).
//
// HBM-bound: 6 arrays x 4 B = 24 B per (t, env) element.  One lane per env, lanes
// along N so every load/store of a time row is one coalesced 256-B wave access; the T
// dependent steps are a register recurrence, loads are issued UNROLL rows ahead so the
// serial chain never waits on memory.
#include "common.h"

namespace rlx {

constexpr int GAE_UNROLL = 8;

__global__ __launch_bounds__(64) void k_gae(const float* __restrict__ rewards, const float* __restrict__ values,
                                            const float* __restrict__ next_values,
                                            const float* __restrict__ terminations, float* __restrict__ advantages,
                                            float* __restrict__ returns, int T, int N, float gamma, float lam) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float gl = gamma * lam;
  float adv = 0.f;  // A[t+1]; the t = T-1 step uses delta only (mask irrelevant: adv = 0)
  int t = T - 1;
  for (; t >= GAE_UNROLL - 1; t -= GAE_UNROLL) {
    float r[GAE_UNROLL], v[GAE_UNROLL], nv[GAE_UNROLL], tm[GAE_UNROLL];
#pragma unroll
    for (int u = 0; u < GAE_UNROLL; ++u) {
      const int64_t o = (int64_t)(t - u) * N + n;
      r[u] = rewards[o];
      v[u] = values[o];
      nv[u] = next_values[o];
      tm[u] = terminations[o];
    }
#pragma unroll
    for (int u = 0; u < GAE_UNROLL; ++u) {
      const int64_t o = (int64_t)(t - u) * N + n;
      const float nt = 1.0f - tm[u];
      const float delta = r[u] + gamma * nv[u] * nt - v[u];
      adv = delta + gl * nt * adv;
      advantages[o] = adv;
      returns[o] = adv + v[u];
    }
  }
  for (; t >= 0; --t) {
    const int64_t o = (int64_t)t * N + n;
    const float nt = 1.0f - terminations[o];
    const float vv = values[o];
    const float delta = rewards[o] + gamma * next_values[o] * nt - vv;
    adv = delta + gl * nt * adv;
    advantages[o] = adv;
    returns[o] = adv + vv;
  }
}

}  // namespace rlx

extern "C" int rlx_gae_f32(rlx_ctx* ctx, const float* rewards, const float* values, const float* next_values,
                           const float* terminations, float* advantages, float* returns, int T, int N, float gamma,
                           float gae_lambda, void* stream) {
  RLX_REQUIRE(ctx && rewards && values && next_values && terminations && advantages && returns, RLX_EINVAL,
              "rlx_gae_f32: NULL pointer");
  RLX_REQUIRE(T >= 0 && N >= 0, RLX_EINVAL, "rlx_gae_f32: negative size");
  if (T == 0 || N == 0) return RLX_OK;
  // 64-thread blocks: N=4096 -> 64 workgroups spread over the XCDs (one wave each)
  hipLaunchKernelGGL(rlx::k_gae, dim3(rlx::div_up(N, 64)), dim3(64), 0, (hipStream_t)stream, rewards, values,
                     next_values, terminations, advantages, returns, T, N, gamma, gae_lambda);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}
