// env.hip -- synthetic random-observation vector env (the build's own benchmark env,
// SURVEY.md 8(d)); contract mirrors the reference's auto-resetting TORCH-interface envs
// (rl_x/environments/custom_mujoco/ant/warp_torch/environment.py:142-186) plus the
// pre-reset `actual_next_observation` of the JAX-interface State
// (rl_x/environments/custom_mujoco/ant/mjx/state.py:7-17).
// CPU restatement for tests: oracle/env.py.
#include "env_device.h"

namespace rlx {

__global__ void k_env_reset(uint32_t seed, int env_id_offset, int N, int O, int horizon, float* __restrict__ obs,
                            int32_t* __restrict__ ep_step, float* __restrict__ ep_ret, float* __restrict__ last_ret,
                            float* __restrict__ last_len) {
  const int pairs = (O + 1) / 2;
  const int64_t items = (int64_t)N * pairs;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(it / pairs), p = (int)(it % pairs);
    float a, b;
    obs_pair(seed, (uint32_t)(n + env_id_offset), ENV_RESET_T, ENV_STREAM_RESET + p, a, b);
    obs[(int64_t)n * O + 2 * p] = a;
    if (2 * p + 1 < O) obs[(int64_t)n * O + 2 * p + 1] = b;
    if (p == 0) {
      ep_step[n] = (int32_t)(((int64_t)(n + env_id_offset) * ENV_PHASE_MULT) % horizon);
      ep_ret[n] = 0.f;
      last_ret[n] = 0.f;
      last_len[n] = 0.f;
    }
  }
}

// One block = EPB consecutive envs (64, 16 or 4 -- see the launcher: 4096 envs x 376 observations in blocks of 64 is
// 64 blocks of 47 threefry + Box-Muller pairs per thread on a quarter of the CUs).
// Phase 0 (one thread per (env, action dim)): the action-cost terms clip(a) - tanh(obs) into LDS (their
// loads and tanh in parallel instead of one dependent chain per env).  Phase 1 (one lane of wave 0 per
// env): the terms squared and added in index order, reward, termination, episode statistics from the
// CURRENT obs.  Phase 2 (one thread per (env, obs pair)): draw next obs; done envs continue from a reset
// draw while final_obs keeps the draw.  Dynamic LDS: EPB * A floats.
template <int EPB>
__global__ __launch_bounds__(256) void k_env_step(uint32_t seed, int env_id_offset, uint32_t t, int N, int O, int A,
                                                  int horizon, float p_term, float reward_noise,
                                                  const float* __restrict__ action, float* __restrict__ obs,
                                                  float* __restrict__ final_obs, float* __restrict__ reward,
                                                  float* __restrict__ terminated, float* __restrict__ truncated,
                                                  int32_t* __restrict__ ep_step, float* __restrict__ ep_ret,
                                                  float* __restrict__ last_ret, float* __restrict__ last_len,
                                                  float* __restrict__ episode_stats, float* __restrict__ prev_obs_out) {
  static_assert(EPB <= 64, "phase 1 is one wave");
  __shared__ int s_done[EPB];
  extern __shared__ float s_diff[];   // [EPB][A]
  const int n0 = blockIdx.x * EPB;
  for (int it = threadIdx.x; it < EPB * A; it += blockDim.x) {
    const int e = it / A, j = it - e * A;
    if (n0 + e < N) s_diff[it] = env_cost_diff(action[(int64_t)(n0 + e) * A + j], obs[(int64_t)(n0 + e) * O + j % O]);
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // all of wave 0 (the wave sums below need every lane)
    const int n = n0 + threadIdx.x;
    int done = 0;
    float fin_ret = 0.f, fin_len = 0.f;
    if (threadIdx.x < EPB && n < N) {
      float acc = 0.f;
      for (int j = 0; j < A; ++j) {
        const float d = s_diff[threadIdx.x * A + j];
        acc += d * d;
      }
      const EnvLaneOut e = env_lane_finish(seed, (uint32_t)(n + env_id_offset), t, A, horizon, p_term, reward_noise, acc,
                                           ep_step, ep_ret, last_ret, last_len, n);
      done = e.done;
      fin_ret = e.fin_ret;
      fin_len = e.fin_len;
      reward[n] = e.reward;
      terminated[n] = e.term ? 1.f : 0.f;
      truncated[n] = e.trunc ? 1.f : 0.f;
    }
    if (threadIdx.x < EPB) s_done[threadIdx.x] = done;
    if (episode_stats) {
      const float c = wave_sum((float)done), sr = wave_sum(fin_ret), sl = wave_sum(fin_len);
      if (threadIdx.x == 0 && c > 0.f) {
        atomicAdd(&episode_stats[0], c);
        atomicAdd(&episode_stats[1], sr);
        atomicAdd(&episode_stats[2], sl);
      }
    }
  }
  __syncthreads();
  const int pairs = (O + 1) / 2;
  const int items = EPB * pairs;
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int e = it / pairs, p = it % pairs;
    const int n = n0 + e;
    if (n >= N) continue;
    float a, b;
    obs_pair(seed, (uint32_t)(n + env_id_offset), t, (uint32_t)p, a, b);
    const int64_t o = (int64_t)n * O + 2 * p;
    final_obs[o] = a;
    if (2 * p + 1 < O) final_obs[o + 1] = b;
    if (s_done[e]) obs_pair(seed, (uint32_t)(n + env_id_offset), t, ENV_STREAM_RESET + p, a, b);
    if (prev_obs_out) {   // the observation the action was computed from (the replay ring's `states` row of this transition)
      prev_obs_out[o] = obs[o];
      if (2 * p + 1 < O) prev_obs_out[o + 1] = obs[o + 1];
    }
    obs[o] = a;
    if (2 * p + 1 < O) obs[o + 1] = b;
  }
}

}  // namespace rlx

using namespace rlx;

extern "C" {

int rlx_env_reset_f32(rlx_ctx* ctx, uint32_t seed, int env_id_offset, int N, int obs_dim, int horizon, float* obs,
                      int32_t* ep_step, float* ep_ret, float* last_ret, float* last_len, void* stream) {
  RLX_REQUIRE(ctx && obs && ep_step && ep_ret && last_ret && last_len, RLX_EINVAL, "rlx_env_reset_f32: NULL pointer");
  RLX_REQUIRE(N > 0 && obs_dim > 0 && horizon > 0, RLX_EINVAL, "rlx_env_reset_f32: bad sizes");
  const int64_t items = (int64_t)N * ((obs_dim + 1) / 2);
  int grid = div_up(items, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_env_reset, dim3(grid), dim3(256), 0, (hipStream_t)stream, seed, env_id_offset, N, obs_dim,
                     horizon, obs, ep_step, ep_ret, last_ret, last_len);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_env_step_copy_f32(rlx_ctx* ctx, uint32_t seed, int env_id_offset, uint32_t t, int N, int obs_dim, int act_dim,
                          int horizon, float p_term, float reward_noise, const float* action, float* obs, float* final_obs,
                          float* reward, float* terminated, float* truncated, int32_t* ep_step, float* ep_ret,
                          float* last_ret, float* last_len, float* episode_stats, float* prev_obs_out, void* stream) {
  RLX_REQUIRE(ctx && action && obs && final_obs && reward && terminated && truncated && ep_step && ep_ret &&
                  last_ret && last_len,
              RLX_EINVAL, "rlx_env_step_f32: NULL pointer");
  RLX_REQUIRE(N > 0 && obs_dim > 0 && act_dim > 0 && horizon > 0, RLX_EINVAL, "rlx_env_step_f32: bad sizes");
  // envs per block: 64 when that already gives every CU two blocks; otherwise the smallest of {4, 16, 64} whose block still
  // has two observation pairs per thread (wide observations, few envs: many small blocks, several waves per SIMD)
  const int pairs = (obs_dim + 1) / 2;
  const int epb = div_up(N, 64) >= 512 ? 64 : (4 * pairs >= 512 ? 4 : (16 * pairs >= 512 ? 16 : 64));
#define RLX_ENV_STEP(EPB)                                                                                                  \
  hipLaunchKernelGGL(k_env_step<EPB>, dim3(div_up(N, EPB)), dim3(256), (size_t)EPB * act_dim * sizeof(float),               \
                     (hipStream_t)stream, seed, env_id_offset, t, N, obs_dim, act_dim, horizon, p_term, reward_noise, action, \
                     obs, final_obs, reward, terminated, truncated, ep_step, ep_ret, last_ret, last_len, episode_stats, prev_obs_out)
  if (epb == 4) RLX_ENV_STEP(4);
  else if (epb == 16) RLX_ENV_STEP(16);
  else RLX_ENV_STEP(64);
#undef RLX_ENV_STEP
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_env_step_f32(rlx_ctx* ctx, uint32_t seed, int env_id_offset, uint32_t t, int N, int obs_dim, int act_dim,
                     int horizon, float p_term, float reward_noise, const float* action, float* obs, float* final_obs,
                     float* reward, float* terminated, float* truncated, int32_t* ep_step, float* ep_ret,
                     float* last_ret, float* last_len, float* episode_stats, void* stream) {
  return rlx_env_step_copy_f32(ctx, seed, env_id_offset, t, N, obs_dim, act_dim, horizon, p_term, reward_noise, action, obs,
                               final_obs, reward, terminated, truncated, ep_step, ep_ret, last_ret, last_len, episode_stats,
                               nullptr, stream);
}

}  // extern "C"
