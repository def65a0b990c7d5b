// env_device.h -- device-side pieces of the synthetic random-observation env shared by
// env.hip (stand-alone step kernel) and rollout.hip (env step fused into the acting kernel).
#pragma once
#include "common.h"

namespace rlx {

constexpr uint32_t ENV_RESET_T = 0xFFFFFFFFu;
constexpr uint32_t ENV_STREAM_MISC = 64u;
constexpr uint32_t ENV_STREAM_RESET = 128u;
constexpr int ENV_PHASE_MULT = 7919;

__device__ __forceinline__ void obs_pair(uint32_t seed, uint32_t n_global, uint32_t t, uint32_t stream, float& a,
                                         float& b) {
  uint32_t x0 = t, x1 = stream;
  threefry2x32(seed, n_global, x0, x1);
  a = normal_from_bits(x0);
  b = normal_from_bits(x1);
}

// reward / termination / episode bookkeeping of ONE env (the part of k_env_step run by one lane)
struct EnvLaneOut {
  float reward;
  int term, trunc, done;
  float fin_ret, fin_len;
};
// one term of the action cost: clip(a, -1, 1) - tanh(obs)
__device__ __forceinline__ float env_cost_diff(float action, float obs) {
  return fminf(fmaxf(action, -1.f), 1.f) - tanhf(obs);
}

// everything after the action cost (acc = sum_j diff_j^2, added in index order); es_in / er_in = ep_step[n] / ep_ret[n] as the caller
// read them (the fused acting kernel requests them a phase early)
__device__ __forceinline__ EnvLaneOut env_lane_finish_v(uint32_t seed, uint32_t n_global, uint32_t t, int A, int horizon,
                                                        float p_term, float reward_noise, float acc, int es_in, float er_in,
                                                        int32_t* __restrict__ ep_step, float* __restrict__ ep_ret,
                                                        float* __restrict__ last_ret, float* __restrict__ last_len, int n) {
  EnvLaneOut o;
  uint32_t x0 = t, x1 = ENV_STREAM_MISC;
  threefry2x32(seed, n_global, x0, x1);
  const float zr = normal_from_bits(x0);
  const float ut = bits_to_unit(x1);
  o.reward = -acc / (float)A + reward_noise * zr;
  o.term = ut < p_term ? 1 : 0;
  int es = es_in + 1;
  o.trunc = es >= horizon ? 1 : 0;
  o.done = (o.term || o.trunc) ? 1 : 0;
  float er = er_in + o.reward;
  o.fin_ret = 0.f;
  o.fin_len = 0.f;
  if (o.done) {
    last_ret[n] = er;
    last_len[n] = (float)es;
    o.fin_ret = er;
    o.fin_len = (float)es;
    er = 0.f;
    es = 0;
  }
  ep_ret[n] = er;
  ep_step[n] = es;
  return o;
}
__device__ __forceinline__ EnvLaneOut env_lane_finish(uint32_t seed, uint32_t n_global, uint32_t t, int A, int horizon,
                                                      float p_term, float reward_noise, float acc,
                                                      int32_t* __restrict__ ep_step, float* __restrict__ ep_ret,
                                                      float* __restrict__ last_ret, float* __restrict__ last_len, int n) {
  return env_lane_finish_v(seed, n_global, t, A, horizon, p_term, reward_noise, acc, ep_step[n], ep_ret[n], ep_step, ep_ret, last_ret,
                           last_len, n);
}

__device__ __forceinline__ EnvLaneOut env_lane_step(uint32_t seed, uint32_t n_global, uint32_t t, int O, int A, int horizon,
                                                    float p_term, float reward_noise, const float* __restrict__ action_row,
                                                    const float* __restrict__ obs_row, int32_t* __restrict__ ep_step,
                                                    float* __restrict__ ep_ret, float* __restrict__ last_ret,
                                                    float* __restrict__ last_len, int n) {
  float acc = 0.f;
  for (int j = 0; j < A; ++j) {
    const float d = env_cost_diff(action_row[j], obs_row[j % O]);
    acc += d * d;
  }
  return env_lane_finish(seed, n_global, t, A, horizon, p_term, reward_noise, acc, ep_step, ep_ret, last_ret, last_len, n);
}

}  // namespace rlx
