// sac_sample.h -- the tanh-Gaussian sampling step of the SAC policy (sac/flax/policy.py:33-41, sac.py:119-125,196-197) as device
// functions shared by k_sac_sample (sac.hip) and the fused forward kernel's sampling epilogue (fwd2h.hip).
#pragma once
#include "common.h"

namespace rlx {

constexpr float SAC_LOG_2PI = 1.8378770664093453f;

// per-sample noise keys of the update.  schedule 0 (host-loop flavour, sac/flax/sac.py:195-197): keys = split(key, 2B+1),
// key = keys[0], keys1 = keys[1::2], keys2 = keys[2::2].  schedule 1 (fully jitted flavour, sac/flax_full_jit/sac.py:273-275):
// keys = split(key, 2B+2), key = keys[0], replay key = keys[1], keys1 = keys[2 : 2+B], keys2 = keys[2+B : 2+2B].
__host__ __device__ __forceinline__ uint32_t sac_key_index(int which /*1 or 2*/, int64_t i, int64_t B, int schedule) {
  return schedule ? (uint32_t)(2 + (which - 1) * B + i) : (uint32_t)(which + 2 * i);
}
__host__ __device__ __forceinline__ uint32_t sac_key_count(int64_t B, int schedule) { return (uint32_t)(2 * B + 1 + (schedule ? 1 : 0)); }

// keys = jax.random.split(key, num)[i]
__device__ __forceinline__ void split_key_at(uint32_t k0, uint32_t k1, uint32_t i, uint32_t num, int scheme,
                                             uint32_t& o0, uint32_t& o1) {
  if (scheme == RLX_THREEFRY_PARTITIONABLE) {
    uint32_t x0 = 0, x1 = i;
    threefry2x32(k0, k1, x0, x1);
    o0 = x0;
    o1 = x1;
  } else {
    o0 = random_bits_at(k0, k1, 2ull * i, 2ull * num, RLX_THREEFRY_LEGACY);
    o1 = random_bits_at(k0, k1, 2ull * i + 1, 2ull * num, RLX_THREEFRY_LEGACY);
  }
}


// everything k_sac_sample needs besides the head output (see its comment in sac.hip for the modes)
struct SacSampleArgs {
  uint32_t k0 = 0, k1 = 0;
  int scheme = 0, mode = 0;
  float* act_out = nullptr;
  int ld_out = 0, col_off = 0;
  float* logp = nullptr;
  int A = 0;
  float ls_min = 0.f, ls_max = 0.f;
  int row_off = 0;
  int64_t N_global = 0;
  int deterministic = 0;
  const float* eps_inject = nullptr;
  int schedule = 0;
  const uint32_t* key_dev = nullptr;
  float* proc_out = nullptr;
  const float* proc_low = nullptr;
  const float* proc_half = nullptr;
};

// the key the noise of row i is drawn from (acting: the call's subkey; update: the row's own key of the split)
__device__ __forceinline__ void sac_row_key(const SacSampleArgs& s, uint32_t k0, uint32_t k1, int64_t i, uint32_t& s0, uint32_t& s1) {
  s0 = k0;
  s1 = k1;
  if (s.mode != 0) split_key_at(k0, k1, sac_key_index(s.mode, i + s.row_off, s.N_global, s.schedule), sac_key_count(s.N_global, s.schedule), s.scheme, s0, s1);
}

// action dim j of row i from (mean, raw log_std): stores the action (and the env-facing one), returns the row's log-prob term
__device__ __forceinline__ float sac_sample_elem(const SacSampleArgs& s, uint32_t s0, uint32_t s1, int64_t i, int j, float mean, float raw) {
  const int A = s.A;
  const float ls = fminf(fmaxf(raw, s.ls_min), s.ls_max);
  float eps;
  if (s.mode == 0) eps = normal_from_bits(random_bits_at(s0, s1, (uint64_t)(i + s.row_off) * A + j, (uint64_t)s.N_global * A, s.scheme));
  else eps = normal_from_bits(random_bits_at(s0, s1, (uint64_t)j, (uint64_t)A, s.scheme));
  if (s.deterministic) eps = 0.f;
  if (s.eps_inject) eps = s.eps_inject[i * A + j];   // test hook (rlx_dbg_set_sac_noise)
  const float u = mean + expf(ls) * eps;
  const float a = tanhf(u);
  s.act_out[i * s.ld_out + s.col_off + j] = a;
  // the action the env receives (get_processed_action, sac/flax/policy.py:44-48): low + 0.5 (clip(a) + 1) (high - low)
  if (s.proc_out) s.proc_out[i * A + j] = s.proc_low[j] + (fminf(fmaxf(a, -1.f), 1.f) + 1.0f) * s.proc_half[j];
  return -0.5f * eps * eps - 0.5f * SAC_LOG_2PI - ls - logf(1.0f - a * a + 1e-6f);
}

}  // namespace rlx
