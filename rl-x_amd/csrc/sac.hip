// sac.hip -- SAC acting, replay ring and the whole `update` for gfx950.
// Replaces the XLA fusions of
//   get_action   rl_x/algorithms/sac/flax/sac.py:119-125
//   loss_fn      rl_x/algorithms/sac/flax/sac.py:133-188 (vmapped, meaned :190-193)
//   update       rl_x/algorithms/sac/flax/sac.py:191-215 (per-sample noise keys, 3 Adam steps, Polyak)
//   ReplayBuffer rl_x/algorithms/sac/flax/replay_buffer.py:4-38 (device-resident ring like
//                rl_x/algorithms/sac/flax_full_jit/sac.py:139-154; index draws stay on the host: numpy PCG64)
// Networks: Policy sac/flax/policy.py:22-41 (two-headed: mean | clipped log_std), VectorCritic
// sac/flax/critic.py:17-53 (two independent Q nets on [obs, action]).  CPU twin: oracle/sac.py.
//
// All dense layers run on the exact-fp32 MFMA GEMM kernels of mlp.hip (wide first layers included);
// this file adds the SAC-specific elementwise / seed kernels and the update schedule.
#include "mlp.h"

namespace rlx {

constexpr float SAC_LOG_2PI = 1.8378770664093453f;
constexpr int SAC_HEAD_ROWS = 16;   // rows per workgroup of the head kernels: 256 workgroups at B = 4096 (64 rows left 3/4 of the CUs idle)

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

// Xc_cur = [s | a | 0], Xc_next[:, :O] = s', Xc_pi[:, :O] = s (action columns are filled by k_sac_sample)
__global__ __launch_bounds__(256) void k_sac_concat(const float* __restrict__ s, const float* __restrict__ s2,
                                                    const float* __restrict__ a, float* __restrict__ xc_cur,
                                                    float* __restrict__ xc_next, float* __restrict__ xc_pi, int64_t B,
                                                    int O, int A, int ld) {
  const int64_t total = B * ld;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / ld;
    const int c = (int)(e - r * ld);
    if (c < O) {
      const float v = s[r * O + c];
      xc_cur[e] = v;
      xc_pi[e] = v;
      xc_next[e] = s2[r * O + c];
    } else {
      xc_cur[e] = c < O + A ? a[r * A + (c - O)] : 0.f;
      xc_next[e] = 0.f;
      xc_pi[e] = 0.f;
    }
  }
}

// tanh-Gaussian sample from head output [B, 2A] = (mean | raw log_std):
//   u = mean + exp(clip(log_std)) * eps, a = tanh(u), logp = sum(-eps^2/2 - log(2pi)/2 - log_std - log(1 - a^2 + 1e-6))
// mode 0: acting (noise = normal(subkey, [N_global, A]) rows [row_off, row_off+B), like get_action)
// mode 1/2: update; sample i uses the per-sample key split(key, 2B+1)[mode + 2i]  (sac.py:196-197)
__global__ __launch_bounds__(256) void k_sac_sample(const float* __restrict__ head, uint32_t k0, uint32_t k1, int scheme,
                                                    int mode, float* __restrict__ act_out, int ld_out, int col_off,
                                                    float* __restrict__ logp, int64_t B, int A, float ls_min,
                                                    float ls_max, int row_off, int64_t N_global, int deterministic,
                                                    const float* __restrict__ eps_inject = nullptr, int schedule = 0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  uint32_t s0 = k0, s1 = k1;
  if (mode != 0) split_key_at(k0, k1, sac_key_index(mode, i, B, schedule), sac_key_count(B, schedule), scheme, s0, s1);
  float lp = 0.f;
  for (int j = 0; j < A; ++j) {
    const float mean = head[i * 2 * A + j];
    const float ls = fminf(fmaxf(head[i * 2 * A + A + j], ls_min), ls_max);
    float eps;
    if (mode == 0) eps = normal_from_bits(random_bits_at(s0, s1, (uint64_t)(i + row_off) * A + j, (uint64_t)N_global * A, scheme));
    else eps = normal_from_bits(random_bits_at(s0, s1, (uint64_t)j, (uint64_t)A, scheme));
    if (deterministic) eps = 0.f;
    if (eps_inject) eps = eps_inject[i * A + j];   // test hook (rlx_dbg_set_sac_noise)
    const float u = mean + expf(ls) * eps;
    const float a = tanhf(u);
    lp += -0.5f * eps * eps - 0.5f * SAC_LOG_2PI - ls - logf(1.0f - a * a + 1e-6f);
    act_out[i * ld_out + col_off + j] = a;
  }
  if (logp) logp[i] = lp;
}

__device__ __forceinline__ float block_sum256(float v, float* s_buf) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = s_buf[0] + s_buf[1] + s_buf[2] + s_buf[3];
  __syncthreads();
  return t;
}

// y = r + gamma (1 - term) (min(qt0, qt1) - alpha logp');  dq_k = (q_k - y) / B;  q_loss = mean_k,i (q_k - y)^2
__global__ __launch_bounds__(256) void k_sac_critic_seed(const float* __restrict__ qt0, const float* __restrict__ qt1,
                                                         const float* __restrict__ logp_n, const float* __restrict__ rew,
                                                         const float* __restrict__ term, const float* __restrict__ log_alpha,
                                                         const float* __restrict__ q0, const float* __restrict__ q1,
                                                         float* __restrict__ dq0, float* __restrict__ dq1,
                                                         float* __restrict__ part, int64_t B, float gamma) {
  __shared__ float s_buf[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float alpha = expf(log_alpha[0]);
  float ql = 0.f;
  if (i < B) {
    const float y = rew[i] + gamma * (1.f - term[i]) * (fminf(qt0[i], qt1[i]) - alpha * logp_n[i]);
    const float e0 = q0[i] - y, e1 = q1[i] - y;
    dq0[i] = e0 / (float)B;
    dq1[i] = e1 / (float)B;
    ql = 0.5f * (e0 * e0 + e1 * e1);
  }
  ql = block_sum256(ql, s_buf);
  if (threadIdx.x == 0) part[blockIdx.x] = ql;
}

// min over the two critics on (s, pi(s)); d(-min_q)/dq_k = -[k == argmin] / B;  sums of min_q, logp
__global__ __launch_bounds__(256) void k_sac_policy_seed(const float* __restrict__ qa0, const float* __restrict__ qa1,
                                                         const float* __restrict__ logp_c, float* __restrict__ d0,
                                                         float* __restrict__ d1, float* __restrict__ part, int nblk,
                                                         int64_t B) {
  __shared__ float s_buf[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float mq = 0.f, lp = 0.f;
  if (i < B) {
    const bool sel0 = qa0[i] <= qa1[i];
    d0[i] = sel0 ? -1.f / (float)B : 0.f;
    d1[i] = sel0 ? 0.f : -1.f / (float)B;
    mq = sel0 ? qa0[i] : qa1[i];
    lp = logp_c[i];
  }
  mq = block_sum256(mq, s_buf);
  lp = block_sum256(lp, s_buf);
  if (threadIdx.x == 0) { part[blockIdx.x] = mq; part[nblk + blockIdx.x] = lp; }
}

// d L / d(head output) of the policy:  d_u = (alpha/B * 2a/(1-a^2+1e-6) + dQ/da) (1 - a^2);  d_mean = d_u;
// d_logstd = [raw inside clip] (d_u std eps - alpha/B)
__global__ __launch_bounds__(256) void k_sac_policy_grad(const float* __restrict__ head, const float* __restrict__ xc_pi,
                                                         int ld, int col_off, const float* __restrict__ da0,
                                                         const float* __restrict__ da1, int ld_da,
                                                         const float* __restrict__ log_alpha, uint32_t k0, uint32_t k1,
                                                         int scheme, float* __restrict__ d_out, int64_t B, int A,
                                                         float ls_min, float ls_max,
                                                         const float* __restrict__ eps_inject = nullptr, int schedule = 0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  const float alpha = expf(log_alpha[0]);
  uint32_t s0, s1;
  split_key_at(k0, k1, sac_key_index(2, i, B, schedule), sac_key_count(B, schedule), scheme, s0, s1);
  const float invB = 1.0f / (float)B;
  for (int j = 0; j < A; ++j) {
    const float raw = head[i * 2 * A + A + j];
    const float ls = fminf(fmaxf(raw, ls_min), ls_max);
    const float eps = eps_inject ? eps_inject[i * A + j]
                                 : normal_from_bits(random_bits_at(s0, s1, (uint64_t)j, (uint64_t)A, scheme));
    const float a = xc_pi[i * ld + col_off + j];
    const float om = 1.0f - a * a;
    const float dq = da0[i * ld_da + j] + da1[i * ld_da + j];
    const float du = (alpha * invB * 2.0f * a / (om + 1e-6f) + dq) * om;
    d_out[i * 2 * A + j] = du;
    const bool inside = raw > ls_min && raw < ls_max;
    d_out[i * 2 * A + A + j] = inside ? du * expf(ls) * eps - alpha * invB : 0.f;
  }
}

// metrics (means) + gradient of log_alpha from the partial sums
//   part_c[nb]: q_loss sums; part_p[2*nb]: min_q sums, logp sums
__global__ void k_sac_finalize(const float* __restrict__ part_c, const float* __restrict__ part_p, int nb,
                               const float* __restrict__ log_alpha, float* __restrict__ g_alpha,
                               float* __restrict__ metrics, int64_t B, float target_entropy) {
  float ql = 0.f, mq = 0.f, lp = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) { ql += part_c[i]; mq += part_p[i]; lp += part_p[nb + i]; }
  ql = wave_sum(ql); mq = wave_sum(mq); lp = wave_sum(lp);
  if (threadIdx.x == 0) {
    const float invB = 1.0f / (float)B;
    const float alpha = expf(log_alpha[0]);
    const float mean_lp = lp * invB, mean_q = mq * invB;
    const float entropy = -mean_lp;
    g_alpha[0] = alpha * (entropy - target_entropy);     // d mean(alpha_g (entropy - target)) / d log_alpha
    metrics[0] = ql * invB;                               // loss/q_loss
    metrics[1] = alpha * mean_lp - mean_q;                // loss/policy_loss
    metrics[2] = alpha * (entropy - target_entropy);      // loss/entropy_loss
    metrics[3] = entropy;                                 // entropy/entropy
    metrics[4] = alpha;                                   // entropy/alpha
    metrics[5] = mean_q;                                  // q_value/q_value
  }
}

// generic head backward: dZ_last = (d_out @ W^T) * act'(H) in place over H; optional per-block partials
// of dW_head[K, OD] and db_head[OD]  (layout [block][K*OD + OD])
__global__ __launch_bounds__(256) void k_head_bwd(float* __restrict__ H, const float* __restrict__ W,
                                                  const float* __restrict__ d_out, float* __restrict__ partials,
                                                  int64_t M, int K, int OD, int act) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int HS = K + 1;
  float* Hs = smem;
  float* Ws = Hs + SAC_HEAD_ROWS * HS;
  float* Ds = Ws + K * OD;
  const int t = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * SAC_HEAD_ROWS;
  for (int i = t; i < SAC_HEAD_ROWS * K; i += 256) {
    const int r = i / K, k = i - r * K;
    Hs[r * HS + k] = (r0 + r < M) ? H[(r0 + r) * K + k] : 0.f;
  }
  for (int i = t; i < K * OD; i += 256) Ws[i] = W[i];
  for (int i = t; i < SAC_HEAD_ROWS * OD; i += 256) {
    const int r = i / OD;
    Ds[i] = (r0 + r < M) ? d_out[r0 * OD + i] : 0.f;
  }
  __syncthreads();
  for (int e = t; e < SAC_HEAD_ROWS * K; e += 256) {
    const int r = e / K, k = e - r * K;
    if (r0 + r < M) {
      float acc = 0.f;
      for (int a = 0; a < OD; ++a) acc = fmaf(Ds[r * OD + a], Ws[k * OD + a], acc);
      H[(r0 + r) * K + k] = acc * act_grad_from_out(Hs[r * HS + k], act);
    }
  }
  if (!partials) return;
  float* pw = partials + (int64_t)blockIdx.x * (K * OD + OD);
  for (int e = t; e < K * OD; e += 256) {
    const int k = e / OD, a = e - k * OD;
    float acc = 0.f;
    for (int r = 0; r < SAC_HEAD_ROWS; ++r) acc = fmaf(Hs[r * HS + k], Ds[r * OD + a], acc);
    pw[e] = acc;
  }
  if (t < OD) {
    float sb = 0.f;
    for (int r = 0; r < SAC_HEAD_ROWS; ++r) sb += Ds[r * OD + t];
    pw[K * OD + t] = sb;
  }
}

__global__ __launch_bounds__(256) void k_polyak(float* __restrict__ target, const float* __restrict__ params, int64_t n,
                                                float tau) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    target[i] = tau * params[i] + (1.f - tau) * target[i];
}

// replay sample: out[i] = ring[idx1[i], idx2[i]]
__global__ __launch_bounds__(256) void k_replay_gather(const float* __restrict__ r_s, const float* __restrict__ r_s2,
                                                       const float* __restrict__ r_a, const float* __restrict__ r_r,
                                                       const float* __restrict__ r_t, const int32_t* __restrict__ idx1,
                                                       const int32_t* __restrict__ idx2, int N, int O, int A, int64_t B,
                                                       float* __restrict__ s, float* __restrict__ s2,
                                                       float* __restrict__ a, float* __restrict__ r,
                                                       float* __restrict__ tm) {
  const int64_t row = 2 * O + A + 2;
  const int64_t total = B * row;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / row;
    const int c = (int)(e - i * row);
    const int64_t src = (int64_t)idx1[i] * N + idx2[i];
    if (c < O) s[i * O + c] = r_s[src * O + c];
    else if (c < 2 * O) s2[i * O + (c - O)] = r_s2[src * O + (c - O)];
    else if (c < 2 * O + A) a[i * A + (c - 2 * O)] = r_a[src * A + (c - 2 * O)];
    else if (c == 2 * O + A) r[i] = r_r[src];
    else tm[i] = r_t[src];
  }
}

// `idx = jax.random.randint(key, (B,), 0, span)` (jax<=0.7.2, restated; oracle/prng.py::randint): k1, k2 = split(key);
// hi, lo = random_bits(k1 / k2, 32, (B,)); mult = ((2^16 % span)^2) % span in uint32;  idx = ((hi % span) * mult + lo % span) % span
__device__ __forceinline__ uint32_t randint_at(uint32_t k0, uint32_t k1, uint64_t i, uint64_t n, uint32_t span, int scheme) {
  uint32_t a0, a1, b0, b1;
  split_key_at(k0, k1, 0u, 2u, scheme, a0, a1);
  split_key_at(k0, k1, 1u, 2u, scheme, b0, b1);
  const uint32_t hi = random_bits_at(a0, a1, i, n, scheme), lo = random_bits_at(b0, b1, i, n, scheme);
  uint32_t mult = 65536u % span;
  mult = (mult * mult) % span;
  return ((hi % span) * mult + (lo % span)) % span;
}

// the fully jitted flavour's sampler (sac/flax_full_jit/sac.py:276-282): BOTH index vectors come from the SAME key
__global__ __launch_bounds__(256) void k_replay_draw(uint32_t k0, uint32_t k1, int scheme, int32_t* __restrict__ idx1,
                                                     int32_t* __restrict__ idx2, int64_t B, uint32_t size, uint32_t nr_envs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  idx1[i] = (int32_t)randint_at(k0, k1, (uint64_t)i, (uint64_t)B, size, scheme);
  idx2[i] = (int32_t)randint_at(k0, k1, (uint64_t)i, (uint64_t)B, nr_envs, scheme);
}

// ---------------------------------------------------------------------------------------
struct NetBufs {
  float* acts[4];
};

static int head_bwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, float* h_last,
                    const float* d_out, float* partials, int64_t M, hipStream_t st) {
  const int K = L.head.in, OD = L.head.out;
  const size_t lds = ((size_t)SAC_HEAD_ROWS * (K + 1) + (size_t)K * OD + (size_t)SAC_HEAD_ROWS * OD) * sizeof(float);
  RLX_REQUIRE(lds <= 150 * 1024, RLX_EUNSUP, "sac: head too wide for the LDS-staged head kernel");
  hipLaunchKernelGGL(k_head_bwd, dim3(div_up(M, SAC_HEAD_ROWS)), dim3(256), lds, st, h_last, params + L.head.W, d_out,
                     partials, M, K, OD, d.act);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// forward of one net keeping activations; out[M, out_dim]
// critics (out_dim 1, two-headed policy has out_dim >= 2) always read the padded concat buffers
static inline bool sac_gemm_l0(const rlx_mlp_desc& d, int ldx) { return d.in_dim > 32 || d.out_dim == 1 || ldx != d.in_dim; }

static int net_fwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x, int ldx,
                   float* const* acts, float* out, int64_t M, hipStream_t st) {
  // rows from the padded concat buffers (the critics' [obs | action] input, wide policy observations) take the GEMM
  // first layer whatever the width; dense narrow observations take the small-input kernel
  int rc = mlp_trunk_fwd(ctx, d, L, params, x, acts, M, st, ldx, sac_gemm_l0(d, ldx));
  if (rc) return rc;
  return launch_head_fwd(acts[d.n_hidden - 1], params + L.head.W, params + L.head.b, out, M, L.head.in, L.head.out, st);
}

// backward of one net from d_out[M, out_dim]; grads may be null (input gradient only)
static int net_bwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x, int ldx,
                   float* const* acts, const float* d_out, float* grads, float* head_part, int64_t M, float* sumsq,
                   int* nsq, const TrunkOpts* opt_in, hipStream_t st) {
  const int K = L.head.in, OD = L.head.out;
  int rc = head_bwd(ctx, d, L, params, acts[d.n_hidden - 1], d_out, grads ? head_part : nullptr, M, st);
  if (rc) return rc;
  TrunkOpts opt;
  if (opt_in) opt = *opt_in;
  opt.ldx = ldx;
  opt.gemm_l0 = sac_gemm_l0(d, ldx);
  ReduceSeg extra[2];
  int ne = 0;
  if (grads) {
    const int nb = div_up(M, SAC_HEAD_ROWS);
    const int64_t PS = (int64_t)K * OD + OD;
    extra[ne++] = ReduceSeg{head_part, grads + L.head.W, (int64_t)K * OD, PS, nb, 0, 1.f, 0.f, 1};
    extra[ne++] = ReduceSeg{head_part + (int64_t)K * OD, grads + L.head.b, (int64_t)OD, PS, nb, 0, 1.f, 0.f, 1};
  }
  return mlp_trunk_bwd(ctx, d, L, params, x, acts, grads, M, extra, ne, sumsq, nsq, st, &opt);
}

}  // namespace rlx

using namespace rlx;

extern "C" {

int rlx_sac_replay_sample_f32(rlx_ctx* ctx, const float* ring_states, const float* ring_next_states,
                              const float* ring_actions, const float* ring_rewards, const float* ring_terminations,
                              int nr_envs, int obs_dim, int act_dim, const int32_t* idx1, const int32_t* idx2, int64_t B,
                              float* states, float* next_states, float* actions, float* rewards, float* terminations,
                              void* stream) {
  RLX_REQUIRE(ctx && ring_states && ring_next_states && ring_actions && ring_rewards && ring_terminations && idx1 && idx2 &&
                  states && next_states && actions && rewards && terminations,
              RLX_EINVAL, "rlx_sac_replay_sample_f32: NULL pointer");
  RLX_REQUIRE(B > 0 && nr_envs > 0 && obs_dim > 0 && act_dim > 0, RLX_EINVAL, "rlx_sac_replay_sample_f32: bad sizes");
  const int64_t total = B * (2 * obs_dim + act_dim + 2);
  int grid = div_up(total, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_replay_gather, dim3(grid), dim3(256), 0, (hipStream_t)stream, ring_states, ring_next_states,
                     ring_actions, ring_rewards, ring_terminations, idx1, idx2, nr_envs, obs_dim, act_dim, B, states,
                     next_states, actions, rewards, terminations);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_sac_replay_draw_i32(rlx_ctx* ctx, const uint32_t update_key[2], int scheme, int64_t B, int size, int nr_envs,
                            int32_t* idx1, int32_t* idx2, void* stream) {
  RLX_REQUIRE(ctx && update_key && idx1 && idx2 && B > 0 && size > 0 && nr_envs > 0, RLX_EINVAL,
              "rlx_sac_replay_draw_i32: bad args");
  // replay key = split(update_key, 2B + 2)[1]
  uint32_t r0, r1;
  const uint64_t nkeys = sac_key_count(B, 1);
  if (scheme == RLX_THREEFRY_PARTITIONABLE) {
    r0 = 0; r1 = 1;
    threefry2x32(update_key[0], update_key[1], r0, r1);
  } else {
    r0 = random_bits_at(update_key[0], update_key[1], 2, 2ull * nkeys, RLX_THREEFRY_LEGACY);
    r1 = random_bits_at(update_key[0], update_key[1], 3, 2ull * nkeys, RLX_THREEFRY_LEGACY);
  }
  hipLaunchKernelGGL(k_replay_draw, dim3(div_up(B, 256)), dim3(256), 0, (hipStream_t)stream, r0, r1, scheme, idx1, idx2, B,
                     (uint32_t)size, (uint32_t)nr_envs);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_sac_act_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams, const float* obs, uint32_t key_io[2],
                    int scheme, float* action, int N, float log_std_min, float log_std_max, int deterministic,
                    int row_offset, int N_global, void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && obs && key_io && action && N > 0 && N_global >= N, RLX_EINVAL,
              "rlx_sac_act_f32: bad args");
  RLX_REQUIRE(pdesc->out_dim % 2 == 0, RLX_EINVAL, "rlx_sac_act_f32: policy out_dim must be 2 * act_dim (mean | log_std)");
  int rc = mlp_check_desc(*pdesc);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int A = pdesc->out_dim / 2;
  float* head = (float*)scratch(ctx, SL_MEAN, (size_t)N * 2 * A * sizeof(float));
  if (!head) return RLX_ENOMEM;
  rc = rlx_mlp_fwd_f32(ctx, pdesc, pparams, obs, head, N, stream);
  if (rc) return rc;
  uint32_t ks[4] = {key_io[0], key_io[1], 0, 0};
  if (!deterministic) {
    split_host(key_io, ks, 2, scheme);  // key, subkey = split(key)   (sac.py:123)
    key_io[0] = ks[0];
    key_io[1] = ks[1];
  }
  hipLaunchKernelGGL(k_sac_sample, dim3(div_up(N, 256)), dim3(256), 0, st, head, ks[2], ks[3], scheme, 0, action, A, 0,
                     (float*)nullptr, (int64_t)N, A, log_std_min, log_std_max, row_offset, (int64_t)N_global,
                     deterministic);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_sac_update_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, float* pparams, float* pm, float* pv,
                       const rlx_mlp_desc* qdesc, float* qparams, float* qm, float* qv, float* qtarget, float* log_alpha,
                       float* am, float* av, const float* states, const float* next_states, const float* actions,
                       const float* rewards, const float* terminations, int64_t B, uint32_t key_io[2], int scheme,
                       int64_t* opt_count_io, const rlx_sac_hparams* hp, float* metrics_out, void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && pm && pv && qdesc && qparams && qm && qv && qtarget && log_alpha && am && av &&
                  states && next_states && actions && rewards && terminations && key_io && opt_count_io && hp && metrics_out,
              RLX_EINVAL, "rlx_sac_update_f32: NULL pointer");
  RLX_REQUIRE(B > 0, RLX_EINVAL, "rlx_sac_update_f32: batch must be positive");
  int rc = mlp_check_desc(*pdesc);
  if (rc) return rc;
  rc = mlp_check_desc(*qdesc);
  if (rc) return rc;
  const int O = pdesc->in_dim, A = pdesc->out_dim / 2;
  RLX_REQUIRE(pdesc->out_dim == 2 * A && qdesc->in_dim == O + A && qdesc->out_dim == 1 && !pdesc->has_logstd, RLX_EINVAL,
              "rlx_sac_update_f32: policy out_dim = 2*act_dim (no logstd param), critic in_dim = obs+act, out_dim = 1");
  hipStream_t st = (hipStream_t)stream;
  const MlpLayout LP = make_layout(*pdesc), LQ = make_layout(*qdesc);
  const int64_t np_ = LP.n_params, nq_ = LQ.n_params;
  const int ldc = (O + A + 3) & ~3;
  const int lda = (A + 3) & ~3;
  const int nb = div_up(B, 256);
  // ---- scratch arena
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 63) & ~size_t(63); return o; };
  int hmax = 0;
  for (int l = 0; l < pdesc->n_hidden; ++l) hmax = pdesc->hidden[l] > hmax ? pdesc->hidden[l] : hmax;
  for (int l = 0; l < qdesc->n_hidden; ++l) hmax = qdesc->hidden[l] > hmax ? qdesc->hidden[l] : hmax;
  const size_t o_xc = take((size_t)B * ldc), o_xn = take((size_t)B * ldc), o_xp = take((size_t)B * ldc);
  // act sets: 0 tmp (next-state policy, target critics), 1 policy on s, 2 / 3 online critics on (s, a),
  //           4 / 5 online critics on (s, pi(s))  -- separate so the critic-loss and policy-loss chains can overlap
  size_t o_acts[6][4];                       // [.][3]: pre-LayerNorm values of the first layer (full-jit nets)
  const bool any_ln = pdesc->ln_first || qdesc->ln_first;
  for (int s = 0; s < 6; ++s)
    for (int l = 0; l < 4; ++l) o_acts[s][l] = (l < 3 || any_ln) ? take((size_t)B * hmax) : 0;
  const size_t o_hn = take((size_t)B * 2 * A), o_hc = take((size_t)B * 2 * A), o_dpi = take((size_t)B * 2 * A);
  const size_t o_vec = take((size_t)B * 12);  // qt0 qt1 q0 q1 qa0 qa1 lpn lpc dq0 dq1 d0 d1
  const size_t o_da0 = take((size_t)B * lda), o_da1 = take((size_t)B * lda);
  const size_t o_gp = take(np_), o_gq = take(2 * nq_), o_ga = take(64);
  const size_t o_part = take((size_t)nb * 3 + 64);
  const size_t hp_floats = (size_t)div_up(B, SAC_HEAD_ROWS) * ((size_t)LP.head.in * LP.head.out + LP.head.out + LQ.head.in + 1);
  const size_t o_hpart = take(hp_floats), o_hpart2 = take(hp_floats);
  float* base = (float*)scratch(ctx, SL_SAC, off * sizeof(float));
  float* sq0 = (float*)scratch(ctx, SL_NORM, REDUCE_MAX_BLOCKS * sizeof(float));
  float* sq1 = (float*)scratch(ctx, SL_NORM2, REDUCE_MAX_BLOCKS * sizeof(float));
  if (!base || !sq0 || !sq1) return RLX_ENOMEM;
  float *xc = base + o_xc, *xn = base + o_xn, *xp = base + o_xp;
  NetBufs nbuf[6];
  for (int s = 0; s < 6; ++s)
    for (int l = 0; l < 4; ++l) nbuf[s].acts[l] = (l < 3 || any_ln) ? base + o_acts[s][l] : nullptr;
  float *hn = base + o_hn, *hc = base + o_hc, *dpi = base + o_dpi;
  float* vec = base + o_vec;
  float *qt0 = vec, *qt1 = vec + B, *q0 = vec + 2 * B, *q1 = vec + 3 * B, *qa0 = vec + 4 * B, *qa1 = vec + 5 * B,
        *lpn = vec + 6 * B, *lpc = vec + 7 * B, *dq0 = vec + 8 * B, *dq1 = vec + 9 * B, *d0 = vec + 10 * B,
        *d1 = vec + 11 * B;
  float *da0 = base + o_da0, *da1 = base + o_da1;
  float *gp = base + o_gp, *gq = base + o_gq, *ga = base + o_ga;
  float *part_c = base + o_part, *part_p = part_c + nb;
  float* hpart = base + o_hpart;
  float* hpart_p = base + o_hpart2;

  // keys = split(key, 2B+1 [+1]); key = keys[0]
  const uint32_t k0 = key_io[0], k1 = key_io[1];
  const int ksched = hp->key_schedule ? 1 : 0;
  {
    uint32_t nk[2];
    const uint64_t nkeys = sac_key_count(B, ksched);
    if (scheme == RLX_THREEFRY_PARTITIONABLE) {
      uint32_t x0 = 0, x1 = 0;
      threefry2x32(k0, k1, x0, x1);
      nk[0] = x0; nk[1] = x1;
    } else {
      nk[0] = random_bits_at(k0, k1, 0, 2ull * nkeys, RLX_THREEFRY_LEGACY);
      nk[1] = random_bits_at(k0, k1, 1, 2ull * nkeys, RLX_THREEFRY_LEGACY);
    }
    key_io[0] = nk[0];
    key_io[1] = nk[1];
  }
  {
    int grid = div_up((int64_t)B * ldc, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_sac_concat, dim3(grid), dim3(256), 0, st, states, next_states, actions, xc, xn, xp, B, O, A, ldc);
    RLX_LAUNCH_CHECK();
  }
  // policy input: narrow observations are read dense [B, O] by the small-input first-layer kernel; wide ones from the
  // observation columns of the padded concat buffers (row stride ldc, a multiple of 4 whatever O is; the action
  // columns meet zero-guarded weight rows)
  const int ldo = O > 32 ? ldc : O;
  const float* pol_next = O > 32 ? xn : next_states;
  const float* pol_cur = O > 32 ? xc : states;
  // Two chains that only meet at the seed kernels and at the optimizer (sac.py:133-188 evaluates both losses on the
  // same, pre-update parameters):
  //   main stream : policy(s') -> a', log pi -> target critics -> [wait q0, q1] -> critic seed -> critic backward
  //   side stream : online critics on (s, a) -> policy(s) -> a~, log pi -> critics on (s, a~) -> seed -> dQ/da -> policy backward
  // (B = 4096 rows fill a quarter of the chip per GEMM: the chains overlap almost for free.)
  // split-bf16 weight images of all five networks for the GEMMs of this update (batches >= 4096 rows; the parameters do not
  // change before the optimizer steps at the end): one launch, in front of the fork
  struct BxAll { rlx_ctx* c; ~BxAll() { bx_release_all(c); } } bx_all{ctx};
  if (B >= 4096) {
    const bool pw = pdesc->in_dim > 32, qw = true;   // critics always run their first layer on the GEMM kernels
    const BxNetSpec nets[5] = {{pdesc, pparams, true, pw}, {qdesc, qparams, true, qw}, {qdesc, qparams + nq_, true, qw},
                               {qdesc, qtarget, false, qw}, {qdesc, qtarget + nq_, false, qw}};
    rc = bx_prepare_nets(ctx, nets, 5, st);
    if (rc) return rc;
  }
  hipStream_t sy = st;
  if (ctx->two_streams) {
    rc = ctx_side_stream(ctx);
    if (rc) return rc;
    sy = ctx->side;
    RLX_HIP_TRY(hipEventRecord(ctx->ev_fork, st));
    RLX_HIP_TRY(hipStreamWaitEvent(sy, ctx->ev_fork, 0));
  }
  int nsq_q0 = 0, nsq_q1 = 0, nsq_p = 0;
  // ---- side chain, part 1: both online critics on (s, a) (critic 0 keeps its activations in set 2, critic 1 in set 3)
  ctx->bank = sy != st ? 1 : 0;
  rc = net_fwd(ctx, *qdesc, LQ, qparams, xc, ldc, nbuf[2].acts, q0, B, sy);
  if (!rc) rc = net_fwd(ctx, *qdesc, LQ, qparams + nq_, xc, ldc, nbuf[3].acts, q1, B, sy);
  ctx->bank = 0;
  if (rc) return rc;
  if (sy != st) RLX_HIP_TRY(hipEventRecord(ctx->ev_join, sy));
  // ---- main chain: critic loss
  rc = net_fwd(ctx, *pdesc, LP, pparams, pol_next, ldo, nbuf[0].acts, hn, B, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sac_sample, dim3(nb), dim3(256), 0, st, hn, k0, k1, scheme, 1, xn, ldc, O, lpn, B, A,
                     hp->log_std_min, hp->log_std_max, 0, B, 0, ctx->dbg_sac_eps[0], ksched);
  RLX_LAUNCH_CHECK();
  rc = net_fwd(ctx, *qdesc, LQ, qtarget, xn, ldc, nbuf[0].acts, qt0, B, st);
  if (rc) return rc;
  rc = net_fwd(ctx, *qdesc, LQ, qtarget + nq_, xn, ldc, nbuf[0].acts, qt1, B, st);
  if (rc) return rc;
  if (sy != st) RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));   // q0, q1 are ready
  hipLaunchKernelGGL(k_sac_critic_seed, dim3(nb), dim3(256), 0, st, qt0, qt1, lpn, rewards, terminations, log_alpha, q0,
                     q1, dq0, dq1, part_c, B, hp->gamma);
  RLX_LAUNCH_CHECK();
  rc = net_bwd(ctx, *qdesc, LQ, qparams, xc, ldc, nbuf[2].acts, dq0, gq, hpart, B, sq0, &nsq_q0, nullptr, st);
  if (rc) return rc;
  // the two critics are ONE optimizer state in the reference: their squared norms are summed
  rc = net_bwd(ctx, *qdesc, LQ, qparams + nq_, xc, ldc, nbuf[3].acts, dq1, gq + nq_, hpart, B, sq0 + nsq_q0, &nsq_q1,
               nullptr, st);
  if (rc) return rc;
  RLX_REQUIRE(nsq_q0 + nsq_q1 <= REDUCE_MAX_BLOCKS, RLX_EUNSUP, "sac: critic too large for the norm partial array");
  // ---- side chain, part 2: policy loss (critic activations of this chain live in sets 4 / 5)
  ctx->bank = sy != st ? 1 : 0;
  rc = net_fwd(ctx, *pdesc, LP, pparams, pol_cur, ldo, nbuf[1].acts, hc, B, sy);
  if (!rc) {
    hipLaunchKernelGGL(k_sac_sample, dim3(nb), dim3(256), 0, sy, hc, k0, k1, scheme, 2, xp, ldc, O, lpc, B, A,
                       hp->log_std_min, hp->log_std_max, 0, B, 0, ctx->dbg_sac_eps[1], ksched);
    rc = net_fwd(ctx, *qdesc, LQ, qparams, xp, ldc, nbuf[4].acts, qa0, B, sy);
  }
  if (!rc) rc = net_fwd(ctx, *qdesc, LQ, qparams + nq_, xp, ldc, nbuf[5].acts, qa1, B, sy);
  if (!rc) {
    hipLaunchKernelGGL(k_sac_policy_seed, dim3(nb), dim3(256), 0, sy, qa0, qa1, lpc, d0, d1, part_p, nb, B);
    TrunkOpts opt;
    opt.dx_c0 = O; opt.dx_nc = A; opt.dx_ld = lda;
    opt.dx_out = da0;
    rc = net_bwd(ctx, *qdesc, LQ, qparams, xp, ldc, nbuf[4].acts, d0, nullptr, nullptr, B, nullptr, nullptr, &opt, sy);
    if (!rc) {
      opt.dx_out = da1;
      rc = net_bwd(ctx, *qdesc, LQ, qparams + nq_, xp, ldc, nbuf[5].acts, d1, nullptr, nullptr, B, nullptr, nullptr, &opt, sy);
    }
  }
  if (!rc) {
    hipLaunchKernelGGL(k_sac_policy_grad, dim3(nb), dim3(256), 0, sy, hc, xp, ldc, O, da0, da1, lda, log_alpha, k0, k1, scheme,
                       dpi, B, A, hp->log_std_min, hp->log_std_max, ctx->dbg_sac_eps[1], ksched);
    rc = net_bwd(ctx, *pdesc, LP, pparams, pol_cur, ldo, nbuf[1].acts, dpi, gp, hpart_p, B, sq1, &nsq_p, nullptr, sy);
  }
  ctx->bank = 0;
  if (rc) return rc;
  RLX_LAUNCH_CHECK();
  if (sy != st) {
    RLX_HIP_TRY(hipEventRecord(ctx->ev_join, sy));
    RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
  }
  // ---- entropy coefficient gradient + metrics
  hipLaunchKernelGGL(k_sac_finalize, dim3(1), dim3(64), 0, st, part_c, part_p, nb, log_alpha, ga, metrics_out, B,
                     hp->target_entropy);
  RLX_LAUNCH_CHECK();
  // ---- three plain Adam steps (no clipping, sac.py:95,102,108) + Polyak (sac.py:208)
  const int64_t step = *opt_count_io + 1;
  rc = launch_clip_adam(pparams, gp, pm, pv, np_, sq1, nsq_p, step, hp->lr_policy, -1.f, hp->adam_b1, hp->adam_b2,
                        hp->adam_eps, metrics_out + 6, st);
  if (rc) return rc;
  rc = launch_clip_adam(qparams, gq, qm, qv, 2 * nq_, sq0, nsq_q0 + nsq_q1, step, hp->lr_critic, -1.f, hp->adam_b1,
                        hp->adam_b2, hp->adam_eps, metrics_out + 7, st);
  if (rc) return rc;
  rc = rlx_clip_adam_step_f32(ctx, log_alpha, ga, am, av, 1, step, hp->lr_alpha, -1.f, hp->adam_b1, hp->adam_b2,
                              hp->adam_eps, metrics_out + 8, stream);
  if (rc) return rc;
  {
    int grid = div_up(2 * nq_, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_polyak, dim3(grid), dim3(256), 0, st, qtarget, qparams, 2 * nq_, hp->tau);
    RLX_LAUNCH_CHECK();
  }
  *opt_count_io += 1;
  return RLX_OK;
}

int rlx_dbg_set_sac_noise(rlx_ctx* ctx, const float* eps_next, const float* eps_cur) {
  RLX_REQUIRE(ctx, RLX_EINVAL, "rlx_dbg_set_sac_noise: ctx is NULL");
  ctx->dbg_sac_eps[0] = eps_next;
  ctx->dbg_sac_eps[1] = eps_cur;
  return RLX_OK;
}

}  // extern "C"
