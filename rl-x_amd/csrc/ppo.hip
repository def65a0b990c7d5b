// ppo.hip -- PPO acting, minibatch loss/gradient and the whole `update` for gfx950.
// Replaces the XLA fusions of
//   get_action_and_value   rl_x/algorithms/ppo/flax/ppo.py:110-119
//   loss_fn                rl_x/algorithms/ppo/flax/ppo.py:142-177 (vmapped :186, meaned :187-188)
//   minibatch_update       rl_x/algorithms/ppo/flax/ppo.py:196-220
//   update                 rl_x/algorithms/ppo/flax/ppo.py:138-232
// CPU twin: oracle/ppo.py.
#include "ppo_internal.h"
#include "gemm_bx.h"
#include "dist.h"

namespace rlx {

constexpr float LOG_2PI = 1.8378770664093453f;
constexpr float HALF_LOG_2PIE = 1.4189385332046727f;  // 0.5 * log(2*pi*e)
constexpr int HEAD_ROWS = 64;

// ---------------------------------------------------------------------------------------
// K5: gather the minibatch rows (flattened index i = t*N + n, ppo.py:180-184) into dense
// [mb, O] / [mb, A] / [mb, 3] buffers (coalesced writes; reads are 68-B / 24-B row pieces)
// and accumulate sum(adv), sum(adv^2), count in double (ppo.py:199-200 statistics).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ states, const float* __restrict__ actions,
                                                const float* __restrict__ logp, const float* __restrict__ returns,
                                                const float* __restrict__ adv, const int32_t* __restrict__ idx,
                                                float* __restrict__ mb_x, float* __restrict__ mb_a,
                                                float* __restrict__ aux, const int32_t* __restrict__ valid_rows,
                                                int64_t mb, int O, int A, const float* __restrict__ cstates,
                                                float* __restrict__ mb_xc, int Oc) {
  // cstates (optional, [B, Oc]): the critic's own observation rows -- an env with critic_observation_indices !=
  // policy_observation_indices (ppo/flax/critic.py:12,24 vs policy.py:13,33): the same rows idx[] gathered into mb_xc
  // valid_rows (device, optional): rows [*valid_rows, mb) are PADDING of a rank-local minibatch (data-parallel update):
  // they gather row 0 of the rollout (finite values; the head/loss kernels give them zero weight).
  // The advantage statistics of ppo.py:199-200 are NOT accumulated here: k_mb_adv_sums (dist.hip) produces them for all
  // minibatches of an update call at once, one workgroup per minibatch in a fixed order -- reproducible bit for bit
  // (no floating-point atomics) and off the per-update critical path.
  const int64_t nv = valid_rows ? (int64_t)*valid_rows : mb;
  const int64_t nx = mb * O, na = mb * A, nxc = cstates ? mb * Oc : 0;
  const int64_t total = nx + na + mb + nxc;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    if (e >= nx + na + mb) {
      const int64_t f = e - nx - na - mb;
      const int64_t r = f / Oc;
      const int d = (int)(f - r * Oc);
      mb_xc[f] = cstates[(int64_t)(r < nv ? idx[r] : 0) * Oc + d];
    } else if (e < nx) {
      const int64_t r = e / O;
      const int d = (int)(e - r * O);
      mb_x[e] = states[(int64_t)(r < nv ? idx[r] : 0) * O + d];
    } else if (e < nx + na) {
      const int64_t f = e - nx;
      const int64_t r = f / A;
      const int d = (int)(f - r * A);
      mb_a[f] = actions[(int64_t)(r < nv ? idx[r] : 0) * A + d];
    } else {
      const int64_t r = e - nx - na;
      const int64_t i = r < nv ? idx[r] : 0;
      aux[r * 3 + 0] = logp[i];
      aux[r * 3 + 1] = returns[i];
      aux[r * 3 + 2] = adv[i];
    }
  }
}

// The whole-update calls gather every rollout row once per epoch, and a sampled row touches six 128-B lines of the five source
// arrays (68-B observation and action rows straddle lines half of the time; log-prob, return and advantage are a line each):
// 26 MB of HBM traffic per 32768-row minibatch for 3.4 MB of rows (profiles/r06_pmc_traffic.md).  So the call first lays
// the rollout out as one aligned record per row -- [obs(O) | action(A) | log_prob, return, advantage | pad] of 32 / 64 / 128
// floats (k_pack_rows: one coalesced pass, 134 MB at T*N = 524288) -- and its E*M gathers read two lines per row.
__global__ __launch_bounds__(256) void k_pack_rows(const float* __restrict__ states, const float* __restrict__ actions,
                                                   const float* __restrict__ logp, const float* __restrict__ returns,
                                                   const float* __restrict__ adv, float4* __restrict__ rec, int64_t B, int O,
                                                   int A, int lg4) {
  const int64_t total = B << lg4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e >> lg4;
    const int q = (int)(e & ((1 << lg4) - 1));
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = 4 * q + j;
      float x = 0.f;
      if (p < O) x = states[r * O + p];
      else if (p < O + A) x = actions[r * A + (p - O)];
      else if (p == O + A) x = logp[r];
      else if (p == O + A + 1) x = returns[r];
      else if (p == O + A + 2) x = adv[r];
      v[j] = x;
    }
    rec[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// k_gather from the records: 1 << lg4 lanes per row, one 16-B load each
// grp_rows > 0: the launch covers mb / grp_rows consecutive updates of grp_rows rows each, valid_rows[j] = valid rows of the j-th
__global__ __launch_bounds__(256) void k_gather_rec(const float4* __restrict__ rec, const int32_t* __restrict__ idx,
                                                    float* __restrict__ mb_x, float* __restrict__ mb_a, float* __restrict__ aux,
                                                    const int32_t* __restrict__ valid_rows, int64_t mb, int O, int A, int lg4,
                                                    int grp_rows) {
  const int64_t total = mb << lg4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e >> lg4;
    const int q = (int)(e & ((1 << lg4) - 1));
    if (4 * q >= O + A + 3) continue;
    bool live = true;
    if (valid_rows) {
      if (grp_rows > 0) {
        const int j = (int)r / grp_rows;
        live = (int)r - j * grp_rows < valid_rows[j];
      } else {
        live = r < (int64_t)*valid_rows;
      }
    }
    const int64_t i = live ? idx[r] : 0;
    const float4 v4 = rec[(i << lg4) + q];
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = 4 * q + j;
      if (p < O) mb_x[r * O + p] = v[j];
      else if (p < O + A) mb_a[r * A + (p - O)] = v[j];
      else if (p < O + A + 3) aux[r * 3 + (p - O - A)] = v[j];
    }
  }
}

__device__ __forceinline__ void adv_norm_from_stats(const double* __restrict__ stats, float& mean, float& inv,
                                                    float& stdv) {
  const double cnt = stats[2] > 0.0 ? stats[2] : 1.0;
  const double m = stats[0] / cnt;
  double var = stats[1] / cnt - m * m;
  if (var < 0.0) var = 0.0;
  const float sd = (float)sqrt(var);
  mean = (float)m;
  stdv = sd;
  inv = 1.0f / (sd + 1e-8f);
}

// ---------------------------------------------------------------------------------------
// K6 head: output layer forward + PPO loss + its gradient seeds + head weight gradients.
//   POLICY: mean = h @ W + b; Gaussian log-prob, ratio, clipped surrogate, approx-KL, clip
//           fraction; d mean, d logstd;             (loss_fn, ppo.py:144-160)
//   CRITIC: v = h @ w + b; 0.5 (v - R)^2; d v.      (ppo.py:162-166)
// then dZ_last = (d_out @ W^T) * act'(h) written IN PLACE over h, and per-block partials of
// dW_head[K,A], db_head[A], dlogstd[A], metric sums -> partials[block][PS].
// 64 rows per workgroup; everything staged in LDS.
// ---------------------------------------------------------------------------------------
template <bool POLICY, bool DISCRETE = false>
__global__ __launch_bounds__(256) void k_head_loss(float* __restrict__ H, const float* __restrict__ W,
                                                   const float* __restrict__ b, const float* __restrict__ logstd,
                                                   const float* __restrict__ mb_a, const float* __restrict__ aux,
                                                   const double* __restrict__ stats, float* __restrict__ partials,
                                                   float* __restrict__ metrics, int64_t M, int K, int A, int PS,
                                                   float inv_mb, float clip, float ent_coef, float critic_coef,
                                                   int act, const int32_t* __restrict__ valid_rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t Mv = valid_rows ? (int64_t)*valid_rows : M;   // rows [Mv, M): padding, zero weight
  const int HS = K + 1;
  float* Hs = smem;                    // [64][K+1]
  float* Ws = Hs + HEAD_ROWS * HS;     // [K][A]
  float* Ms = Ws + K * A;              // [64][A]  mean -> d_out
  float* DL = Ms + HEAD_ROWS * A;      // [64][A]  d logstd terms
  float* bs = DL + HEAD_ROWS * A;      // [A]
  float* ls = bs + A;                  // [A]
  const int t = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * HEAD_ROWS;
  for (int i = t; i < HEAD_ROWS * K; i += 256) {
    const int r = i / K, k = i - r * K;
    Hs[r * HS + k] = (r0 + r < M) ? H[(r0 + r) * K + k] : 0.f;
  }
  lds_stage<256>(Ws, W, K * A);
  if (t < A) {
    bs[t] = b[t];
    ls[t] = (POLICY && !DISCRETE) ? logstd[t] : 0.f;
  }
  __syncthreads();
  {  // head forward
    const int r = t & 63;
    for (int a = t >> 6; a < A; a += 4) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(Hs[r * HS + k], Ws[k * A + a], acc);
      Ms[r * A + a] = acc + bs[a];
    }
  }
  __syncthreads();
  if (t < 64) {  // one lane per row
    const int r = t;
    const int64_t row = r0 + r;
    const bool valid = row < Mv;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    if (POLICY) {
      float nlp = 0.f;
      // Categorical head (ppo/pytorch/policy.py:118-130): log-softmax of the logits, log-prob of the stored action index,
      // per-sample entropy; A <= 8
      float lsm[8], ent_row = 0.f;
      int ai = 0;
      if (DISCRETE) {
        float zmax = -3.0e38f;
        for (int a = 0; a < A; ++a) zmax = fmaxf(zmax, Ms[r * A + a]);
        float se = 0.f;
        for (int a = 0; a < A; ++a) se += expf(Ms[r * A + a] - zmax);
        const float lse = zmax + logf(se);
        for (int a = 0; a < A; ++a) {
          lsm[a] = Ms[r * A + a] - lse;
          ent_row -= expf(lsm[a]) * lsm[a];
        }
        if (valid) {
          ai = (int)mb_a[row];
          ai = ai < 0 ? 0 : (ai >= A ? A - 1 : ai);
          nlp = lsm[ai];
          m3 = ent_row;
        }
      } else if (valid) {
        for (int a = 0; a < A; ++a) {
          const float sd = expf(ls[a]);
          const float zs = (mb_a[row * A + a] - Ms[r * A + a]) / sd;
          nlp += -0.5f * zs * zs - 0.5f * LOG_2PI - ls[a];
        }
      }
      float amean, ainv, astd;
      adv_norm_from_stats(stats, amean, ainv, astd);
      const float logp_old = valid ? aux[row * 3 + 0] : 0.f;
      const float advn = valid ? (aux[row * 3 + 2] - amean) * ainv : 0.f;
      const float logratio = nlp - logp_old;
      const float ratio = valid ? expf(logratio) : 1.f;
      const float pg1 = -advn * ratio;
      const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
      const float pg2 = -advn * rc;
      const bool inside = (ratio >= 1.f - clip) && (ratio <= 1.f + clip);
      const float d_ratio = (inside || pg1 > pg2) ? -advn : 0.f;
      const float d_logp = valid ? d_ratio * ratio * inv_mb : 0.f;
      for (int a = 0; a < A; ++a) {
        float dm = 0.f, dl = 0.f;
        if (valid) {
          if (DISCRETE) {
            // d logp / d z_j = [j == a] - p_j;  d(-ent_coef * mean H) / d z_j = ent_coef / mb * p_j (log p_j + H)
            const float pj = expf(lsm[a]);
            dm = d_logp * ((a == ai ? 1.f : 0.f) - pj) + ent_coef * inv_mb * pj * (lsm[a] + ent_row);
          } else {
            const float sd = expf(ls[a]);
            const float zs = (mb_a[row * A + a] - Ms[r * A + a]) / sd;
            dm = d_logp * zs / sd;
            dl = d_logp * (zs * zs - 1.f) - ent_coef * inv_mb;   // + d(-ent_coef * entropy) / d logstd, per weighted row
          }
        }
        Ms[r * A + a] = dm;
        DL[r * A + a] = dl;
      }
      m0 = valid ? fmaxf(pg1, pg2) : 0.f;
      m1 = valid ? (ratio - 1.f) - logratio : 0.f;
      m2 = (valid && fabsf(ratio - 1.f) > clip) ? 1.f : 0.f;
      if (blockIdx.x == 0 && t == 0) {
        float ent = 0.f, sstd = 0.f;
        if (!DISCRETE) {
          for (int a = 0; a < A; ++a) { ent += ls[a] + HALF_LOG_2PIE; sstd += expf(ls[a]); }
          metrics[2] = ent;          // (Categorical: the mean per-sample entropy arrives through the partials)
        }
        metrics[5] = amean;
        metrics[6] = astd;
        metrics[7] = sstd / (float)A;  // mean policy std (metric policy/std_dev, ppo.py:230) before this update; 0 for Categorical
      }
    } else {
      float dv = 0.f;
      if (valid) {
        const float diff = Ms[r] - aux[row * 3 + 1];
        m0 = 0.5f * diff * diff;
        dv = critic_coef * inv_mb * diff;
      }
      Ms[r] = dv;
    }
    m0 = wave_sum(m0);
    m1 = wave_sum(m1);
    m2 = wave_sum(m2);
    m3 = wave_sum(m3);
    if (t == 0) {
      float* pm = partials + (int64_t)blockIdx.x * PS + K * A + 2 * A;
      pm[0] = m0; pm[1] = m1; pm[2] = m2; pm[3] = m3;
    }
  }
  __syncthreads();
  // dZ_last (in place over H)
  for (int e = t; e < HEAD_ROWS * K; e += 256) {
    const int r = e / K, k = e - r * K;
    if (r0 + r < M) {
      float acc = 0.f;
      for (int a = 0; a < A; ++a) acc = fmaf(Ms[r * A + a], Ws[k * A + a], acc);
      H[(r0 + r) * K + k] = acc * act_grad_from_out(Hs[r * HS + k], act);
    }
  }
  // head weight / bias / logstd partials
  float* pw = partials + (int64_t)blockIdx.x * PS;
  for (int e = t; e < K * A; e += 256) {
    const int k = e / A, a = e - k * A;
    float acc = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) acc = fmaf(Hs[r * HS + k], Ms[r * A + a], acc);
    pw[e] = acc;
  }
  if (t < A) {
    float sb = 0.f, sl = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) {
      sb += Ms[r * A + t];
      if (POLICY) sl += DL[r * A + t];
    }
    pw[K * A + t] = sb;
    pw[K * A + A + t] = sl;
  }
}

// ---------------------------------------------------------------------------------------
// Acting epilogue: a = mean + exp(logstd) * eps, log-prob, optional clip+rescale
// (get_processed_action_function, ppo/flax/policy.py:43-50), Batch.states[t] copy.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample(const float* __restrict__ mean, const float* __restrict__ logstd,
                                                uint32_t k0, uint32_t k1, int scheme, float* __restrict__ action,
                                                float* __restrict__ processed, float* __restrict__ logp,
                                                const float* __restrict__ obs, float* __restrict__ states_row, int N,
                                                int A, int O, int clip_and_rescale, const float* __restrict__ lo,
                                                const float* __restrict__ hi, int env_off, int N_global,
                                                int deterministic) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const uint64_t total = (uint64_t)N_global * A;
  float lp = 0.f;
  for (int a = 0; a < A; ++a) {
    const uint64_t i = (uint64_t)(n + env_off) * A + a;
    const float eps = deterministic ? 0.f : normal_from_bits(random_bits_at(k0, k1, i, total, scheme));
    const float ls = logstd[a];
    const float sd = expf(ls);
    const float mu = mean[(int64_t)n * A + a];
    const float act = mu + sd * eps;
    const float zs = (act - mu) / sd;
    lp += -0.5f * zs * zs - 0.5f * LOG_2PI - ls;
    action[(int64_t)n * A + a] = act;
    if (processed) {
      float p = act;
      if (clip_and_rescale) {
        const float c = fminf(fmaxf(act, -1.f), 1.f);
        p = lo[a] + 0.5f * (c + 1.0f) * (hi[a] - lo[a]);
      }
      processed[(int64_t)n * A + a] = p;
    }
  }
  logp[n] = lp;
  if (states_row)
    for (int d = 0; d < O; ++d) states_row[(int64_t)n * O + d] = obs[(int64_t)n * O + d];
}

// Categorical acting epilogue (DiscreteFlatValuesPolicy.get_action_logprob, ppo/pytorch/policy.py:118-124, with the noise
// drawn the JAX way: jax.random.categorical = argmax(logits + Gumbel(key)), Gumbel = -log(-log(U)), U uniform in
// [tiny, 1) from the threefry bits of element n * A + a).  action[n] = index as float, logp[n] = log_softmax[index].
__global__ __launch_bounds__(256) void k_sample_categorical(const float* __restrict__ logits, uint32_t k0, uint32_t k1,
                                                            int scheme, float* __restrict__ action, float* __restrict__ logp,
                                                            const float* __restrict__ obs, float* __restrict__ states_row,
                                                            int N, int A, int O, int env_off, int N_global,
                                                            int deterministic) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const uint64_t total = (uint64_t)N_global * A;
  float zmax = -3.0e38f;
  for (int a = 0; a < A; ++a) zmax = fmaxf(zmax, logits[(int64_t)n * A + a]);
  float se = 0.f, best = -3.0e38f;
  int ai = 0;
  for (int a = 0; a < A; ++a) {
    const float z = logits[(int64_t)n * A + a];
    se += expf(z - zmax);
    float score = z;
    if (!deterministic) {
      const uint32_t bits = random_bits_at(k0, k1, (uint64_t)(n + env_off) * A + a, total, scheme);
      float u = __uint_as_float((bits >> 9) | 0x3f800000u) - 1.0f;          // jax.random.uniform: [0, 1)
      u = fmaxf(1.17549435e-38f, u * (1.0f - 1.17549435e-38f) + 1.17549435e-38f);   // minval = tiny
      score = z - logf(-logf(u));
    }
    if (score > best) { best = score; ai = a; }                            // argmax: first maximum
  }
  action[n] = (float)ai;
  logp[n] = logits[(int64_t)n * A + ai] - (zmax + logf(se));
  if (states_row)
    for (int d = 0; d < O; ++d) states_row[(int64_t)n * O + d] = obs[(int64_t)n * O + d];
}

// one launch of k_gather (+ one workgroup of k_mb_adv_sums when this minibatch's statistics {sum adv, sum adv^2, count}
// are wanted at `stats`; the whole-update entry points compute them for all minibatches up front instead)
static int launch_gather(rlx_ctx* ctx, const float* states, const float* actions, const float* log_probs,
                         const float* returns, const float* advantages, const int32_t* idx, const MbScratch& s,
                         double* stats, const int32_t* valid_rows, int64_t mb, int O, int A_act, hipStream_t st,
                         const float* cstates = nullptr, int Oc = 0, int grp_rows = 0) {
  // grp_rows > 0 (record source only): mb = several updates of grp_rows rows, valid_rows = their counts
  if (!s.mb_xc) cstates = nullptr;
  if (s.rec && !cstates) {
    const int64_t total = mb << s.rec_lg4;
    int grid = div_up(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_gather_rec, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(s.rec), idx, s.mb_x, s.mb_a,
                       s.aux, valid_rows, mb, O, A_act, s.rec_lg4, grp_rows);
    RLX_LAUNCH_CHECK();
    if (stats) return dist_adv_sums(advantages, idx, valid_rows, 1, (int)mb, (int)mb, stats, st);
    return RLX_OK;
  }
  const int64_t total = mb * (O + A_act + 1 + (cstates ? Oc : 0));
  int grid = div_up(total, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, st, states, actions, log_probs, returns, advantages, idx, s.mb_x,
                     s.mb_a, s.aux, valid_rows, mb, O, A_act, cstates, s.mb_xc, Oc);
  RLX_LAUNCH_CHECK();
  if (stats) return dist_adv_sums(advantages, idx, valid_rows, 1, (int)mb, (int)mb, stats, st);
  return RLX_OK;
}

// the rollout as aligned row records for the gathers of a whole-update call (k_pack_rows); leaves s[0..n) without records when the
// row does not fit 128 floats, the critic reads its own observation columns, or the option "gather_records" is 0
static int pack_rollout_rows(rlx_ctx* ctx, const float* states, const float* actions, const float* log_probs, const float* returns,
                             const float* advantages, int64_t B, int O, int A_act, bool critic_rows, MbScratch* s, int n,
                             hipStream_t st) {
  const int need = O + A_act + 3;
  if (!ctx->gather_records || critic_rows || need > 128) return RLX_OK;
  const int lg4 = need <= 32 ? 3 : (need <= 64 ? 4 : 5);
  float* rec = (float*)scratch(ctx, SL_ROW_REC, ((size_t)B << lg4) * 4 * sizeof(float));
  if (!rec) return RLX_ENOMEM;
  const int64_t total = B << lg4;
  int grid = div_up(total, 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, st, states, actions, log_probs, returns, advantages,
                     reinterpret_cast<float4*>(rec), B, O, A_act, lg4);
  RLX_LAUNCH_CHECK();
  for (int i = 0; i < n; ++i) { s[i].rec = rec; s[i].rec_lg4 = lg4; }
  return RLX_OK;
}

// ---------------------------------------------------------------------------------------
static int mb_scratch(rlx_ctx* ctx, const rlx_mlp_desc& pd, const rlx_mlp_desc& cd, int64_t mb, MbScratch* s,
                      bool critic_rows = false) {
  const int O = pd.in_dim, A = pd.out_dim;
  s->mb_x = (float*)scratch(ctx, SL_MB_X, (size_t)mb * (O + A) * sizeof(float));
  s->mb_xc = nullptr;
  if (critic_rows) {
    s->mb_xc = (float*)scratch(ctx, SL_MB_XC, (size_t)mb * cd.in_dim * sizeof(float));
    if (!s->mb_xc) return RLX_ENOMEM;
  }
  s->aux = (float*)scratch(ctx, SL_MB_AUX, (size_t)mb * 3 * sizeof(float));
  s->stats = (double*)scratch(ctx, SL_STATS, 64);
  if (!s->mb_x || !s->aux || !s->stats) return RLX_ENOMEM;
  s->mb_a = s->mb_x + (size_t)mb * O;
  const int nh = pd.n_hidden > cd.n_hidden ? pd.n_hidden : cd.n_hidden;
  for (int l = 0; l < nh; ++l) {
    int h = 0;
    if (l < pd.n_hidden) h = pd.hidden[l];
    if (l < cd.n_hidden && cd.hidden[l] > h) h = cd.hidden[l];
    s->acts[l] = (float*)scratch(ctx, (ScratchSlot)(SL_ACT_P0 + l), (size_t)mb * h * sizeof(float));
    if (!s->acts[l]) return RLX_ENOMEM;
  }
  if ((O > 32 || cd.in_dim > 32) && (pd.ln_first || cd.ln_first)) {   // wide observations + LayerNorm: the pre-LayerNorm values are kept
    const int h0 = pd.hidden[0] > cd.hidden[0] ? pd.hidden[0] : cd.hidden[0];
    s->acts[3] = (float*)scratch(ctx, SL_ACT_C0, (size_t)mb * h0 * sizeof(float));
    if (!s->acts[3]) return RLX_ENOMEM;
  }
  const int Kp = pd.hidden[pd.n_hidden - 1], Kc = cd.hidden[cd.n_hidden - 1];
  const size_t psp = (size_t)Kp * A + 2 * A + 8, psc = (size_t)Kc + 2 + 8;
  const size_t nb = (size_t)div_up(mb, 32);   // (the 32-row tail form leaves one partial block per 32 rows)
  s->head_part = (float*)scratch(ctx, SL_HEAD_PART, nb * (psp > psc ? psp : psc) * sizeof(float));
  if (!s->head_part) return RLX_ENOMEM;
  return RLX_OK;
}

// ---------------------------------------------------------------------------------------
// K6 head, register-resident form for K in {64, 128, 256} and A <= 8 (every configuration of BASELINE.json).
// Four threads per row: thread q keeps the K/4-wide slice q of the row's last hidden activation in registers
// (16-B loads), computes its share of the head dot products against W (LDS, [K][8] zero padded, broadcast
// reads), and a quad DPP reduction gives all four the outputs.  The loss and its seeds are computed per row in
// registers; dZ_last = (d_out @ W^T) * act'(h) is written back in place from the same registers (16-B stores).
// Only the head weight gradient needs the tile in LDS: dW[k][a] = sum_r h[r][k] d[r][a] over the 64 rows,
// summed in a fixed order (deterministic).  Same partials layout as k_head_loss.
// ---------------------------------------------------------------------------------------
typedef float hl_f4 __attribute__((ext_vector_type(4)));

template <bool POLICY, int KQ>
__device__ __forceinline__ void head_loss_fast_body(float* __restrict__ H, const float* __restrict__ W,
                                                    const float* __restrict__ b, const float* __restrict__ logstd,
                                                    const float* __restrict__ mb_a, const float* __restrict__ aux,
                                                    const double* __restrict__ stats, float* __restrict__ partials,
                                                    float* __restrict__ metrics, int64_t M, int A, int PS, float inv_mb,
                                                    float clip, float ent_coef, float critic_coef, int act,
                                                    const int32_t* __restrict__ valid_rows) {
  constexpr int K = 4 * KQ, HS = K + 1, AP = 8, NP = 256 / K > 0 ? 256 / K : 1, RP = HEAD_ROWS / NP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                     // [K][8]
  float* Ds = Ws + K * AP;              // [64][8]  d_out rows
  float* DLs = Ds + HEAD_ROWS * AP;     // [64][8]  d logstd terms
  float* Hs = DLs + HEAD_ROWS * AP;     // [64][K+1]
  float* red = Hs + HEAD_ROWS * HS;     // [NP][K][8] dW partial sums of the row groups
  const int t = threadIdx.x, r = t >> 2, q = t & 3;
  const int64_t row = (int64_t)blockIdx.x * HEAD_ROWS + r;
  const bool inb = row < M;                                                   // row exists in memory
  const bool valid = row < (valid_rows ? (int64_t)*valid_rows : M);           // row carries weight (not padding)
  for (int i = t; i < K * AP; i += 256) {
    const int k = i >> 3, a = i & 7;
    Ws[i] = a < A ? W[k * A + a] : 0.f;
  }
  hl_f4 h[KQ / 4];
  {
    const hl_f4* hp = reinterpret_cast<const hl_f4*>(H + row * K + q * KQ);
#pragma unroll
    for (int j = 0; j < KQ / 4; ++j) h[j] = inb ? hp[j] : hl_f4{0.f, 0.f, 0.f, 0.f};
  }
  float bias[AP], ls[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) {
    bias[a] = a < A ? b[a] : 0.f;
    ls[a] = (POLICY && a < A) ? logstd[a] : 0.f;
  }
  __syncthreads();
  float out[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) out[a] = 0.f;
#pragma unroll
  for (int j = 0; j < KQ / 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* wr = Ws + (q * KQ + 4 * j + e) * AP;
      const hl_f4 w0 = *reinterpret_cast<const hl_f4*>(wr), w1 = *reinterpret_cast<const hl_f4*>(wr + 4);
      const float hv = h[j][e];
#pragma unroll
      for (int a = 0; a < 4; ++a) { out[a] = fmaf(hv, w0[a], out[a]); out[4 + a] = fmaf(hv, w1[a], out[4 + a]); }
    }
#pragma unroll
  for (int a = 0; a < AP; ++a) {   // fold the four K-slices of the row (fixed order: (q0+q1)+(q2+q3) in every lane)
    out[a] += dpp_f(out[a], 0);
    out[a] += dpp_f(out[a], 1);
    out[a] += bias[a];
  }
  // ---- loss and seeds, per row (the four threads of a row compute the same values)
  float d[AP], m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int a = 0; a < AP; ++a) d[a] = 0.f;
  if (POLICY) {
    float nlp = 0.f, zs[AP], isd[AP];
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      isd[a] = 1.0f / expf(ls[a]);
      const float act_a = (valid && a < A) ? mb_a[row * A + a] : out[a];
      zs[a] = (act_a - out[a]) * isd[a];
      if (a < A) nlp += -0.5f * zs[a] * zs[a] - 0.5f * LOG_2PI - ls[a];
    }
    float amean, ainv, astd;
    adv_norm_from_stats(stats, amean, ainv, astd);
    const float logp_old = valid ? aux[row * 3 + 0] : 0.f;
    const float advn = valid ? (aux[row * 3 + 2] - amean) * ainv : 0.f;
    const float logratio = valid ? nlp - logp_old : 0.f;
    const float ratio = expf(logratio);
    const float pg1 = -advn * ratio;
    const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    const float pg2 = -advn * rc;
    const bool inside = (ratio >= 1.f - clip) && (ratio <= 1.f + clip);
    const float d_ratio = (inside || pg1 > pg2) ? -advn : 0.f;
    const float d_logp = valid ? d_ratio * ratio * inv_mb : 0.f;
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      d[a] = a < A ? d_logp * zs[a] * isd[a] : 0.f;
      // second term: d(-entropy_coef * entropy) / d logstd_a = -entropy_coef, carried by every weighted row with 1 / mb so
      // that the sum over a (possibly sharded, padded) global minibatch is exactly one such term
      if (q == 0) DLs[r * AP + a] = a < A ? d_logp * (zs[a] * zs[a] - 1.f) - (valid ? ent_coef * inv_mb : 0.f) : 0.f;
    }
    if (valid && q == 0) {
      m0 = fmaxf(pg1, pg2);
      m1 = (ratio - 1.f) - logratio;
      m2 = fabsf(ratio - 1.f) > clip ? 1.f : 0.f;
    }
    if (blockIdx.x == 0 && t == 0) {
      float ent = 0.f, sstd = 0.f;
      for (int a = 0; a < A; ++a) { ent += ls[a] + HALF_LOG_2PIE; sstd += expf(ls[a]); }
      metrics[2] = ent;
      metrics[5] = amean;
      metrics[6] = astd;
      metrics[7] = sstd / (float)A;  // mean policy std (metric policy/std_dev, ppo.py:230) before this update
    }
  } else {
    if (valid) {
      const float diff = out[0] - aux[row * 3 + 1];
      if (q == 0) m0 = 0.5f * diff * diff;
      d[0] = critic_coef * inv_mb * diff;
    }
  }
  if (q == 0) {
#pragma unroll
    for (int a = 0; a < AP; ++a) Ds[r * AP + a] = d[a];
  }
  m0 = wave_sum(m0);
  m1 = wave_sum(m1);
  m2 = wave_sum(m2);
  if ((t & 63) == 0) { red[(t >> 6) * 4 + 0] = m0; red[(t >> 6) * 4 + 1] = m1; red[(t >> 6) * 4 + 2] = m2; }
  // ---- park h for the weight gradient, then dZ_last in place from registers
#pragma unroll
  for (int j = 0; j < KQ / 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) Hs[r * HS + q * KQ + 4 * j + e] = h[j][e];
  if (inb) {                                                                  // padding rows: d == 0 -> zeros are stored
    hl_f4* hp = reinterpret_cast<hl_f4*>(H + row * K + q * KQ);
#pragma unroll
    for (int j = 0; j < KQ / 4; ++j) {
      hl_f4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* wr = Ws + (q * KQ + 4 * j + e) * AP;
        const hl_f4 w0 = *reinterpret_cast<const hl_f4*>(wr), w1 = *reinterpret_cast<const hl_f4*>(wr + 4);
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) { sacc = fmaf(d[a], w0[a], sacc); sacc = fmaf(d[4 + a], w1[a], sacc); }
        o[e] = sacc * act_grad_from_out(h[j][e], act);
      }
      hp[j] = o;
    }
  }
  __syncthreads();
  float* pw = partials + (int64_t)blockIdx.x * PS;
  if (t == 0) {
    float* pm = pw + K * A + 2 * A;
    pm[0] = (red[0] + red[4]) + (red[8] + red[12]);
    pm[1] = (red[1] + red[5]) + (red[9] + red[13]);
    pm[2] = (red[2] + red[6]) + (red[10] + red[14]);
  }
  __syncthreads();
  // ---- head weight gradient: thread (k, row group): RP rows in order, then the NP groups in order
  {
    const int k = t % K, part = t / K;
    float dw[AP];
#pragma unroll
    for (int a = 0; a < AP; ++a) dw[a] = 0.f;
    if (part < NP) {
      for (int rr = part * RP; rr < (part + 1) * RP; ++rr) {
        const float hv = Hs[rr * HS + k];
        const hl_f4 d0 = *reinterpret_cast<const hl_f4*>(Ds + rr * AP), d1 = *reinterpret_cast<const hl_f4*>(Ds + rr * AP + 4);
#pragma unroll
        for (int a = 0; a < 4; ++a) { dw[a] = fmaf(hv, d0[a], dw[a]); dw[4 + a] = fmaf(hv, d1[a], dw[4 + a]); }
      }
      if (NP > 1) {
#pragma unroll
        for (int a = 0; a < AP; ++a) red[(part * K + k) * AP + a] = dw[a];
      }
    }
    if (NP > 1) __syncthreads();
    if (part == 0) {
      if (NP > 1) {
        for (int pp = 1; pp < NP; ++pp)
#pragma unroll
          for (int a = 0; a < AP; ++a) dw[a] += red[(pp * K + k) * AP + a];
      }
      for (int a = 0; a < A; ++a) pw[k * A + a] = dw[a];
    }
  }
  if (t < A) {
    float sb = 0.f, sl = 0.f;
    for (int rr = 0; rr < HEAD_ROWS; ++rr) {
      sb += Ds[rr * AP + t];
      if (POLICY) sl += DLs[rr * AP + t];
    }
    pw[K * A + t] = sb;
    pw[K * A + A + t] = sl;
  }
}

template <bool POLICY, int KQ>
__global__ __launch_bounds__(256) void k_head_loss_fast(float* __restrict__ H, const float* __restrict__ W,
                                                        const float* __restrict__ b, const float* __restrict__ logstd,
                                                        const float* __restrict__ mb_a, const float* __restrict__ aux,
                                                        const double* __restrict__ stats, float* __restrict__ partials,
                                                        float* __restrict__ metrics, int64_t M, int A, int PS, float inv_mb,
                                                        float clip, float ent_coef, float critic_coef, int act,
                                                        const int32_t* __restrict__ valid_rows) {
  head_loss_fast_body<POLICY, KQ>(H, W, b, logstd, mb_a, aux, stats, partials, metrics, M, A, PS, inv_mb, clip, ent_coef,
                                  critic_coef, act, valid_rows);
}

// Twin launch (grid.y == 2): blockIdx.y == 0 is the policy's head + PPO surrogate loss, blockIdx.y == 1 the critic's head + value
// loss, on the same gathered rows -- the two networks of ppo.py:196-210 have the same last hidden width here, so the two bodies
// share one launch geometry (64 rows per workgroup, 4 threads per row).  Same code, same results as the two single launches.
struct HeadNet {
  float* H;             // [M, K] last hidden activation -> dZ_last in place
  const float* W;       // head weights [K, A]
  const float* b;
  const float* logstd;  // policy only
  float* partials;      // [blocks, PS]
  int A, PS;
};
template <int KQ>
__global__ __launch_bounds__(256) void k_head_loss_pc(HeadNet p, HeadNet c, const float* __restrict__ mb_a,
                                                      const float* __restrict__ aux, const double* __restrict__ stats,
                                                      float* __restrict__ metrics, int64_t M, float inv_mb, float clip,
                                                      float ent_coef, float critic_coef, int act,
                                                      const int32_t* __restrict__ valid_rows) {
  if (blockIdx.y == 0)
    head_loss_fast_body<true, KQ>(p.H, p.W, p.b, p.logstd, mb_a, aux, stats, p.partials, metrics, M, p.A, p.PS, inv_mb, clip,
                                  ent_coef, critic_coef, act, valid_rows);
  else
    head_loss_fast_body<false, KQ>(c.H, c.W, c.b, nullptr, mb_a, aux, stats, c.partials, metrics, M, c.A, c.PS, inv_mb, clip,
                                   ent_coef, critic_coef, act, valid_rows);
}

static inline size_t head_fast_lds_bytes(int K) {
  const int NP = 256 / K > 0 ? 256 / K : 1;
  return ((size_t)K * 8 + 2 * HEAD_ROWS * 8 + (size_t)HEAD_ROWS * (K + 1) + (size_t)NP * K * 8 + 16) * sizeof(float);
}

static int launch_head_loss_pc(const HeadNet& p, const HeadNet& c, const MbScratch& s, float* metrics, int64_t mb, int K,
                               float inv_mb, const rlx_ppo_hparams& hp, int act, hipStream_t st) {
  const int nb = div_up(mb, HEAD_ROWS);
  const size_t lds = head_fast_lds_bytes(K);
#define RLX_HL_PC(KQV)                                                                                            \
  {                                                                                                               \
    static AttrOnce attr_set;                                                                                       \
    if (!attr_set.done()) {                                                                                              \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_loss_pc<KQV>),                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                   \
      attr_set.mark();                                                                                              \
    }                                                                                                             \
    hipLaunchKernelGGL((k_head_loss_pc<KQV>), dim3(nb, 2), dim3(256), lds, st, p, c, s.mb_a, s.aux, s.stats, metrics, mb,      \
                       inv_mb, hp.clip_range, hp.entropy_coef, hp.critic_coef, act, s.valid_rows);                \
  }
  if (K == 64) RLX_HL_PC(16)
  else if (K == 128) RLX_HL_PC(32)
  else if (K == 256) RLX_HL_PC(64)
  else RLX_REQUIRE(false, RLX_EUNSUP, "ppo twin head: last hidden width must be 64, 128 or 256");
#undef RLX_HL_PC
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// ---------------------------------------------------------------------------------------
// Row-tile-local TAIL of a network's minibatch pass: for a tile of 64 rows, in ONE launch
//     H3  = act(H2 @ W3 + b3)                         (forward of the last hidden layer)
//     out = H3 @ Wh + bh -> PPO loss, its seeds, the head's gradient partials            (k_head_loss_fast's arithmetic)
//     dZ3 = (d_out @ Wh^T) * act'(H3)                 -> HBM (the weight-gradient kernel of layer 3 reads it)
//     dZ2 = (dZ3 @ W3^T) * act'(H2)                   -> HBM (weight gradient of layer 2, fused first-layer backward)
// H3 never exists in HBM and dZ3 is not read back for the input gradient: per network and update the tail reads H2 once and
// writes dZ3 and dZ2 once (+ one L2-resident re-read of H2 for act') = 84 MB at 32768 rows where the three launches it
// replaces -- k_gemm_bx<0> (layer 3), k_head_loss_fast, k_gemm_bx<1> (layer 3) -- moved 168 MB, and one dependent launch stands
// where three stood (the 4096-row regime is bound by the chain of dependent launches).
//
// Built from the pieces it replaces: bx_kloop (gemm_bx.h) for both products -- A operand staged as fp16 planes through LDS,
// weight fragments from the forward / transposed split image of W3 in L2 --, the head arithmetic of head_loss_fast_body in the
// same thread geometry (64 rows x 4 threads).  The 64 x 128 tile of H3 (then dZ3) lives in LDS as fp32 between the phases; the
// head phase's scratch (Wh, seeds, reduction slots) aliases the plane stages, idle at that point: 65 KiB of LDS, 256 threads --
// two workgroups per CU.  Last hidden width 128; M a multiple of 64 (the callers fall back to the three launches otherwise).
// ---------------------------------------------------------------------------------------
constexpr int TL_TS = 132;            // row stride (floats) of the LDS tile: 16-byte aligned rows, 4-bank skew
constexpr int TL_K3 = 128;            // last hidden width
struct TailNet {
  const float* H2;      // [M, N2]
  const u32x4* W3f;     // forward split image of W3 (K = N2, N = 128)
  const u32x4* W3t;     // transposed split image  (K = 128, N = N2)
  const float* b3;
  const float* Wh;      // head [128, A]
  const float* bh;
  const float* logstd;  // policy only
  float* dZ3;           // [M, 128]
  float* dZ2;           // [M, N2]
  float* partials;      // head partials [M / 64][PS]
  int A, PS;
};

template <bool POLICY, int ACT>
__device__ __forceinline__ void tail_body(const TailNet& n, char* __restrict__ smem, const float* __restrict__ mb_a,
                                          const float* __restrict__ aux, const double* __restrict__ stats,
                                          float* __restrict__ metrics, int64_t M, int N2, float inv_mb, float clip, float ent_coef,
                                          float critic_coef, const int32_t* __restrict__ valid_rows, float gs) {
  constexpr int K = TL_K3, KQ = K / 4, AP = 8, NP = 2, RP = HEAD_ROWS / NP;
  char* planes = smem;                                            // 2 x X_OPER
  float* T = reinterpret_cast<float*>(smem + 2 * X_OPER);          // [64][TL_TS]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
  const int a_r = t >> 3, a_c = (t & 7) * 4;
  const int64_t m0 = (int64_t)blockIdx.x * HEAD_ROWS;
  const int A = n.A, PS = n.PS;
  // ------------------------------------------------------------------ phase A: H3 tile = act(H2 @ W3 + b3) -> T
  {
    f32x16 acc[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    const int nk = N2 / X_BK;
    const float* ap = n.H2 + (m0 + a_r) * N2 + a_c;
    auto load = [&](int kt, float4 (&rr)[2]) {
      const int kk = (kt < nk ? kt : nk - 1) * X_BK;
#pragma unroll
      for (int p = 0; p < 2; ++p) rr[p] = *reinterpret_cast<const float4*>(ap + (int64_t)(32 * p) * N2 + kk);
    };
    bx_kloop<1>(planes, n.W3f, nk, 4, wn * 2, wm, lane, a_r, a_c, load, acc, X_ASCALE);
    const float so = X_AINV * X_WINV;
    float* tb = T + (wm * 32 + 4 * (lane >> 5)) * TL_TS + wn * 64 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float bv = n.b3[wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
      for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2)) * TL_TS + j * 32] = act_fwd_t<ACT>(fmaf(acc[0][j][r], so, bv));
    }
  }
  // ------------------------------------------------------------------ phase B: head, loss, seeds, head-gradient partials
  float* Ws = reinterpret_cast<float*>(planes);    // [K][8]      (the plane stages are idle until phase C)
  float* Ds = Ws + K * AP;                         // [64][8]  d_out rows
  float* DLs = Ds + HEAD_ROWS * AP;                // [64][8]  d logstd terms
  float* red = DLs + HEAD_ROWS * AP;               // [16] metric sums, then [NP][K][8] dW partial sums
  const int r = t >> 2, q = t & 3;
  const int64_t row = m0 + r;
  const bool valid = row < (valid_rows ? (int64_t)*valid_rows : M);
  for (int i = t; i < K * AP; i += 256) {
    const int k = i >> 3, a = i & 7;
    Ws[i] = a < A ? n.Wh[k * A + a] : 0.f;
  }
  float bias[AP], ls[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) {
    bias[a] = a < A ? n.bh[a] : 0.f;
    ls[a] = (POLICY && a < A) ? n.logstd[a] : 0.f;
  }
  __syncthreads();                                 // T (phase A's stores) and Ws are visible
  hl_f4 h[KQ / 4];
  {
    const hl_f4* hp = reinterpret_cast<const hl_f4*>(T + r * TL_TS + q * KQ);
#pragma unroll
    for (int j = 0; j < KQ / 4; ++j) h[j] = hp[j];
  }
  float out[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) out[a] = 0.f;
#pragma unroll
  for (int j = 0; j < KQ / 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* wr = Ws + (q * KQ + 4 * j + e) * AP;
      const hl_f4 w0 = *reinterpret_cast<const hl_f4*>(wr), w1 = *reinterpret_cast<const hl_f4*>(wr + 4);
      const float hv = h[j][e];
#pragma unroll
      for (int a = 0; a < 4; ++a) { out[a] = fmaf(hv, w0[a], out[a]); out[4 + a] = fmaf(hv, w1[a], out[4 + a]); }
    }
#pragma unroll
  for (int a = 0; a < AP; ++a) {   // fold the four K-slices of the row (fixed order: (q0+q1)+(q2+q3) in every lane)
    out[a] += dpp_f(out[a], 0);
    out[a] += dpp_f(out[a], 1);
    out[a] += bias[a];
  }
  float d[AP], m0s = 0.f, m1s = 0.f, m2s = 0.f;
#pragma unroll
  for (int a = 0; a < AP; ++a) d[a] = 0.f;
  if (POLICY) {
    float nlp = 0.f, zs[AP], isd[AP];
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      isd[a] = 1.0f / expf(ls[a]);
      const float act_a = (valid && a < A) ? mb_a[row * A + a] : out[a];
      zs[a] = (act_a - out[a]) * isd[a];
      if (a < A) nlp += -0.5f * zs[a] * zs[a] - 0.5f * LOG_2PI - ls[a];
    }
    float amean, ainv, astd;
    adv_norm_from_stats(stats, amean, ainv, astd);
    const float logp_old = valid ? aux[row * 3 + 0] : 0.f;
    const float advn = valid ? (aux[row * 3 + 2] - amean) * ainv : 0.f;
    const float logratio = valid ? nlp - logp_old : 0.f;
    const float ratio = expf(logratio);
    const float pg1 = -advn * ratio;
    const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    const float pg2 = -advn * rc;
    const bool inside = (ratio >= 1.f - clip) && (ratio <= 1.f + clip);
    const float d_ratio = (inside || pg1 > pg2) ? -advn : 0.f;
    const float d_logp = valid ? d_ratio * ratio * inv_mb : 0.f;
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      d[a] = a < A ? d_logp * zs[a] * isd[a] : 0.f;
      if (q == 0) DLs[r * AP + a] = a < A ? d_logp * (zs[a] * zs[a] - 1.f) - (valid ? ent_coef * inv_mb : 0.f) : 0.f;
    }
    if (valid && q == 0) {
      m0s = fmaxf(pg1, pg2);
      m1s = (ratio - 1.f) - logratio;
      m2s = fabsf(ratio - 1.f) > clip ? 1.f : 0.f;
    }
    if (blockIdx.x == 0 && t == 0) {
      float ent = 0.f, sstd = 0.f;
      for (int a = 0; a < A; ++a) { ent += ls[a] + HALF_LOG_2PIE; sstd += expf(ls[a]); }
      metrics[2] = ent;
      metrics[5] = amean;
      metrics[6] = astd;
      metrics[7] = sstd / (float)A;
    }
  } else {
    if (valid) {
      const float diff = out[0] - aux[row * 3 + 1];
      if (q == 0) m0s = 0.5f * diff * diff;
      d[0] = critic_coef * inv_mb * diff;
    }
  }
  if (q == 0) {
#pragma unroll
    for (int a = 0; a < AP; ++a) Ds[r * AP + a] = d[a];
  }
  m0s = wave_sum(m0s);
  m1s = wave_sum(m1s);
  m2s = wave_sum(m2s);
  if ((t & 63) == 0) { red[(t >> 6) * 4 + 0] = m0s; red[(t >> 6) * 4 + 1] = m1s; red[(t >> 6) * 4 + 2] = m2s; }
  __syncthreads();
  float* pw = n.partials + (int64_t)blockIdx.x * PS;
  if (t == 0) {
    float* pm = pw + K * A + 2 * A;
    pm[0] = (red[0] + red[4]) + (red[8] + red[12]);
    pm[1] = (red[1] + red[5]) + (red[9] + red[13]);
    pm[2] = (red[2] + red[6]) + (red[10] + red[14]);
  }
  __syncthreads();
  {  // head weight gradient: thread (k, row group): RP rows in order, then the NP groups in order (H3 rows from the tile)
    const int k = t % K, part = t / K;
    float dw[AP];
#pragma unroll
    for (int a = 0; a < AP; ++a) dw[a] = 0.f;
    for (int rr = part * RP; rr < (part + 1) * RP; ++rr) {
      const float hv = T[rr * TL_TS + k];
      const hl_f4 d0 = *reinterpret_cast<const hl_f4*>(Ds + rr * AP), d1 = *reinterpret_cast<const hl_f4*>(Ds + rr * AP + 4);
#pragma unroll
      for (int a = 0; a < 4; ++a) { dw[a] = fmaf(hv, d0[a], dw[a]); dw[4 + a] = fmaf(hv, d1[a], dw[4 + a]); }
    }
#pragma unroll
    for (int a = 0; a < AP; ++a) red[(part * K + k) * AP + a] = dw[a];
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int a = 0; a < AP; ++a) dw[a] += red[(K + k) * AP + a];
      for (int a = 0; a < A; ++a) pw[k * A + a] = dw[a];
    }
  }
  if (t < A) {
    float sb = 0.f, sl = 0.f;
    for (int rr = 0; rr < HEAD_ROWS; ++rr) {
      sb += Ds[rr * AP + t];
      if (POLICY) sl += DLs[rr * AP + t];
    }
    pw[K * A + t] = sb;
    pw[K * A + A + t] = sl;
  }
  __syncthreads();                                 // every read of T as H3 is done
  {  // dZ3 = (d_out @ Wh^T) * act'(H3): into the tile (phase C's A operand) and to HBM (the weight-gradient kernel's operand)
    hl_f4* tp = reinterpret_cast<hl_f4*>(T + r * TL_TS + q * KQ);
    hl_f4* gp = reinterpret_cast<hl_f4*>(n.dZ3 + row * K + q * KQ);
#pragma unroll
    for (int j = 0; j < KQ / 4; ++j) {
      hl_f4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* wr = Ws + (q * KQ + 4 * j + e) * AP;
        const hl_f4 w0 = *reinterpret_cast<const hl_f4*>(wr), w1 = *reinterpret_cast<const hl_f4*>(wr + 4);
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) { sacc = fmaf(d[a], w0[a], sacc); sacc = fmaf(d[4 + a], w1[a], sacc); }
        o[e] = sacc * act_grad_t<ACT>(h[j][e]);
      }
      tp[j] = o;
      gp[j] = o;
    }
  }
  __syncthreads();                                 // T = dZ3; the plane stages are free again (Ws / Ds / red are dead)
  // ------------------------------------------------------------------ phase C: dZ2 = (dZ3 @ W3^T) * act'(H2), 128 columns at a time
  {
    constexpr int nk3 = K / X_BK;
    auto loadT = [&](int kt, float4 (&rr)[2]) {
      const int kk = (kt < nk3 ? kt : nk3 - 1) * X_BK;
#pragma unroll
      for (int p = 0; p < 2; ++p) rr[p] = *reinterpret_cast<const float4*>(T + (a_r + 32 * p) * TL_TS + kk + a_c);
    };
    const float so2 = X_WINV / gs;
    const int ntc = N2 / G_BN;
    for (int nt = 0; nt < ntc; ++nt) {
      f32x16 acc[1][2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) acc[0][j][rr] = 0.f;
      bx_kloop<1>(planes, n.W3t, nk3, ntc * 4, nt * 4 + wn * 2, wm, lane, a_r, a_c, loadT, acc, gs);
      const int64_t off = (m0 + wm * 32 + 4 * (lane >> 5)) * N2 + nt * G_BN + wn * 64 + (lane & 31);
      const float* hsb = n.H2 + off;
      float* cb = n.dZ2 + off;
      float hh[2][16];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) hh[j][rr] = hsb[(int64_t)((rr & 3) + 8 * (rr >> 2)) * N2 + j * 32];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
          cb[(int64_t)((rr & 3) + 8 * (rr >> 2)) * N2 + j * 32] = acc[0][j][rr] * so2 * act_grad_t<ACT>(hh[j][rr]);
    }
  }
}

// both == 1: grid.y == 2 -- blockIdx.y == 0 the policy (p), 1 the critic (c); both == 0: one network, `which` (0 policy, 1 critic)
template <int ACT>
__global__ __launch_bounds__(256, 2) void k_tail_bx(TailNet p, TailNet c, const float* __restrict__ mb_a,
                                                    const float* __restrict__ aux, const double* __restrict__ stats,
                                                    float* __restrict__ metrics, int64_t M, int N2, float inv_mb, float clip,
                                                    float ent_coef, float critic_coef, const int32_t* __restrict__ valid_rows,
                                                    float gs, int both, int which) {
  extern __shared__ __attribute__((aligned(16))) char tl_smem[];
  const bool critic = both ? blockIdx.y != 0 : which != 0;
  if (!critic)
    tail_body<true, ACT>(p, tl_smem, mb_a, aux, stats, metrics, M, N2, inv_mb, clip, ent_coef, critic_coef, valid_rows, gs);
  else
    tail_body<false, ACT>(c, tl_smem, mb_a, aux, stats, metrics, M, N2, inv_mb, clip, ent_coef, critic_coef, valid_rows, gs);
}

// ---------------------------------------------------------------------------------------
#ifndef RLX_T32_PFC
#define RLX_T32_PFC 2
#endif
// Tail, second form (option ppo_tail = 2): 32-row tiles with the WHOLE H2 tile resident in LDS as fp16 planes.  All of the tile's
// H2 rows are requested up front (32 KB in flight per workgroup, one exposed memory latency instead of eight), both products run
// barrier-free K loops (A fragments straight from the resident planes / the dZ3 planes the head phase leaves, weight fragments from
// the images in L2, as k_l12fwd and k_dx_l1bwd do), and the act'(H2) factor of the dZ2 epilogue is rebuilt from the resident planes
// (hi + lo, 2^-22 relative) instead of a second HBM read of H2: 34 + 17 + 34 MB per network and update at 32768 rows where the
// first form moves 122.  Padded LDS rows (stride + 16 bytes) keep every fragment / store address a per-lane base + an immediate.
// 256 threads: wave w owns output columns [32 w, 32 w + 32) of the forward and column tiles 2 w, 2 w + 1 of the input gradient;
// the head runs 8 threads per row.  65 KB of LDS: two workgroups per CU.
// ---------------------------------------------------------------------------------------
constexpr int T32_ROWS = 32;
template <bool POLICY, int ACT, int N2>
__device__ __forceinline__ void tail32_body(const TailNet& n, char* __restrict__ smem, const float* __restrict__ mb_a,
                                            const float* __restrict__ aux, const double* __restrict__ stats,
                                            float* __restrict__ metrics, int64_t M, float inv_mb, float clip, float ent_coef,
                                            float critic_coef, const int32_t* __restrict__ valid_rows, float gs) {
  constexpr int K = TL_K3, AP = 8, NP = 2, RP = T32_ROWS / NP, KQ = K / 8;     // head: 8 threads per row, 16 columns each
  constexpr int HROW = 2 * N2 + 16, HPL = T32_ROWS * HROW;                     // H2 planes: padded rows
  constexpr int DROW = 2 * K + 16, DPL = T32_ROWS * DROW;                      // dZ3 planes
  constexpr int TREG = (T32_ROWS * TL_TS * 4 > 2 * DPL) ? T32_ROWS * TL_TS * 4 : 2 * DPL;   // fp32 H3 tile, then the dZ3 planes
  char* H2P = smem;
  float* T = reinterpret_cast<float*>(smem + 2 * HPL);
  char* D3P = smem + 2 * HPL;
  float* Ws = reinterpret_cast<float*>(smem + 2 * HPL + TREG);    // [K][8]
  float* Ds = Ws + K * AP;                                        // [32][8]
  float* DLs = Ds + T32_ROWS * AP;                                // [32][8]
  float* red = DLs + T32_ROWS * AP;                               // [16] metric sums, then [NP][K][8]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * T32_ROWS;
  const int A = n.A, PS = n.PS;
  // ------------------------------------------------------------------ the H2 tile -> resident planes
  {
    constexpr int NV = T32_ROWS * N2 / 4 / 256;                   // float4 pieces per thread (8 at N2 = 256)
    float4 hv[NV];
    const int sr = t >> 3, sc = (t & 7) * 4;                      // row, first column of the thread's pieces (then + 32 i)
    const float* hp = n.H2 + (m0 + sr) * N2 + sc;
#pragma unroll
    for (int i = 0; i < NV; ++i) hv[i] = *reinterpret_cast<const float4*>(hp + 32 * i);
    for (int i = t; i < K * AP; i += 256) {                       // (head weights, under the loads' latency)
      const int k = i >> 3, a = i & 7;
      Ws[i] = a < A ? n.Wh[k * A + a] : 0.f;
    }
    char* d0 = H2P + sr * HROW + sc * 2;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      uint32_t a0, a1, b0, b1;
      bx_split2(hv[i].x * X_ASCALE, hv[i].y * X_ASCALE, a0, a1);
      bx_split2(hv[i].z * X_ASCALE, hv[i].w * X_ASCALE, b0, b1);
      *reinterpret_cast<u32x2*>(d0 + 64 * i) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(d0 + 64 * i + HPL) = u32x2{a1, b1};
    }
  }
  __syncthreads();
  // ------------------------------------------------------------------ phase A: H3 = act(H2 @ W3 + b3), wave w: columns [32 w, 32 w + 32)
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NB = N2 / 16, PF = 4;
    const char* ard = H2P + li * HROW + lh * 16;
    const u32x4* wf = n.W3f + (int64_t)w * X_NP * 64 + lane;      // image: [kb][4 column tiles][2 planes][64]
    constexpr int wstep = 4 * X_NP * 64;
    u32x4 bq[PF][X_NP];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int p = 0; p < X_NP; ++p) bq[u][p] = wf[(int64_t)u * wstep + p * 64];
    // (unconditional refills inside the loop, the last group peeled: with the refill behind `if (q + u + PF < NB)` hipcc opens every
    //  block with s_waitcnt vmcnt(0) and the fragments requested ahead are drained each time -- fwd2h.hip has the numbers)
#define T32_BLOCK(QU, REFILL)                                                                                         \
  {                                                                                                                   \
    u32x4 av[X_NP];                                                                                                   \
    _Pragma("unroll") for (int p = 0; p < X_NP; ++p) av[p] = *reinterpret_cast<const u32x4*>(ard + (QU) * 32 + p * HPL); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bq[u][1]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[1]), __builtin_bit_cast(f16x8, bq[u][0]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bq[u][0]), acc, 0, 0, 0); \
    if (REFILL) {                                                                                                     \
      _Pragma("unroll") for (int p = 0; p < X_NP; ++p) bq[u][p] = wf[(int64_t)((QU) + PF) * wstep + p * 64];           \
    }                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  }
#pragma unroll 1
    for (int q = 0; q < NB - PF; q += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) T32_BLOCK(q + u, true)
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) T32_BLOCK(NB - PF + u, false)
#undef T32_BLOCK
    const float so = X_AINV * X_WINV, bv = n.b3[w * 32 + li];
    float* tb = T + (4 * lh) * TL_TS + w * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2)) * TL_TS] = act_fwd_t<ACT>(fmaf(acc[r], so, bv));
  }
  __syncthreads();
  // ------------------------------------------------------------------ phase B: head, loss, seeds, head-gradient partials (8 threads per row)
  const int r = t >> 3, q8 = t & 7;
  const int64_t row = m0 + r;
  const bool valid = row < (valid_rows ? (int64_t)*valid_rows : M);
  float bias[AP], ls[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) {
    bias[a] = a < A ? n.bh[a] : 0.f;
    ls[a] = (POLICY && a < A) ? n.logstd[a] : 0.f;
  }
  hl_f4 h[KQ / 4];
  {
    const hl_f4* hp = reinterpret_cast<const hl_f4*>(T + r * TL_TS + q8 * KQ);
#pragma unroll
    for (int j = 0; j < KQ / 4; ++j) h[j] = hp[j];
  }
  float out[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) out[a] = 0.f;
#pragma unroll
  for (int j = 0; j < KQ / 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* wr = Ws + (q8 * KQ + 4 * j + e) * AP;
      const hl_f4 w0 = *reinterpret_cast<const hl_f4*>(wr), w1 = *reinterpret_cast<const hl_f4*>(wr + 4);
      const float hv = h[j][e];
#pragma unroll
      for (int a = 0; a < 4; ++a) { out[a] = fmaf(hv, w0[a], out[a]); out[4 + a] = fmaf(hv, w1[a], out[4 + a]); }
    }
#pragma unroll
  for (int a = 0; a < AP; ++a) {   // fold the eight K-slices of the row: xor 1, xor 2, then the two quads (fixed order in every lane)
    out[a] += dpp_f(out[a], 0);
    out[a] += dpp_f(out[a], 1);
    out[a] += dpp_f(out[a], 2);
    out[a] += bias[a];
  }
  float d[AP], m0s = 0.f, m1s = 0.f, m2s = 0.f;
#pragma unroll
  for (int a = 0; a < AP; ++a) d[a] = 0.f;
  if (POLICY) {
    float nlp = 0.f, zs[AP], isd[AP];
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      isd[a] = 1.0f / expf(ls[a]);
      const float act_a = (valid && a < A) ? mb_a[row * A + a] : out[a];
      zs[a] = (act_a - out[a]) * isd[a];
      if (a < A) nlp += -0.5f * zs[a] * zs[a] - 0.5f * LOG_2PI - ls[a];
    }
    float amean, ainv, astd;
    adv_norm_from_stats(stats, amean, ainv, astd);
    const float logp_old = valid ? aux[row * 3 + 0] : 0.f;
    const float advn = valid ? (aux[row * 3 + 2] - amean) * ainv : 0.f;
    const float logratio = valid ? nlp - logp_old : 0.f;
    const float ratio = expf(logratio);
    const float pg1 = -advn * ratio;
    const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    const float pg2 = -advn * rc;
    const bool inside = (ratio >= 1.f - clip) && (ratio <= 1.f + clip);
    const float d_ratio = (inside || pg1 > pg2) ? -advn : 0.f;
    const float d_logp = valid ? d_ratio * ratio * inv_mb : 0.f;
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      d[a] = a < A ? d_logp * zs[a] * isd[a] : 0.f;
      if (q8 == 0) DLs[r * AP + a] = a < A ? d_logp * (zs[a] * zs[a] - 1.f) - (valid ? ent_coef * inv_mb : 0.f) : 0.f;
    }
    if (valid && q8 == 0) {
      m0s = fmaxf(pg1, pg2);
      m1s = (ratio - 1.f) - logratio;
      m2s = fabsf(ratio - 1.f) > clip ? 1.f : 0.f;
    }
    if (blockIdx.x == 0 && t == 0) {
      float ent = 0.f, sstd = 0.f;
      for (int a = 0; a < A; ++a) { ent += ls[a] + HALF_LOG_2PIE; sstd += expf(ls[a]); }
      metrics[2] = ent;
      metrics[5] = amean;
      metrics[6] = astd;
      metrics[7] = sstd / (float)A;
    }
  } else {
    if (valid) {
      const float diff = out[0] - aux[row * 3 + 1];
      if (q8 == 0) m0s = 0.5f * diff * diff;
      d[0] = critic_coef * inv_mb * diff;
    }
  }
  if (q8 == 0) {
#pragma unroll
    for (int a = 0; a < AP; ++a) Ds[r * AP + a] = d[a];
  }
  m0s = wave_sum(m0s);
  m1s = wave_sum(m1s);
  m2s = wave_sum(m2s);
  if ((t & 63) == 0) { red[(t >> 6) * 4 + 0] = m0s; red[(t >> 6) * 4 + 1] = m1s; red[(t >> 6) * 4 + 2] = m2s; }
  __syncthreads();
  float* pw = n.partials + (int64_t)blockIdx.x * PS;
  if (t == 0) {
    float* pm = pw + K * A + 2 * A;
    pm[0] = (red[0] + red[4]) + (red[8] + red[12]);
    pm[1] = (red[1] + red[5]) + (red[9] + red[13]);
    pm[2] = (red[2] + red[6]) + (red[10] + red[14]);
  }
  __syncthreads();
  {  // head weight gradient: thread (k, row group): RP rows in order, then the NP groups in order
    const int k = t % K, part = t / K;
    float dw[AP];
#pragma unroll
    for (int a = 0; a < AP; ++a) dw[a] = 0.f;
    for (int rr = part * RP; rr < (part + 1) * RP; ++rr) {
      const float hv = T[rr * TL_TS + k];
      const hl_f4 d0 = *reinterpret_cast<const hl_f4*>(Ds + rr * AP), d1 = *reinterpret_cast<const hl_f4*>(Ds + rr * AP + 4);
#pragma unroll
      for (int a = 0; a < 4; ++a) { dw[a] = fmaf(hv, d0[a], dw[a]); dw[4 + a] = fmaf(hv, d1[a], dw[4 + a]); }
    }
#pragma unroll
    for (int a = 0; a < AP; ++a) red[(part * K + k) * AP + a] = dw[a];
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int a = 0; a < AP; ++a) dw[a] += red[(K + k) * AP + a];
      for (int a = 0; a < A; ++a) pw[k * A + a] = dw[a];
    }
  }
  if (t < A) {
    float sb = 0.f, sl = 0.f;
    for (int rr = 0; rr < T32_ROWS; ++rr) {
      sb += Ds[rr * AP + t];
      if (POLICY) sl += DLs[rr * AP + t];
    }
    pw[K * A + t] = sb;
    pw[K * A + A + t] = sl;
  }
  __syncthreads();                                 // every read of T as H3 is done
  {  // dZ3 = (d_out @ Wh^T) * act'(H3): to HBM (fp32, the weight-gradient job's operand) and as fp16 planes into the tile
    hl_f4* gp = reinterpret_cast<hl_f4*>(n.dZ3 + row * K + q8 * KQ);
    char* dp = D3P + r * DROW + q8 * KQ * 2;
#pragma unroll
    for (int j = 0; j < KQ / 4; ++j) {
      hl_f4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* wr = Ws + (q8 * KQ + 4 * j + e) * AP;
        const hl_f4 w0 = *reinterpret_cast<const hl_f4*>(wr), w1 = *reinterpret_cast<const hl_f4*>(wr + 4);
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) { sacc = fmaf(d[a], w0[a], sacc); sacc = fmaf(d[4 + a], w1[a], sacc); }
        o[e] = sacc * act_grad_t<ACT>(h[j][e]);
      }
      gp[j] = o;
      uint32_t a0, a1, b0, b1;
      bx_split2(o[0] * gs, o[1] * gs, a0, a1);
      bx_split2(o[2] * gs, o[3] * gs, b0, b1);
      *reinterpret_cast<u32x2*>(dp + 8 * j) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(dp + 8 * j + DPL) = u32x2{a1, b1};
    }
  }
  __syncthreads();
  // ------------------------------------------------------------------ phase C: dZ2 = (dZ3 @ W3^T) * act'(H2); wave w: column tiles 2 w, 2 w + 1
  {
    constexpr int NTC = N2 / 32, JW = NTC / 4;                    // column tiles of the output, per wave
    f32x16 acc[JW];
#pragma unroll
    for (int j = 0; j < JW; ++j)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[j][rr] = 0.f;
    const char* ard = D3P + li * DROW + lh * 16;
    const u32x4* wt = n.W3t + (int64_t)(w * JW) * X_NP * 64 + lane;   // image: [kb][NTC column tiles][2 planes][64]
    constexpr int wstep = NTC * X_NP * 64;
    constexpr int PFC = RLX_T32_PFC, NBC = K / 16;                          // weight fragments four 16-k blocks ahead (L2 latency per block otherwise)
    u32x4 bc[PFC][JW][X_NP];
#pragma unroll
    for (int u = 0; u < PFC; ++u)
#pragma unroll
      for (int j = 0; j < JW; ++j)
#pragma unroll
        for (int p = 0; p < X_NP; ++p) bc[u][j][p] = wt[(int64_t)u * wstep + (j * X_NP + p) * 64];
#define T32C_BLOCK(QU, REFILL)                                                                                        \
  {                                                                                                                   \
    u32x4 av[X_NP];                                                                                                   \
    _Pragma("unroll") for (int p = 0; p < X_NP; ++p) av[p] = *reinterpret_cast<const u32x4*>(ard + (QU) * 32 + p * DPL); \
    _Pragma("unroll") for (int j = 0; j < JW; ++j) {                                                                  \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bc[u][j][1]), acc[j], 0, 0, 0); \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[1]), __builtin_bit_cast(f16x8, bc[u][j][0]), acc[j], 0, 0, 0); \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bc[u][j][0]), acc[j], 0, 0, 0); \
    }                                                                                                                 \
    if (REFILL) {                                                                                                     \
      _Pragma("unroll") for (int j = 0; j < JW; ++j) _Pragma("unroll") for (int p = 0; p < X_NP; ++p)                  \
          bc[u][j][p] = wt[(int64_t)((QU) + PFC) * wstep + (j * X_NP + p) * 64];                                       \
    }                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  }
    static_assert(NBC % PFC == 0 && NBC >= PFC, "phase C: whole groups of PFC blocks");
#pragma unroll 1
    for (int q = 0; q < NBC - PFC; q += PFC) {       // (unconditional refills, last group peeled: counted waits -- see phase A)
#pragma unroll
      for (int u = 0; u < PFC; ++u) T32C_BLOCK(q + u, true)
    }
#pragma unroll
    for (int u = 0; u < PFC; ++u) T32C_BLOCK(NBC - PFC + u, false)
#undef T32C_BLOCK
    const float so2 = X_WINV / gs;
#pragma unroll
    for (int j = 0; j < JW; ++j) {
      const int col = (w * JW + j) * 32 + li;
      float* cb = n.dZ2 + (m0 + 4 * lh) * N2 + col;
      const char* hb = H2P + (4 * lh) * HROW + col * 2;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int rho = (rr & 3) + 8 * (rr >> 2);
        const _Float16 hi = *reinterpret_cast<const _Float16*>(hb + rho * HROW);
        const _Float16 lo = *reinterpret_cast<const _Float16*>(hb + rho * HROW + HPL);
        const float h2 = ((float)hi + (float)lo) * X_AINV;        // the resident planes hold H2 x 16 as hi + lo
        cb[(int64_t)rho * N2] = acc[j][rr] * so2 * act_grad_t<ACT>(h2);
      }
    }
  }
}

template <int ACT, int N2>
__global__ __launch_bounds__(256, 2) void k_tail32_bx(TailNet p, TailNet c, const float* __restrict__ mb_a,
                                                      const float* __restrict__ aux, const double* __restrict__ stats,
                                                      float* __restrict__ metrics, int64_t M, float inv_mb, float clip,
                                                      float ent_coef, float critic_coef, const int32_t* __restrict__ valid_rows,
                                                      float gs, int both, int which) {
  extern __shared__ __attribute__((aligned(16))) char tl32_smem[];
  const bool critic = both ? blockIdx.y != 0 : which != 0;
  if (!critic)
    tail32_body<true, ACT, N2>(p, tl32_smem, mb_a, aux, stats, metrics, M, inv_mb, clip, ent_coef, critic_coef, valid_rows, gs);
  else
    tail32_body<false, ACT, N2>(c, tl32_smem, mb_a, aux, stats, metrics, M, inv_mb, clip, ent_coef, critic_coef, valid_rows, gs);
}

// rows per workgroup (= per block of head partials) of the tail form in use
// Which tail form a minibatch of mb rows takes.  ppo_tail = -1 (default): the 32-row form up to 8192 rows (4096 rows: 99.5 vs 107.0 us
// per update -- twice the workgroups when there are fewer 64-row tiles than CUs) and above 16384 rows, i.e. in the two-chain
// schedule (32768 rows: 85 instead of 122 MB of HBM traffic per launch beside the other chain's kernels; in-process A/B, ten blocks of
// five iterations each: 69.75 vs 70.06 ms per iteration -- with one gather per update it had been 71.80 vs 70.79); the 64-row form
// in between (16384 rows, twin launches: 228.7 vs 234.0 us per update); 1 / 2 force a form.
static inline int tail_rows(const rlx_ctx* ctx, int N2, int act, int64_t mb) {
  const bool can32 = N2 == 256 && act == RLX_ACT_ELU && mb % T32_ROWS == 0;
  const bool want32 = ctx->ppo_tail == 2 || (ctx->ppo_tail < 0 && (mb <= 8192 || mb > 16384));
  return (can32 && want32) ? T32_ROWS : HEAD_ROWS;
}

static bool tail_shape_ok(const rlx_ctx* ctx, const rlx_mlp_desc& d, int64_t mb, const rlx_ppo_hparams& hp) {
  return ctx->ppo_tail && ctx->gemm_bx && !hp.discrete_actions && d.n_hidden == 3 && d.hidden[2] == TL_K3 && d.hidden[1] % G_BN == 0 &&
         d.hidden[1] <= 512 && d.out_dim <= 8 && mb >= 4096 && mb % HEAD_ROWS == 0;
}

// p / c: nullptr = not in this launch (a single network); both given = twin launch
static int launch_tail(rlx_ctx* ctx, const TailNet* p, const TailNet* c, const MbScratch& s, float* metrics, int64_t mb, int N2,
                       int mb_global, const rlx_ppo_hparams& hp, int act, hipStream_t st) {
  const TailNet dummy{};
  const TailNet& tp = p ? *p : dummy;
  const TailNet& tc = c ? *c : dummy;
  const int both = (p && c) ? 1 : 0, which = p ? 0 : 1;
  const float gs = bx_grad_scale(mb_global);
  const float inv_mb = 1.0f / (float)mb_global;
  const size_t lds = 2 * X_OPER + (size_t)HEAD_ROWS * TL_TS * sizeof(float);
  const double nets = both ? 2.0 : 1.0;
  // algorithmic: both products of the last hidden layer (forward and input gradient) + the head; H2 in, dZ3 and dZ2 out
  ProfScope prof(s.valid_rows ? nullptr : ctx, PK_TAIL, nets * 4.0 * (double)mb * N2 * TL_K3, st,
                 nets * 4.0 * ((double)mb * (2 * N2 + TL_K3) + 2.0 * N2 * TL_K3), mb, TL_K3, N2, 1);
  if (tail_rows(ctx, N2, act, mb) == T32_ROWS) {
    constexpr int HPL = T32_ROWS * (2 * 256 + 16), DPL = T32_ROWS * (2 * TL_K3 + 16);
    constexpr int TREG = (T32_ROWS * TL_TS * 4 > 2 * DPL) ? T32_ROWS * TL_TS * 4 : 2 * DPL;
    const size_t lds32 = 2 * HPL + TREG + ((size_t)TL_K3 * 8 + 2 * T32_ROWS * 8 + 16 + 2 * TL_K3 * 8) * sizeof(float);
    static AttrOnce attr32;
    if (!attr32.done()) {
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail32_bx<RLX_ACT_ELU, 256>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr32.mark();
    }
    RLX_PLAUNCH((k_tail32_bx<RLX_ACT_ELU, 256>), dim3((unsigned)(mb / T32_ROWS), both ? 2 : 1), dim3(256), lds32, st, tp, tc, s.mb_a,
                s.aux, s.stats, metrics, mb, inv_mb, hp.clip_range, hp.entropy_coef, hp.critic_coef, s.valid_rows, gs, both, which);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
  }
#define RLX_TAIL_LAUNCH(ACTV)                                                                                        \
  {                                                                                                                  \
    static AttrOnce attr_set;                                                                                          \
    if (!attr_set.done()) {                                                                                                 \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail_bx<ACTV>),                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                      \
      attr_set.mark();                                                                                                 \
    }                                                                                                                \
    RLX_PLAUNCH((k_tail_bx<ACTV>), dim3((unsigned)(mb / HEAD_ROWS), both ? 2 : 1), dim3(256), lds, st, tp, tc, s.mb_a, s.aux,   \
                s.stats, metrics, mb, N2, inv_mb, hp.clip_range, hp.entropy_coef, hp.critic_coef, s.valid_rows, gs, both,      \
                which);                                                                                              \
  }
  if (act == RLX_ACT_ELU) RLX_TAIL_LAUNCH(RLX_ACT_ELU)
  else if (act == RLX_ACT_TANH) RLX_TAIL_LAUNCH(RLX_ACT_TANH)
  else RLX_TAIL_LAUNCH(RLX_ACT_RELU)
#undef RLX_TAIL_LAUNCH
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// picks the register-resident head kernel when the shape allows it
template <bool POLICY>
static int launch_head_loss(float* H, const float* W, const float* b, const float* logstd, const MbScratch& s, float* metrics,
                            int64_t mb, int K, int A, int PS, float inv_mb, const rlx_ppo_hparams& hp, int act, hipStream_t st,
                            rlx_ctx* prof_ctx = nullptr) {
  const int nb = div_up(mb, HEAD_ROWS);
  // algorithmic bytes: H_last in, dZ_last out (in place), the gathered per-row scalars, the per-block partial slabs
  ProfScope prof(s.valid_rows ? nullptr : prof_ctx, PK_HEAD_LOSS, 0.0, st,
                 4.0 * ((double)mb * (2 * K + A + 3) + (double)nb * PS + (double)K * A), mb, K, A, PROF_ENGINE_HBM);
  if (POLICY && hp.discrete_actions) {
    RLX_REQUIRE(A >= 2 && A <= 8, RLX_EUNSUP, "ppo: the Categorical head supports 2..8 actions");
    const size_t ldsd = ((size_t)HEAD_ROWS * (K + 1) + (size_t)K * A + 2 * (size_t)HEAD_ROWS * A + 2 * A) * sizeof(float);
    RLX_REQUIRE(ldsd <= 160 * 1024, RLX_EUNSUP, "ppo head: last hidden layer too wide for the LDS-staged head kernel");
    hipLaunchKernelGGL((k_head_loss<POLICY, true>), dim3(nb), dim3(256), ldsd, st, H, W, b, logstd, s.mb_a, s.aux, s.stats,
                       s.head_part, metrics, mb, K, A, PS, inv_mb, hp.clip_range, hp.entropy_coef, hp.critic_coef, act,
                       s.valid_rows);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
  }
  if (A <= 8 && (K == 64 || K == 128 || K == 256)) {
    const int NP = 256 / K > 0 ? 256 / K : 1;
    const size_t lds = ((size_t)K * 8 + 2 * HEAD_ROWS * 8 + (size_t)HEAD_ROWS * (K + 1) + (size_t)NP * K * 8 + 16) * sizeof(float);
#define RLX_HL_FAST(KQV)                                                                                        \
  {                                                                                                             \
    static AttrOnce attr_set;                                                                                     \
    if (!attr_set.done()) {                                                                                            \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_loss_fast<POLICY, KQV>),              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                 \
      attr_set.mark();                                                                                            \
    }                                                                                                           \
    RLX_PLAUNCH((k_head_loss_fast<POLICY, KQV>), dim3(nb), dim3(256), lds, st, H, W, b, logstd, s.mb_a, s.aux,          \
                s.stats, s.head_part, metrics, mb, A, PS, inv_mb, hp.clip_range, hp.entropy_coef, hp.critic_coef,        \
                act, s.valid_rows);                                                                             \
  }
    if (K == 64) RLX_HL_FAST(16)
    else if (K == 128) RLX_HL_FAST(32)
    else RLX_HL_FAST(64)
#undef RLX_HL_FAST
    RLX_LAUNCH_CHECK();
    return RLX_OK;
  }
  const size_t lds = ((size_t)HEAD_ROWS * (K + 1) + (size_t)K * A + 2 * (size_t)HEAD_ROWS * A + 2 * A) * sizeof(float);
  RLX_REQUIRE(lds <= 160 * 1024, RLX_EUNSUP, "ppo head: last hidden layer too wide for the LDS-staged head kernel");
  hipLaunchKernelGGL(k_head_loss<POLICY>, dim3(nb), dim3(256), lds, st, H, W, b, logstd, s.mb_a, s.aux, s.stats, s.head_part,
                     metrics, mb, K, A, PS, inv_mb, hp.clip_range, hp.entropy_coef, hp.critic_coef, act, s.valid_rows);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

template <bool POLICY>
static int net_fwd_bwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const float* params, float* grads, float* metrics,
                       const MbScratch& s, int64_t mb, int mb_global, const rlx_ppo_hparams& hp, float* sumsq,
                       int* n_sumsq, hipStream_t st, hipEvent_t ev_after_fwd = nullptr) {
  const MlpLayout L = make_layout(d);
  // weight images of the hidden layers for the fp16-pipe GEMMs of this pass (one launch; stale after the optimizer step)
  // (inside a whole-update call the images stay registered between the passes of a bank's network and the clip + Adam
  //  kernel keeps them current: bx_keep)
  const bool kept = ctx->bx_keep[ctx->bank] && ctx->bx_n[ctx->bank] > 0 && d.n_hidden >= 2 &&
                    ctx->bx_img[ctx->bank][0].W == params + L.layer[1].W;
  int rc = (mb >= 4096 && !kept) ? bx_prepare_mlp(ctx, d, L, params, true, st) : RLX_OK;
  if (rc) return rc;
  struct BxScope { rlx_ctx* c; ~BxScope() { if (!c->bx_keep[c->bank]) bx_release(c); } } bx_scope{ctx};
  const float* x_in = (!POLICY && s.mb_xc) ? s.mb_xc : s.mb_x;   // the critic's own observation columns, if it has them
  const int K = L.head.in, A = L.head.out;
  const int PS = K * A + 2 * A + 8;
  int nb = div_up(mb, HEAD_ROWS);
  const float inv_mb = 1.0f / (float)mb_global;
  const bool discrete = POLICY && hp.discrete_actions != 0;
  // row-tile-local tail (k_tail_bx): last hidden layer forward + head + loss + dZ_last + dZ of the layer below in one launch
  const void *w3f = nullptr, *w3t = nullptr;
  float* dz2 = nullptr;
  if (tail_shape_ok(ctx, d, mb, hp)) {
    const LayerOff& o3 = L.layer[2];
    w3f = bx_lookup(ctx, params + o3.W, 0, o3.in, o3.out);
    w3t = bx_lookup(ctx, params + o3.W, 1, o3.out, o3.in);
    if (w3f && w3t) dz2 = (float*)scratch(ctx, SL_DACT_0, (size_t)mb * o3.in * sizeof(float));
  }
  const bool tail = dz2 != nullptr;
  if (tail) nb = (int)(mb / tail_rows(ctx, L.layer[2].in, d.act, mb));
  XmaxScope xscope(ctx, ctx->xmax_slot[(!POLICY && s.mb_xc) ? 1 : 0]);   // scale of the raw-observation operand (k_l12fwd, k_dx_l1bwd)
  // first-layer activations never stored: k_l12fwd leaves the rows' LayerNorm statistics, the merged weight-gradient launch
  // rebuilds its operand (needs the tail's dZ pair -> the two-job launch, and the fused two-layer forward)
  struct L12Scope { rlx_ctx* c; ~L12Scope() { c->l12_stats = nullptr; c->l12_ran = false; } } l12scope{ctx};
  ctx->l12_ran = false;
  ctx->l12_stats = nullptr;
  if (ctx->dw_recompute && tail && ctx->l12_fused && l12fwd_supported(d) && dw_merge_ok(ctx, d, L, mb) && d.hidden[0] == 512 &&
      d.act == RLX_ACT_ELU && grads) {
    ctx->l12_stats = (float*)scratch(ctx, SL_LN_P, (size_t)2 * mb * sizeof(float));
    if (!ctx->l12_stats) return RLX_ENOMEM;
  }
  rc = mlp_trunk_fwd(ctx, d, L, params, x_in, s.acts, mb, st, 0, false, nullptr, tail ? d.n_hidden - 1 : -1);
  if (rc) return rc;
  if (ev_after_fwd) RLX_HIP_TRY(hipEventRecord(ev_after_fwd, st));
  if (tail) {
    const LayerOff& o3 = L.layer[2];
    const TailNet tn{s.acts[1], (const u32x4*)w3f, (const u32x4*)w3t, params + o3.b, params + L.head.W, params + L.head.b,
                     POLICY ? params + L.logstd : nullptr, s.acts[2], dz2, s.head_part, A, PS};
    rc = launch_tail(ctx, POLICY ? &tn : nullptr, POLICY ? nullptr : &tn, s, metrics, mb, o3.in, mb_global, hp, d.act, st);
  } else {
    rc = launch_head_loss<POLICY>(s.acts[d.n_hidden - 1], params + L.head.W, params + L.head.b,
                                  (POLICY && !discrete) ? params + L.logstd : nullptr, s, metrics, mb, K, A, PS, inv_mb, hp,
                                  d.act, st, ctx);
  }
  if (rc) return rc;
  ReduceSeg extra[8];
  int ne = 0;
  extra[ne++] = ReduceSeg{s.head_part, grads + L.head.W, (int64_t)K * A, (int64_t)PS, nb, 0, 1.f, 0.f, 1};
  extra[ne++] = ReduceSeg{s.head_part + K * A, grads + L.head.b, (int64_t)A, (int64_t)PS, nb, 0, 1.f, 0.f, 1};
  if (POLICY) {
    // (d/dlogstd of -entropy_coef * sum_a(logstd_a + c) = -entropy_coef rides in the per-row partials: see the head kernels)
    if (!discrete)
      extra[ne++] = ReduceSeg{s.head_part + K * A + A, grads + L.logstd, (int64_t)A, (int64_t)PS, nb, 0, 1.f, 0.f, 1};
    else   // mean per-sample entropy of the Categorical policy
      extra[ne++] = ReduceSeg{s.head_part + K * A + 2 * A + 3, metrics + 2, 1, (int64_t)PS, nb, 0, inv_mb, 0.f, 0};
    extra[ne++] = ReduceSeg{s.head_part + K * A + 2 * A + 0, metrics + 0, 1, (int64_t)PS, nb, 0, inv_mb, 0.f, 0};
    extra[ne++] = ReduceSeg{s.head_part + K * A + 2 * A + 1, metrics + 3, 1, (int64_t)PS, nb, 0, inv_mb, 0.f, 0};
    extra[ne++] = ReduceSeg{s.head_part + K * A + 2 * A + 2, metrics + 4, 1, (int64_t)PS, nb, 0, inv_mb, 0.f, 0};
  } else {
    extra[ne++] = ReduceSeg{s.head_part + K * A + 2 * A + 0, metrics + 1, 1, (int64_t)PS, nb, 0, inv_mb, 0.f, 0};
  }
  GradScaleScope gscope(ctx, bx_grad_scale(mb_global));   // dZ ~ 1 / mb_global
  TrunkOpts topt;
  topt.dz_below_last = dz2;
  return mlp_trunk_bwd(ctx, d, L, params, x_in, s.acts, grads, mb, extra, ne, sumsq, n_sumsq, st, tail ? &topt : nullptr);
}

// ---------------------------------------------------------------------------------------
// Policy || critic as TWIN launches.  The two networks of the reference's full-jit PPO (ppo/flax_full_jit/policy.py:30-42,
// critic.py:21-32) have the same trunk (in -> 512 LN -> 256 -> 128) and read the same gathered rows, so every kernel of the
// minibatch pass can take both in ONE launch (grid.y = 2; blockIdx.y selects the network's pointers): 11 launches per update on
// one stream and no event operations, instead of 23 launches + 4 event operations on two streams.  That is what the
// launch-latency regime wants (4096-row minibatches: the per-rank share of configs[2], where the chain of dependent 8-27 us
// kernels bounds the update); at 32768 rows the two-stream schedule hides more (one net's memory-bound kernels under the
// other's) and stays the default.  Same kernels and tiles per network; the weight-gradient slabs are half as
// many per network, i.e. the gradients agree with the two-chain schedule up to fp32 summation order.
// ---------------------------------------------------------------------------------------
// max |x| of the observation rows a PPO update call will read (policy rows -> slot 0, the critic's own rows -> slot 1), for the
// lifetime of the call: the fused first-layer backward scales its X planes by it (common.h: l1_xmax)
struct XmaxCall {
  rlx_ctx* c;
  explicit XmaxCall(rlx_ctx* ctx) : c(ctx) {}
  int set(const float* states, int64_t n, const float* cstates, int64_t nc, hipStream_t st) {
    if (!c->gemm_bx) return RLX_OK;
    int rc = x_max_update(c, states, n, 0, st, &c->xmax_slot[0]);
    if (!rc && cstates) rc = x_max_update(c, cstates, nc, 1, st, &c->xmax_slot[1]);
    return rc;
  }
  ~XmaxCall() { c->xmax_slot[0] = c->xmax_slot[1] = nullptr; }
};

static bool twin_shapes_ok(const rlx_ctx* ctx, const rlx_mlp_desc& pd, const rlx_mlp_desc& cd, const rlx_ppo_hparams& hp,
                           int64_t mb) {
  // MEASURED (round 5, whole update of 4096 envs x 128 steps x 10 epochs): 8192 rows 90.4 (twin) vs 90.8 ms (two chains), 16384 rows
  // 71.4 vs 75.5, 32768 rows 71.0 vs 67.3 -- twin launches up to 16384 rows
  if (ctx->ppo_twin == 0 || (ctx->ppo_twin < 0 && mb > 16384)) return false;
  // ... and below 6144 rows when the gathers can be grouped (row records: pack_rollout_rows): kernels of 128 workgroups fill half of
  // the CUs, so two chains that meet once per gather group run side by side -- MEASURED (round 6, 1280 updates of 4096 rows per
  // iteration, same box): 116.1 (two chains) vs 119.6 ms (twin); 8192 rows: 89.4 vs 89.1; 16384 rows: 75.6 vs 73.6
  if (ctx->ppo_twin < 0 && mb < 6144 && ctx->gather_records && ctx->gather_group_rows > mb && !hp.critic_states &&
      pd.in_dim + (hp.discrete_actions ? 1 : pd.out_dim) + 3 <= 128)
    return false;
  // With a real communicator (world > 1 over RCCL) the default is the two-chain schedule at every size: the twin schedule has ONE
  // stream, so its all-reduce (1.4 MB, every update) sits fully exposed between the slab reduction and the Adam launch, while a
  // chain's all-reduce runs under the other network's kernels.  One-GPU cost of the choice: 136.2 vs 122-129 ms per iteration of
  // 1280 4096-row updates -- it pays once a collective takes more than ~10 us.  NOT measured on real links.
  if (ctx->ppo_twin < 0 && ctx->comm && ctx->world > 1) return false;
  if (!ctx->gemm_bx || !ctx->adam_emit || ctx->disable_l1fused || !ctx->l1fwd_mfma || hp.discrete_actions || hp.critic_states)
    return false;
  if (pd.n_hidden != cd.n_hidden || pd.n_hidden < 2 || pd.in_dim != cd.in_dim || pd.act != cd.act ||
      pd.ln_first != cd.ln_first || !pd.has_logstd || cd.out_dim != 1 || pd.out_dim > 8)
    return false;
  for (int l = 0; l < pd.n_hidden; ++l)
    if (pd.hidden[l] != cd.hidden[l] || (l >= 1 && pd.hidden[l] % 4 != 0)) return false;
  const int K = pd.hidden[pd.n_hidden - 1];
  if (!(K == 64 || K == 128 || K == 256)) return false;
  if (mb < 4096 || !l1fwd_mfma_supported(pd) || !l1fused_supported(pd)) return false;   // (split-fp16 images and kernels from 4096 rows)
  for (int l = 1; l < pd.n_hidden; ++l)
    if (!bx_dw_usable(ctx, mb, pd.hidden[l - 1], pd.hidden[l - 1], pd.hidden[l])) return false;
  return true;
}

struct TwinImages {
  const void* f[4][2];     // forward image of hidden layer l (>= 1), network q
  const void* t[4][2];     // transposed image
  const void* w2x[2];      // fused first-layer backward: transposed image of layer 1 / forward image of layer 0
  const void* w1x[2];
};

// lays out (first update of a call) or finds (kept current by the clip + Adam launches) the weight images of both networks;
// false: an image is missing -- the caller takes the two-chain schedule
static int twin_images(rlx_ctx* ctx, const rlx_mlp_desc& pd, const MlpLayout& LP, const float* pparams, const rlx_mlp_desc& cd,
                       const MlpLayout& LC, const float* cparams, hipStream_t st, TwinImages* im, bool* ok) {
  *ok = false;
  const rlx_mlp_desc* dd[2] = {&pd, &cd};
  const MlpLayout* LL[2] = {&LP, &LC};
  const float* pp[2] = {pparams, cparams};
  const int bank0 = ctx->bank;
  for (int q = 0; q < 2; ++q) {
    ctx->bank = q;
    const bool kept = ctx->bx_keep[q] && ctx->bx_n[q] > 0 && ctx->bx_img[q][0].W == pp[q] + LL[q]->layer[1].W;
    if (!kept) {
      const int rc = bx_prepare_mlp(ctx, *dd[q], *LL[q], pp[q], true, st);
      if (rc) { ctx->bank = bank0; return rc; }
    }
    bool all = l1fused_bx_images(ctx, *LL[q], pp[q], &im->w2x[q], &im->w1x[q]);
    for (int l = 1; l < dd[q]->n_hidden && all; ++l) {
      const LayerOff& o = LL[q]->layer[l];
      im->f[l][q] = bx_lookup(ctx, pp[q] + o.W, 0, o.in, o.out);
      im->t[l][q] = bx_lookup(ctx, pp[q] + o.W, 1, o.out, o.in);
      all = im->f[l][q] && im->t[l][q];
    }
    if (!all) { ctx->bank = bank0; return RLX_OK; }
  }
  ctx->bank = bank0;
  *ok = true;
  return RLX_OK;
}

// forward + loss + backward of BOTH networks on the rows of sp (mb_x / mb_a / aux / stats / valid_rows); activations and partial
// slabs of the policy in sp (scratch bank 0), of the critic in sc (bank 1).  Leaves the flat gradients in pg / cg and the
// sum-of-squares partials of both at sq: [0, *npb) the policy's, [*npb, *npb + *ncb) the critic's.
static int twin_fwd_bwd(rlx_ctx* ctx, const rlx_mlp_desc& pd, const MlpLayout& LP, const float* pparams, float* pg,
                        const rlx_mlp_desc& cd, const MlpLayout& LC, const float* cparams, float* cg, const TwinImages& im,
                        float* met, const MbScratch& sp, const MbScratch& sc, int64_t mb, int mb_global,
                        const rlx_ppo_hparams& hp, float* sq, int* npb, int* ncb, hipStream_t st) {
  const int nh = pd.n_hidden, last = nh - 1;
  Twin t;
  XmaxScope xscope(ctx, ctx->xmax_slot[0]);     // scale of the raw-observation operand (k_l12fwd, k_dx_l1bwd)
  int rc;
  int l_first = 1;
  float* stats2[2] = {nullptr, nullptr};     // [2][mb] LayerNorm statistics per network when the first-layer activations are not stored
  const bool l12 = ctx->l12_fused && l12fwd_supported(pd) && im.w1x[0] && im.w1x[1];
  if (l12 && ctx->dw_recompute && ctx->dw_merge && nh == 3 && tail_shape_ok(ctx, pd, mb, hp) && pd.hidden[0] == 512 &&
      pd.act == RLX_ACT_ELU) {
    const int b0 = ctx->bank;
    for (int q = 0; q < 2; ++q) {
      ctx->bank = q;
      stats2[q] = (float*)scratch(ctx, SL_LN_P, (size_t)2 * mb * sizeof(float));
    }
    ctx->bank = b0;
    if (!stats2[0] || !stats2[1]) return RLX_ENOMEM;
  }
  if (l12) {
    // first + second layer of both networks in one launch
    L12Twin tw{cparams, sc.acts[0], sc.acts[1], im.w1x[1], im.f[1][1]};
    tw.stats = stats2[1];
    rc = launch_l12fwd(ctx, pd, LP, pparams, sp.mb_x, sp.acts[0], sp.acts[1], im.w1x[0], im.f[1][0], mb, st, &tw, stats2[0]);
    l_first = 2;
  } else {
    rc = launch_l1fwd_mfma(pd, LP, pparams, sp.mb_x, sp.acts[0], mb, ctx->num_cus, st, nullptr, ctx, cparams, sc.acts[0]);
  }
  if (rc) return rc;
  // row-tile-local tail (k_tail_bx) for both networks in one launch when the shape allows it
  float* dz2[2] = {nullptr, nullptr};
  const float* dzb[4][2];
  for (int l = 0; l < 4; ++l) { dzb[l][0] = sp.acts[l]; dzb[l][1] = sc.acts[l]; }
  if (tail_shape_ok(ctx, pd, mb, hp)) {
    const int bank0_ = ctx->bank;
    for (int q = 0; q < 2; ++q) {
      ctx->bank = q;
      dz2[q] = (float*)scratch(ctx, SL_DACT_0, (size_t)mb * LP.layer[2].in * sizeof(float));
    }
    ctx->bank = bank0_;
    if (!dz2[0] || !dz2[1]) return RLX_ENOMEM;
    dzb[1][0] = dz2[0];
    dzb[1][1] = dz2[1];
  }
  const bool tail = dz2[0] != nullptr;
  for (int l = l_first; l < (tail ? nh - 1 : nh); ++l) {
    const LayerOff& o = LP.layer[l];
    t.p[0] = sc.acts[l - 1]; t.p[1] = im.f[l][1]; t.p[2] = cparams + LC.layer[l].b; t.p[3] = sc.acts[l];
    rc = bx_launch_fwd(ctx, sp.acts[l - 1], im.f[l][0], pparams + o.b, sp.acts[l], mb, o.out, o.in, pd.act, st, 0, nullptr, &t);
    if (rc) return rc;
  }
  const int K = LP.head.in, A = LP.head.out;
  const int PSp = K * A + 2 * A + 8, PSc = K + 2 + 8;
  const int nb = tail ? (int)(mb / tail_rows(ctx, LP.layer[2].in, pd.act, mb)) : div_up(mb, HEAD_ROWS);
  const float inv_mb = 1.0f / (float)mb_global;
  if (tail) {
    const LayerOff& o3 = LP.layer[2];
    const TailNet tp{sp.acts[1], (const u32x4*)im.f[2][0], (const u32x4*)im.t[2][0], pparams + o3.b, pparams + LP.head.W,
                     pparams + LP.head.b, pparams + LP.logstd, sp.acts[2], dz2[0], sp.head_part, A, PSp};
    const TailNet tc{sc.acts[1], (const u32x4*)im.f[2][1], (const u32x4*)im.t[2][1], cparams + LC.layer[2].b, cparams + LC.head.W,
                     cparams + LC.head.b, nullptr, sc.acts[2], dz2[1], sc.head_part, 1, PSc};
    rc = launch_tail(ctx, &tp, &tc, sp, met, mb, o3.in, mb_global, hp, pd.act, st);
    if (rc) return rc;
  } else {
    const HeadNet hn_p{sp.acts[last], pparams + LP.head.W, pparams + LP.head.b, pparams + LP.logstd, sp.head_part, A, PSp};
    const HeadNet hn_c{sc.acts[last], cparams + LC.head.W, cparams + LC.head.b, nullptr, sc.head_part, 1, PSc};
    rc = launch_head_loss_pc(hn_p, hn_c, sp, met, mb, K, inv_mb, hp, pd.act, st);
    if (rc) return rc;
  }
  GradScaleScope gscope(ctx, bx_grad_scale(mb_global));   // dZ ~ 1 / mb_global
  // partial-slab arenas: per network [for l = last..1: W slabs, b slabs][fused first-layer slabs], one workgroup per CU over BOTH nets
  const int cus = ctx->num_cus / 2 > 0 ? ctx->num_cus / 2 : 1;
  int S[4] = {0, 0, 0, 0};
  int64_t Mc[4] = {0, 0, 0, 0};
  size_t per_net = 0;
  for (int l = 1; l < nh; ++l) {
    const LayerOff& o = LP.layer[l];
    Mc[l] = choose_mc(mb, div_up(o.in, G_BM) * div_up(o.out, G_BN), cus, &S[l]);
  }
  const bool dw_merge = ctx->dw_merge && tail && nh == 3;      // both dZ are ready: one two-job launch (bx_launch_dw2), shared M-slabs
  if (dw_merge) {
    const int tiles = div_up(LP.layer[2].in, G_BM) * div_up(LP.layer[2].out, G_BN) + div_up(LP.layer[1].in, G_BM) * div_up(LP.layer[1].out, G_BN);
    Mc[2] = Mc[1] = choose_mc_fit(mb, tiles, cus, &S[2]);
    S[1] = S[2];
  }
  for (int l = 1; l < nh; ++l) per_net += (size_t)S[l] * ((size_t)LP.layer[l].in * LP.layer[l].out + LP.layer[l].out);
  const int lf_grid = l1fused_grid(mb, cus);
  per_net += l1fused_partial_floats(pd, lf_grid);
  float* arena[2];
  const int bank0 = ctx->bank;
  for (int q = 0; q < 2; ++q) {
    ctx->bank = q;
    arena[q] = (float*)scratch(ctx, SL_PARTIAL, per_net * sizeof(float));
  }
  ctx->bank = bank0;
  if (!arena[0] || !arena[1]) return RLX_ENOMEM;
  float *pW[4][2] = {}, *pB[4][2] = {}, *lf[2];
  for (int q = 0; q < 2; ++q) {
    float* cur = arena[q];
    for (int l = last; l >= 1; --l) {
      const LayerOff& o = LP.layer[l];
      pW[l][q] = cur; cur += (size_t)S[l] * o.in * o.out;
      pB[l][q] = cur; cur += (size_t)S[l] * o.out;
    }
    lf[q] = cur;
  }
  float* gr[2] = {pg, cg};
  const MlpLayout* LL[2] = {&LP, &LC};
  ReduceTable tab[2];
  tab[0].n = tab[1].n = 0;
  for (int l = last; l >= 1; --l) {
    const LayerOff& o = LP.layer[l];
    if (dw_merge) {
      if (l == 1) {
        const LayerOff& o3 = LP.layer[2];
        BxDwJob j2{sp.acts[0], dzb[1][0], pW[1][0], pB[1][0], o.in, o.in, o.out, Mc[1], S[1], div_up(o.in, G_BM), div_up(o.out, G_BN)};
        BxDwRecompute rcd;
        if (stats2[0]) {       // the first-layer activations were not stored: the layer-2 job rebuilds them
          const LayerOff& o0 = LP.layer[0];
          rcd.X = sp.mb_x; rcd.W1x = im.w1x[0]; rcd.b1 = pparams + o0.b; rcd.g = pparams + o0.g; rcd.be = pparams + o0.be;
          rcd.stats = stats2[0]; rcd.xmax = ctx->l1_xmax; rcd.W1x1 = im.w1x[1]; rcd.stats1 = stats2[1];
          rcd.pdelta1 = (int64_t)(cparams - pparams); rcd.O = o0.in; rcd.NT1 = o0.out / 32;
          j2.rc = &rcd;
        }
        const BxDwJob j3{sp.acts[1], dzb[2][0], pW[2][0], pB[2][0], o3.in, o3.in, o3.out, Mc[2], S[2], div_up(o3.in, G_BM), div_up(o3.out, G_BN)};
        Twin t2, t3;
        t2.p[0] = sc.acts[0]; t2.p[1] = dzb[1][1]; t2.p[2] = pW[1][1]; t2.p[3] = pB[1][1];
        t3.p[0] = sc.acts[1]; t3.p[1] = dzb[2][1]; t3.p[2] = pW[2][1]; t3.p[3] = pB[2][1];
        rc = bx_launch_dw2(ctx, j2, j3, mb, st, &t2, &t3);
        if (rc) return rc;
      }
    } else {
      t.p[0] = sc.acts[l - 1]; t.p[1] = dzb[l][1]; t.p[2] = pW[l][1]; t.p[3] = pB[l][1];
      rc = bx_launch_dw(ctx, sp.acts[l - 1], dzb[l][0], pW[l][0], pB[l][0], mb, o.in, o.in, o.out, Mc[l], S[l], div_up(o.in, G_BM),
                        div_up(o.out, G_BN), st, &t);
      if (rc) return rc;
    }
    for (int q = 0; q < 2; ++q) {
      const LayerOff& oq = LL[q]->layer[l];
      tab[q].seg[tab[q].n++] = ReduceSeg{pW[l][q], gr[q] + oq.W, (int64_t)o.in * o.out, (int64_t)o.in * o.out, S[l], 0, 1.f, 0.f, 1};
      tab[q].seg[tab[q].n++] = ReduceSeg{pB[l][q], gr[q] + oq.b, (int64_t)o.out, (int64_t)o.out, S[l], 0, 1.f, 0.f, 1};
    }
    if (l == 1) break;   // the layer-1 input gradient is folded into the fused first-layer backward below
    if (tail && l == last) continue;   // dZ of the layer below came out of the tail launch
    t.p[0] = dzb[l][1]; t.p[1] = im.t[l][1]; t.p[2] = nullptr; t.p[3] = sc.acts[l - 1];
    rc = bx_launch_dx(ctx, dzb[l][0], im.t[l][0], sp.acts[l - 1], mb, o.out, o.in, o.in, pd.act, 1, st, &t);
    if (rc) return rc;
  }
  {
    L1FusedTwin tw{cparams, dzb[1][1], lf[1], cg, im.w2x[1], im.w1x[1], &tab[1]};
    ctx->bank = 0;       // (the policy's images are the ones registered under bank 0)
    rc = launch_l1fused(ctx, pd, LP, pparams, sp.mb_x, dzb[1][0], lf[0], lf_grid, pg, mb, &tab[0], st, &tw);
    ctx->bank = bank0;
    if (rc) return rc;
  }
  // head partials and metric sums: the segment order of net_fwd_bwd per network
  tab[0].seg[tab[0].n++] = ReduceSeg{sp.head_part, pg + LP.head.W, (int64_t)K * A, (int64_t)PSp, nb, 0, 1.f, 0.f, 1};
  tab[0].seg[tab[0].n++] = ReduceSeg{sp.head_part + K * A, pg + LP.head.b, (int64_t)A, (int64_t)PSp, nb, 0, 1.f, 0.f, 1};
  tab[0].seg[tab[0].n++] = ReduceSeg{sp.head_part + K * A + A, pg + LP.logstd, (int64_t)A, (int64_t)PSp, nb, 0, 1.f, 0.f, 1};
  tab[0].seg[tab[0].n++] = ReduceSeg{sp.head_part + K * A + 2 * A + 0, met + 0, 1, (int64_t)PSp, nb, 0, inv_mb, 0.f, 0};
  tab[0].seg[tab[0].n++] = ReduceSeg{sp.head_part + K * A + 2 * A + 1, met + 3, 1, (int64_t)PSp, nb, 0, inv_mb, 0.f, 0};
  tab[0].seg[tab[0].n++] = ReduceSeg{sp.head_part + K * A + 2 * A + 2, met + 4, 1, (int64_t)PSp, nb, 0, inv_mb, 0.f, 0};
  tab[1].seg[tab[1].n++] = ReduceSeg{sc.head_part, cg + LC.head.W, (int64_t)K, (int64_t)PSc, nb, 0, 1.f, 0.f, 1};
  tab[1].seg[tab[1].n++] = ReduceSeg{sc.head_part + K, cg + LC.head.b, 1, (int64_t)PSc, nb, 0, 1.f, 0.f, 1};
  tab[1].seg[tab[1].n++] = ReduceSeg{sc.head_part + K + 2 + 0, met + 1, 1, (int64_t)PSc, nb, 0, inv_mb, 0.f, 0};
  ReduceTable all;
  all.n = 0;
  RLX_REQUIRE(tab[0].n + tab[1].n <= REDUCE_MAX_SEGS, RLX_EUNSUP, "ppo twin update: too many reduction segments");
  for (int q = 0; q < 2; ++q)
    for (int i = 0; i < tab[q].n; ++i) all.seg[all.n++] = tab[q].seg[i];
  int total = 0;
  rc = launch_reduce_segments(all, sq, &total, st, ctx);
  if (rc) return rc;
  int nbp = 0;
  for (int i = 0; i < tab[0].n; ++i) nbp += all.seg[i].nblocks;
  *npb = nbp;
  *ncb = total - nbp;
  return RLX_OK;
}

static int minibatch_core(rlx_ctx* ctx, const rlx_mlp_desc& pd, const float* pparams, float* pgrads,
                          const rlx_mlp_desc& cd, const float* cparams, float* cgrads, float* metrics,
                          const float* states, const float* actions, const float* log_probs, const float* returns,
                          const float* advantages, const int32_t* idx, int mb_local, int mb_global, double* stats_io,
                          int phase, const rlx_ppo_hparams& hp, float* p_sumsq, int* p_nsq, float* c_sumsq, int* c_nsq,
                          hipStream_t st, hipStream_t st_c = nullptr, double* prezeroed_stats = nullptr, bool allow_twin = false) {
  // prezeroed_stats: the whole-update caller zeroed `metrics` and this 4-double statistics slot up front (one memset
  // per update call instead of two per minibatch on the critical path)
  int rc = mlp_check_desc(pd);
  if (rc) return rc;
  rc = mlp_check_desc(cd);
  if (rc) return rc;
  const bool crows = hp.critic_states != nullptr;
  RLX_REQUIRE(crows || pd.in_dim == cd.in_dim, RLX_EUNSUP,
              "ppo: policy and critic must share the observation (or the critic's rows come through hparams.critic_states)");
  RLX_REQUIRE((hp.discrete_actions ? !pd.has_logstd : pd.has_logstd) && cd.out_dim == 1, RLX_EINVAL,
              "ppo: a Gaussian policy needs logstd, a Categorical one must not have it; critic out_dim must be 1");
  MbScratch s;
  rc = mb_scratch(ctx, pd, cd, mb_local > 0 ? mb_local : 1, &s, crows);
  if (rc) return rc;
  if (prezeroed_stats) s.stats = prezeroed_stats;
  // single-minibatch entries set the observation-scale slots for THIS call only (below, after their gather)
  struct XmaxReset { rlx_ctx* c; bool on; ~XmaxReset() { if (on) c->xmax_slot[0] = c->xmax_slot[1] = nullptr; } } xreset{ctx, prezeroed_stats == nullptr};
  const int O = pd.in_dim, A = pd.out_dim;
  // phase 0: gather + write local stats, return.   phase 1: consume all-reduced stats (rows were
  // gathered by the preceding phase-0 call).   phase 2: gather AND consume externally supplied
  // (already global) stats in one call -- the batched-statistics protocol of the multi-GPU loop.
  // phase 3 / 4: phase 2 split in halves (policy net / critic net on the rows gathered by phase 3), so the host can
  // all-reduce the policy gradients while the critic runs.  phase 5 / 6 / 4: gather + statistics only, then the policy
  // (6) and the critic (4) as separate calls the host may issue on two streams; the critic uses scratch bank 1.
  if ((phase == 4 || phase == 6) && mb_local == 0) {
    RLX_HIP_TRY(hipMemsetAsync(metrics, 0, 8 * sizeof(float), st));
    float* gz = phase == 4 ? cgrads : pgrads;
    RLX_HIP_TRY(hipMemsetAsync(gz, 0, (size_t)make_layout(phase == 4 ? cd : pd).n_params * sizeof(float), st));
    return RLX_OK;
  }
  if (phase == 4) {
    MbScratch s2 = s, tmp;
    ctx->bank = 1;
    rc = mb_scratch(ctx, pd, cd, mb_local, &tmp);
    if (!rc) {
      for (int l = 0; l < 4; ++l) s2.acts[l] = tmp.acts[l];
      s2.head_part = tmp.head_part;
      rc = hipMemsetAsync(metrics, 0, 8 * sizeof(float), st) == hipSuccess ? RLX_OK : RLX_EHIP;
      if (!rc) rc = net_fwd_bwd<false>(ctx, cd, cparams, cgrads, metrics, s2, mb_local, mb_global, hp, c_sumsq, c_nsq, st);
    }
    ctx->bank = 0;
    return rc;
  }
  if (phase == 6) {
    RLX_HIP_TRY(hipMemsetAsync(metrics, 0, 8 * sizeof(float), st));
    return net_fwd_bwd<true>(ctx, pd, pparams, pgrads, metrics, s, mb_local, mb_global, hp, p_sumsq, p_nsq, st);
  }
  const bool do_gather = (stats_io == nullptr) || phase == 0 || phase == 2 || phase == 3 || phase == 5;
  if (do_gather) {
    if (!prezeroed_stats) RLX_HIP_TRY(hipMemsetAsync(s.stats, 0, 32, st));
    if (mb_local > 0) {
      const int A_act = hp.discrete_actions ? 1 : A;   // Categorical: one action index per sample
      rc = launch_gather(ctx, states, actions, log_probs, returns, advantages, idx, s, prezeroed_stats ? nullptr : s.stats,
                         nullptr, (int64_t)mb_local, O, A_act, st, hp.critic_states, cd.in_dim);
      if (rc) return rc;
      if (!prezeroed_stats && ctx->gemm_bx && phase != 0 && phase != 5) {
        // single-minibatch entries: max |x| of the rows just gathered (the whole-update calls take it over their rollout once)
        rc = x_max_update(ctx, s.mb_x, (int64_t)mb_local * O, 0, st, &ctx->xmax_slot[0]);
        if (!rc && s.mb_xc) rc = x_max_update(ctx, s.mb_xc, (int64_t)mb_local * cd.in_dim, 1, st, &ctx->xmax_slot[1]);
        if (rc) return rc;
      }
    }
    if (stats_io && phase == 0) {
      RLX_HIP_TRY(hipMemcpyAsync(stats_io, s.stats, 32, hipMemcpyDeviceToDevice, st));
      return RLX_OK;  // phase 0 ends here; the host all-reduces stats_io
    }
    if (stats_io && (phase == 2 || phase == 3 || phase == 5))
      RLX_HIP_TRY(hipMemcpyAsync(s.stats, stats_io, 32, hipMemcpyDeviceToDevice, st));
    if (stats_io && phase == 5) return RLX_OK;
  } else {
    RLX_HIP_TRY(hipMemcpyAsync(s.stats, stats_io, 32, hipMemcpyDeviceToDevice, st));
  }
  if (!prezeroed_stats) RLX_HIP_TRY(hipMemsetAsync(metrics, 0, 8 * sizeof(float), st));
  if (mb_local == 0) {   // a rank without rows of this global minibatch contributes zero gradients (and still all-reduces)
    if (pgrads) RLX_HIP_TRY(hipMemsetAsync(pgrads, 0, (size_t)make_layout(pd).n_params * sizeof(float), st));
    if (cgrads && !(stats_io && phase == 3)) RLX_HIP_TRY(hipMemsetAsync(cgrads, 0, (size_t)make_layout(cd).n_params * sizeof(float), st));
    if (p_nsq) *p_nsq = 0;
    if (c_nsq) *c_nsq = 0;
    return RLX_OK;
  }
  if (allow_twin && ctx->ppo_twin == 1 && pgrads && cgrads && !(stats_io && phase == 3) &&
      twin_shapes_ok(ctx, pd, cd, hp, mb_local)) {
    // the twin-launch pass of the whole-update calls, reachable for ONE minibatch (option ppo_twin = 1; tests hold its gradients
    // against the float64 oracle through this entry)
    const MlpLayout LP = make_layout(pd), LC = make_layout(cd);
    TwinImages im;
    bool ok = false;
    rc = twin_images(ctx, pd, LP, pparams, cd, LC, cparams, st, &im, &ok);
    struct Rel { rlx_ctx* c; ~Rel() { if (!c->bx_keep[0] && !c->bx_keep[1]) bx_release_all(c); } } rel{ctx};
    if (rc) return rc;
    if (ok) {
      MbScratch s2 = s, tmp;
      ctx->bank = 1;
      rc = mb_scratch(ctx, pd, cd, mb_local, &tmp);
      ctx->bank = 0;
      if (rc) return rc;
      for (int l = 0; l < 4; ++l) s2.acts[l] = tmp.acts[l];
      s2.head_part = tmp.head_part;
      int npb = 0, ncb = 0;
      rc = twin_fwd_bwd(ctx, pd, LP, pparams, pgrads, cd, LC, cparams, cgrads, im, metrics, s, s2, mb_local, mb_global, hp, p_sumsq,
                        &npb, &ncb, st);
      if (p_nsq) *p_nsq = npb;
      if (c_nsq) *c_nsq = 0;      // (the critic's partials follow the policy's in p_sumsq: this entry's callers do not consume them)
      return rc;
    }
  }
  if (st_c && st_c != st) {
    // policy || critic: the two nets are independent once the rows are gathered.  The critic runs on the side
    // stream with its own activation / slab arenas (scratch bank 1); the caller joins after the optimizer steps.
    RLX_HIP_TRY(hipEventRecord(ctx->ev_fork, st));
    RLX_HIP_TRY(hipStreamWaitEvent(st_c, ctx->ev_fork, 0));
    MbScratch s2 = s;
    ctx->bank = 1;
    MbScratch tmp;
    rc = mb_scratch(ctx, pd, cd, mb_local, &tmp);
    if (!rc) {
      for (int l = 0; l < 4; ++l) s2.acts[l] = tmp.acts[l];
      s2.head_part = tmp.head_part;
      rc = net_fwd_bwd<false>(ctx, cd, cparams, cgrads, metrics, s2, mb_local, mb_global, hp, c_sumsq, c_nsq, st_c);
    }
    ctx->bank = 0;
    if (rc) return rc;
    return net_fwd_bwd<true>(ctx, pd, pparams, pgrads, metrics, s, mb_local, mb_global, hp, p_sumsq, p_nsq, st);
  }
  rc = net_fwd_bwd<true>(ctx, pd, pparams, pgrads, metrics, s, mb_local, mb_global, hp, p_sumsq, p_nsq, st);
  if (rc || (stats_io && phase == 3)) return rc;
  return net_fwd_bwd<false>(ctx, cd, cparams, cgrads, metrics, s, mb_local, mb_global, hp, c_sumsq, c_nsq, st);
}

int ppo_sample(const float* mean, const float* logstd, uint32_t k0, uint32_t k1, int scheme, float* action, float* processed,
               float* logp, const float* obs, float* states_row, int N, int A, int O, int clip_and_rescale, const float* lo,
               const float* hi, int row_off, int N_global, hipStream_t st, int deterministic) {
  hipLaunchKernelGGL(k_sample, dim3(div_up(N, 256)), dim3(256), 0, st, mean, logstd, k0, k1, scheme, action, processed, logp,
                     obs, states_row, N, A, O, clip_and_rescale, lo, hi, row_off, N_global, deterministic);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int ppo_mb_scratch(rlx_ctx* ctx, int O, int A, const rlx_mlp_desc& cd, int Kp, int64_t mb, MbScratch* s, bool critic_rows) {
  s->mb_x = (float*)scratch(ctx, SL_MB_X, (size_t)mb * (O + A) * sizeof(float));
  s->mb_xc = nullptr;
  if (critic_rows) {
    s->mb_xc = (float*)scratch(ctx, SL_MB_XC, (size_t)mb * cd.in_dim * sizeof(float));
    if (!s->mb_xc) return RLX_ENOMEM;
  }
  s->aux = (float*)scratch(ctx, SL_MB_AUX, (size_t)mb * 3 * sizeof(float));
  s->stats = (double*)scratch(ctx, SL_STATS, 64);
  if (!s->mb_x || !s->aux || !s->stats) return RLX_ENOMEM;
  s->mb_a = s->mb_x + (size_t)mb * O;
  for (int l = 0; l < cd.n_hidden; ++l) {
    s->acts[l] = (float*)scratch(ctx, (ScratchSlot)(SL_ACT_P0 + l), (size_t)mb * cd.hidden[l] * sizeof(float));
    if (!s->acts[l]) return RLX_ENOMEM;
  }
  const int Kc = cd.hidden[cd.n_hidden - 1];
  const size_t psp = (size_t)Kp * A + 2 * A + 8, psc = (size_t)Kc + 2 + 8;
  const size_t nb = (size_t)div_up(mb, 32);   // (the 32-row tail form leaves one partial block per 32 rows)
  s->head_part = (float*)scratch(ctx, SL_HEAD_PART, nb * (psp > psc ? psp : psc) * sizeof(float));
  return s->head_part ? RLX_OK : RLX_ENOMEM;
}

int ppo_gather(rlx_ctx* ctx, const float* states, const float* actions, const float* log_probs, const float* returns,
               const float* advantages, const int32_t* idx, int64_t mb, int O, int A, const MbScratch& s, hipStream_t st,
               const float* cstates, int Oc, bool local_stats) {
  if (local_stats) RLX_HIP_TRY(hipMemsetAsync(s.stats, 0, 32, st));
  int rc = launch_gather(ctx, states, actions, log_probs, returns, advantages, idx, s, local_stats ? s.stats : nullptr, nullptr, mb, O, A, st,
                         cstates, Oc);
  if (rc) return rc;
  return RLX_OK;
}

int ppo_policy_head_loss(rlx_ctx* ctx, float* h_last, const float* Wh, const float* bh, const float* logstd,
                         const MbScratch& s, float* metrics, int64_t mb, int mb_global, int K, int A, int act,
                         const rlx_ppo_hparams& hp, float* gW, float* gb, float* glogstd, float* sumsq, int* nsq,
                         hipStream_t st) {
  const int PS = K * A + 2 * A + 8;
  const int nb = div_up(mb, HEAD_ROWS);
  const float inv_mb = 1.0f / (float)mb_global;
  int rc = launch_head_loss<true>(h_last, Wh, bh, logstd, s, metrics, mb, K, A, PS, inv_mb, hp, act, st);
  if (rc) return rc;
  ReduceTable tab;
  tab.n = 0;
  tab.seg[tab.n++] = ReduceSeg{s.head_part, gW, (int64_t)K * A, (int64_t)PS, nb, 0, 1.f, 0.f, 1};
  tab.seg[tab.n++] = ReduceSeg{s.head_part + K * A, gb, (int64_t)A, (int64_t)PS, nb, 0, 1.f, 0.f, 1};
  tab.seg[tab.n++] = ReduceSeg{s.head_part + K * A + A, glogstd, (int64_t)A, (int64_t)PS, nb, 0, 1.f, 0.f, 1};
  tab.seg[tab.n++] = ReduceSeg{s.head_part + K * A + 2 * A + 0, metrics + 0, 1, (int64_t)PS, nb, 0, inv_mb, 0.f, 0};
  tab.seg[tab.n++] = ReduceSeg{s.head_part + K * A + 2 * A + 1, metrics + 3, 1, (int64_t)PS, nb, 0, inv_mb, 0.f, 0};
  tab.seg[tab.n++] = ReduceSeg{s.head_part + K * A + 2 * A + 2, metrics + 4, 1, (int64_t)PS, nb, 0, inv_mb, 0.f, 0};
  return stage_reduce(ctx, tab, sumsq, nsq, st);
}

int ppo_critic_fwd_bwd(rlx_ctx* ctx, const rlx_mlp_desc& cd, const float* cparams, float* cgrads, float* metrics,
                       const MbScratch& s, int64_t mb, int mb_global, const rlx_ppo_hparams& hp, float* sumsq, int* nsq,
                       hipStream_t st) {
  return net_fwd_bwd<false>(ctx, cd, cparams, cgrads, metrics, s, mb, mb_global, hp, sumsq, nsq, st);
}

}  // namespace rlx

using namespace rlx;

extern "C" {

int rlx_ppo_minibatch_fwd_bwd_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams, float* pgrads,
                                  const rlx_mlp_desc* cdesc, const float* cparams, float* cgrads, float* metrics,
                                  const float* states, const float* actions, const float* log_probs,
                                  const float* returns, const float* advantages, const int32_t* idx, int mb_local,
                                  int mb_global, double* stats_io, int phase, const rlx_ppo_hparams* hp, void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && cdesc && cparams && metrics && states && actions && log_probs && returns &&
                  advantages && idx && hp && (pgrads || phase == 4 || phase == 5) &&
                  (cgrads || phase == 3 || phase == 5 || phase == 6),
              RLX_EINVAL, "rlx_ppo_minibatch_fwd_bwd_f32: NULL pointer");
  RLX_REQUIRE(phase >= 0 && phase <= 6 && (stats_io || phase < 3), RLX_EINVAL,
              "rlx_ppo_minibatch_fwd_bwd_f32: phase must be 0..6 (3..6 need stats_io)");
  RLX_REQUIRE(mb_local >= 0 && mb_global >= mb_local && mb_global > 0, RLX_EINVAL,
              "rlx_ppo_minibatch_fwd_bwd_f32: need 0 <= mb_local <= mb_global, mb_global > 0");
  float* psq = (float*)scratch(ctx, SL_NORM, REDUCE_MAX_BLOCKS * sizeof(float));
  float* csq = (float*)scratch(ctx, SL_NORM2, REDUCE_MAX_BLOCKS * sizeof(float));
  if (!psq || !csq) return RLX_ENOMEM;
  int np = 0, nc = 0;
  return minibatch_core(ctx, *pdesc, pparams, pgrads, *cdesc, cparams, cgrads, metrics, states, actions, log_probs,
                        returns, advantages, idx, mb_local, mb_global, stats_io, phase, *hp, psq, &np, csq, &nc,
                        (hipStream_t)stream, nullptr, nullptr, /*allow_twin=*/true);
}

int rlx_ppo_prefetch_permutation(rlx_ctx* ctx, const uint32_t key_at_update[2], int nr_epochs, int64_t B, int scheme,
                                 void* stream) {
  RLX_REQUIRE(ctx && key_at_update && nr_epochs > 0 && B > 0, RLX_EINVAL, "rlx_ppo_prefetch_permutation: bad args");
  int32_t* perm = (int32_t*)scratch(ctx, SL_PERM, (size_t)nr_epochs * B * sizeof(int32_t));
  if (!perm) return RLX_ENOMEM;
  if (!ctx->pf_done) RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->pf_done, hipEventDisableTiming));
  uint32_t k[2] = {key_at_update[0], key_at_update[1]};
  ctx->pf_valid = false;
  // runs on the library's side stream (idle outside the updates), ordered after the previous update's last read of the
  // permutation buffer -- NOT after what the caller has queued on `stream` since: called once the rollout is queued, the
  // sorts run under it without the host's ~250 sort launches delaying the first acting step
  int rc = ctx_side_stream(ctx);
  if (rc) return rc;
  if (ctx->perm_free_recorded) {
    RLX_HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_perm_free, 0));
  } else {
    RLX_HIP_TRY(hipEventRecord(ctx->ev_fork, (hipStream_t)stream));
    RLX_HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
  }
  rc = rlx_permutation_i32(ctx, k, perm, nr_epochs, B, scheme, ctx->side);
  if (rc) return rc;
  RLX_HIP_TRY(hipEventRecord(ctx->pf_done, ctx->side));
  ctx->pf_key_in[0] = key_at_update[0]; ctx->pf_key_in[1] = key_at_update[1];
  ctx->pf_key_out[0] = k[0]; ctx->pf_key_out[1] = k[1];
  ctx->pf_E = nr_epochs; ctx->pf_B = B; ctx->pf_scheme = scheme;
  ctx->pf_dist = false;
  ctx->pf_valid = true;
  return RLX_OK;
}

int rlx_ppo_update_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, float* pparams, float* pm, float* pv,
                       const rlx_mlp_desc* cdesc, float* cparams, float* cm, float* cv, const float* states,
                       const float* actions, const float* log_probs, const float* returns, const float* advantages,
                       int T, int N, int nr_epochs, int minibatch_size, uint32_t key_io[2], int scheme,
                       int64_t* opt_count_io, const float* lr_schedule, const rlx_ppo_hparams* hp, float* metrics_out,
                       void* stream) {
  if (ctx) ctx->ro_img.valid = false;   // the acting nets' weight images go stale with this call
  // weight images of the two networks persist over the updates of this call (re-emitted by clip + Adam), dropped at its end
  struct BxKeep {
    rlx_ctx* c;
    ~BxKeep() {
      if (!c) return;
      c->bx_keep[0] = c->bx_keep[1] = false;
      bx_release_all(c);
    }
  } bx_keep_scope{ctx};
  if (ctx && ctx->adam_emit && ctx->gemm_bx) ctx->bx_keep[0] = ctx->bx_keep[1] = true;
  RLX_REQUIRE(ctx && pdesc && pparams && pm && pv && cdesc && cparams && cm && cv && states && actions && log_probs &&
                  returns && advantages && key_io && opt_count_io && lr_schedule && hp && metrics_out,
              RLX_EINVAL, "rlx_ppo_update_f32: NULL pointer");
  const int64_t B = (int64_t)T * N;
  RLX_REQUIRE(T > 0 && N > 0 && nr_epochs > 0 && minibatch_size > 0 && B % minibatch_size == 0, RLX_EINVAL,
              "rlx_ppo_update_f32: batch (T*N) must be a positive multiple of minibatch_size");
  hipStream_t st = (hipStream_t)stream;
  const int M = (int)(B / minibatch_size);
  const int64_t np_ = rlx_mlp_param_count(pdesc), nc_ = rlx_mlp_param_count(cdesc);
  RLX_REQUIRE(np_ > 0 && nc_ > 0, RLX_EINVAL, "rlx_ppo_update_f32: bad MLP descriptor");
  int32_t* perm = (int32_t*)scratch(ctx, SL_PERM, (size_t)nr_epochs * B * sizeof(int32_t));
  float* pg = (float*)scratch(ctx, SL_GRAD_P, (size_t)np_ * sizeof(float));
  float* cg = (float*)scratch(ctx, SL_GRAD_C, (size_t)nc_ * sizeof(float));
  float* psq = (float*)scratch(ctx, SL_NORM, REDUCE_MAX_BLOCKS * sizeof(float));
  float* csq = (float*)scratch(ctx, SL_NORM2, REDUCE_MAX_BLOCKS * sizeof(float));
  if (!perm || !pg || !cg || !psq || !csq) return RLX_ENOMEM;
  int rc = RLX_OK;
  if (ctx->pf_valid && !ctx->pf_dist && ctx->pf_key_in[0] == key_io[0] && ctx->pf_key_in[1] == key_io[1] &&
      ctx->pf_E == nr_epochs && ctx->pf_B == B && ctx->pf_scheme == scheme) {
    // the permutation for exactly this key was generated ahead of time (on another stream): just order after it
    RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->pf_done, 0));
    key_io[0] = ctx->pf_key_out[0];
    key_io[1] = ctx->pf_key_out[1];
  } else {
    // a prefetch for another key / shape may still be writing the permutation and sort buffers on the side stream
    if (ctx->pf_valid && ctx->pf_done) RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->pf_done, 0));
    rc = rlx_permutation_i32(ctx, key_io, perm, nr_epochs, B, scheme, stream);
    if (rc) return rc;
  }
  ctx->pf_valid = false;
  hipStream_t st_c = st;
  if (ctx->two_streams) {
    rc = ctx_side_stream(ctx);
    if (rc) return rc;
    st_c = ctx->side;
  }
  const int n_upd = nr_epochs * M;
  double* stats_all = (double*)scratch(ctx, SL_STATS_ALL, (size_t)n_upd * 4 * sizeof(double));
  if (!stats_all) return RLX_ENOMEM;
  XmaxCall xmax_call(ctx);
  rc = xmax_call.set(states, B * pdesc->in_dim, hp->critic_states, B * cdesc->in_dim, st);
  if (rc) return rc;
  // policy || critic as twin launches on ONE stream (small minibatches by default: twin_shapes_ok); needs both networks' images
  const MlpLayout LPt = make_layout(*pdesc), LCt = make_layout(*cdesc);
  TwinImages tim;
  bool twin = twin_shapes_ok(ctx, *pdesc, *cdesc, *hp, minibatch_size);
  if (twin) {
    rc = twin_images(ctx, *pdesc, LPt, pparams, *cdesc, LCt, cparams, st, &tim, &twin);
    if (rc) return rc;
  }
  if (twin || (st_c != st && ctx->pipeline_updates)) {
    // Policy chain on the main stream, critic chain on the side stream, and NO join between updates: the gathered rows are
    // double buffered (scratch banks 0 / 1 by update parity), so gather(u+1) and policy(u+1) start while critic(u) is still
    // running.  The two chains drift out of phase and one net's bandwidth-bound kernels (first layer, head/loss,
    // slab reduction, Adam) run under the other net's MFMA kernels instead of next to their twins.  Same kernels, same
    // inputs, same order per net: the results are those of the joined schedule bit for bit.
    const int O = pdesc->in_dim, A = pdesc->out_dim;
    MbScratch sb[2];
    for (int b = 0; b < 2; ++b) {
      ctx->bank = b;
      rc = mb_scratch(ctx, *pdesc, *cdesc, minibatch_size, &sb[b], hp->critic_states != nullptr);
      ctx->bank = 0;
      if (rc) return rc;
    }
    rc = pack_rollout_rows(ctx, states, actions, log_probs, returns, advantages, B, O, hp->discrete_actions ? 1 : A,
                           hp->critic_states != nullptr, sb, 2, st);
    if (rc) return rc;
    // per-update {lr, 1 - b1^step, 1 - b2^step} in a device table (pinned staging ring -> one H2D copy per call)
    float* sched_dev = (float*)scratch(ctx, SL_SCHED, (size_t)n_upd * 4 * sizeof(float));
    if (!sched_dev) return RLX_ENOMEM;
    {
      const size_t need = (size_t)n_upd * 4 * sizeof(float);
      if (ctx->sched_cap < need) {
        for (int q = 0; q < 4; ++q) {
          if (ctx->sched_ev[q]) RLX_HIP_TRY(hipEventSynchronize(ctx->sched_ev[q]));
          if (ctx->sched_host[q]) RLX_HIP_TRY(hipHostFree(ctx->sched_host[q]));
          ctx->sched_host[q] = nullptr;
          RLX_HIP_TRY(hipHostMalloc((void**)&ctx->sched_host[q], need, hipHostMallocDefault));
          if (!ctx->sched_ev[q]) RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->sched_ev[q], hipEventDisableTiming));
        }
        ctx->sched_cap = need;
      }
      const int slot = ctx->sched_pos;
      ctx->sched_pos = (slot + 1) & 3;
      RLX_HIP_TRY(hipEventSynchronize(ctx->sched_ev[slot]));   // the copy issued four calls ago has long completed
      float* h = ctx->sched_host[slot];
      for (int u = 0; u < n_upd; ++u) {
        adam_schedule_entry(h + 4 * u, *opt_count_io + u + 1, lr_schedule[u], hp->adam_b1, hp->adam_b2);
        h[4 * u + 3] = 0.f;
      }
      RLX_HIP_TRY(hipMemcpyAsync(sched_dev, h, need, hipMemcpyHostToDevice, st));
      RLX_HIP_TRY(hipEventRecord(ctx->sched_ev[slot], st));
    }
    auto issue = [&](hipStream_t s0) -> int {
      // advantage statistics of all E*M minibatches: one workgroup each, fixed summation order (dist.hip)
      int r = dist_adv_sums(advantages, perm, nullptr, n_upd, minibatch_size, minibatch_size, stats_all, s0);
      if (r) return r;
      RLX_HIP_TRY(hipMemsetAsync(metrics_out, 0, (size_t)n_upd * 10 * sizeof(float), s0));
      // Small minibatches (the per-rank share of a sharded job): ~11 dependent kernels of 5-17 us per chain, and the update is
      // bound by the issuing thread (measured on this box: 2.9 us per launch, 4.4 us per event record, ~4 us per stream wait;
      // tools/probes/launch_cost.hip).  Each chain then gathers ITS OWN copy of the rows on its own stream: one more 4 us
      // kernel, four event operations fewer per update, and nothing couples the chains between the fork and the final join.
      if (twin) {
        // The rows of G consecutive updates are gathered by ONE launch (the permutation of the whole call exists up front and the
        // updates' index ranges are adjacent): at 4096 rows the gather was 5.7 of the update's 107 us, a dependent launch of its own.
        const int grp_rows = ctx->gather_group_rows > 32768 && sb[0].rec ? ctx->gather_group_rows : 32768;
        const int G = n_upd < grp_rows / minibatch_size ? n_upd : (grp_rows / minibatch_size > 0 ? grp_rows / minibatch_size : 1);
        MbScratch grp = sb[0];
        if (G > 1) {
          const int64_t rows = (int64_t)G * minibatch_size;
          grp.mb_x = (float*)scratch(ctx, SL_MB_GROUP_X, (size_t)rows * (O + A) * sizeof(float));
          grp.aux = (float*)scratch(ctx, SL_MB_GROUP_AUX, (size_t)rows * 3 * sizeof(float));
          if (!grp.mb_x || !grp.aux) return RLX_ENOMEM;
          grp.mb_a = grp.mb_x + (size_t)rows * O;
          grp.mb_xc = nullptr;
        }
        for (int u = 0; u < n_upd; ++u) {
          float* met = metrics_out + (int64_t)u * 10;
          const int j = u % G;
          if (j == 0) {
            const int g = n_upd - u < G ? n_upd - u : G;
            grp.mb_a = grp.mb_x + (size_t)g * minibatch_size * O;                 // (a shorter last group: [g*mb, O] then [g*mb, A])
            r = launch_gather(ctx, states, actions, log_probs, returns, advantages, perm + (int64_t)u * minibatch_size, grp,
                              nullptr, nullptr, (int64_t)g * minibatch_size, O, A, s0);
            if (r) return r;
          }
          MbScratch sp = sb[0], sc = sb[1];
          sp.mb_x = grp.mb_x + (size_t)j * minibatch_size * O;
          sp.mb_a = grp.mb_a + (size_t)j * minibatch_size * A;
          sp.aux = grp.aux + (size_t)j * minibatch_size * 3;
          sp.stats = sc.stats = stats_all + (int64_t)u * 4;
          int npb = 0, ncb = 0;
          r = twin_fwd_bwd(ctx, *pdesc, LPt, pparams, pg, *cdesc, LCt, cparams, cg, tim, met, sp, sc, minibatch_size,
                           minibatch_size, *hp, psq, &npb, &ncb, s0);
          if (r) return r;
          ctx->bank = 0;
          const BxEmit pe = bx_emit_table(ctx, *pdesc, pparams);
          ctx->bank = 1;
          const BxEmit ce = bx_emit_table(ctx, *cdesc, cparams);
          ctx->bank = 0;
          r = launch_clip_adam2(pparams, pg, pm, pv, np_, psq, npb, met + 8, &pe, cparams, cg, cm, cv, nc_, psq + npb, ncb, met + 9,
                                &ce, *opt_count_io + u + 1, lr_schedule[u], hp->max_grad_norm, hp->adam_b1, hp->adam_b2,
                                hp->adam_eps, s0, sched_dev + 4 * u);
          if (r) return r;
        }
        return RLX_OK;
      }
      // Row buffers.  G > 1 (record source, option "gather_group_rows"): the rows of G consecutive updates are gathered by ONE launch
      // into one of two alternating group buffers and the chains meet once per group (the critic waits for the group's rows, the
      // gather of group g + 2 for the critic's last read of group g).  G == 1: one gather per update into the two row buffers by
      // update parity -- or, for minibatches of at most 8192 rows, each chain gathers its own copy (own_rows; nothing couples the
      // chains between the fork and the final join).
      const int A_act = hp->discrete_actions ? 1 : A;
      int G = 1;
      if (sb[0].rec && !hp->critic_states && ctx->gather_group_rows > minibatch_size) {
        G = ctx->gather_group_rows / minibatch_size;
        if (G > n_upd) G = n_upd;
      }
      MbScratch grp[2] = {sb[0], sb[1]};
      if (G > 1) {
        const int64_t rows = (int64_t)G * minibatch_size;
        for (int b = 0; b < 2; ++b) {
          ctx->bank = b;
          grp[b].mb_x = (float*)scratch(ctx, SL_MB_GROUP_X, (size_t)rows * (O + A_act) * sizeof(float));
          grp[b].aux = (float*)scratch(ctx, SL_MB_GROUP_AUX, (size_t)rows * 3 * sizeof(float));
          ctx->bank = 0;
          if (!grp[b].mb_x || !grp[b].aux) return RLX_ENOMEM;
          grp[b].mb_xc = nullptr;
        }
      }
      const bool own_rows = G == 1 && minibatch_size <= 8192;
      for (int u = 0; u < n_upd; ++u) {
        const int gi = u / G, j = u % G;
        const int par = own_rows ? 0 : (G > 1 ? (gi & 1) : (u & 1));
        const bool group_start = j == 0, group_end = j == G - 1 || u == n_upd - 1;
        float* met = metrics_out + (int64_t)u * 10;
        double* stats = stats_all + (int64_t)u * 4;
        const float* sch = sched_dev + 4 * u;
        if (group_start) {
          if (gi >= 2 && !own_rows) RLX_HIP_TRY(hipStreamWaitEvent(s0, ctx->ev_cdone[par], 0));   // critic done with the rows of group gi - 2
          if (G > 1) {
            const int g = n_upd - u < G ? n_upd - u : G;
            grp[par].mb_a = grp[par].mb_x + (size_t)g * minibatch_size * O;
            r = launch_gather(ctx, states, actions, log_probs, returns, advantages, perm + (int64_t)u * minibatch_size, grp[par],
                              nullptr, nullptr, (int64_t)g * minibatch_size, O, A_act, s0);
          } else {
            r = launch_gather(ctx, states, actions, log_probs, returns, advantages, perm + (int64_t)u * minibatch_size, sb[par],
                              nullptr, nullptr, (int64_t)minibatch_size, O, A_act, s0, hp->critic_states, cdesc->in_dim);
          }
          if (r) return r;
          if (!own_rows) RLX_HIP_TRY(hipEventRecord(ctx->ev_rows[par], s0));
        }
        MbScratch rows_u = sb[par];           // where this update's rows are
        if (G > 1) {
          rows_u.mb_x = grp[par].mb_x + (size_t)j * minibatch_size * O;
          rows_u.mb_a = grp[par].mb_a + (size_t)j * minibatch_size * A_act;
          rows_u.aux = grp[par].aux + (size_t)j * minibatch_size * 3;
          rows_u.mb_xc = nullptr;
        }
        int npb = 0, ncb = 0;
        const int64_t step = *opt_count_io + u + 1;
        MbScratch sp = sb[0];                 // policy: activation / slab arenas of bank 0, rows of this update
        sp.mb_x = rows_u.mb_x; sp.mb_a = rows_u.mb_a; sp.aux = rows_u.aux; sp.stats = stats;
        // the first critic pass starts when the first policy pass has finished its forward half: from then on the two
        // chains stay about half an update apart (nothing joins them before the end of the call)
        r = net_fwd_bwd<true>(ctx, *pdesc, pparams, pg, met, sp, minibatch_size, minibatch_size, *hp, psq, &npb, s0,
                              u == 0 ? ctx->ev_fork : nullptr);
        if (r) return r;
        const BxEmit pe = bx_emit_table(ctx, *pdesc, pparams);
        r = launch_clip_adam(pparams, pg, pm, pv, np_, psq, npb, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1,
                             hp->adam_b2, hp->adam_eps, met + 8, s0, sch, &pe);
        if (r) return r;
        if (u == 0 || (!own_rows && group_start)) RLX_HIP_TRY(hipStreamWaitEvent(st_c, u == 0 ? ctx->ev_fork : ctx->ev_rows[par], 0));
        MbScratch sc = sb[1];                 // critic: arenas of bank 1
        if (own_rows) {
          r = launch_gather(ctx, states, actions, log_probs, returns, advantages, perm + (int64_t)u * minibatch_size, sb[1],
                            nullptr, nullptr, (int64_t)minibatch_size, O, A_act, st_c, hp->critic_states, cdesc->in_dim);
          if (r) return r;
        } else {
          sc.mb_x = rows_u.mb_x; sc.mb_xc = rows_u.mb_xc; sc.mb_a = rows_u.mb_a; sc.aux = rows_u.aux;
        }
        sc.stats = stats;
        ctx->bank = 1;
        r = net_fwd_bwd<false>(ctx, *cdesc, cparams, cg, met, sc, minibatch_size, minibatch_size, *hp, csq, &ncb, st_c);
        const BxEmit ce = bx_emit_table(ctx, *cdesc, cparams);   // (bank 1 still selected)
        ctx->bank = 0;
        if (r) return r;
        r = launch_clip_adam(cparams, cg, cm, cv, nc_, csq, ncb, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1,
                             hp->adam_b2, hp->adam_eps, met + 9, st_c, sch, &ce);
        if (r) return r;
        if (!own_rows && group_end) RLX_HIP_TRY(hipEventRecord(ctx->ev_cdone[par], st_c));
      }
      RLX_HIP_TRY(hipEventRecord(ctx->ev_join, st_c));     // the call's work completes on the main stream
      RLX_HIP_TRY(hipStreamWaitEvent(s0, ctx->ev_join, 0));
      return RLX_OK;
    };
    rc = issue(st);
    if (rc) return rc;
    *opt_count_io += (int64_t)nr_epochs * M;
    if (!ctx->ev_perm_free) RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_perm_free, hipEventDisableTiming));
    RLX_HIP_TRY(hipEventRecord(ctx->ev_perm_free, st));
    ctx->perm_free_recorded = true;
    return RLX_OK;
  }
  // advantage statistics of all E*M minibatches: one workgroup each, fixed summation order (dist.hip)
  rc = dist_adv_sums(advantages, perm, nullptr, n_upd, minibatch_size, minibatch_size, stats_all, st);
  if (rc) return rc;
  RLX_HIP_TRY(hipMemsetAsync(metrics_out, 0, (size_t)n_upd * 10 * sizeof(float), st));
  ctx->bx_keep[0] = ctx->bx_keep[1] = false;   // this schedule lays the weight images out per pass (its Adam launches do not emit)
  for (int u = 0; u < n_upd; ++u) {
    float* met = metrics_out + (int64_t)u * 10;
    int npb = 0, ncb = 0;
    rc = minibatch_core(ctx, *pdesc, pparams, pg, *cdesc, cparams, cg, met, states, actions, log_probs, returns,
                        advantages, perm + (int64_t)u * minibatch_size, minibatch_size, minibatch_size, nullptr, 1, *hp,
                        psq, &npb, csq, &ncb, st, st_c, stats_all + (int64_t)u * 4);
    if (rc) return rc;
    const int64_t step = *opt_count_io + u + 1;
    rc = launch_clip_adam(cparams, cg, cm, cv, nc_, csq, ncb, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1,
                          hp->adam_b2, hp->adam_eps, met + 9, st_c);
    if (rc) return rc;
    rc = launch_clip_adam(pparams, pg, pm, pv, np_, psq, npb, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1,
                          hp->adam_b2, hp->adam_eps, met + 8, st);
    if (rc) return rc;
    if (st_c != st) {  // the next gather overwrites the rows the critic reads
      RLX_HIP_TRY(hipEventRecord(ctx->ev_join, st_c));
      RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
    }
  }
  *opt_count_io += (int64_t)nr_epochs * M;
  if (!ctx->ev_perm_free) RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_perm_free, hipEventDisableTiming));
  RLX_HIP_TRY(hipEventRecord(ctx->ev_perm_free, st));
  ctx->perm_free_recorded = true;
  return RLX_OK;
}

void* rlx_ctx_side_stream(rlx_ctx* ctx) {
  if (!ctx || ctx_side_stream(ctx) != RLX_OK) return nullptr;
  return (void*)ctx->side;
}

// permutation of the GLOBAL index space + restriction to this rank's rows, on `st` (shared by the prefetch and the
// in-line path).  perm: SL_PERM [E * Bg]; lidx: SL_LIDX [n_upd, cap]; counts: SL_COUNTS [n_upd]
static int dist_index_plumbing(rlx_ctx* ctx, uint32_t key_io[2], int nr_epochs, int T, int n_local, int n_global, int env_off,
                               int mb, int scheme, hipStream_t st) {
  const int64_t Bg = (int64_t)T * n_global;
  const int n_upd = (int)(nr_epochs * (Bg / mb));
  const int cap = dist_row_capacity(mb, n_local, n_global);
  int32_t* perm = (int32_t*)scratch(ctx, SL_PERM, (size_t)nr_epochs * Bg * sizeof(int32_t));
  int32_t* lidx = (int32_t*)scratch(ctx, SL_LIDX, (size_t)n_upd * cap * sizeof(int32_t));
  int32_t* counts = (int32_t*)scratch(ctx, SL_COUNTS, (size_t)2 * n_upd * sizeof(int32_t));   // counts | dropped rows
  int32_t* ovf = dist_overflow_slot(ctx);
  if (!perm || !lidx || !counts || !ovf) return RLX_ENOMEM;
  int rc = rlx_permutation_i32(ctx, key_io, perm, nr_epochs, Bg, scheme, st);
  if (rc) return rc;
  return dist_compact(ctx, perm, n_upd, mb, n_local, n_global, env_off, cap, lidx, counts, ovf, counts + n_upd, st);
}

int rlx_ppo_dist_prefetch(rlx_ctx* ctx, const uint32_t key_at_update[2], int nr_epochs, int T, int n_local, int n_global,
                          int env_id_offset, int minibatch_size, int scheme, void* stream) {
  RLX_REQUIRE(ctx && key_at_update && nr_epochs > 0 && T > 0 && n_local > 0 && n_global >= n_local && env_id_offset >= 0 &&
                  env_id_offset + n_local <= n_global && minibatch_size > 0 &&
                  ((int64_t)T * n_global) % minibatch_size == 0,
              RLX_EINVAL, "rlx_ppo_dist_prefetch: bad args");
  if (!ctx->pf_done) RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->pf_done, hipEventDisableTiming));
  ctx->pf_valid = false;
  int rc = ctx_side_stream(ctx);
  if (rc) return rc;
  if (ctx->perm_free_recorded) {
    RLX_HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_perm_free, 0));
  } else {
    RLX_HIP_TRY(hipEventRecord(ctx->ev_fork, (hipStream_t)stream));
    RLX_HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
  }
  uint32_t k[2] = {key_at_update[0], key_at_update[1]};
  rc = dist_index_plumbing(ctx, k, nr_epochs, T, n_local, n_global, env_id_offset, minibatch_size, scheme, ctx->side);
  if (rc) return rc;
  RLX_HIP_TRY(hipEventRecord(ctx->pf_done, ctx->side));
  ctx->pf_key_in[0] = key_at_update[0]; ctx->pf_key_in[1] = key_at_update[1];
  ctx->pf_key_out[0] = k[0]; ctx->pf_key_out[1] = k[1];
  ctx->pf_E = nr_epochs; ctx->pf_B = (int64_t)T * n_global; ctx->pf_scheme = scheme;
  ctx->pf_dist = true;
  ctx->pf_T = T; ctx->pf_nl = n_local; ctx->pf_ng = n_global; ctx->pf_off = env_id_offset; ctx->pf_mb = minibatch_size;
  ctx->pf_valid = true;
  return RLX_OK;
}

int rlx_ppo_update_dist_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, float* pparams, float* pm, float* pv,
                            const rlx_mlp_desc* cdesc, float* cparams, float* cm, float* cv, const float* states,
                            const float* actions, const float* log_probs, const float* returns, const float* advantages,
                            int T, int n_local, int n_global, int env_id_offset, int nr_epochs, int minibatch_size,
                            uint32_t key_io[2], int scheme, int64_t* opt_count_io, const float* lr_schedule,
                            const rlx_ppo_hparams* hp, float* metrics_out, void* stream) {
  if (ctx) ctx->ro_img.valid = false;   // the acting nets' weight images go stale with this call
  // weight images of the two networks persist over the updates of this call (re-emitted by clip + Adam), dropped at its end
  struct BxKeep {
    rlx_ctx* c;
    ~BxKeep() {
      if (!c) return;
      c->bx_keep[0] = c->bx_keep[1] = false;
      bx_release_all(c);
    }
  } bx_keep_scope{ctx};
  if (ctx && ctx->adam_emit && ctx->gemm_bx) ctx->bx_keep[0] = ctx->bx_keep[1] = true;
  RLX_REQUIRE(ctx && pdesc && pparams && pm && pv && cdesc && cparams && cm && cv && states && actions && log_probs &&
                  returns && advantages && key_io && opt_count_io && lr_schedule && hp && metrics_out,
              RLX_EINVAL, "rlx_ppo_update_dist_f32: NULL pointer");
  const int64_t Bg = (int64_t)T * n_global;
  RLX_REQUIRE(T > 0 && n_local > 0 && n_global >= n_local && env_id_offset >= 0 && env_id_offset + n_local <= n_global &&
                  nr_epochs > 0 && minibatch_size > 0 && Bg % minibatch_size == 0,
              RLX_EINVAL, "rlx_ppo_update_dist_f32: global batch (T * n_global) must be a positive multiple of minibatch_size");
  int rc = mlp_check_desc(*pdesc);
  if (rc) return rc;
  rc = mlp_check_desc(*cdesc);
  if (rc) return rc;
  RLX_REQUIRE((hp->critic_states || pdesc->in_dim == cdesc->in_dim) && cdesc->out_dim == 1 &&
                  (hp->discrete_actions ? !pdesc->has_logstd : pdesc->has_logstd),
              RLX_EINVAL, "rlx_ppo_update_dist_f32: policy / critic descriptors do not fit the PPO losses");
  hipStream_t st = (hipStream_t)stream;
  rc = ctx_side_stream(ctx);
  if (rc) return rc;
  hipStream_t st_c = ctx->side;
  const int n_upd = (int)(nr_epochs * (Bg / minibatch_size));
  const int cap = dist_row_capacity(minibatch_size, n_local, n_global);
  const bool collective = dist_active(ctx);
  // ---- index plumbing: prefetched under the rollout, or generated now
  if (ctx->pf_valid && ctx->pf_dist && ctx->pf_key_in[0] == key_io[0] && ctx->pf_key_in[1] == key_io[1] &&
      ctx->pf_E == nr_epochs && ctx->pf_B == Bg && ctx->pf_scheme == scheme && ctx->pf_T == T && ctx->pf_nl == n_local &&
      ctx->pf_ng == n_global && ctx->pf_off == env_id_offset && ctx->pf_mb == minibatch_size) {
    RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->pf_done, 0));
    key_io[0] = ctx->pf_key_out[0];
    key_io[1] = ctx->pf_key_out[1];
  } else {
    if (ctx->pf_valid && ctx->pf_done) RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->pf_done, 0));   // a stale prefetch may still be writing the buffers
    rc = dist_index_plumbing(ctx, key_io, nr_epochs, T, n_local, n_global, env_id_offset, minibatch_size, scheme, st);
    if (rc) return rc;
  }
  ctx->pf_valid = false;
  ctx->pf_dist = false;
  const int32_t* lidx = (const int32_t*)ctx->slots[0][SL_LIDX].ptr;
  const int32_t* counts = (const int32_t*)ctx->slots[0][SL_COUNTS].ptr;
  const int64_t np_ = rlx_mlp_param_count(pdesc), nc_ = rlx_mlp_param_count(cdesc);
  // twin launches (one stream, policy || critic per kernel; twin_shapes_ok): the two gradient vectors are ONE buffer
  // [policy | pad to 16 B | critic], so each update needs one all-reduce instead of two
  const MlpLayout LPt = make_layout(*pdesc), LCt = make_layout(*cdesc);
  TwinImages tim;
  bool twin = twin_shapes_ok(ctx, *pdesc, *cdesc, *hp, cap);
  if (twin) {
    rc = twin_images(ctx, *pdesc, LPt, pparams, *cdesc, LCt, cparams, st, &tim, &twin);
    if (rc) return rc;
  }
  XmaxCall xmax_call(ctx);
  rc = xmax_call.set(states, (int64_t)T * n_local * pdesc->in_dim, hp->critic_states, (int64_t)T * n_local * cdesc->in_dim, st);
  if (rc) return rc;
  const int64_t np4 = (np_ + 3) & ~(int64_t)3;
  float* pg = (float*)scratch(ctx, SL_GRAD_P, (size_t)(twin ? np4 + nc_ : np_) * sizeof(float));
  float* cg = twin ? (pg ? pg + np4 : nullptr) : (float*)scratch(ctx, SL_GRAD_C, (size_t)nc_ * sizeof(float));
  float* psq = (float*)scratch(ctx, SL_NORM, REDUCE_MAX_BLOCKS * sizeof(float));
  float* csq = (float*)scratch(ctx, SL_NORM2, REDUCE_MAX_BLOCKS * sizeof(float));
  double* stats_all = (double*)scratch(ctx, SL_STATS_ALL, (size_t)n_upd * 4 * sizeof(double));
  if (!pg || !cg || !psq || !csq || !stats_all) return RLX_ENOMEM;
  // ---- advantage statistics of every GLOBAL minibatch: local fp64 sums (one workgroup per minibatch, fixed order -- the
  // same kernel rlx_ppo_update_f32 uses), then ONE all-reduce
  const bool whole = n_local == n_global;
  rc = dist_adv_sums(advantages, lidx, whole ? nullptr : counts, n_upd, cap, cap, stats_all, st, whole ? nullptr : counts + n_upd);
  if (rc) return rc;
  if (collective) {
    rc = dist_allreduce(ctx, stats_all, (int64_t)n_upd * 4, 1, st);
    if (rc) return rc;
  }
  if (!whole) {   // rows dropped by ANY rank (slot 3 of the all-reduced records): the same count on every rank
    rc = dist_sum_dropped(ctx, stats_all, n_upd, st);
    if (rc) return rc;
  }
  const int O = pdesc->in_dim, A = pdesc->out_dim, A_act = hp->discrete_actions ? 1 : A;
  MbScratch sb[2];
  for (int b = 0; b < 2; ++b) {
    ctx->bank = b;
    rc = mb_scratch(ctx, *pdesc, *cdesc, cap, &sb[b], hp->critic_states != nullptr);
    ctx->bank = 0;
    if (rc) return rc;
  }
  rc = pack_rollout_rows(ctx, states, actions, log_probs, returns, advantages, (int64_t)T * n_local, O, A_act,
                         hp->critic_states != nullptr, sb, 2, st);
  if (rc) return rc;
  RLX_HIP_TRY(hipMemsetAsync(metrics_out, 0, (size_t)n_upd * 10 * sizeof(float), st));
  if (twin) {
    if (np4 > np_) RLX_HIP_TRY(hipMemsetAsync(pg + np_, 0, (size_t)(np4 - np_) * sizeof(float), st));
    // the rows of G consecutive updates by ONE gather launch, as in rlx_ppo_update_f32 (lidx is [n_upd, cap]: adjacent ranges;
    // the per-update valid counts go along as an array) -- needs the record source (k_gather_rec)
    const int grp_rows = ctx->gather_group_rows > 32768 ? ctx->gather_group_rows : 32768;
    const int G = !sb[0].rec ? 1 : (n_upd < grp_rows / cap ? n_upd : (grp_rows / cap > 0 ? grp_rows / cap : 1));
    MbScratch grp = sb[0];
    if (G > 1) {
      const int64_t rows = (int64_t)G * cap;
      grp.mb_x = (float*)scratch(ctx, SL_MB_GROUP_X, (size_t)rows * (O + A_act) * sizeof(float));
      grp.aux = (float*)scratch(ctx, SL_MB_GROUP_AUX, (size_t)rows * 3 * sizeof(float));
      if (!grp.mb_x || !grp.aux) return RLX_ENOMEM;
      grp.mb_xc = nullptr;
    }
    for (int u = 0; u < n_upd; ++u) {
      float* met = metrics_out + (int64_t)u * 10;
      const int32_t* cnt_u = whole ? nullptr : counts + u;
      const int j = u % G;
      if (j == 0) {
        const int g = n_upd - u < G ? n_upd - u : G;
        grp.mb_a = grp.mb_x + (size_t)g * cap * O;
        rc = launch_gather(ctx, states, actions, log_probs, returns, advantages, lidx + (int64_t)u * cap, grp, nullptr, cnt_u,
                           (int64_t)g * cap, O, A_act, st, nullptr, 0, G > 1 ? cap : 0);
        if (rc) return rc;
      }
      MbScratch sp = sb[0], sc = sb[1];
      sp.mb_x = grp.mb_x + (size_t)j * cap * O;
      sp.mb_a = grp.mb_a + (size_t)j * cap * A_act;
      sp.aux = grp.aux + (size_t)j * cap * 3;
      sp.stats = sc.stats = stats_all + (int64_t)u * 4;
      sp.valid_rows = sc.valid_rows = cnt_u;
      int npb = 0, ncb = 0;
      rc = twin_fwd_bwd(ctx, *pdesc, LPt, pparams, pg, *cdesc, LCt, cparams, cg, tim, met, sp, sc, cap, minibatch_size, *hp, psq,
                        &npb, &ncb, st);
      if (rc) return rc;
      const float *pp_ = psq, *cp_ = psq + npb;
      if (collective) {
        rc = dist_allreduce(ctx, pg, np4 + nc_, 0, st);
        if (rc) return rc;
        rc = launch_sumsq_partials2(pg, np_, psq, &npb, cg, nc_, csq, &ncb, st);   // norms of the REDUCED gradients
        if (rc) return rc;
        cp_ = csq;
      }
      ctx->bank = 0;
      const BxEmit pe = bx_emit_table(ctx, *pdesc, pparams);
      ctx->bank = 1;
      const BxEmit ce = bx_emit_table(ctx, *cdesc, cparams);
      ctx->bank = 0;
      rc = launch_clip_adam2(pparams, pg, pm, pv, np_, pp_, npb, met + 8, &pe, cparams, cg, cm, cv, nc_, cp_, ncb, met + 9, &ce,
                             *opt_count_io + u + 1, lr_schedule[u], hp->max_grad_norm, hp->adam_b1, hp->adam_b2, hp->adam_eps, st);
      if (rc) return rc;
    }
  } else {
  // the side stream starts after everything queued on `stream` so far (statistics, index plumbing)
  RLX_HIP_TRY(hipEventRecord(ctx->ev_join, st));
  RLX_HIP_TRY(hipStreamWaitEvent(st_c, ctx->ev_join, 0));
  // The rows of G consecutive updates are gathered by ONE launch into one of two alternating group buffers (as in the twin
  // schedule above): the chains then meet once per group instead of once per update -- the critic waits for the group's rows, the
  // gather of group g + 2 for the critic's last read of group g.  (G = 1 without the record source or with critic-only rows.)
  const int grp_rows2 = ctx->gather_group_rows > 32768 ? ctx->gather_group_rows : 32768;
  const int G = (!sb[0].rec || hp->critic_states) ? 1 : (n_upd < grp_rows2 / cap ? n_upd : (grp_rows2 / cap > 0 ? grp_rows2 / cap : 1));
  MbScratch grp[2] = {sb[0], sb[1]};
  if (G > 1) {
    const int64_t rows = (int64_t)G * cap;
    for (int b = 0; b < 2; ++b) {
      ctx->bank = b;
      grp[b].mb_x = (float*)scratch(ctx, SL_MB_GROUP_X, (size_t)rows * (O + A_act) * sizeof(float));
      grp[b].aux = (float*)scratch(ctx, SL_MB_GROUP_AUX, (size_t)rows * 3 * sizeof(float));
      ctx->bank = 0;
      if (!grp[b].mb_x || !grp[b].aux) return RLX_ENOMEM;
      grp[b].mb_xc = nullptr;
    }
  }
  for (int u = 0; u < n_upd; ++u) {
    const int gi = u / G, j = u % G;
    const int par = G > 1 ? (gi & 1) : (u & 1);
    const bool group_start = j == 0, group_end = j == G - 1 || u == n_upd - 1;
    float* met = metrics_out + (int64_t)u * 10;
    double* stats = stats_all + (int64_t)u * 4;
    const int32_t* cnt_u = whole ? nullptr : counts + u;   // a rank that holds every row has no padding
    if (group_start) {
      if (gi >= 2) RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->ev_cdone[par], 0));   // the critic is done with the rows of group gi - 2
      if (G > 1) {
        const int g = n_upd - u < G ? n_upd - u : G;
        grp[par].mb_a = grp[par].mb_x + (size_t)g * cap * O;
        rc = launch_gather(ctx, states, actions, log_probs, returns, advantages, lidx + (int64_t)u * cap, grp[par], nullptr, cnt_u,
                           (int64_t)g * cap, O, A_act, st, nullptr, 0, cap);
      } else {
        rc = launch_gather(ctx, states, actions, log_probs, returns, advantages, lidx + (int64_t)u * cap, sb[par],
                           nullptr, cnt_u, (int64_t)cap, O, A_act, st, hp->critic_states, cdesc->in_dim);
      }
      if (rc) return rc;
      RLX_HIP_TRY(hipEventRecord(ctx->ev_rows[par], st));
    }
    MbScratch rows_u = sb[par];               // where this update's rows are
    if (G > 1) {
      rows_u.mb_x = grp[par].mb_x + (size_t)j * cap * O;
      rows_u.mb_a = grp[par].mb_a + (size_t)j * cap * A_act;
      rows_u.aux = grp[par].aux + (size_t)j * cap * 3;
      rows_u.mb_xc = nullptr;
    }
    int npb = 0, ncb = 0;
    const int64_t step = *opt_count_io + u + 1;
    MbScratch sp = sb[0];
    sp.mb_x = rows_u.mb_x; sp.mb_a = rows_u.mb_a; sp.aux = rows_u.aux; sp.stats = stats; sp.valid_rows = cnt_u;
    rc = net_fwd_bwd<true>(ctx, *pdesc, pparams, pg, met, sp, cap, minibatch_size, *hp, psq, &npb, st,
                           u == 0 ? ctx->ev_fork : nullptr);
    if (rc) return rc;
    const BxEmit pe = bx_emit_table(ctx, *pdesc, pparams);
    if (collective) {
      rc = dist_allreduce(ctx, pg, np_, 0, st);
      if (rc) return rc;
      rc = clip_adam_step(ctx, pparams, pg, pm, pv, np_, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1,
                          hp->adam_b2, hp->adam_eps, met + 8, st, &pe);
    } else {
      rc = launch_clip_adam(pparams, pg, pm, pv, np_, psq, npb, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1,
                            hp->adam_b2, hp->adam_eps, met + 8, st, nullptr, &pe);
    }
    if (rc) return rc;
    if (group_start) RLX_HIP_TRY(hipStreamWaitEvent(st_c, u == 0 ? ctx->ev_fork : ctx->ev_rows[par], 0));
    MbScratch sc = sb[1];
    sc.mb_x = rows_u.mb_x; sc.mb_xc = rows_u.mb_xc; sc.mb_a = rows_u.mb_a; sc.aux = rows_u.aux; sc.stats = stats; sc.valid_rows = cnt_u;
    ctx->bank = 1;
    rc = net_fwd_bwd<false>(ctx, *cdesc, cparams, cg, met, sc, cap, minibatch_size, *hp, csq, &ncb, st_c);
    const BxEmit ce = bx_emit_table(ctx, *cdesc, cparams);   // (bank 1 still selected)
    ctx->bank = 0;
    if (rc) return rc;
    if (collective) {
      rc = dist_allreduce(ctx, cg, nc_, 0, st_c);
      if (rc) return rc;
      rc = clip_adam_step(ctx, cparams, cg, cm, cv, nc_, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1,
                          hp->adam_b2, hp->adam_eps, met + 9, st_c, &ce);
    } else {
      rc = launch_clip_adam(cparams, cg, cm, cv, nc_, csq, ncb, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1,
                            hp->adam_b2, hp->adam_eps, met + 9, st_c, nullptr, &ce);
    }
    if (rc) return rc;
    if (group_end) RLX_HIP_TRY(hipEventRecord(ctx->ev_cdone[par], st_c));
  }
  RLX_HIP_TRY(hipEventRecord(ctx->ev_join, st_c));
  RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
  }
  if (collective) {
    // per-update metrics: partial sums over this rank's rows -> ONE all-reduce per iteration
    rc = dist_mask_metrics(metrics_out, n_upd, ctx->rank, hp->discrete_actions, st);
    if (rc) return rc;
    rc = dist_allreduce(ctx, metrics_out, (int64_t)n_upd * 10, 0, st);
    if (rc) return rc;
  }
  *opt_count_io += n_upd;
  if (!ctx->ev_perm_free) RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_perm_free, hipEventDisableTiming));
  RLX_HIP_TRY(hipEventRecord(ctx->ev_perm_free, st));
  ctx->perm_free_recorded = true;
  return RLX_OK;
}

int rlx_actor_critic_fwd_sample_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams,
                                    const rlx_mlp_desc* cdesc, const float* cparams, const float* obs,
                                    const float* critic_obs, uint32_t key_io[2], int scheme, float* action, float* processed,
                                    float* value,
                                    float* logp, float* states_row, int N, int clip_and_rescale, const float* act_low,
                                    const float* act_high, int env_id_offset, int N_global, void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && cdesc && cparams && obs && key_io && action && value && logp, RLX_EINVAL,
              "rlx_actor_critic_fwd_sample_f32: NULL pointer");
  RLX_REQUIRE(N > 0 && N_global >= N && env_id_offset >= 0, RLX_EINVAL, "rlx_actor_critic_fwd_sample_f32: bad sizes");
  RLX_REQUIRE(!clip_and_rescale || (act_low && act_high), RLX_EINVAL,
              "rlx_actor_critic_fwd_sample_f32: clip_and_rescale needs act_low/act_high");
  RLX_REQUIRE(pdesc->has_logstd, RLX_EINVAL, "rlx_actor_critic_fwd_sample_f32: policy desc needs has_logstd");
  hipStream_t st = (hipStream_t)stream;
  const int A = pdesc->out_dim;
  float* mean = (float*)scratch(ctx, SL_MEAN, (size_t)N * A * sizeof(float));
  if (!mean) return RLX_ENOMEM;
  int rc = rlx_mlp_fwd_f32(ctx, pdesc, pparams, obs, mean, N, stream);
  if (rc) return rc;
  RLX_REQUIRE(critic_obs || pdesc->in_dim == cdesc->in_dim, RLX_EINVAL,
              "rlx_actor_critic_fwd_sample_f32: a critic with its own observation width needs critic_obs");
  rc = rlx_mlp_fwd_f32(ctx, cdesc, cparams, critic_obs ? critic_obs : obs, value, N, stream);
  if (rc) return rc;
  uint32_t ks[4];
  split_host(key_io, ks, 2, scheme);  // key, subkey = split(key)
  key_io[0] = ks[0];
  key_io[1] = ks[1];
  const MlpLayout L = make_layout(*pdesc);
  return ppo_sample(mean, pparams + L.logstd, ks[2], ks[3], scheme, action, processed, logp, obs, states_row, N, A,
                    pdesc->in_dim, clip_and_rescale, act_low, act_high, env_id_offset, N_global, st);
}

int rlx_actor_critic_fwd_sample_discrete_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams,
                                             const rlx_mlp_desc* cdesc, const float* cparams, const float* obs,
                                             uint32_t key_io[2], int scheme, float* action, float* value, float* logp,
                                             float* states_row, int N, int env_id_offset, int N_global, int deterministic,
                                             void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && cdesc && cparams && obs && key_io && action && value && logp, RLX_EINVAL,
              "rlx_actor_critic_fwd_sample_discrete_f32: NULL pointer");
  RLX_REQUIRE(N > 0 && N_global >= N && env_id_offset >= 0, RLX_EINVAL, "rlx_actor_critic_fwd_sample_discrete_f32: bad sizes");
  RLX_REQUIRE(!pdesc->has_logstd && pdesc->out_dim >= 2 && pdesc->out_dim <= 8, RLX_EINVAL,
              "rlx_actor_critic_fwd_sample_discrete_f32: Categorical policy = no logstd, 2..8 actions");
  hipStream_t st = (hipStream_t)stream;
  const int A = pdesc->out_dim;
  float* logits = (float*)scratch(ctx, SL_MEAN, (size_t)N * A * sizeof(float));
  if (!logits) return RLX_ENOMEM;
  int rc = rlx_mlp_fwd_f32(ctx, pdesc, pparams, obs, logits, N, stream);
  if (rc) return rc;
  rc = rlx_mlp_fwd_f32(ctx, cdesc, cparams, obs, value, N, stream);
  if (rc) return rc;
  uint32_t ks[4] = {key_io[0], key_io[1], 0u, 0u};
  if (!deterministic) {
    split_host(key_io, ks, 2, scheme);  // key, subkey = split(key)
    key_io[0] = ks[0];
    key_io[1] = ks[1];
  }
  hipLaunchKernelGGL(k_sample_categorical, dim3(div_up(N, 256)), dim3(256), 0, st, logits, ks[2], ks[3], scheme, action, logp,
                     obs, states_row, N, A, pdesc->in_dim, env_id_offset, N_global, deterministic);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

}  // extern "C"
