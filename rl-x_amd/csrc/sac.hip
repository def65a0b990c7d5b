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
#include <cstring>
#include "mlp.h"
#include "adam_body.h"
#include "gemm_bx.h"
#include "dist.h"
#include "ln_kernels.h"
#include "sac_sample.h"

namespace rlx {

constexpr int SAC_HEAD_ROWS = 16;   // rows per workgroup of the head kernels: 256 workgroups at B = 4096 (64 rows left 3/4 of the CUs idle)

// per-call values of an update that the kernels read from device memory (the Adam schedules of the three optimizers, the key)
struct SacConsts { float sched[12]; uint32_t key[2]; };

// Xc_cur = [s | a | 0], Xc_next[:, :O] = s', Xc_pi[:, :O] = s (action columns are filled by k_sac_sample)
// cdst (optional): the update's per-call values ride along -- one launch less in front of the chains (eager issue only: a
// replayed graph gets them from k_sac_consts, which stays outside the capture)
__global__ __launch_bounds__(256) void k_sac_concat(const float* __restrict__ s, const float* __restrict__ s2,
                                                    const float* __restrict__ a, float* __restrict__ xc_cur,
                                                    float* __restrict__ xc_next, float* __restrict__ xc_pi, int64_t B,
                                                    int O, int A, int ld, SacConsts cval = SacConsts(), SacConsts* cdst = nullptr) {
  if (cdst && blockIdx.x == 0 && threadIdx.x == 0) *cdst = cval;
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

// thread geometry of the per-(row, action dim) kernels: AP = lanes per row (power of two >= A, at most 64; wider action vectors
// loop), 256 / AP rows per workgroup.  (One thread per ROW was 16 workgroups at B = 4096 and A serial erfinv/tanh/log chains.)
static inline int sac_lanes_per_row(int A) {
  int ap = 1;
  while (ap < A && ap < 64) ap <<= 1;
  return ap;
}

// tanh-Gaussian sample from head output [B, 2A] = (mean | raw log_std):
//   u = mean + exp(clip(log_std)) * eps, a = tanh(u), logp = sum(-eps^2/2 - log(2pi)/2 - log_std - log(1 - a^2 + 1e-6))
// mode 0: acting (noise = normal(subkey, [N_global, A]) rows [row_off, row_off+B), like get_action)
// mode 1/2: update; sample i uses the per-sample key split(key, 2B+1)[mode + 2i]  (sac.py:196-197)
// One thread per (row, action dim); the log-prob terms of a row meet in LDS and are added in index order by one lane (the
// same order as a serial loop over the action dims).  Dynamic LDS: (256 / AP) * A floats.
__global__ __launch_bounds__(256) void k_sac_sample(const float* __restrict__ head, SacSampleArgs sa, int64_t B, int AP) {
  extern __shared__ float s_term[];   // [rows per block][A]
  uint32_t k0 = sa.k0, k1 = sa.k1;
  if (sa.key_dev) { k0 = sa.key_dev[0]; k1 = sa.key_dev[1]; }   // the update's key lives in device memory (replayed graphs)
  const int A = sa.A;
  const int rpb = 256 / AP;
  const int rl = threadIdx.x / AP, jl = threadIdx.x - rl * AP;
  const int64_t i = (int64_t)blockIdx.x * rpb + rl;
  if (i < B) {
    uint32_t s0, s1;
    // update modes: row_off / N_global = this rank's first row / the size of the (global) batch the keys are split for
    sac_row_key(sa, k0, k1, i, s0, s1);
    for (int j = jl; j < A; j += AP) s_term[rl * A + j] = sac_sample_elem(sa, s0, s1, i, j, head[i * 2 * A + j], head[i * 2 * A + A + j]);
  }
  if (!sa.logp) return;
  __syncthreads();
  if (i < B && jl == 0) {
    float lp = 0.f;
    for (int j = 0; j < A; ++j) lp += s_term[rl * A + j];
    sa.logp[i] = lp;
  }
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
                                                         float* __restrict__ part, int64_t B, float gamma, int64_t Bg) {
  __shared__ float s_buf[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float alpha = expf(log_alpha[0]);
  float ql = 0.f;
  if (i < B) {
    const float y = rew[i] + gamma * (1.f - term[i]) * (fminf(qt0[i], qt1[i]) - alpha * logp_n[i]);
    const float e0 = q0[i] - y, e1 = q1[i] - y;
    dq0[i] = e0 / (float)Bg;                 // Bg: size of the batch the mean runs over (= B on one GPU)
    dq1[i] = e1 / (float)Bg;
    ql = 0.5f * (e0 * e0 + e1 * e1);
  }
  ql = block_sum256(ql, s_buf);
  if (threadIdx.x == 0) part[blockIdx.x] = ql;
}

// min over the two critics on (s, pi(s)); d(-min_q)/dq_k = -[k == argmin] / B;  sums of min_q, logp
__global__ __launch_bounds__(256) void k_sac_policy_seed(const float* __restrict__ qa0, const float* __restrict__ qa1,
                                                         const float* __restrict__ logp_c, float* __restrict__ d0,
                                                         float* __restrict__ d1, float* __restrict__ part, int nblk,
                                                         int64_t B, int64_t Bg) {
  __shared__ float s_buf[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float mq = 0.f, lp = 0.f;
  if (i < B) {
    const bool sel0 = qa0[i] <= qa1[i];
    d0[i] = sel0 ? -1.f / (float)Bg : 0.f;
    d1[i] = sel0 ? 0.f : -1.f / (float)Bg;
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
                                                         int scheme, float* __restrict__ d_out, int64_t B, int A, int AP,
                                                         float ls_min, float ls_max,
                                                         const float* __restrict__ eps_inject = nullptr, int schedule = 0,
                                                         const uint32_t* __restrict__ key_dev = nullptr, int64_t row_off = 0,
                                                         int64_t Bg = 0) {
  // one thread per (row, action dim), like k_sac_sample
  if (key_dev) { k0 = key_dev[0]; k1 = key_dev[1]; }
  const int rpb = 256 / AP;
  const int rl = threadIdx.x / AP, jl = threadIdx.x - rl * AP;
  const int64_t i = (int64_t)blockIdx.x * rpb + rl;
  if (i >= B) return;
  const float alpha = expf(log_alpha[0]);
  uint32_t s0, s1;
  if (Bg == 0) Bg = B;
  split_key_at(k0, k1, sac_key_index(2, i + row_off, Bg, schedule), sac_key_count(Bg, schedule), scheme, s0, s1);
  const float invB = 1.0f / (float)Bg;
  for (int j = jl; j < A; j += AP) {
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

// data parallel: this rank's three loss sums (q_loss, min_q, logp) -> out[0..2], in the layout sac_finalize_body reads with nb = 1
__global__ void k_sac_loss_sums(const float* __restrict__ part_c, const float* __restrict__ part_p, int nb, float* __restrict__ out) {
  float ql = 0.f, mq = 0.f, lp = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) { ql += part_c[i]; mq += part_p[i]; lp += part_p[nb + i]; }
  ql = wave_sum(ql); mq = wave_sum(mq); lp = wave_sum(lp);
  if (threadIdx.x == 0) { out[0] = ql; out[1] = mq; out[2] = lp; }
}

// metrics (means) + gradient of log_alpha from the partial sums, and -- the coefficient being ONE parameter -- its plain Adam
// step right here (same arithmetic as k_clip_adam without clipping; sched = DEVICE {lr, 1 - b1^t, 1 - b2^t})
//   part_c[nb]: q_loss sums; part_p[2*nb]: min_q sums, logp sums
struct SacFinalize {
  const float *part_c, *part_p;
  int nb;
  float *log_alpha, *g_alpha, *metrics;
  int64_t B;
  float target_entropy;
  float *am, *av;
  const float* sched;
};
__device__ __forceinline__ void sac_finalize_body(const float* __restrict__ part_c, const float* __restrict__ part_p, int nb,
                               float* __restrict__ log_alpha, float* __restrict__ g_alpha,
                               float* __restrict__ metrics, int64_t B, float target_entropy, float* __restrict__ am,
                               float* __restrict__ av, const float* __restrict__ sched, float b1, float b2, float eps) {
  if (threadIdx.x >= 64) return;
  float ql = 0.f, mq = 0.f, lp = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) { ql += part_c[i]; mq += part_p[i]; lp += part_p[nb + i]; }
  ql = wave_sum(ql); mq = wave_sum(mq); lp = wave_sum(lp);
  if (threadIdx.x == 0) {
    const float invB = 1.0f / (float)B;
    const float la = log_alpha[0];
    const float alpha = expf(la);
    const float mean_lp = lp * invB, mean_q = mq * invB;
    const float entropy = -mean_lp;
    const float ga = alpha * (entropy - target_entropy);   // d mean(alpha_g (entropy - target)) / d log_alpha
    g_alpha[0] = ga;
    metrics[0] = ql * invB;                               // loss/q_loss
    metrics[1] = alpha * mean_lp - mean_q;                // loss/policy_loss
    metrics[2] = alpha * (entropy - target_entropy);      // loss/entropy_loss
    metrics[3] = entropy;                                 // entropy/entropy
    metrics[4] = alpha;                                   // entropy/alpha
    metrics[5] = mean_q;                                  // q_value/q_value
    if (sched) {
      metrics[8] = sqrtf(ga * ga);                        // gradient norm of the one-parameter optimizer
      const float mi = b1 * am[0] + (1.f - b1) * ga;
      const float vi = b2 * av[0] + (1.f - b2) * ga * ga;
      am[0] = mi;
      av[0] = vi;
      const float mhat = mi / sched[1];
      const float vhat = vi / sched[2];
      log_alpha[0] = la - sched[0] * (mhat / (sqrtf(vhat) + eps));
    }
  }
}
// The three optimizer steps of one update in ONE launch (they read disjoint gradients and write disjoint parameters): block 0 =
// metrics + the entropy coefficient's step (sac_finalize_body), blocks [1, 1 + nb_p) the policy's Adam step, the rest the critics'
// (with the Polyak update of the targets).  nb_p / nb_q = the jobs' tile + rest blocks (adam_body.h).
__global__ __launch_bounds__(256) void k_sac_optimizers(SacFinalize F, AdamJob P, AdamJob Q, int nb_p, int nb_q, float b1,
                                                        float b2, float eps) {
  __shared__ float s_buf[4];
  __shared__ __attribute__((aligned(16))) uint16_t s_tile[ADAM_LDS_HALVES];
  const int b = blockIdx.x;
  if (b == 0) {
    sac_finalize_body(F.part_c, F.part_p, F.nb, F.log_alpha, F.g_alpha, F.metrics, F.B, F.target_entropy, F.am, F.av, F.sched,
                      b1, b2, eps);
  } else if (b <= nb_p) {
    clip_adam_job(P, b - 1, b1, b2, eps, s_buf, s_tile);
  } else {
    clip_adam_job(Q, b - 1 - nb_p, b1, b2, eps, s_buf, s_tile);
  }
}

// head backward: dZ_last = (d_out @ W^T) * act'(H) in place over H; optional per-block partials of dW_head[K, OD] and
// db_head[OD]  (layout [block][K*OD + OD]).  SAC_HEAD_ROWS rows per workgroup.
// k_head_bwd (K <= 256): thread <-> hidden column k with the block's 16 activations of that column in registers (coalesced
// loads, no LDS round trip for H); W in LDS, d_out transposed in LDS and read as broadcast 16-B words.  Every output is the
// same ascending fmaf chain as in the generic kernel below.  The dW tile goes back through LDS for coalesced stores.
// RLX_HB_NOPACK (bisection of docs/PACKED_F32_HAZARD.md, vectorized builds only): bit 1 keeps the SLP vectorizer from pairing
// the dz chains into v_pk_fma_f32, bit 2 from pairing the epilogue's multiplies into v_pk_mul_f32
#ifndef RLX_HB_NOPACK
#define RLX_HB_NOPACK 0
#endif
#define RLX_HB_FENCE(BIT, V) if (RLX_HB_NOPACK & BIT) asm volatile("" : "+v"(V));
__global__ __launch_bounds__(256) void k_head_bwd(float* __restrict__ H, const float* __restrict__ W,
                                                  const float* __restrict__ d_out, float* __restrict__ partials,
                                                  int64_t M, int K, int OD, int act, Twin tw) {
  static_assert(SAC_HEAD_ROWS == 16, "register tile below is 16 rows");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.y) {   // twin launch: {H, W, d_out, partials} of the second net
    H = const_cast<float*>(static_cast<const float*>(tw.p[0]));
    W = static_cast<const float*>(tw.p[1]);
    d_out = static_cast<const float*>(tw.p[2]);
    partials = const_cast<float*>(static_cast<const float*>(tw.p[3]));
  }
  float* Dt = smem;                       // [OD][16]
  float* Ws = smem + 16 * OD;             // [K][OD]; later the dW tile
  const int t = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * SAC_HEAD_ROWS;
  const bool kv = t < K;
  float h[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) h[r] = (kv && r0 + r < M) ? H[(r0 + r) * K + t] : 0.f;
  // (all global loads of the tile in flight before the first LDS store: one round trip instead of one per unrolled group)
  constexpr int WB = 36;
  const int KO = K * OD;
  float wreg[WB], dreg[4];
#pragma unroll
  for (int j = 0; j < WB; ++j) {
    const int i = t + 256 * j;
    wreg[j] = i < KO ? W[i] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = t + 256 * j;
    dreg[j] = (i < 16 * OD && r0 + i / OD < M) ? d_out[r0 * OD + i] : 0.f;
  }
  for (int base = 0; base < KO; base += 256 * WB) {
    if (base) {
#pragma unroll
      for (int j = 0; j < WB; ++j) {
        const int i = base + t + 256 * j;
        wreg[j] = i < KO ? W[i] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < WB; ++j) {
      const int i = base + t + 256 * j;
      if (i < KO) Ws[i] = wreg[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = t + 256 * j;
    if (i < 16 * OD) {
      const int r = i / OD, a = i - r * OD;
      Dt[a * 16 + r] = dreg[j];
    }
  }
  __syncthreads();
  if (kv) {
    float dz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[r] = 0.f;
    for (int a = 0; a < OD; ++a) {
      const float w = Ws[t * OD + a];
      const float4* d4 = reinterpret_cast<const float4*>(Dt + a * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 d = d4[q];
        dz[4 * q + 0] = fmaf(d.x, w, dz[4 * q + 0]);
        RLX_HB_FENCE(1, dz[4 * q + 0])
        dz[4 * q + 1] = fmaf(d.y, w, dz[4 * q + 1]);
        RLX_HB_FENCE(1, dz[4 * q + 1])
        dz[4 * q + 2] = fmaf(d.z, w, dz[4 * q + 2]);
        RLX_HB_FENCE(1, dz[4 * q + 2])
        dz[4 * q + 3] = fmaf(d.w, w, dz[4 * q + 3]);
        RLX_HB_FENCE(1, dz[4 * q + 3])
      }
      // bit 4: 32 idle cycles between this output column's VALU work and the next column's LDS reads (whose destination registers
      // the register allocator recycles from operands the packed FMAs above have just read); bit 16: the same compiler barrier
      // without the wait states (control)
      if (RLX_HB_NOPACK & 4) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      if (RLX_HB_NOPACK & 16) asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float o = dz[r] * act_grad_from_out(h[r], act);
      RLX_HB_FENCE(2, o)
      if (r0 + r < M) H[(r0 + r) * K + t] = o;
    }
  }
  if (!partials) return;
  if (kv) {   // thread k reads and writes row k of Ws only
    for (int a = 0; a < OD; ++a) {
      const float4* d4 = reinterpret_cast<const float4*>(Dt + a * 16);
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 d = d4[q];
        acc = fmaf(h[4 * q + 0], d.x, acc);
        acc = fmaf(h[4 * q + 1], d.y, acc);
        acc = fmaf(h[4 * q + 2], d.z, acc);
        acc = fmaf(h[4 * q + 3], d.w, acc);
      }
      Ws[t * OD + a] = acc;
    }
  }
  __syncthreads();
  float* pw = partials + (int64_t)blockIdx.x * (K * OD + OD);
  for (int i = t; i < K * OD; i += 256) pw[i] = Ws[i];
  if (t < OD) {
    float sb = 0.f;
    for (int r = 0; r < 16; ++r) sb += Dt[t * 16 + r];
    pw[K * OD + t] = sb;
  }
}

// any K (hidden width above 256): the block's activations staged in LDS
__global__ __launch_bounds__(256) void k_head_bwd_wide(float* __restrict__ H, const float* __restrict__ W,
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

// replay sample: out[i] = ring[idx1[i], idx2[i]].  One wave per sampled transition (its ring slot is looked up once; the
// two observation rows are copied as 16-B words when vec, i.e. O % 4 == 0 and 16-B aligned bases).
__global__ __launch_bounds__(256) void k_replay_gather(const float* __restrict__ r_s, const float* __restrict__ r_s2,
                                                       const float* __restrict__ r_a, const float* __restrict__ r_r,
                                                       const float* __restrict__ r_t, const int32_t* __restrict__ idx1,
                                                       const int32_t* __restrict__ idx2, int N, int O, int A, int64_t B,
                                                       float* __restrict__ s, float* __restrict__ s2,
                                                       float* __restrict__ a, float* __restrict__ r,
                                                       float* __restrict__ tm, int vec) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < B; i += nw) {
    const int64_t src = (int64_t)idx1[i] * N + idx2[i];
    if (vec) {
      const float4* p1 = reinterpret_cast<const float4*>(r_s + src * O);
      const float4* p2 = reinterpret_cast<const float4*>(r_s2 + src * O);
      float4* q1 = reinterpret_cast<float4*>(s + i * O);
      float4* q2 = reinterpret_cast<float4*>(s2 + i * O);
      for (int c = lane; c < (O >> 2); c += 64) {
        q1[c] = p1[c];
        q2[c] = p2[c];
      }
    } else {
      for (int c = lane; c < O; c += 64) {
        s[i * O + c] = r_s[src * O + c];
        s2[i * O + c] = r_s2[src * O + c];
      }
    }
    for (int c = lane; c < A; c += 64) a[i * A + c] = r_a[src * A + c];
    if (lane == 0) {
      r[i] = r_r[src];
      tm[i] = r_t[src];
    }
  }
}

// k_replay_gather + k_sac_concat in one launch (the update's optional ring source, rlx_sac_hparams::ring_*): a wave per sampled
// transition copies it into the batch arrays AND lays out the critics' input rows
//   xc_cur = [s | a | 0], xc_next = [s' | 0], xc_pi = [s | 0]   (row stride ld; action columns of the last two: k_sac_sample)
__global__ __launch_bounds__(256) void k_sac_gather_concat(const float* __restrict__ r_s, const float* __restrict__ r_s2,
                                                           const float* __restrict__ r_a, const float* __restrict__ r_r,
                                                           const float* __restrict__ r_t, const int32_t* __restrict__ idx1,
                                                           const int32_t* __restrict__ idx2, int N, int O, int A, int64_t B,
                                                           float* __restrict__ s, float* __restrict__ s2, float* __restrict__ a,
                                                           float* __restrict__ r, float* __restrict__ tm,
                                                           float* __restrict__ xc_cur, float* __restrict__ xc_next,
                                                           float* __restrict__ xc_pi, int ld, int vec, SacConsts cval,
                                                           SacConsts* cdst) {
  if (cdst && blockIdx.x == 0 && threadIdx.x == 0) *cdst = cval;
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < B; i += nw) {
    const int64_t src = (int64_t)idx1[i] * N + idx2[i];
    if (vec) {   // O % 4 == 0, ld % 4 == 0, 16-byte aligned bases
      const float4* p1 = reinterpret_cast<const float4*>(r_s + src * O);
      const float4* p2 = reinterpret_cast<const float4*>(r_s2 + src * O);
      float4* q1 = reinterpret_cast<float4*>(s + i * O);
      float4* q2 = reinterpret_cast<float4*>(s2 + i * O);
      float4* c1 = reinterpret_cast<float4*>(xc_cur + i * ld);
      float4* c2 = reinterpret_cast<float4*>(xc_next + i * ld);
      float4* c3 = reinterpret_cast<float4*>(xc_pi + i * ld);
      for (int c = lane; c < (O >> 2); c += 64) {
        const float4 v1 = p1[c], v2 = p2[c];
        if (s) {      // (NULL: the caller does not want the gathered observation rows themselves, only the critics' input rows)
          q1[c] = v1;
          q2[c] = v2;
        }
        c1[c] = v1;
        c3[c] = v1;
        c2[c] = v2;
      }
    } else {
      for (int c = lane; c < O; c += 64) {
        const float v1 = r_s[src * O + c], v2 = r_s2[src * O + c];
        if (s) {
          s[i * O + c] = v1;
          s2[i * O + c] = v2;
        }
        xc_cur[i * ld + c] = v1;
        xc_pi[i * ld + c] = v1;
        xc_next[i * ld + c] = v2;
      }
    }
    for (int c = lane; c < ld - O; c += 64) {   // action columns and the zero padding behind them
      float av = 0.f;
      if (c < A) {
        av = r_a[src * A + c];
        a[i * A + c] = av;
      }
      xc_cur[i * ld + O + c] = av;
      xc_next[i * ld + O + c] = 0.f;
      xc_pi[i * ld + O + c] = 0.f;
    }
    if (lane == 0) {
      r[i] = r_r[src];
      tm[i] = r_t[src];
    }
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

// tw (optional, K <= 256): {h_last, head W, d_out, partials} of a second net of the same shape, same launch
static int head_bwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, float* h_last,
                    const float* d_out, float* partials, int64_t M, hipStream_t st, const Twin* tw = nullptr) {
  const int K = L.head.in, OD = L.head.out;
  static AttrOnce lds_opt_in;
  if (!lds_opt_in.done()) {
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_bwd_wide), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    150 * 1024));
    lds_opt_in.mark();
  }
  if (K <= 256) {
    const size_t lds = ((size_t)SAC_HEAD_ROWS * OD + (size_t)K * OD) * sizeof(float);
    RLX_REQUIRE(lds <= 150 * 1024, RLX_EUNSUP, "sac: head too wide for the LDS-staged head kernel");
    hipLaunchKernelGGL(k_head_bwd, dim3(div_up(M, SAC_HEAD_ROWS), tw ? 2 : 1), dim3(256), lds, st, h_last, params + L.head.W,
                       d_out, partials, M, K, OD, d.act, tw ? *tw : Twin{});
  } else {
    RLX_REQUIRE(!tw, RLX_EUNSUP, "sac: twin head backward needs a last hidden width of at most 256");
    const size_t lds = ((size_t)SAC_HEAD_ROWS * (K + 1) + (size_t)K * OD + (size_t)SAC_HEAD_ROWS * OD) * sizeof(float);
    RLX_REQUIRE(lds <= 150 * 1024, RLX_EUNSUP, "sac: head too wide for the LDS-staged head kernel");
    hipLaunchKernelGGL(k_head_bwd_wide, dim3(div_up(M, SAC_HEAD_ROWS)), dim3(256), lds, st, h_last, params + L.head.W, d_out,
                       partials, M, K, OD, d.act);
  }
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

static inline const void* si_arena(const rlx_ctx* ctx) { return ctx->sac_img_arena; }

// the registered images of one network's weight matrices (vector offset `off` of its first parameter inside the optimizer job's flat
// vector) -> emit table entries
static void sac_emit_add(const rlx_ctx* ctx, BxEmitN* e, const rlx_mlp_desc& d, const float* params, int64_t off, bool want_t) {
  const MlpLayout L = make_layout(d);
  for (int l = 0; l < d.n_hidden && e->n < 8; ++l) {
    const LayerOff& o = L.layer[l];
    const void* nn = bx_lookup(ctx, params + o.W, 0, o.in, o.out);
    const void* tt = (want_t && l > 0) ? bx_lookup(ctx, params + o.W, 1, o.out, o.in) : nullptr;
    if (!nn && !tt) continue;
    BxEmitLayer& q = e->l[e->n++];
    q.w_off = off + o.W;
    q.in = o.in;
    q.out = o.out;
    q.nn = const_cast<void*>(nn);
    q.tt = const_cast<void*>(tt);
    q.nt_nn = 4 * div_up(o.out, G_BN);
    q.nt_tt = 4 * div_up(o.in, G_BN);
  }
}

// forward of one net keeping activations; out[M, out_dim]
// critics (out_dim 1, two-headed policy has out_dim >= 2) always read the padded concat buffers
static inline bool sac_gemm_l0(const rlx_mlp_desc& d, int ldx) { return d.in_dim > 32 || d.out_dim == 1 || ldx != d.in_dim; }

// sample (optional, policies): the sampling step that follows the forward (k_sac_sample).  MEASURED-NEGATIVE twice (rounds 4 and 5): the
// sampling as the epilogue of the forward kernel -- 2.2 us of serial threefry / erfinv / tanh / log per tile behind the head's barrier
// cost what the 5 us launch did (22.5 vs 21.6 us per acting call, 262.0 vs 263.3 us per vector step); not kept
static int launch_sac_sample(const float* head, const SacSampleArgs& sa, int64_t B, hipStream_t st) {
  const int AP = sac_lanes_per_row(sa.A);
  hipLaunchKernelGGL(k_sac_sample, dim3(div_up(B, 256 / AP)), dim3(256), (size_t)(256 / AP) * sa.A * sizeof(float), st, head, sa, B, AP);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

static int net_fwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x, int ldx,
                   float* const* acts, float* out, int64_t M, hipStream_t st, const SacSampleArgs* sample = nullptr) {
  if (sample) {
    const int rcs = net_fwd(ctx, d, L, params, x, ldx, acts, out, M, st);
    return rcs ? rcs : launch_sac_sample(out, *sample, M, st);
  }
  // rows from the padded concat buffers (the critics' [obs | action] input, wide policy observations) take the GEMM
  // first layer whatever the width; dense narrow observations take the small-input kernel
  {
    const void *w1x = nullptr, *w2x = nullptr;      // 256-256 nets: trunk + head in one launch (fwd2h.hip), activations kept for the backward
    if (sac_gemm_l0(d, ldx) && fwd2h_supported(ctx, d, L, params, M, ldx, &w1x, &w2x))
      return launch_fwd2h(ctx, d, L, params, w1x, w2x, x, ldx, acts[0], acts[1], out, M, st);
    const void* wx[3];                              // 512-LayerNorm-256-128 nets (full-jit flavour): the same, k_fwd3h; acts[3] = pre-LayerNorm values
    if (sac_gemm_l0(d, ldx) && acts[3] && fwd3h_supported(ctx, d, L, params, M, ldx, wx))
      return launch_fwd3h(ctx, d, L, params, wx, x, ldx, acts[3], acts[0], acts[1], acts[2], out, M, st);
  }
  int rc = mlp_trunk_fwd(ctx, d, L, params, x, acts, M, st, ldx, sac_gemm_l0(d, ldx));
  if (rc) return rc;
  return launch_head_fwd(acts[d.n_hidden - 1], params + L.head.W, params + L.head.b, out, M, L.head.in, L.head.out, st);
}

// backward of one net from d_out[M, out_dim]; grads may be null (input gradient only)
static int net_bwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x, int ldx,
                   float* const* acts, const float* d_out, float* grads, float* head_part, int64_t M, float* sumsq,
                   int* nsq, const TrunkOpts* opt_in, hipStream_t st) {
  const int K = L.head.in, OD = L.head.out;
  if (!grads && opt_in && opt_in->dx_out && dxa2h_supported(ctx, d, M, opt_in->dx_nc)) {      // dQ/da of one critic: one launch (fwd2h.hip)
    if (const void* w2t = bx_lookup(ctx, params + L.layer[1].W, 1, L.layer[1].out, L.layer[1].in))
      return launch_dxa2h(ctx, d, L, params, w2t, acts[0], acts[1], d_out, opt_in->dx_out, opt_in->dx_c0, opt_in->dx_nc, opt_in->dx_ld, M, st);
  }
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

// ---------------------------------------------------------------------------------------
// Twin critics.  The reference's critic is ONE vmapped module (sac/flax/critic.py:44-53): both Q nets see the same input and
// have the same shape.  At B = 4096 a single net's GEMM fills a quarter to a half of the chip and takes its ~10 us of launch,
// prologue and epilogue whatever the grid, so the pair goes out as ONE launch per layer (grid.y = 2, Twin in common.h): half
// the launches on the host, one kernel latency instead of two on the chain.  Plain two-hidden-layer nets whose weight images
// are registered (B >= 4096, gemm_bx on); everything else takes the two sequential passes.  Same kernels, same tiles per net:
// forward and input-gradient results are those of the sequential passes bit for bit; the weight gradients are summed over
// half as many (twice as long) M-slabs, i.e. in a different fp32 order.
// ---------------------------------------------------------------------------------------
struct TwinImgs { const void* f[3][2]; const void* t[3][2]; };   // forward / transposed image of layer l, net q
static bool twin_usable(const rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* p0, const float* p1,
                        int64_t M, int ldx, bool need_t, bool need_dw, int dx_nc, TwinImgs* im) {
  if (dx_nc > 64 || (size_t)(16 * (L.layer[0].out + 4) + L.layer[0].out * dx_nc) * sizeof(float) > 128 * 1024) return false;   // (launch_dx_cols)
  if (!ctx->sac_twin || !ctx->gemm_bx || d.n_hidden < 2 || d.n_hidden > 3 || d.out_dim != 1 || L.head.in > 256) return false;
  if (d.ln_first && (L.layer[0].out % 64 != 0 || L.layer[0].out > 512)) return false;
  const float* pp[2] = {p0, p1};
  for (int l = 0; l < d.n_hidden; ++l) {
    const LayerOff& o = L.layer[l];
    for (int q = 0; q < 2; ++q) {
      im->f[l][q] = bx_lookup(ctx, pp[q] + o.W, 0, o.in, o.out);
      im->t[l][q] = (need_t && l >= 1) ? bx_lookup(ctx, pp[q] + o.W, 1, o.out, o.in) : nullptr;
      if (!im->f[l][q] || (need_t && l >= 1 && !im->t[l][q])) return false;
    }
    if (!bx_twin_usable(ctx, M, o.out) || (need_t && l >= 1 && !bx_twin_usable(ctx, M, o.in))) return false;
    if (need_dw && !bx_dw_usable(ctx, M, o.in, l == 0 ? ldx : o.in, o.out)) return false;
  }
  return ldx % 4 == 0 && ldx >= d.in_dim;
}

static inline int ln_grid_rows(const rlx_ctx* ctx, int64_t M, int per_cu) {
  int grid = div_up(M, 4);
  if (grid > ctx->num_cus * per_cu) grid = ctx->num_cus * per_cu;
  return grid;
}

// LayerNorm backward: every workgroup leaves 2 * D partial sums, and the reduction's workgroups that own a D-long segment add its
// slabs in a chain of S / 64 dependent round trips (16 loads in flight per thread) -- with one row per wave (1024 workgroups at
// 4096 rows) that chain was the tail of the whole reduction launch (30 us); 16 rows per workgroup leave 256 slabs.
static inline int ln_bwd_grid(const rlx_ctx* ctx, int64_t M) {
  int grid = div_up(M, 16);
  if (grid > ctx->num_cus * 4) grid = ctx->num_cus * 4;
  return grid < 1 ? 1 : grid;
}

static int twin_fwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* p0, const float* p1, const TwinImgs& im,
                    const float* x, int ldx, float* const* acts0, float* const* acts1, float* out0, float* out1, int64_t M,
                    hipStream_t st) {
  const LayerOff& o0 = L.layer[0];
  Twin t;
  int rc;
  {
    const void *w1x = nullptr, *w2x = nullptr;      // 256-256 critics: both nets' whole forward in one launch (fwd2h.hip)
    if (fwd2h_supported(ctx, d, L, p0, M, ldx, &w1x, &w2x)) {
      const Fwd2hTwin tw{p1, im.f[0][1], im.f[1][1], acts1[0], acts1[1], out1};
      return launch_fwd2h(ctx, d, L, p0, w1x, w2x, x, ldx, acts0[0], acts0[1], out0, M, st, &tw);
    }
    const void* wx[3];
    if (acts0[3] && acts1[3] && fwd3h_supported(ctx, d, L, p0, M, ldx, wx)) {
      const Fwd3hTwin tw{p1, {im.f[0][1], im.f[1][1], im.f[2][1]}, acts1[3], acts1[0], acts1[1], acts1[2], out1};
      return launch_fwd3h(ctx, d, L, p0, wx, x, ldx, acts0[3], acts0[0], acts0[1], acts0[2], out0, M, st, &tw);
    }
  }
  if (d.ln_first) {
    // Dense -> acts[3] (kept: the backward needs the pre-LayerNorm values), LayerNorm + activation -> acts[0]
    RLX_REQUIRE(acts0[3] && acts1[3], RLX_EUNSUP, "sac: wide LayerNorm layer without its pre-activation buffer");
    t.p[0] = x; t.p[1] = im.f[0][1]; t.p[2] = p1 + o0.b; t.p[3] = acts1[3];
    rc = bx_launch_fwd(ctx, x, im.f[0][0], p0 + o0.b, acts0[3], M, o0.out, o0.in, RLX_ACT_NONE, st, ldx, nullptr, &t);
    if (rc) return rc;
    t.p[0] = acts1[3]; t.p[1] = acts1[0]; t.p[2] = p1 + o0.g; t.p[3] = nullptr;
    hipLaunchKernelGGL(k_ln_act_twin<false>, dim3(ln_grid_rows(ctx, M, 8), 2), dim3(256), 0, st, (const float*)acts0[3], acts0[0],
                       p0 + o0.g, p0 + o0.be, (float*)nullptr, M, o0.out, d.act, t);
    RLX_LAUNCH_CHECK();
  } else {
    t.p[0] = x; t.p[1] = im.f[0][1]; t.p[2] = p1 + o0.b; t.p[3] = acts1[0];
    rc = bx_launch_fwd(ctx, x, im.f[0][0], p0 + o0.b, acts0[0], M, o0.out, o0.in, d.act, st, ldx, nullptr, &t);
    if (rc) return rc;
  }
  for (int l = 1; l < d.n_hidden; ++l) {
    const LayerOff& o = L.layer[l];
    t.p[0] = acts1[l - 1]; t.p[1] = im.f[l][1]; t.p[2] = p1 + o.b; t.p[3] = acts1[l];
    rc = bx_launch_fwd(ctx, acts0[l - 1], im.f[l][0], p0 + o.b, acts0[l], M, o.out, o.in, d.act, st, 0, nullptr, &t);
    if (rc) return rc;
  }
  const int last = d.n_hidden - 1;
  t.p[0] = acts1[last]; t.p[1] = p1 + L.head.W; t.p[2] = p1 + L.head.b; t.p[3] = out1;
  return launch_head_fwd(acts0[last], p0 + L.head.W, p0 + L.head.b, out0, M, L.head.in, L.head.out, st, nullptr, &t);
}

// backward of both critics from d_out0 / d_out1 [M, 1] (the layer order of mlp_trunk_bwd, every launch a twin).  grads0 / grads1
// != NULL: parameter gradients of both nets, reduced by ONE launch (net 0's segments first: the norm partials keep the order of
// two sequential passes); NULL: input gradient only, of the columns [dx_c0, dx_c0 + dx_nc) into dx0 / dx1 (row stride dx_ld).
static int twin_bwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* p0, const float* p1, const TwinImgs& im,
                    const float* x, int ldx, float* const* acts0, float* const* acts1, const float* d_out0, const float* d_out1,
                    float* grads0, float* grads1, float* hpart0, float* hpart1, float* dx0, float* dx1, int dx_c0, int dx_nc,
                    int dx_ld, int64_t M, float* sumsq, int* nsq, hipStream_t st) {
  const LayerOff& o0 = L.layer[0];
  const bool pg = grads0 != nullptr;
  const int nh = d.n_hidden, last = nh - 1;
  if (!pg && im.t[1][0] && im.t[1][1] && dxa2h_supported(ctx, d, M, dx_nc)) {      // the whole dQ/da chain of both critics in one launch
    const Dxa2hTwin tw{p1, im.t[1][1], acts1[0], acts1[1], d_out1, dx1};
    return launch_dxa2h(ctx, d, L, p0, im.t[1][0], acts0[0], acts0[1], d_out0, dx0, dx_c0, dx_nc, dx_ld, M, st, &tw);
  }
  float* gr[2] = {grads0, grads1};
  Twin t;
  t.p[0] = acts1[last]; t.p[1] = p1 + L.head.W; t.p[2] = d_out1; t.p[3] = pg ? hpart1 : nullptr;
  int rc = head_bwd(ctx, d, L, p0, acts0[last], d_out0, pg ? hpart0 : nullptr, M, st, &t);
  if (rc) return rc;
  // M-slabs of the weight gradients: one workgroup per CU over BOTH nets (the kernel's 96 KB tile leaves room for one per CU)
  const int cus = ctx->num_cus / 2 > 0 ? ctx->num_cus / 2 : 1;
  const int ln_grid = ln_bwd_grid(ctx, M);
  int S[3] = {0, 0, 0};
  int64_t Mc[3] = {0, 0, 0};
  size_t per_net = 0;
  float *pW[3][2] = {}, *pB[3][2] = {}, *pLN[2] = {nullptr, nullptr};
  if (pg) {
    for (int l = 0; l < nh; ++l) {
      const LayerOff& o = L.layer[l];
      Mc[l] = choose_mc(M, div_up(o.in, G_BM) * div_up(o.out, G_BN), cus, &S[l]);
      per_net += (size_t)S[l] * ((size_t)o.in * o.out + o.out);
    }
    if (d.ln_first) per_net += (size_t)ln_grid * 2 * o0.out;
    per_net = (per_net + 63) & ~size_t(63);
    float* arena = (float*)scratch(ctx, SL_PARTIAL, 2 * per_net * sizeof(float));
    if (!arena) return RLX_ENOMEM;
    for (int q = 0; q < 2; ++q) {
      float* cur = arena + q * per_net;
      for (int l = 0; l < nh; ++l) {
        const LayerOff& o = L.layer[l];
        pW[l][q] = cur; cur += (size_t)S[l] * o.in * o.out;
        pB[l][q] = cur; cur += (size_t)S[l] * o.out;
      }
      if (d.ln_first) pLN[q] = cur;
    }
  } else if (d.ln_first) {
    // the LayerNorm backward writes its scale / bias partials unconditionally: park them in the arena
    per_net = ((size_t)ln_grid * 2 * o0.out + 63) & ~size_t(63);
    float* arena = (float*)scratch(ctx, SL_PARTIAL, 2 * per_net * sizeof(float));
    if (!arena) return RLX_ENOMEM;
    pLN[0] = arena;
    pLN[1] = arena + per_net;
  }
  // Plain two-hidden-layer critics: the input gradient of layer 1 goes OUT OF PLACE (dZ0 in its own buffer, act' from h1), so h1
  // survives and both weight gradients -- h1^T dZ1 and x^T dZ0 -- are ONE two-job twin launch instead of two launches with the
  // input gradient between them: one dependent kernel less on the critic chain.
  const bool merge = pg && ctx->dw_merge && nh == 2 && !d.ln_first && bx_dw_usable(ctx, M, L.layer[1].in, L.layer[1].in, L.layer[1].out) &&
                     bx_dw_usable(ctx, M, o0.in, ldx, o0.out);
  if (merge) {
    const LayerOff& o1 = L.layer[1];
    float* dz0 = (float*)scratch(ctx, SL_DZ0, (size_t)2 * M * o0.out * sizeof(float));
    if (!dz0) return RLX_ENOMEM;
    float* dz0q[2] = {dz0, dz0 + (size_t)M * o0.out};
    t.p[0] = acts1[1]; t.p[1] = im.t[1][1]; t.p[2] = acts1[0]; t.p[3] = dz0q[1];
    rc = bx_launch_dx(ctx, acts0[1], im.t[1][0], dz0q[0], M, o1.out, o1.in, o1.in, d.act, 1, st, &t, acts0[0]);
    if (rc) return rc;
    // the same M-slabs for both jobs, never more workgroups than CUs over both nets
    const int tiles = div_up(o1.in, G_BM) * div_up(o1.out, G_BN) + div_up(o0.in, G_BM) * div_up(o0.out, G_BN);
    int Sm = 0;
    const int64_t Mcm = choose_mc_fit(M, tiles, cus, &Sm);
    RLX_REQUIRE(Sm <= S[0] && Sm <= S[1], RLX_EUNSUP, "sac: merged weight-gradient slabs exceed the arena");
    S[0] = S[1] = Sm;
    Mc[0] = Mc[1] = Mcm;
    const BxDwJob j0{x, dz0q[0], pW[0][0], pB[0][0], o0.in, ldx, o0.out, Mcm, Sm, div_up(o0.in, G_BM), div_up(o0.out, G_BN)};
    const BxDwJob j1{acts0[0], acts0[1], pW[1][0], pB[1][0], o1.in, o1.in, o1.out, Mcm, Sm, div_up(o1.in, G_BM), div_up(o1.out, G_BN)};
    Twin t0, t1;
    t0.p[0] = x; t0.p[1] = dz0q[1]; t0.p[2] = pW[0][1]; t0.p[3] = pB[0][1];
    t1.p[0] = acts1[0]; t1.p[1] = acts1[1]; t1.p[2] = pW[1][1]; t1.p[3] = pB[1][1];
    rc = bx_launch_dw2(ctx, j0, j1, M, st, &t0, &t1);
    if (rc) return rc;
  }
  for (int l = last; l >= 1 && !merge; --l) {
    const LayerOff& o = L.layer[l];
    if (pg) {
      t.p[0] = acts1[l - 1]; t.p[1] = acts1[l]; t.p[2] = pW[l][1]; t.p[3] = pB[l][1];
      rc = bx_launch_dw(ctx, acts0[l - 1], acts0[l], pW[l][0], pB[l][0], M, o.in, o.in, o.out, Mc[l], S[l], div_up(o.in, G_BM),
                        div_up(o.out, G_BN), st, &t);
      if (rc) return rc;
    }
    // dZ_{l-1} = (dZ_l @ W_l^T) * act'(H_{l-1}) in place over acts[l-1]; below a LayerNorm the act' and LN' come afterwards
    const int apply = (l - 1 == 0 && d.ln_first) ? 0 : 1;
    t.p[0] = acts1[l]; t.p[1] = im.t[l][1]; t.p[2] = nullptr; t.p[3] = acts1[l - 1];
    rc = bx_launch_dx(ctx, acts0[l], im.t[l][0], acts0[l - 1], M, o.out, o.in, o.in, d.act, apply, st, &t);
    if (rc) return rc;
  }
  if (d.ln_first) {
    // acts[0] holds dL/dH0 (raw); LayerNorm' and act' from the kept pre-LayerNorm values -> dZ0 in place
    t.p[0] = acts1[3]; t.p[1] = acts1[0]; t.p[2] = p1 + o0.g; t.p[3] = pLN[1];
    hipLaunchKernelGGL(k_ln_act_twin<true>, dim3(ln_grid, 2), dim3(256), (size_t)8 * o0.out * sizeof(float), st,
                       (const float*)acts0[3], acts0[0], p0 + o0.g, p0 + o0.be, pLN[0], M, o0.out, d.act, t);
    RLX_LAUNCH_CHECK();
  }
  if (!pg) {
    t.p[0] = acts1[0]; t.p[1] = p1 + o0.W + (int64_t)dx_c0 * o0.out; t.p[2] = nullptr; t.p[3] = dx1;
    return launch_dx_cols(acts0[0], p0 + o0.W + (int64_t)dx_c0 * o0.out, dx0, M, o0.out, dx_nc, dx_ld, st, &t);
  }
  if (!merge) {
    t.p[0] = x; t.p[1] = acts1[0]; t.p[2] = pW[0][1]; t.p[3] = pB[0][1];
    rc = bx_launch_dw(ctx, x, acts0[0], pW[0][0], pB[0][0], M, o0.in, ldx, o0.out, Mc[0], S[0], div_up(o0.in, G_BM),
                      div_up(o0.out, G_BN), st, &t);
    if (rc) return rc;
  }
  ReduceTable tab;
  tab.n = 0;
  float* hp_[2] = {hpart0, hpart1};
  const int nb = div_up(M, SAC_HEAD_ROWS);
  const int64_t PS = (int64_t)L.head.in * L.head.out + L.head.out;
  for (int q = 0; q < 2; ++q) {   // per net: the segment order of mlp_trunk_bwd
    for (int l = last; l >= 1; --l) {
      const LayerOff& o = L.layer[l];
      tab.seg[tab.n++] = ReduceSeg{pW[l][q], gr[q] + o.W, (int64_t)o.in * o.out, (int64_t)o.in * o.out, S[l], 0, 1.f, 0.f, 1};
      tab.seg[tab.n++] = ReduceSeg{pB[l][q], gr[q] + o.b, (int64_t)o.out, (int64_t)o.out, S[l], 0, 1.f, 0.f, 1};
    }
    if (d.ln_first) {
      tab.seg[tab.n++] = ReduceSeg{pLN[q], gr[q] + o0.g, (int64_t)o0.out, (int64_t)2 * o0.out, ln_grid, 0, 1.f, 0.f, 1};
      tab.seg[tab.n++] = ReduceSeg{pLN[q] + o0.out, gr[q] + o0.be, (int64_t)o0.out, (int64_t)2 * o0.out, ln_grid, 0, 1.f, 0.f, 1};
    }
    tab.seg[tab.n++] = ReduceSeg{pW[0][q], gr[q] + o0.W, (int64_t)o0.in * o0.out, (int64_t)o0.in * o0.out, S[0], 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{pB[0][q], gr[q] + o0.b, (int64_t)o0.out, (int64_t)o0.out, S[0], 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{hp_[q], gr[q] + L.head.W, (int64_t)L.head.in * L.head.out, PS, nb, 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{hp_[q] + (int64_t)L.head.in * L.head.out, gr[q] + L.head.b, (int64_t)L.head.out, PS, nb, 0, 1.f, 0.f, 1};
  }
  return launch_reduce_segments(tab, sumsq, nsq, st);
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
  int grid = div_up(B, 4);
  if (grid > 4096) grid = 4096;
  const uintptr_t al = (uintptr_t)ring_states | (uintptr_t)ring_next_states | (uintptr_t)states | (uintptr_t)next_states;
  const int vec = (obs_dim % 4 == 0 && al % 16 == 0) ? 1 : 0;
  hipLaunchKernelGGL(k_replay_gather, dim3(grid), dim3(256), 0, (hipStream_t)stream, ring_states, ring_next_states,
                     ring_actions, ring_rewards, ring_terminations, idx1, idx2, nr_envs, obs_dim, act_dim, B, states,
                     next_states, actions, rewards, terminations, vec);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_sac_invalidate_images(rlx_ctx* ctx) {
  RLX_REQUIRE(ctx, RLX_EINVAL, "rlx_sac_invalidate_images: NULL context");
  ctx->sac_img.valid = false;
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

// acting forward + sampling.  With the policy's images kept current by the update calls (rlx_sac_hparams::keep_images) they are
// registered, not laid out again.
static int sac_policy_act(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams, const float* obs, float* head, int N,
                          const SacSampleArgs& sa, hipStream_t st) {
  const rlx_ctx::SacImages& si = ctx->sac_img;
  const bool reg = si.valid && si.pp == pparams && std::memcmp(&si.pd, pdesc, sizeof(rlx_mlp_desc)) == 0 &&
                   N >= 4096 && ctx->bx_n[0] == 0 && ctx->bx_n[1] == 0;
  struct Rel { rlx_ctx* c; bool on; ~Rel() { if (on) bx_release_all(c); } } rel{ctx, false};
  int rc;
  if (reg) {
    const BxNetSpec net = {pdesc, pparams, true, pdesc->in_dim > 32};
    rc = bx_prepare_nets(ctx, &net, 1, st, SL_WFRAG_SAC, false);
    if (rc) return rc;
    rel.on = true;
    if (ctx->bx_n[0] > 0 && ctx->bx_img[0][0].img != si_arena(ctx)) { bx_release_all(ctx); rel.on = false; }
  }
  rc = rlx_mlp_fwd_f32(ctx, pdesc, pparams, obs, head, N, (void*)st);
  if (rc) return rc;
  return launch_sac_sample(head, sa, N, st);
}

static int sac_act_impl(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams, const float* obs, uint32_t key_io[2],
                        int scheme, float* action, int N, float log_std_min, float log_std_max, int deterministic,
                        int row_offset, int N_global, const float* low, const float* half_range, float* processed,
                        void* stream, const char* who) {
  RLX_REQUIRE(ctx && pdesc && pparams && obs && key_io && action && N > 0 && N_global >= N, RLX_EINVAL,
              (std::string(who) + ": bad args").c_str());
  RLX_REQUIRE(pdesc->out_dim % 2 == 0, RLX_EINVAL, (std::string(who) + ": policy out_dim must be 2 * act_dim (mean | log_std)").c_str());
  int rc = mlp_check_desc(*pdesc);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int A = pdesc->out_dim / 2;
  float* head = (float*)scratch(ctx, SL_MEAN, (size_t)N * 2 * A * sizeof(float));
  if (!head) return RLX_ENOMEM;
  uint32_t ks[4] = {key_io[0], key_io[1], 0, 0};
  if (!deterministic) {
    split_host(key_io, ks, 2, scheme);  // key, subkey = split(key)   (sac.py:123)
    key_io[0] = ks[0];
    key_io[1] = ks[1];
  }
  SacSampleArgs sa;
  sa.k0 = ks[2]; sa.k1 = ks[3]; sa.scheme = scheme; sa.mode = 0; sa.act_out = action; sa.ld_out = A; sa.col_off = 0; sa.A = A;
  sa.ls_min = log_std_min; sa.ls_max = log_std_max; sa.row_off = row_offset; sa.N_global = (int64_t)N_global;
  sa.deterministic = deterministic; sa.proc_out = processed; sa.proc_low = low; sa.proc_half = half_range;
  return sac_policy_act(ctx, pdesc, pparams, obs, head, N, sa, st);
}

int rlx_sac_act_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams, const float* obs, uint32_t key_io[2],
                    int scheme, float* action, int N, float log_std_min, float log_std_max, int deterministic,
                    int row_offset, int N_global, void* stream) {
  return sac_act_impl(ctx, pdesc, pparams, obs, key_io, scheme, action, N, log_std_min, log_std_max, deterministic, row_offset,
                      N_global, nullptr, nullptr, nullptr, stream, "rlx_sac_act_f32");
}

int rlx_sac_act_processed_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams, const float* obs,
                              uint32_t key_io[2], int scheme, float* action, int N, float log_std_min, float log_std_max,
                              int deterministic, int row_offset, int N_global, const float* low, const float* half_range,
                              float* processed, void* stream) {
  RLX_REQUIRE(low && half_range && processed, RLX_EINVAL, "rlx_sac_act_processed_f32: NULL pointer");
  return sac_act_impl(ctx, pdesc, pparams, obs, key_io, scheme, action, N, log_std_min, log_std_max, deterministic, row_offset,
                      N_global, low, half_range, processed, stream, "rlx_sac_act_processed_f32");
}

int rlx_sac_update_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, float* pparams, float* pm, float* pv,
                       const rlx_mlp_desc* qdesc, float* qparams, float* qm, float* qv, float* qtarget, float* log_alpha,
                       float* am, float* av, const float* states, const float* next_states, const float* actions,
                       const float* rewards, const float* terminations, int64_t B, uint32_t key_io[2], int scheme,
                       int64_t* opt_count_io, const rlx_sac_hparams* hp, float* metrics_out, void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && pm && pv && qdesc && qparams && qm && qv && qtarget && log_alpha && am && av &&
                  actions && rewards && terminations && key_io && opt_count_io && hp && metrics_out,
              RLX_EINVAL, "rlx_sac_update_f32: NULL pointer");
  // states / next_states may be NULL together with the ring source: the caller does not want the gathered observation rows back
  const bool elide_states = !states && !next_states && hp->ring_states != nullptr;
  RLX_REQUIRE((states && next_states) || elide_states, RLX_EINVAL,
              "rlx_sac_update_f32: states / next_states may only be NULL both at once and with the ring source (hp->ring_states)");
  RLX_REQUIRE(B > 0, RLX_EINVAL, "rlx_sac_update_f32: batch must be positive");
  int rc = mlp_check_desc(*pdesc);
  if (rc) return rc;
  rc = mlp_check_desc(*qdesc);
  if (rc) return rc;
  const int O = pdesc->in_dim, A = pdesc->out_dim / 2;
  // Oc: observation columns the critics read.  An env with critic_observation_indices != policy_observation_indices
  // (sac/flax/critic.py:11,23 vs policy.py:14,31) hands the critics' columns of the sampled transitions over through
  // hp->critic_states / critic_next_states; states / next_states then hold the policy's columns.
  const int Oc = qdesc->in_dim - A;
  const bool asym = hp->critic_states != nullptr;
  RLX_REQUIRE(pdesc->out_dim == 2 * A && Oc > 0 && qdesc->out_dim == 1 && !pdesc->has_logstd, RLX_EINVAL,
              "rlx_sac_update_f32: policy out_dim = 2*act_dim (no logstd param), critic in_dim = obs+act, out_dim = 1");
  RLX_REQUIRE(asym ? hp->critic_next_states != nullptr : Oc == O, RLX_EINVAL,
              "rlx_sac_update_f32: critic in_dim != obs + act needs hp->critic_states AND hp->critic_next_states");
  RLX_REQUIRE(!elide_states || O > 32, RLX_EINVAL,
              "rlx_sac_update_f32: states / next_states = NULL needs obs_dim > 32 (the policy reads narrow observation rows from them)");
  const float* cstates = asym ? hp->critic_states : states;
  const float* cnext = asym ? hp->critic_next_states : next_states;
  hipStream_t st = (hipStream_t)stream;
  // data parallel: this rank's B rows are rows [roff, roff + B) of a global batch of Bg samples (rlx_sac_hparams)
  const int64_t Bg = hp->batch_global > 0 ? hp->batch_global : B;
  const int64_t roff = hp->batch_global > 0 ? hp->batch_row_offset : 0;
  const bool sharded = Bg != B;
  RLX_REQUIRE(roff >= 0 && roff + B <= Bg, RLX_EINVAL, "rlx_sac_update_f32: batch_row_offset + B exceeds batch_global");
  RLX_REQUIRE(!hp->ring_states || (hp->ring_next_states && hp->ring_actions && hp->ring_rewards && hp->ring_terminations &&
                                   hp->ring_idx1 && hp->ring_idx2 && hp->ring_nr_envs > 0 && !hp->critic_states),
              RLX_EINVAL, "rlx_sac_update_f32: the ring source needs all five ring arrays, both index vectors, nr_envs > 0 and "
                          "no separate critic observation columns");
  RLX_REQUIRE(!sharded || dist_active(ctx), RLX_EINVAL,
              "rlx_sac_update_f32: batch_global > B needs a context with a communicator (rlx_ctx_create with world > 1) or an all-reduce hook");
  const MlpLayout LP = make_layout(*pdesc), LQ = make_layout(*qdesc);
  const int64_t np_ = LP.n_params, nq_ = LQ.n_params;
  const int ldc = (Oc + A + 3) & ~3;
  const int ldp = (O + 3) & ~3;                              // asym, wide policy observation with a ragged width: padded copy
  const int lda = (A + 3) & ~3;
  const int nb = div_up(B, 256);
  const int AP = sac_lanes_per_row(A);                       // per-(row, action dim) kernels
  const int nb_rc = div_up(B, 256 / AP);
  const size_t lds_rc = (size_t)(256 / AP) * A * sizeof(float);
  // ---- scratch arena
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 63) & ~size_t(63); return o; };
  int hmax = 0;
  for (int l = 0; l < pdesc->n_hidden; ++l) hmax = pdesc->hidden[l] > hmax ? pdesc->hidden[l] : hmax;
  for (int l = 0; l < qdesc->n_hidden; ++l) hmax = qdesc->hidden[l] > hmax ? qdesc->hidden[l] : hmax;
  const size_t o_xc = take((size_t)B * ldc), o_xn = take((size_t)B * ldc), o_xp = take((size_t)B * ldc);
  const bool pol_pad = asym && O > 32 && O % 4 != 0;
  const size_t o_pp = pol_pad ? take((size_t)B * ldp) : 0, o_pn = pol_pad ? take((size_t)B * ldp) : 0;
  // act sets: 0 tmp (next-state policy, target critics), 1 policy on s, 2 / 3 online critics on (s, a),
  //           4 / 5 online critics on (s, pi(s))  -- separate so the critic-loss and policy-loss chains can overlap;
  //           6 second target critic when the pair runs as one twin launch per layer
  size_t o_acts[7][4];                       // [.][3]: pre-LayerNorm values of the first layer (full-jit nets)
  const bool any_ln = pdesc->ln_first || qdesc->ln_first;
  for (int s = 0; s < 7; ++s)
    for (int l = 0; l < 4; ++l) o_acts[s][l] = (l < 3 || any_ln) ? take((size_t)B * hmax) : 0;
  const size_t o_hn = take((size_t)B * 2 * A), o_hc = take((size_t)B * 2 * A), o_dpi = take((size_t)B * 2 * A);
  const size_t o_vec = take((size_t)B * 12);  // qt0 qt1 q0 q1 qa0 qa1 lpn lpc dq0 dq1 d0 d1
  const size_t o_da0 = take((size_t)B * lda), o_da1 = take((size_t)B * lda);
  const size_t o_gp = take(np_), o_gq = take(2 * nq_), o_ga = take(64);
  const size_t o_part = take((size_t)nb * 3 + 64);
  const size_t hp_floats = (size_t)div_up(B, SAC_HEAD_ROWS) * ((size_t)LP.head.in * LP.head.out + LP.head.out + LQ.head.in + 1);
  const size_t o_hpart = take(hp_floats), o_hpart2 = take(hp_floats), o_hpart3 = take(hp_floats);
  float* base = (float*)scratch(ctx, SL_SAC, off * sizeof(float));
  float* sq0 = (float*)scratch(ctx, SL_NORM, REDUCE_MAX_BLOCKS * sizeof(float));
  float* sq1 = (float*)scratch(ctx, SL_NORM2, REDUCE_MAX_BLOCKS * sizeof(float));
  if (!base || !sq0 || !sq1) return RLX_ENOMEM;
  float *xc = base + o_xc, *xn = base + o_xn, *xp = base + o_xp;
  NetBufs nbuf[7];
  for (int s = 0; s < 7; ++s)
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
  float* hpart_q1 = base + o_hpart3;   // second critic's head partials (twin backward)

  // keys = split(key, 2B+1 [+1]); key = keys[0]
  const uint32_t k0 = key_io[0], k1 = key_io[1];
  const int ksched = hp->key_schedule ? 1 : 0;
  {
    uint32_t nk[2];
    const uint64_t nkeys = sac_key_count(Bg, ksched);
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
  // policy inputs: narrow observations straight from the caller's rows; wide ones from 16-byte-pitched rows -- the critics'
  // concat buffers when both read the same columns, the caller's rows (or a padded copy) when the critics have their own
  const int ldo = O > 32 ? (asym ? ldp : ldc) : O;
  const float* pol_next = O > 32 ? (asym ? (pol_pad ? base + o_pn : next_states) : xn) : next_states;
  const float* pol_cur = O > 32 ? (asym ? (pol_pad ? base + o_pp : states) : xc) : states;
  // ---- per-call values -> device memory (stream-ordered, in front of the update's launches)
  const int64_t step = *opt_count_io + 1;
  SacConsts hc_{};
  adam_schedule_entry(hc_.sched + 0, step, hp->lr_policy, hp->adam_b1, hp->adam_b2);
  adam_schedule_entry(hc_.sched + 4, step, hp->lr_critic, hp->adam_b1, hp->adam_b2);
  adam_schedule_entry(hc_.sched + 8, step, hp->lr_alpha, hp->adam_b1, hp->adam_b2);
  hc_.key[0] = k0;
  hc_.key[1] = k1;
  SacConsts* cst = (SacConsts*)scratch(ctx, SL_SCHED, sizeof(SacConsts));
  if (!cst) return RLX_ENOMEM;
  const uint32_t* key_dev = cst->key;
  const int nch = ctx->two_streams ? 2 : 1;
  if (nch > 1) {
    rc = ctx_sac_streams(ctx);
    if (rc) return rc;
  }
  // Three chains that only meet at the seed kernels and at the optimizer (sac.py:133-188 evaluates both losses on the same,
  // pre-update parameters):
  //   A (s0)  : policy(s') -> a', log pi -> target critics -> [wait C] -> critic seed -> critic backward
  //   B (side): policy(s) -> a~, log pi -> critics on (s, a~) -> seed -> dQ/da -> policy backward
  //   C       : online critics on (s, a)              (in front of chain A on its stream)
  // (B = 4096 rows fill a quarter of the chip per GEMM: the chains overlap almost for free -- once the host is out of the way:
  //  issued eagerly, ~75 launches take longer to SUBMIT than to run, and chain B starts when chain A has been submitted.)
  GradScaleScope gscope(ctx, bx_grad_scale(Bg));   // every loss of the update is a mean over the global batch
  auto issue = [&](hipStream_t s0) -> int {
    int r;
    {
      const bool ring = hp->ring_states != nullptr;
      if (ring) {   // transitions from the replay ring: into the caller's batch arrays (outputs here) ...
        float *ws = const_cast<float*>(states), *ws2 = const_cast<float*>(next_states), *wa = const_cast<float*>(actions),
              *wr = const_cast<float*>(rewards), *wt = const_cast<float*>(terminations);
        // wide symmetric observations: every pass of the update reads the critics' input rows; the two gathered observation arrays
        // (12 of the launch's 44 MB) are written only for a caller that passes them (states / next_states NULL: left out)
        int grid = div_up(B, 4);
        if (grid > 4096) grid = 4096;
        const uintptr_t al = (uintptr_t)hp->ring_states | (uintptr_t)hp->ring_next_states | (uintptr_t)ws | (uintptr_t)ws2;
        const int vec = (O % 4 == 0 && al % 16 == 0) ? 1 : 0;
        // ... and the critics' input rows in the same launch
        hipLaunchKernelGGL(k_sac_gather_concat, dim3(grid), dim3(256), 0, s0, hp->ring_states, hp->ring_next_states,
                           hp->ring_actions, hp->ring_rewards, hp->ring_terminations, hp->ring_idx1, hp->ring_idx2,
                           (int)hp->ring_nr_envs, O, A, B, ws, ws2, wa, wr, wt, xc, xn, xp, ldc, (vec && ldc % 4 == 0) ? 1 : 0, hc_, cst);
        RLX_LAUNCH_CHECK();
      } else {
        int grid = div_up((int64_t)B * ldc, 256);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(k_sac_concat, dim3(grid), dim3(256), 0, s0, cstates, cnext, actions, xc, xn, xp, B, Oc, A, ldc, hc_,
                           cst);
        RLX_LAUNCH_CHECK();
      }
      if (pol_pad) {   // [s | 0] and [s' | 0] at pitch ldp (A = 0: no action columns; the third output aliases the first)
        float *pp_ = base + o_pp, *pn_ = base + o_pn;
        int g2 = div_up((int64_t)B * ldp, 256);
        if (g2 > 4096) g2 = 4096;
        hipLaunchKernelGGL(k_sac_concat, dim3(g2), dim3(256), 0, s0, states, next_states, actions, pp_, pn_, pp_, B, O, 0, ldp,
                           SacConsts(), (SacConsts*)nullptr);
        RLX_LAUNCH_CHECK();
      }
    }
    // split-fp16 weight images of all five networks for the GEMMs of this update (batches >= 4096 rows; the parameters do
    // not change before the optimizer steps at the end): one launch, in front of the fork
    struct BxAll { rlx_ctx* c; ~BxAll() { bx_release_all(c); } } bx_all{ctx};
    bool emit_images = false;
    if (B >= 4096) {
      const bool pw = pdesc->in_dim > 32, qw = true;   // critics always run their first layer on the GEMM kernels
      const BxNetSpec nets[5] = {{pdesc, pparams, true, pw}, {qdesc, qparams, true, qw}, {qdesc, qparams + nq_, true, qw},
                                 {qdesc, qtarget, false, qw}, {qdesc, qtarget + nq_, false, qw}};
      const bool keep = hp->keep_images != 0;
      rlx_ctx::SacImages& si = ctx->sac_img;
      bool reuse = keep && si.valid && si.pp == pparams && si.qp == qparams && si.qt == qtarget &&
                   std::memcmp(&si.pd, pdesc, sizeof(rlx_mlp_desc)) == 0 && std::memcmp(&si.qd, qdesc, sizeof(rlx_mlp_desc)) == 0;
      r = bx_prepare_nets(ctx, nets, 5, s0, keep ? SL_WFRAG_SAC : SL_WFRAG, !reuse);
      if (r) return r;
      if (reuse && ctx->bx_n[0] > 0 && ctx->bx_img[0][0].img != si_arena(ctx)) {      // the arena moved (it was enlarged): lay the images out again
        r = bx_prepare_nets(ctx, nets, 5, s0, SL_WFRAG_SAC, true);
        if (r) return r;
      }
      si.valid = false;
      if (keep && ctx->bx_n[0] > 0) {
        si.pp = pparams; si.qp = qparams; si.qt = qtarget; si.pd = *pdesc; si.qd = *qdesc; si.np = np_; si.nq2 = 2 * nq_;
        ctx->sac_img_arena = ctx->bx_img[0][0].img;
        emit_images = true;      // k_sac_optimizers keeps them current; valid again once it has been issued
      }
    } else {
      ctx->sac_img.valid = false;      // an update without images changes the parameters under any kept ones
    }
    // Chain B (policy loss: two backward passes) is the LONGER one and stays on the caller's stream; chains C + A (critic loss) run
    // on the side stream.  The fork's and the join's cross-stream hand-offs (an event record + wait costs ~10 us on the device
    // before the waiting queue resumes) then sit on the SHORTER chain: B starts right behind the gather and the optimizer launch
    // right behind B's last kernel, while the side stream's event has long fired (round 5 had A on the caller's stream: 12 us idle
    // in front of k_sac_optimizers and 7 us behind the gather, profiles/r05_sac_timeline.txt).
    hipStream_t sA = nch >= 2 ? ctx->side : s0;
    hipStream_t sB = s0;
    hipStream_t sC = sA;
    if (nch >= 2) {
      RLX_HIP_TRY(hipEventRecord(ctx->sac_ev[0], s0));
      RLX_HIP_TRY(hipStreamWaitEvent(sA, ctx->sac_ev[0], 0));
    }
    int nsq_q0 = 0, nsq_q1 = 0, nsq_p = 0;
    TwinImgs im;
    // ---- chain C: both online critics on (s, a) (critic 0 keeps its activations in set 2, critic 1 in set 3)
    auto C1 = [&]() -> int {
      ctx->bank = 0;
      if (twin_usable(ctx, *qdesc, LQ, qparams, qparams + nq_, B, ldc, false, false, 0, &im)) {
        r = twin_fwd(ctx, *qdesc, LQ, qparams, qparams + nq_, im, xc, ldc, nbuf[2].acts, nbuf[3].acts, q0, q1, B, sC);
      } else {
        r = net_fwd(ctx, *qdesc, LQ, qparams, xc, ldc, nbuf[2].acts, q0, B, sC);
        if (!r) r = net_fwd(ctx, *qdesc, LQ, qparams + nq_, xc, ldc, nbuf[3].acts, q1, B, sC);
      }
      return r;       // (same stream as chain A: q0, q1 are ready when its seed kernel runs)
    };
    // ---- chain A: critic loss
    auto A1 = [&]() -> int {
      ctx->bank = 0;
      SacSampleArgs sa;
      sa.scheme = scheme; sa.mode = 1; sa.act_out = xn; sa.ld_out = ldc; sa.col_off = Oc; sa.logp = lpn; sa.A = A;
      sa.ls_min = hp->log_std_min; sa.ls_max = hp->log_std_max; sa.row_off = (int)roff; sa.N_global = Bg;
      sa.eps_inject = ctx->dbg_sac_eps[0]; sa.schedule = ksched; sa.key_dev = key_dev;
      return net_fwd(ctx, *pdesc, LP, pparams, pol_next, ldo, nbuf[0].acts, hn, B, sA, &sa);
    };
    auto A2 = [&]() -> int {
      ctx->bank = 0;
      if (twin_usable(ctx, *qdesc, LQ, qtarget, qtarget + nq_, B, ldc, false, false, 0, &im)) {
        r = twin_fwd(ctx, *qdesc, LQ, qtarget, qtarget + nq_, im, xn, ldc, nbuf[0].acts, nbuf[6].acts, qt0, qt1, B, sA);
      } else {
        r = net_fwd(ctx, *qdesc, LQ, qtarget, xn, ldc, nbuf[0].acts, qt0, B, sA);
        if (!r) r = net_fwd(ctx, *qdesc, LQ, qtarget + nq_, xn, ldc, nbuf[0].acts, qt1, B, sA);
      }
      if (r) return r;
      hipLaunchKernelGGL(k_sac_critic_seed, dim3(nb), dim3(256), 0, sA, qt0, qt1, lpn, rewards, terminations, log_alpha, q0,
                         q1, dq0, dq1, part_c, B, hp->gamma, Bg);
      RLX_LAUNCH_CHECK();
      return RLX_OK;
    };
    auto A3 = [&]() -> int {
      ctx->bank = 0;
      // the two critics are ONE optimizer state in the reference: their squared norms are summed
      if (twin_usable(ctx, *qdesc, LQ, qparams, qparams + nq_, B, ldc, true, true, 0, &im)) {
        r = twin_bwd(ctx, *qdesc, LQ, qparams, qparams + nq_, im, xc, ldc, nbuf[2].acts, nbuf[3].acts, dq0, dq1, gq, gq + nq_,
                     hpart, hpart_q1, nullptr, nullptr, 0, 0, 0, B, sq0, &nsq_q0, sA);
      } else {
        r = net_bwd(ctx, *qdesc, LQ, qparams, xc, ldc, nbuf[2].acts, dq0, gq, hpart, B, sq0, &nsq_q0, nullptr, sA);
        if (!r) r = net_bwd(ctx, *qdesc, LQ, qparams + nq_, xc, ldc, nbuf[3].acts, dq1, gq + nq_, hpart, B, sq0 + nsq_q0,
                            &nsq_q1, nullptr, sA);
      }
      if (r) return r;
      RLX_REQUIRE(nsq_q0 + nsq_q1 <= REDUCE_MAX_BLOCKS, RLX_EUNSUP, "sac: critic too large for the norm partial array");
      return RLX_OK;
    };
    // ---- chain B: policy loss (critic activations of this chain live in sets 4 / 5)
    const int bankB = sA != s0 ? 1 : 0;
    auto B1 = [&]() -> int {
      ctx->bank = bankB;
      SacSampleArgs sa;
      sa.scheme = scheme; sa.mode = 2; sa.act_out = xp; sa.ld_out = ldc; sa.col_off = Oc; sa.logp = lpc; sa.A = A;
      sa.ls_min = hp->log_std_min; sa.ls_max = hp->log_std_max; sa.row_off = (int)roff; sa.N_global = Bg;
      sa.eps_inject = ctx->dbg_sac_eps[1]; sa.schedule = ksched; sa.key_dev = key_dev;
      return net_fwd(ctx, *pdesc, LP, pparams, pol_cur, ldo, nbuf[1].acts, hc, B, sB, &sa);
    };
    auto B2 = [&]() -> int {
      ctx->bank = bankB;
      if (twin_usable(ctx, *qdesc, LQ, qparams, qparams + nq_, B, ldc, false, false, 0, &im)) {
        r = twin_fwd(ctx, *qdesc, LQ, qparams, qparams + nq_, im, xp, ldc, nbuf[4].acts, nbuf[5].acts, qa0, qa1, B, sB);
      } else {
        r = net_fwd(ctx, *qdesc, LQ, qparams, xp, ldc, nbuf[4].acts, qa0, B, sB);
        if (!r) r = net_fwd(ctx, *qdesc, LQ, qparams + nq_, xp, ldc, nbuf[5].acts, qa1, B, sB);
      }
      if (r) return r;
      hipLaunchKernelGGL(k_sac_policy_seed, dim3(nb), dim3(256), 0, sB, qa0, qa1, lpc, d0, d1, part_p, nb, B, Bg);
      RLX_LAUNCH_CHECK();
      return RLX_OK;
    };
    auto B3 = [&]() -> int {
      ctx->bank = bankB;
      if (twin_usable(ctx, *qdesc, LQ, qparams, qparams + nq_, B, ldc, true, false, A, &im)) {
        r = twin_bwd(ctx, *qdesc, LQ, qparams, qparams + nq_, im, xp, ldc, nbuf[4].acts, nbuf[5].acts, d0, d1, nullptr, nullptr,
                     nullptr, nullptr, da0, da1, Oc, A, lda, B, nullptr, nullptr, sB);
      } else {
        TrunkOpts opt;
        opt.dx_c0 = Oc; opt.dx_nc = A; opt.dx_ld = lda;
        opt.dx_out = da0;
        r = net_bwd(ctx, *qdesc, LQ, qparams, xp, ldc, nbuf[4].acts, d0, nullptr, nullptr, B, nullptr, nullptr, &opt, sB);
        if (!r) {
          opt.dx_out = da1;
          r = net_bwd(ctx, *qdesc, LQ, qparams + nq_, xp, ldc, nbuf[5].acts, d1, nullptr, nullptr, B, nullptr, nullptr, &opt, sB);
        }
      }
      if (r) return r;
      hipLaunchKernelGGL(k_sac_policy_grad, dim3(nb_rc), dim3(256), 0, sB, hc, xp, ldc, Oc, da0, da1, lda, log_alpha, 0u, 0u,
                         scheme, dpi, B, A, AP, hp->log_std_min, hp->log_std_max, ctx->dbg_sac_eps[1], ksched, key_dev, roff, Bg);
      RLX_LAUNCH_CHECK();
      return RLX_OK;
    };
    auto B4 = [&]() -> int {
      ctx->bank = bankB;
      return net_bwd(ctx, *pdesc, LP, pparams, pol_cur, ldo, nbuf[1].acts, dpi, gp, hpart_p, B, sq1, &nsq_p, nullptr, sB);
    };
    // The host SUBMITS the chains' launches in this order, and a chain cannot run ahead of its submission: the blocks of the
    // two chains alternate (B, the longer one, first) so that both streams have work from the start -- submitted one chain
    // after the other, chain B idled until chain A's ~17 launches were out.
    struct BankReset { rlx_ctx* c; ~BankReset() { c->bank = 0; } } bank_reset{ctx};
    if ((r = B1()) || (r = C1()) || (r = A1()) || (r = B2()) || (r = A2()) || (r = B3()) || (r = A3()) || (r = B4())) return r;
    ctx->bank = 0;
    if (sA != s0) {
      RLX_HIP_TRY(hipEventRecord(ctx->sac_ev[2], sA));
      RLX_HIP_TRY(hipStreamWaitEvent(s0, ctx->sac_ev[2], 0));
    }
    // ---- metrics, entropy-coefficient gradient and its Adam step, the two plain Adam steps: ONE launch (k_sac_optimizers)
    SacFinalize F{part_c, part_p, nb, log_alpha, ga, metrics_out, B, hp->target_entropy, am, av, (const float*)(cst->sched + 8)};
    if (sharded) {
      // ONE collective per update: [policy grads | critic grads | this rank's three loss sums] are one span of the arena
      // (gp .. ga + 4; the 64-float alignment gaps ride along).  Every rank then finishes the update redundantly on the
      // reduced values -- norms, the entropy coefficient's gradient and step, the two Adam steps, Polyak.
      hipLaunchKernelGGL(k_sac_loss_sums, dim3(1), dim3(64), 0, s0, part_c, part_p, nb, ga + 1);
      RLX_LAUNCH_CHECK();
      r = dist_allreduce(ctx, gp, (int64_t)(ga + 4 - gp), 0, s0);
      if (r) return r;
      nsq_p = launch_sumsq_partials(gp, np_, sq1, s0);            // norms of the REDUCED gradients (metrics 6 / 7)
      nsq_q0 = launch_sumsq_partials(gq, 2 * nq_, sq0, s0);
      nsq_q1 = 0;
      RLX_LAUNCH_CHECK();
      F.part_c = ga + 1; F.part_p = ga + 2; F.nb = 1; F.B = Bg;
    }
    // plain Adam steps (no clipping, sac.py:95,102,108), the critics' with the Polyak update of the targets folded in
    // (sac.py:208); schedule values from `cst`
    AdamJob P{pparams, gp, pm, pv, np_, sq1, nsq_p, -1.f, metrics_out + 6, cst->sched + 0, nullptr, 0.f, 0.f};
    AdamJob Q{qparams, gq, qm, qv, 2 * nq_, sq0, nsq_q0 + nsq_q1, -1.f, metrics_out + 7, cst->sched + 4, qtarget, hp->tau, 0.f};
    BxEmitN ep, eq, eqt;
    if (emit_images) {
      sac_emit_add(ctx, &ep, *pdesc, pparams, 0, true);
      sac_emit_add(ctx, &eq, *qdesc, qparams, 0, true);
      sac_emit_add(ctx, &eq, *qdesc, qparams + nq_, nq_, true);
      sac_emit_add(ctx, &eqt, *qdesc, qtarget, 0, false);
      sac_emit_add(ctx, &eqt, *qdesc, qtarget + nq_, nq_, false);
      ctx->sac_img.valid = true;
    }
    adam_job_plan(P, ep, BxEmitN{});
    adam_job_plan(Q, eq, eqt);
    const int nb_p = P.n_tile_blocks + P.n_rest_blocks, nb_q = Q.n_tile_blocks + Q.n_rest_blocks;
    hipLaunchKernelGGL(k_sac_optimizers, dim3(1 + nb_p + nb_q), dim3(256), 0, s0, F, P, Q, nb_p, nb_q, hp->adam_b1, hp->adam_b2,
                       hp->adam_eps);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
  };
  ctx->ro_img.valid = false;
  rc = issue(st);
  if (rc) return rc;
  *opt_count_io += 1;
  return RLX_OK;
}

int rlx_dbg_set_sac_noise(rlx_ctx* ctx, const float* eps_next, const float* eps_cur) {
  RLX_REQUIRE(ctx, RLX_EINVAL, "rlx_dbg_set_sac_noise: ctx is NULL");
  ctx->dbg_sac_eps[0] = eps_next;
  ctx->dbg_sac_eps[1] = eps_cur;
  return RLX_OK;
}

int rlx_dbg_set_stamps(rlx_ctx* ctx, void* stamps) {
  RLX_REQUIRE(ctx, RLX_EINVAL, "rlx_dbg_set_stamps: ctx is NULL");
  ctx->dbg_stamps = stamps;
  return RLX_OK;
}

}  // extern "C"
