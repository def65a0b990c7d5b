// lstm_kernels.h -- device kernels of the recurrent PPO policy (ppo_lstm.hip):
//   k_ln_act<BWD>      LayerNorm + activation over rows of width D (post-LSTM latent, torso layer 1)
//   k_lstm_seq_fwd     whole-sequence LSTM forward for a block of envs (persistent over T)
//   k_lstm_seq_bwd     BPTT for the same block
//   k_seq_index / k_gather_seq_aux / k_concat2 / k_split2   sequence-minibatch plumbing
// Reference: flax.linen.OptimizedLSTMCell inside rl_x/algorithms/ppo_lstm/flax_full_jit/policy.py:56,
// used by apply_one_step :121-131 and forward_sequence :134-142 (carry * (1 - done[t-1]) before obs[t]).
// CPU twin: oracle/ppo_lstm.py.
#pragma once
#include "gemm.h"

namespace rlx {

constexpr int LSTM_H = 64;     // hidden units (4 waves <-> 4 gate blocks of 64 columns)
constexpr int LSTM_ROWS = 32;  // envs per workgroup
constexpr int LSTM_G = 4 * LSTM_H;

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------
// LayerNorm + activation over [M, D], D % 64 == 0, D <= 512.  One wave per row.
//   fwd: Y = act(LN(Z) * g + b)                       (Z kept for the backward)
//   bwd: dY (in place) -> dZ; per-block partial dg, db -> partials[grid][2*D]
// ---------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void k_ln_act(const float* __restrict__ Z, float* __restrict__ Y /*fwd out; bwd: dY -> dZ*/,
                                                const float* __restrict__ g, const float* __restrict__ be,
                                                float* __restrict__ partials, int64_t M, int D, int act) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // bwd: [4][2*D]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int NJ = D >> 6;
  float gam[8], bet[8], dg[8], db[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    gam[j] = j < NJ ? g[lane + 64 * j] : 0.f;
    bet[j] = j < NJ ? be[lane + 64 * j] : 0.f;
    dg[j] = db[j] = 0.f;
  }
  const float invD = 1.0f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < M; row += (int64_t)gridDim.x * 4) {
    float z[8], dy[8];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      z[j] = j < NJ ? Z[row * D + lane + 64 * j] : 0.f;
      if (BWD) dy[j] = j < NJ ? Y[row * D + lane + 64 * j] : 0.f;
      s += z[j];
      ss += z[j] * z[j];
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    const float mean = s * invD;
    const float rstd = rsqrtf(fmaxf(0.f, ss * invD - mean * mean) + 1e-6f);
    if (!BWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < NJ) Y[row * D + lane + 64 * j] = act_fwd((z[j] - mean) * rstd * gam[j] + bet[j], act);
    } else {
      float m1 = 0.f, m2 = 0.f, xh[8], dxh[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[j] = (z[j] - mean) * rstd;
        const float h = act_fwd(xh[j] * gam[j] + bet[j], act);
        const float d = (j < NJ) ? dy[j] * act_grad_from_out(h, act) : 0.f;
        dg[j] += d * xh[j];
        db[j] += d;
        dxh[j] = d * gam[j];
        m1 += dxh[j];
        m2 += dxh[j] * xh[j];
      }
      m1 = wave_sum(m1) * invD;
      m2 = wave_sum(m2) * invD;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < NJ) Y[row * D + lane + 64 * j] = rstd * (dxh[j] - m1 - xh[j] * m2);
    }
  }
  if (BWD) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < NJ) {
        smem[w * 2 * D + lane + 64 * j] = dg[j];
        smem[w * 2 * D + D + lane + 64 * j] = db[j];
      }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256)
      partials[(int64_t)blockIdx.x * 2 * D + i] = (smem[i] + smem[2 * D + i]) + (smem[4 * D + i] + smem[6 * D + i]);
  }
}

// ---------------------------------------------------------------------------------------
// LSTM forward over a whole sequence for 32 envs per workgroup (256 threads = 4 waves; wave w owns gate
// block w = {i, f, g, o}: 64 columns = two 32x32 MFMA tiles).  h @ Wh runs on the exact-fp32 MFMA with
// h (LDS [32][65]) as A and Wh (LDS [64][260]) as B; the input projection Gx = E_l @ Wi was computed for
// all T at once by the GEMM kernel.
//   GA   [T, n, 4H]  in: x-projection (no bias)   out: ACTIVATED gates (sig i, sig f, tanh g, sig o)
//   hout [T, n, H]   h_t (unmasked, consumed by the decoder)
//   cout [T, n, H]   c_t (unmasked)
//   hin / cin [T, n, H]  the carry actually fed to step t (after the done[t-1] reset) -- for dWh, df
//   done [T, n] (1 = episode ended AFTER step t); c0/h0 [n, H] carry valid for step 0; cT/hT: final carry
//   (masked with done[T-1] when mask_final != 0: the rollout convention, ppo_lstm.py:148-149)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lstm_seq_fwd(float* __restrict__ GA, const float* __restrict__ Wh,
                                                      const float* __restrict__ bh, const float* __restrict__ c0,
                                                      const float* __restrict__ h0, const float* __restrict__ done,
                                                      float* __restrict__ hout, float* __restrict__ cout,
                                                      float* __restrict__ hin, float* __restrict__ cin,
                                                      float* __restrict__ cT, float* __restrict__ hT, int T, int n,
                                                      int mask_final) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int HS = LSTM_H + 1, US = LSTM_G + 4;
  float* Us = smem;                         // [64][260]
  float* hs = Us + LSTM_H * US;             // [32][65]   carry h fed to the current step
  float* cs = hs + LSTM_ROWS * HS;          // [32][65]   carry c
  float* gs = cs + LSTM_ROWS * HS;          // [4][32][65] activated gates of the current step
  const int t_ = threadIdx.x, lane = t_ & 63, w = t_ >> 6, li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * LSTM_ROWS;
  for (int i = t_; i < LSTM_H * LSTM_G; i += 256) Us[(i / LSTM_G) * US + (i % LSTM_G)] = Wh[i];
  for (int i = t_; i < LSTM_ROWS * LSTM_H; i += 256) {
    const int r = i / LSTM_H, u = i % LSTM_H;
    const bool v = r0 + r < n;
    hs[r * HS + u] = v ? h0[(int64_t)(r0 + r) * LSTM_H + u] : 0.f;
    cs[r * HS + u] = v ? c0[(int64_t)(r0 + r) * LSTM_H + u] : 0.f;
  }
  float bias[2];
  bias[0] = bh[w * 64 + li];
  bias[1] = bh[w * 64 + 32 + li];
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    // record the carry fed to this step
    for (int i = t_; i < LSTM_ROWS * LSTM_H; i += 256) {
      const int r = i / LSTM_H, u = i % LSTM_H;
      if (r0 + r < n) {
        const int64_t o = ((int64_t)t * n + r0 + r) * LSTM_H + u;
        hin[o] = hs[r * HS + u];
        cin[o] = cs[r * HS + u];
      }
    }
    // gates = Gx[t] + bias + h @ Wh   (C layout: row = (r&3) + 8*(r>>2) + 4*lh, col = w*64 + 32*j + li)
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        acc[j][r] = (r0 + row < n) ? GA[((int64_t)t * n + r0 + row) * LSTM_G + w * 64 + 32 * j + li] + bias[j] : 0.f;
      }
    {
      const float* a0 = hs + li * HS + lh;
      const float* b0 = Us + lh * US + w * 64 + li;
#pragma unroll
      for (int kk = 0; kk < LSTM_H; kk += 2) {
        const float av = a0[kk];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[kk * US], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[kk * US + 32], acc[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = (w == 2) ? tanhf(acc[j][r]) : sigmoid_f(acc[j][r]);
        gs[(w * LSTM_ROWS + row) * HS + 32 * j + li] = v;
        if (r0 + row < n) GA[((int64_t)t * n + r0 + row) * LSTM_G + w * 64 + 32 * j + li] = v;
      }
    __syncthreads();
    // cell update, one (row, unit) per thread slot
    for (int i = t_; i < LSTM_ROWS * LSTM_H; i += 256) {
      const int r = i / LSTM_H, u = i % LSTM_H;
      const float ig = gs[(0 * LSTM_ROWS + r) * HS + u], fg = gs[(1 * LSTM_ROWS + r) * HS + u];
      const float gg = gs[(2 * LSTM_ROWS + r) * HS + u], og = gs[(3 * LSTM_ROWS + r) * HS + u];
      const float c2 = fg * cs[r * HS + u] + ig * gg;
      const float h2 = og * tanhf(c2);
      float m = 1.f;
      if (r0 + r < n) {
        const int64_t o = ((int64_t)t * n + r0 + r) * LSTM_H + u;
        hout[o] = h2;
        cout[o] = c2;
        m = 1.f - done[(int64_t)t * n + r0 + r];
      }
      const bool last = t == T - 1;
      const float mm = (last && !mask_final) ? 1.f : m;
      hs[r * HS + u] = h2 * mm;   // carry for the next step: reset where the episode ended after step t
      cs[r * HS + u] = c2 * mm;
    }
    __syncthreads();
  }
  if (cT && hT)
    for (int i = t_; i < LSTM_ROWS * LSTM_H; i += 256) {
      const int r = i / LSTM_H, u = i % LSTM_H;
      if (r0 + r < n) {
        cT[(int64_t)(r0 + r) * LSTM_H + u] = cs[r * HS + u];
        hT[(int64_t)(r0 + r) * LSTM_H + u] = hs[r * HS + u];
      }
    }
}

// ---------------------------------------------------------------------------------------
// BPTT.  In: activated gates GA, cout, cin, done, dh_ext [T,n,H] (gradient arriving at h_t from the decoder).
// Out: GA overwritten with dL/d(pre-activation gates) [T,n,4H] (feeds dWi, dWh, dbh and dE_l = dG @ Wi^T).
// dh_{t-1} += dG_t @ Wh^T runs on the MFMA: wave w contracts gate block w (K = 64), partials meet in LDS.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lstm_seq_bwd(float* __restrict__ GA, const float* __restrict__ Wh,
                                                      const float* __restrict__ cout, const float* __restrict__ cin,
                                                      const float* __restrict__ done, const float* __restrict__ dh_ext,
                                                      int T, int n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int HS = LSTM_H + 1;
  float* UT = smem;                          // [4][64 u'][65] : UT[w][u'][u] = Wh[u][w*64 + u']
  float* dGs = UT + 4 * LSTM_H * HS;         // [4][32][65]  gate gradients of the current step
  float* part = dGs + 4 * LSTM_ROWS * HS;    // [4][32][65]  per-gate partial of dG @ Wh^T
  float* dhr = part + 4 * LSTM_ROWS * HS;    // [32][65]     recurrent dh arriving at step t
  float* dcr = dhr + LSTM_ROWS * HS;         // [32][65]     recurrent dc arriving at step t
  const int t_ = threadIdx.x, lane = t_ & 63, w = t_ >> 6, li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * LSTM_ROWS;
  for (int i = t_; i < LSTM_H * LSTM_G; i += 256) {
    const int u = i / LSTM_G, col = i % LSTM_G;
    UT[((col >> 6) * LSTM_H + (col & 63)) * HS + u] = Wh[i];
  }
  for (int i = t_; i < LSTM_ROWS * HS; i += 256) { dhr[i] = 0.f; dcr[i] = 0.f; }
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    for (int i = t_; i < LSTM_ROWS * LSTM_H; i += 256) {
      const int r = i / LSTM_H, u = i % LSTM_H;
      float di = 0.f, df = 0.f, dg = 0.f, dob = 0.f, dcp = 0.f;
      if (r0 + r < n) {
        const int64_t row = (int64_t)t * n + r0 + r;
        const float* ga = GA + row * LSTM_G;
        const float ig = ga[u], fg = ga[LSTM_H + u], gg = ga[2 * LSTM_H + u], og = ga[3 * LSTM_H + u];
        const float tc = tanhf(cout[row * LSTM_H + u]);
        const float dh = dh_ext[row * LSTM_H + u] + dhr[r * HS + u];
        const float dc = dcr[r * HS + u] + dh * og * (1.f - tc * tc);
        di = dc * gg * ig * (1.f - ig);
        df = dc * cin[row * LSTM_H + u] * fg * (1.f - fg);
        dg = dc * ig * (1.f - gg * gg);
        dob = dh * tc * og * (1.f - og);
        dcp = dc * fg;
        float* go = GA + row * LSTM_G;
        go[u] = di; go[LSTM_H + u] = df; go[2 * LSTM_H + u] = dg; go[3 * LSTM_H + u] = dob;
      }
      dGs[(0 * LSTM_ROWS + r) * HS + u] = di;
      dGs[(1 * LSTM_ROWS + r) * HS + u] = df;
      dGs[(2 * LSTM_ROWS + r) * HS + u] = dg;
      dGs[(3 * LSTM_ROWS + r) * HS + u] = dob;
      // gradient w.r.t. the carry fed to step t; the carry was (carry_out[t-1] * (1 - done[t-1]))
      float m = 0.f;
      if (t > 0 && r0 + r < n) m = 1.f - done[(int64_t)(t - 1) * n + r0 + r];
      dcr[r * HS + u] = dcp * m;
      dhr[r * HS + u] = m;  // holds the mask until the MFMA partials are folded in below
    }
    __syncthreads();
    {
      f32x16 acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      const float* a0 = dGs + (w * LSTM_ROWS + li) * HS + lh;       // A[i = row][k = u']
      const float* b0 = UT + (w * LSTM_H + lh) * HS + li;           // B[k = u'][j = u]
#pragma unroll
      for (int kk = 0; kk < LSTM_H; kk += 2) {
        const float av = a0[kk];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[kk * HS], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[kk * HS + 32], acc[1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          part[(w * LSTM_ROWS + row) * HS + 32 * j + li] = acc[j][r];
        }
    }
    __syncthreads();
    for (int i = t_; i < LSTM_ROWS * LSTM_H; i += 256) {
      const int r = i / LSTM_H, u = i % LSTM_H;
      const float s = (part[(0 * LSTM_ROWS + r) * HS + u] + part[(1 * LSTM_ROWS + r) * HS + u]) +
                      (part[(2 * LSTM_ROWS + r) * HS + u] + part[(3 * LSTM_ROWS + r) * HS + u]);
      dhr[r * HS + u] = s * dhr[r * HS + u];
    }
    __syncthreads();
  }
}

// idx_flat[t*ne + e] = t*N + env_idx[e]   (rows of a sequence minibatch in the flattened [T*N] rollout arrays)
__global__ void k_seq_index(const int32_t* __restrict__ env_idx, int32_t* __restrict__ idx_flat, int T, int ne, int N) {
  const int64_t total = (int64_t)T * ne;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    idx_flat[i] = (int32_t)((i / ne) * N + env_idx[i % ne]);
}

// done_mb[t, e] = dones[t, env_idx[e]];  c0_mb / h0_mb[e] = carry0[env_idx[e]]
__global__ void k_gather_seq_aux(const float* __restrict__ dones, const float* __restrict__ c0, const float* __restrict__ h0,
                                 const int32_t* __restrict__ env_idx, float* __restrict__ done_mb,
                                 float* __restrict__ c0_mb, float* __restrict__ h0_mb, int T, int ne, int N) {
  const int64_t nd = (int64_t)T * ne, nc = (int64_t)ne * LSTM_H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nd + nc; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nd) done_mb[i] = dones[(i / ne) * N + env_idx[i % ne]];
    else {
      const int64_t j = i - nd;
      const int64_t src = (int64_t)env_idx[j / LSTM_H] * LSTM_H + (j % LSTM_H);
      c0_mb[j] = c0[src];
      h0_mb[j] = h0[src];
    }
  }
}

// out[M, Da+Db] = [a | b]
__global__ void k_concat2(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t M,
                          int Da, int Db) {
  const int D = Da + Db;
  const int64_t total = M * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    out[i] = c < Da ? a[r * Da + c] : b[r * Db + (c - Da)];
  }
}

// a[M, Da], b[M, Db] = split(in[M, ld] columns [0, Da+Db))
__global__ void k_split2(const float* __restrict__ in, int ld, float* __restrict__ a, float* __restrict__ b, int64_t M,
                         int Da, int Db) {
  const int D = Da + Db;
  const int64_t total = M * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    const float v = in[r * ld + c];
    if (c < Da) a[r * Da + c] = v;
    else b[r * Db + (c - Da)] = v;
  }
}

}  // namespace rlx
