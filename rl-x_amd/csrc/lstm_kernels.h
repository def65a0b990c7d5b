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
#include "gemm_bx.h"
#include "ln_kernels.h"

namespace rlx {

constexpr int LSTM_H = 64;     // hidden units (4 waves x 16 units)
constexpr int LSTM_ROWS = 16;  // envs per workgroup
constexpr int LSTM_G = 4 * LSTM_H;


// ---------------------------------------------------------------------------------------
// LSTM forward over a whole sequence, 16 envs per workgroup (256 threads = 4 waves), persistent over T.
// Wave w owns hidden units [16w, 16w+16) of ALL FOUR gates: four 16x16 accumulator tiles of the exact-fp32
// MFMA (v_mfma_f32_16x16x4_f32) whose C layouts coincide (row = 4*(lane>>4)+r, unit = 16w + (lane&15)), so the
// cell update c' = f c + i g, h' = o tanh(c') is register-local: the carry never leaves VGPRs.  Wh lives in
// registers for the whole sequence (64 B-fragments per lane); only h goes through LDS (double buffered,
// [16][68] -> conflict-free ds_read_b128 A-fragments), one barrier per step.  The x-projection
// Gx = E_l @ Wi was computed for all T by the GEMM kernel and is prefetched one step ahead.
//   GA   [T, n, 4H]  in: x-projection (no bias)   out: ACTIVATED gates (sig i, sig f, tanh g, sig o)
//   hout [T, n, H]   h_t (unmasked, consumed by the decoder);  cout [T, n, H]  c_t (unmasked)
//   hin / cin [T, n, H]  the carry actually fed to step t (after the done[t-1] reset) -- for dWh, df
//   done [T, n] (1 = episode ended AFTER step t); c0/h0 [n, H] carry valid for step 0; cT/hT: final carry
//   (masked with done[T-1] when mask_final != 0: the rollout convention, ppo_lstm.py:148-149)
// ---------------------------------------------------------------------------------------
typedef float lstm_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_fast(float x) {
  const float xc = fminf(fmaxf(x, -30.f), 30.f);
  return rcp_fast(1.0f + __expf(-xc));
}

// BF: the recurrent product h @ Wh on the half-precision matrix pipe with split-fp32 operands (gemm_bx.h): v_mfma_f32_16x16x32_f16
// has the C layout of the f32 form (row 4 * (lane >> 4) + r, unit lane & 15), so the register-local cell update is unchanged;
// 2 k-steps x 3 plane products x 4 gates = 24 MFMAs of ~17 cycles instead of 64 of 32.  Wh (times X_WSCALE) lives in VGPRs as two
// fp16 planes per gate and k-step (64 registers); h (|h| < 1, times X_ASCALE) goes through LDS as two fp16 planes WRITTEN BY ITS
// PRODUCER LANES (4 values each -- splitting the fragment on the consumer side would cost more VALU cycles than the MFMAs save),
// rows 144 B apart (conflict-free ds_read_b128 fragments).  The accumulators hold X_ASCALE * X_WSCALE * (bias + h Wh); the scale leaves in the
// fused multiply-add that joins the x-projection.
__device__ __forceinline__ void lstm_split1(float x, uint16_t (&h)[X_NP]) {
  uint32_t p0, p1;
  bx_split2(x * X_ASCALE, 0.f, p0, p1);
  h[0] = (uint16_t)(p0 & 0xffffu);
  h[1] = (uint16_t)(p1 & 0xffffu);
}
constexpr int LSTM_HB = 72;                              // fp16 per LDS row of an h plane (64 + 8 pad = 144 B)
constexpr int LSTM_HPLANE = LSTM_ROWS * LSTM_HB * 2;     // bytes per plane
constexpr int LSTM_HBUF = X_NP * LSTM_HPLANE;            // bytes per (double-buffered) h image

template <bool FULL, bool BF = false>   // FULL: every row tile of the launch has 16 valid rows -> no per-row guards (exec-masked branches)
__global__ __launch_bounds__(256) void k_lstm_seq_fwd(float* __restrict__ GA, const float* __restrict__ Wh,
                                                      const float* __restrict__ bh, const float* __restrict__ c0,
                                                      const float* __restrict__ h0, const float* __restrict__ done,
                                                      float* __restrict__ hout, float* __restrict__ cout,
                                                      float* __restrict__ hin, float* __restrict__ cin,
                                                      float* __restrict__ cT, float* __restrict__ hT, int T, int n,
                                                      int mask_final) {
  constexpr int HS = 68;
  __shared__ __attribute__((aligned(16))) char smem_h[BF ? 2 * LSTM_HBUF : 2 * LSTM_ROWS * HS * 4];
  float (*hs)[LSTM_ROWS * HS] = reinterpret_cast<float (*)[LSTM_ROWS * HS]>(smem_h);   // !BF: fp32 h, double buffered
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * LSTM_ROWS;
  const int u = 16 * w + col;
  // B fragments: MFMA step s of lane group q contracts k = 16q + s
  float Bv[4][16];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) Bv[g][s_] = Wh[(16 * q + s_) * LSTM_G + g * LSTM_H + u];
  // BF: element e of the 8-wide fp16 operand of k-step ks <-> k = 16q + 8ks + e, for A (LDS order) and B alike
  u32x4 Bp[4][2][X_NP];
  if (BF) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          uint32_t p0, p1;
          bx_split2(Bv[g][8 * ks + 2 * m] * X_WSCALE, Bv[g][8 * ks + 2 * m + 1] * X_WSCALE, p0, p1);
          Bp[g][ks][0][m] = p0;
          Bp[g][ks][1][m] = p1;
        }
  }
  auto store_h = [&](int buf, int row, float v) {   // the carry of (row, u) for the next step's product
    if (BF) {
      uint16_t hb[X_NP];
      lstm_split1(v, hb);
#pragma unroll
      for (int p = 0; p < X_NP; ++p)
        *reinterpret_cast<uint16_t*>(smem_h + buf * LSTM_HBUF + p * LSTM_HPLANE + (row * LSTM_HB + u) * 2) = hb[p];
    } else {
      hs[buf][row * HS + u] = v;
    }
  };
  float bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bias[g] = bh[g * LSTM_H + u] * (BF ? X_WSCALE * X_ASCALE : 1.f);
  float c[4], hp[4];
  bool valid[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + 4 * q + r;
    valid[r] = FULL || row < n;
    c[r] = valid[r] ? c0[(int64_t)row * LSTM_H + u] : 0.f;
    hp[r] = valid[r] ? h0[(int64_t)row * LSTM_H + u] : 0.f;
    store_h(0, 4 * q + r, hp[r]);
  }
  // Memory operations retire in order on the VM counter (CDNA4 counts stores on it too), and hipcc's wait at the top of a loop
  // body is conservative across the back edge.  The time loop is therefore unrolled by two with two NAMED operand sets: step t
  // computes from set `cur` while the x-projection of step t + 1 is requested into set `nxt` IN FRONT of this step's 24 stores,
  // unconditionally (the index is clamped: no branch around the loads) -- the wait for `nxt` one step later is then a counted
  // vmcnt that leaves this step's stores in flight.  (Loads behind the stores, or one register set copied at the end of the
  // body, made every step sit out a full memory round trip: ~2 us per step instead of the ~0.5 us its arithmetic needs.)
  struct FwdOps { float gx[4][4], dn[4]; };
  FwdOps opA, opB;
  auto fetch = [&](int tt, FwdOps& o) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t ro = (int64_t)tt * n + r0 + 4 * q + r;
      o.dn[r] = valid[r] ? done[ro] : 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) o.gx[g][r] = valid[r] ? GA[ro * LSTM_G + g * LSTM_H + u] : 0.f;
    }
  };
  fetch(0, opA);
  __syncthreads();
  auto step = [&](int t, const FwdOps& oc, FwdOps& on) {
    const int cur = t & 1;
    lstm_f4 acc[4];
    fetch(t + 1 < T ? t + 1 : T - 1, on);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (valid[r]) {   // the carry fed to this step
        const int64_t o = ((int64_t)t * n + r0 + 4 * q + r) * LSTM_H + u;
        hin[o] = hp[r];
        cin[o] = c[r];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g][r] = bias[g];
    }
    if (BF) {
      u32x4 Ap[2][X_NP];
      const char* hb = smem_h + cur * LSTM_HBUF + (col * LSTM_HB + 16 * q) * 2;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < X_NP; ++p) Ap[ks][p] = *reinterpret_cast<const u32x4*>(hb + p * LSTM_HPLANE + 16 * ks);
      // smallest products first; the four gate accumulators alternate so no MFMA waits on its predecessor
#define LSTM_BX_STEP(P, Q)                                                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int g = 0; g < 4; ++g)                    \
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Ap[ks][P]),                         \
                                                      __builtin_bit_cast(f16x8, Bp[g][ks][Q]), acc[g], 0, 0, 0);
      LSTM_BX_STEP(0, 1)
      LSTM_BX_STEP(1, 0)
      LSTM_BX_STEP(0, 0)
#undef LSTM_BX_STEP
    } else {
      const lstm_f4* ap = reinterpret_cast<const lstm_f4*>(hs[cur] + col * HS + 16 * q);
      lstm_f4 a4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a4[j] = ap[j];
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s_ >> 2][s_ & 3], Bv[g][s_], acc[g], 0, 0, 0);
    }
    // the x-projection of this step (fetched during the PREVIOUS step) joins only now, so its loads had a whole
    // step to land; adding it ahead of the MFMAs made every step wait for HBM
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[g][r] = BF ? fmaf(acc[g][r], X_WINV * X_AINV, oc.gx[g][r]) : acc[g][r] + oc.gx[g][r];
    float dnc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dnc[r] = oc.dn[r];
    const bool keep = (t == T - 1) && !mask_final;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ig = sigmoid_fast(acc[0][r]), fg = sigmoid_fast(acc[1][r]);
      const float gg = act_fwd_t<RLX_ACT_TANH>(acc[2][r]), og = sigmoid_fast(acc[3][r]);
      const float c2 = fg * c[r] + ig * gg;
      const float h2 = og * act_fwd_t<RLX_ACT_TANH>(c2);
      if (valid[r]) {
        const int64_t ro = (int64_t)t * n + r0 + 4 * q + r;
        float* ga = GA + ro * LSTM_G + u;
        ga[0] = ig; ga[LSTM_H] = fg; ga[2 * LSTM_H] = gg; ga[3 * LSTM_H] = og;
        hout[ro * LSTM_H + u] = h2;
        cout[ro * LSTM_H + u] = c2;
      }
      const float mm = keep ? 1.f : 1.f - dnc[r];   // carry for the next step: reset where the episode ended after step t
      c[r] = c2 * mm;
      hp[r] = h2 * mm;
      store_h(cur ^ 1, 4 * q + r, hp[r]);
    }
    __syncthreads();
  };
  int t = 0;
  for (; t + 1 < T; t += 2) {
    step(t, opA, opB);
    step(t + 1, opB, opA);
  }
  if (t < T) step(t, opA, opB);
  if (cT && hT) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (valid[r]) {
        cT[(int64_t)(r0 + 4 * q + r) * LSTM_H + u] = c[r];
        hT[(int64_t)(r0 + 4 * q + r) * LSTM_H + u] = hp[r];
      }
  }
}

// ---------------------------------------------------------------------------------------
// BPTT, same ownership (wave w <-> units [16w, 16w+16), lane <-> 4 rows x 1 unit x 4 gates).
// In: activated gates GA, cout, cin, done, dh_ext [T,n,H] (gradient arriving at h_t from the decoder).
// Out: GA overwritten with dL/d(pre-activation gates) [T,n,4H] (feeds dWi, dWh, dbh and dE_l = dG @ Wi^T).
// dh_{t-1} = dG_t @ Wh^T on the MFMA: A = dG_t (LDS, [16][260], double buffered), B = Wh rows of the wave's
// units kept in registers (64 fragments per lane); the result lands in the C layout = the lane's own (row, unit)
// slots, so dh / dc stay in registers.  Everything step t-1 needs is prefetched during step t.
// ---------------------------------------------------------------------------------------
// BF: the product on the half-precision pipe with split operands (gemm_bx.h): 8 k-steps x 3 plane products of
// v_mfma_f32_16x16x32_f16 (24 instructions of ~17 cycles) instead of 64 v_mfma_f32_16x16x4_f32 of 32 -- on a single-wave-per-SIMD
// latency chain every cycle of the step counts.  dG (times the pass's gradient scale gs) goes through LDS as two fp16 planes
// written by its producer lanes, rows 544 B apart (conflict-free ds_read_b128 fragments); Wh^T (times X_WSCALE) lives in VGPRs
// as planes.  Element e of the 8-wide operand of k-step ks <-> k = 32 ks + 8 q + e on both sides.
constexpr int LSTM_GB = LSTM_G + 16;                     // fp16 per LDS row of a dG plane
constexpr int LSTM_GPLANE = LSTM_ROWS * LSTM_GB * 2;     // bytes per plane
constexpr int LSTM_GBUF = X_NP * LSTM_GPLANE;            // bytes per (double-buffered) dG image

template <bool FULL, bool BF = false>
__global__ __launch_bounds__(256) void k_lstm_seq_bwd(float* __restrict__ GA, const float* __restrict__ Wh,
                                                      const float* __restrict__ cout, const float* __restrict__ cin,
                                                      const float* __restrict__ done, const float* __restrict__ dh_ext,
                                                      int T, int n, float gs) {
  constexpr int GS = LSTM_G + 4;
  __shared__ __attribute__((aligned(16))) char smem_g[BF ? 2 * LSTM_GBUF : 2 * LSTM_ROWS * GS * 4];
  float (*dGs)[LSTM_ROWS * GS] = reinterpret_cast<float (*)[LSTM_ROWS * GS]>(smem_g);   // !BF: fp32 dG, double buffered
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * LSTM_ROWS;
  const int u = 16 * w + col;
  // B[k][j] = Wh[16w + j][k]; exact form: lane group q contracts k = 64q + s (gate q, unit s)
  lstm_f4 Bv[BF ? 1 : 16];
  u32x4 Bp[BF ? 8 : 1][X_NP];
  if (BF) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const lstm_f4* bp = reinterpret_cast<const lstm_f4*>(Wh + (int64_t)u * LSTM_G + 32 * ks + 8 * q);
      const lstm_f4 b0 = bp[0], b1 = bp[1];
      const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        uint32_t p0, p1;
        bx_split2(bv[2 * m] * X_WSCALE, bv[2 * m + 1] * X_WSCALE, p0, p1);
        Bp[ks][0][m] = p0;
        Bp[ks][1][m] = p1;
      }
    }
  } else {
    const lstm_f4* bp = reinterpret_cast<const lstm_f4*>(Wh + (int64_t)u * LSTM_G + 64 * q);
#pragma unroll
    for (int j = 0; j < 16; ++j) Bv[j] = bp[j];
  }
  const float gso = X_WINV / gs;
  bool valid[4];
  float dh[4], dc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    valid[r] = FULL || r0 + 4 * q + r < n;
    dh[r] = dc[r] = 0.f;
  }
  // (two named operand sets and a time loop unrolled by two, the next step's operands requested in front of this step's 16
  //  stores: see k_lstm_seq_fwd)
  struct BwdOps { float ga[4][4], co[4], ci[4], de[4], dp[4]; };
  BwdOps opA, opB;
  auto fetch = [&](int tt, BwdOps& o) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t ro = (int64_t)tt * n + r0 + 4 * q + r;
      o.co[r] = valid[r] ? cout[ro * LSTM_H + u] : 0.f;
      o.ci[r] = valid[r] ? cin[ro * LSTM_H + u] : 0.f;
      o.de[r] = valid[r] ? dh_ext[ro * LSTM_H + u] : 0.f;
      o.dp[r] = (valid[r] && tt > 0) ? done[ro - n] : 1.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) o.ga[g][r] = valid[r] ? GA[ro * LSTM_G + g * LSTM_H + u] : 0.f;
    }
  };
  fetch(T - 1, opA);
  auto step = [&](int t, const BwdOps& oc, BwdOps& on) {
    const int cur = t & 1;
    fetch(t > 0 ? t - 1 : 0, on);
    float mk[4], dg_[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ig = oc.ga[0][r], fg = oc.ga[1][r], gg = oc.ga[2][r], og = oc.ga[3][r];
      const float tc = act_fwd_t<RLX_ACT_TANH>(oc.co[r]);
      const float dht = oc.de[r] + dh[r];
      const float dct = dc[r] + dht * og * (1.f - tc * tc);
      dg_[0][r] = dct * gg * ig * (1.f - ig);
      dg_[1][r] = dct * oc.ci[r] * fg * (1.f - fg);
      dg_[2][r] = dct * ig * (1.f - gg * gg);
      dg_[3][r] = dht * tc * og * (1.f - og);
      // the carry fed to step t was carry_out[t-1] * (1 - done[t-1]); no gradient flows into the initial carry
      mk[r] = 1.f - oc.dp[r];
      dc[r] = dct * fg * mk[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (valid[r]) {
        float* go = GA + ((int64_t)t * n + r0 + 4 * q + r) * LSTM_G + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) go[g * LSTM_H] = dg_[g][r];
      }
      if (!BF) {
#pragma unroll
        for (int g = 0; g < 4; ++g) dGs[cur][(4 * q + r) * GS + g * LSTM_H + u] = dg_[g][r];
      }
    }
    if (BF) {
      // rows 4q + 2m and 4q + 2m + 1 of column k = g * 64 + u: one split per pair, four 2-byte stores
      char* gb = smem_g + cur * LSTM_GBUF;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          uint32_t p0, p1;
          bx_split2(dg_[g][2 * m] * gs, dg_[g][2 * m + 1] * gs, p0, p1);
          char* d0 = gb + ((4 * q + 2 * m) * LSTM_GB + g * LSTM_H + u) * 2;
          *reinterpret_cast<uint16_t*>(d0) = (uint16_t)p0;
          *reinterpret_cast<uint16_t*>(d0 + LSTM_GB * 2) = (uint16_t)(p0 >> 16);
          *reinterpret_cast<uint16_t*>(d0 + LSTM_GPLANE) = (uint16_t)p1;
          *reinterpret_cast<uint16_t*>(d0 + LSTM_GPLANE + LSTM_GB * 2) = (uint16_t)(p1 >> 16);
        }
    }
    __syncthreads();
    if (BF) {
      lstm_f4 ac[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ac[i] = lstm_f4{0.f, 0.f, 0.f, 0.f};
      const char* ab = smem_g + cur * LSTM_GBUF + (col * LSTM_GB + 8 * q) * 2;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(ab + 64 * ks), a1 = *reinterpret_cast<const u32x4*>(ab + LSTM_GPLANE + 64 * ks);
        // (four accumulators in rotation: no MFMA waits on its predecessor)
        ac[(3 * ks) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, Bp[ks][1]), ac[(3 * ks) & 3], 0, 0, 0);
        ac[(3 * ks + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, Bp[ks][0]), ac[(3 * ks + 1) & 3], 0, 0, 0);
        ac[(3 * ks + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, Bp[ks][0]), ac[(3 * ks + 2) & 3], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[r] = ((ac[0][r] + ac[1][r]) + (ac[2][r] + ac[3][r])) * gso * mk[r];
    } else {
      lstm_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const lstm_f4* ap = reinterpret_cast<const lstm_f4*>(dGs[cur] + col * GS + 64 * q);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const lstm_f4 a = ap[j];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], Bv[j][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], Bv[j][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], Bv[j][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], Bv[j][3], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[r] = (acc0[r] + acc1[r]) * mk[r];
    }
  };
  int t = T - 1;
  for (; t >= 1; t -= 2) {
    step(t, opA, opB);
    step(t - 1, opB, opA);
  }
  if (t == 0) step(0, opA, opB);
}

// ---------------------------------------------------------------------------------------
// GRU (flax.linen.GRUCell inside rl_x/algorithms/ppo_gru/flax_full_jit/policy.py:52, used by apply_one_step / forward_sequence
// exactly like the LSTM): r = sig(gx_r + h Whr), z = sig(gx_z + h Whz), n = tanh(gx_n + r * (h Whn + bhn)),
// h' = (1 - z) n + z h, with gx = E_l @ Wi + bi computed for all T by the GEMM kernel.  Same design as the LSTM kernels:
// 16 envs per workgroup, wave w owns units [16w, 16w+16) of the three gates (three 16x16 MFMA tiles with identical C
// layouts), h stays in registers, the recurrent kernels Wh_rz [H,2H] / Wh_n [H,H] live in VGPRs for the whole sequence.
//   GX  [T, n, 3H]  x-projection incl. bias (read only)
//   GA  [T, n, 4H]  out: r, z, n (activated) and hnp = h Whn + bhn   (saved for the backward)
//   hout [T, n, H] h_t (unmasked);  hin [T, n, H] the carry fed to step t;  done / h0 / hT / mask_final as the LSTM
// ---------------------------------------------------------------------------------------
template <bool FULL>
__global__ __launch_bounds__(256) void k_gru_seq_fwd(const float* __restrict__ GX, float* __restrict__ GA,
                                                     const float* __restrict__ Whrz, const float* __restrict__ Whn,
                                                     const float* __restrict__ bhn, const float* __restrict__ h0,
                                                     const float* __restrict__ done, float* __restrict__ hout,
                                                     float* __restrict__ hin, float* __restrict__ hT, int T, int n,
                                                     int mask_final) {
  constexpr int HS = 68, G3 = 3 * LSTM_H;
  __shared__ __attribute__((aligned(16))) float hs[2][LSTM_ROWS * HS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * LSTM_ROWS;
  const int u = 16 * w + col;
  float Bv[3][16];   // lane group q contracts k = 16q + s
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) {
    Bv[0][s_] = Whrz[(16 * q + s_) * 2 * LSTM_H + u];
    Bv[1][s_] = Whrz[(16 * q + s_) * 2 * LSTM_H + LSTM_H + u];
    Bv[2][s_] = Whn[(16 * q + s_) * LSTM_H + u];
  }
  const float bn = bhn[u];
  float hp[4];
  bool valid[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + 4 * q + r;
    valid[r] = FULL || row < n;
    hp[r] = valid[r] ? h0[(int64_t)row * LSTM_H + u] : 0.f;
    hs[0][(4 * q + r) * HS + u] = hp[r];
  }
  float gx[3][4], dn[4];
#define GRU_FWD_PREFETCH(tt)                                                               \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                          \
    const int64_t ro = (int64_t)(tt) * n + r0 + 4 * q + r;                                 \
    dn[r] = valid[r] ? done[ro] : 0.f;                                                     \
    _Pragma("unroll") for (int g = 0; g < 3; ++g)                                          \
      gx[g][r] = valid[r] ? GX[ro * G3 + g * LSTM_H + u] : 0.f;                            \
  }
  GRU_FWD_PREFETCH(0)
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    lstm_f4 acc[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (valid[r]) hin[((int64_t)t * n + r0 + 4 * q + r) * LSTM_H + u] = hp[r];
      acc[0][r] = 0.f; acc[1][r] = 0.f; acc[2][r] = bn;
    }
    const lstm_f4* ap = reinterpret_cast<const lstm_f4*>(hs[cur] + col * HS + 16 * q);
    lstm_f4 a4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a4[j] = ap[j];
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_)
#pragma unroll
      for (int g = 0; g < 3; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s_ >> 2][s_ & 3], Bv[g][s_], acc[g], 0, 0, 0);
    const bool keep = (t == T - 1) && !mask_final;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float rg = sigmoid_fast(acc[0][r] + gx[0][r]), zg = sigmoid_fast(acc[1][r] + gx[1][r]);
      const float hnp = acc[2][r];
      const float ng = act_fwd_t<RLX_ACT_TANH>(gx[2][r] + rg * hnp);
      const float h2 = (1.f - zg) * ng + zg * hp[r];
      if (valid[r]) {
        const int64_t ro = (int64_t)t * n + r0 + 4 * q + r;
        float* ga = GA + ro * LSTM_G + u;
        ga[0] = rg; ga[LSTM_H] = zg; ga[2 * LSTM_H] = ng; ga[3 * LSTM_H] = hnp;
        hout[ro * LSTM_H + u] = h2;
      }
      const float mm = keep ? 1.f : 1.f - dn[r];
      hp[r] = h2 * mm;
      hs[cur ^ 1][(4 * q + r) * HS + u] = hp[r];
    }
    if (t + 1 < T) { GRU_FWD_PREFETCH(t + 1) }
    __syncthreads();
  }
#undef GRU_FWD_PREFETCH
  if (hT) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (valid[r]) hT[(int64_t)(r0 + 4 * q + r) * LSTM_H + u] = hp[r];
  }
}

// BPTT of the GRU.  In: GA (r, z, n, hnp), hin, done, dh_ext.  Out (dense, for the GEMM kernels):
//   dGX  [T,n,3H] = d/d(x-projection) = (dr_pre, dz_pre, dn_pre)          -> dWi, dbi, dE_l
//   dGRZ [T,n,2H] = (dr_pre, dz_pre),  dHN [T,n,H] = d hnp = dn_pre * r    -> dWh_rz, dWh_n, dbhn
// dh_{t-1} = dh z + (dr_pre, dz_pre) @ Wh_rz^T + dhnp @ Wh_n^T: K = 3H on the MFMA, lane group q contracts 48 of them.
template <bool FULL>
__global__ __launch_bounds__(256) void k_gru_seq_bwd(const float* __restrict__ GA, const float* __restrict__ Whrz,
                                                     const float* __restrict__ Whn, const float* __restrict__ hin,
                                                     const float* __restrict__ done, const float* __restrict__ dh_ext,
                                                     float* __restrict__ dGX, float* __restrict__ dGRZ,
                                                     float* __restrict__ dHN, int T, int n) {
  constexpr int G3 = 3 * LSTM_H, GS = G3 + 4;
  __shared__ __attribute__((aligned(16))) float dGs[2][LSTM_ROWS * GS];   // (dr_pre | dz_pre | dhnp) rows
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * LSTM_ROWS;
  const int u = 16 * w + col;
  // B[k][j] = d h_prev[u=16w+j] / d (column k of (dr_pre|dz_pre|dhnp)): Wh_rz[u][k] for k < 2H, Wh_n[u][k-2H] above
  float Bv[48];
#pragma unroll
  for (int s_ = 0; s_ < 48; ++s_) {
    const int k = 48 * q + s_;
    Bv[s_] = k < 2 * LSTM_H ? Whrz[(int64_t)u * 2 * LSTM_H + k] : Whn[(int64_t)u * LSTM_H + (k - 2 * LSTM_H)];
  }
  bool valid[4];
  float dh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    valid[r] = FULL || r0 + 4 * q + r < n;
    dh[r] = 0.f;
  }
  float ga[4][4], hi[4], de[4], dp[4];
#define GRU_BWD_PREFETCH(tt)                                                               \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                          \
    const int64_t ro = (int64_t)(tt) * n + r0 + 4 * q + r;                                 \
    hi[r] = valid[r] ? hin[ro * LSTM_H + u] : 0.f;                                         \
    de[r] = valid[r] ? dh_ext[ro * LSTM_H + u] : 0.f;                                      \
    dp[r] = (valid[r] && (tt) > 0) ? done[ro - n] : 1.f;                                   \
    _Pragma("unroll") for (int g = 0; g < 4; ++g)                                          \
      ga[g][r] = valid[r] ? GA[ro * LSTM_G + g * LSTM_H + u] : 0.f;                        \
  }
  GRU_BWD_PREFETCH(T - 1)
  for (int t = T - 1; t >= 0; --t) {
    const int cur = t & 1;
    float mk[4], dhz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float rg = ga[0][r], zg = ga[1][r], ng = ga[2][r], hnp = ga[3][r];
      const float dht = de[r] + dh[r];
      const float dn_pre = dht * (1.f - zg) * (1.f - ng * ng);
      const float dz_pre = dht * (hi[r] - ng) * zg * (1.f - zg);
      const float dr_pre = dn_pre * hnp * rg * (1.f - rg);
      const float dhnp = dn_pre * rg;
      mk[r] = 1.f - dp[r];          // the carry fed to step t was h_{t-1} * (1 - done[t-1]); none into the initial carry
      dhz[r] = dht * zg;
      if (valid[r]) {
        const int64_t ro = (int64_t)t * n + r0 + 4 * q + r;
        dGX[ro * G3 + u] = dr_pre; dGX[ro * G3 + LSTM_H + u] = dz_pre; dGX[ro * G3 + 2 * LSTM_H + u] = dn_pre;
        dGRZ[ro * 2 * LSTM_H + u] = dr_pre; dGRZ[ro * 2 * LSTM_H + LSTM_H + u] = dz_pre;
        dHN[ro * LSTM_H + u] = dhnp;
      }
      float* ds = dGs[cur] + (4 * q + r) * GS + u;
      ds[0] = dr_pre; ds[LSTM_H] = dz_pre; ds[2 * LSTM_H] = dhnp;
    }
    if (t > 0) { GRU_BWD_PREFETCH(t - 1) }
    __syncthreads();
    lstm_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const lstm_f4* ap = reinterpret_cast<const lstm_f4*>(dGs[cur] + col * GS + 48 * q);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const lstm_f4 a = ap[j];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], Bv[4 * j + 0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], Bv[4 * j + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], Bv[4 * j + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], Bv[4 * j + 3], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[r] = (dhz[r] + acc0[r] + acc1[r]) * mk[r];
  }
#undef GRU_BWD_PREFETCH
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
