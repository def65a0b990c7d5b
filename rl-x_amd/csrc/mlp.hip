// mlp.hip -- MLP actor/critic layers for gfx950: forward, input-gradient and
// weight-gradient kernels.  Replaces the XLA fusions of
//   Policy.__call__  rl_x/algorithms/ppo/flax/policy.py:31-40, ppo/flax_full_jit/policy.py:30-42
//   Critic.__call__  rl_x/algorithms/ppo/flax/critic.py:22-30,  ppo/flax_full_jit/critic.py:21-32
// and their reverse-mode gradients (jax.value_and_grad, ppo/flax/ppo.py:189,202-210).
// CPU twin: oracle/nets.py.
#include "mlp.h"
#include "ln_kernels.h"

namespace rlx {

// =======================================================================================
// First layer, small in_dim (<= 32): Dense + optional LayerNorm + activation on the VALU,
// forward and backward.  K = obs_dim is 17 for the benchmark: 2.5 % of the FLOPs, so this
// kernel should run at HBM speed (forward writes the [M,H] activation once; backward reads
// dH once and writes dZ1 in place, recomputing the forward instead of storing z / xhat).
//
// 512-thread workgroups (8 waves) share one LDS copy of W1/b/ln params; each wave owns whole
// rows (lane l owns columns l + 64 j), so LayerNorm statistics are a pure in-wave DPP
// reduction -- no barriers in the row loop.  The next row group's x / dH loads are issued
// before the current group's math (register prefetch) so HBM latency overlaps the VALU work.
// =======================================================================================
constexpr int L1_R_FWD = 4, L1_R_BWD = 2;  // rows per wave iteration (bwd holds 3 row-sized register sets)
constexpr int L1_MAXJ = 8;  // H <= 512
constexpr int L1_THREADS = 512;

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

struct L1Args {
  const float* X;
  float* H;  // fwd: out; bwd: dH in -> dZ out (in place)
  int64_t M;
  int O, Hd, NJ;
};

template <bool BWD, int ACT, bool LN, int L1R>
__device__ __forceinline__ void l1_rows(const L1Args& a, const float* __restrict__ Ws, const float* __restrict__ bs,
                                        const float* __restrict__ gs, const float* __restrict__ bes,
                                        float (&dg)[L1_MAXJ], float (&dbe)[L1_MAXJ], int lane, int64_t wave_g,
                                        int64_t nwaves) {
  const float invH = 1.0f / (float)a.Hd;
  const int NJ = a.NJ;
  float xv[L1R], dh[L1R][L1_MAXJ];
  // prologue: loads of the first row group
  int64_t row0 = wave_g * L1R;
#pragma unroll
  for (int r = 0; r < L1R; ++r) {
    xv[r] = (lane < a.O && row0 + r < a.M) ? a.X[(row0 + r) * a.O + lane] : 0.f;
    if (BWD) {
#pragma unroll
      for (int j = 0; j < L1_MAXJ; ++j)
        dh[r][j] = (j < NJ && row0 + r < a.M) ? a.H[(row0 + r) * a.Hd + lane + 64 * j] : 0.f;
    }
  }
  for (; row0 < a.M; row0 += nwaves * L1R) {
    // current group's inputs -> working registers; issue next group's loads now
    float xc[L1R], dc[L1R][L1_MAXJ];
#pragma unroll
    for (int r = 0; r < L1R; ++r) {
      xc[r] = xv[r];
      if (BWD) {
#pragma unroll
        for (int j = 0; j < L1_MAXJ; ++j) dc[r][j] = dh[r][j];
      }
    }
    const int64_t nrow0 = row0 + nwaves * L1R;
#pragma unroll
    for (int r = 0; r < L1R; ++r) {
      xv[r] = (lane < a.O && nrow0 + r < a.M) ? a.X[(nrow0 + r) * a.O + lane] : 0.f;
      if (BWD) {
#pragma unroll
        for (int j = 0; j < L1_MAXJ; ++j)
          dh[r][j] = (j < NJ && nrow0 + r < a.M) ? a.H[(nrow0 + r) * a.Hd + lane + 64 * j] : 0.f;
      }
    }
    float z[L1R][L1_MAXJ];
#pragma unroll
    for (int r = 0; r < L1R; ++r)
#pragma unroll
      for (int j = 0; j < L1_MAXJ; ++j) z[r][j] = 0.f;
    for (int k = 0; k < a.O; ++k) {
      float xs[L1R];
#pragma unroll
      for (int r = 0; r < L1R; ++r) xs[r] = readlane_f(xc[r], k);
#pragma unroll
      for (int j = 0; j < L1_MAXJ; ++j) {
        if (j < NJ) {
          const float wv = Ws[k * a.Hd + lane + 64 * j];
#pragma unroll
          for (int r = 0; r < L1R; ++r) z[r][j] = fmaf(xs[r], wv, z[r][j]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < L1R; ++r) {
      const int64_t row = row0 + r;
      if (row >= a.M) break;  // wave-uniform
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int j = 0; j < L1_MAXJ; ++j)
        if (j < NJ) {
          z[r][j] += bs[lane + 64 * j];
          s += z[r][j];
          ss += z[r][j] * z[r][j];
        }
      float mean = 0.f, rstd = 1.f;
      if (LN) {
        s = wave_sum(s);
        ss = wave_sum(ss);
        mean = s * invH;
        const float var = fmaxf(0.f, ss * invH - mean * mean);  // flax "fast variance"
        rstd = rsqrtf(var + 1e-6f);
      }
      if (!BWD) {
#pragma unroll
        for (int j = 0; j < L1_MAXJ; ++j)
          if (j < NJ) {
            const int c = lane + 64 * j;
            float y = z[r][j];
            if (LN) y = (y - mean) * rstd * gs[c] + bes[c];
            a.H[row * a.Hd + c] = act_fwd_t<ACT>(y);
          }
      } else {
        // recompute h, then dZ1 = LN'(dH * act'(h))
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int j = 0; j < L1_MAXJ; ++j)
          if (j < NJ) {
            const int c = lane + 64 * j;
            const float gam = LN ? gs[c] : 1.f;
            const float xh = (z[r][j] - mean) * rstd;
            const float y = LN ? xh * gam + bes[c] : z[r][j];
            const float h = act_fwd_t<ACT>(y);
            const float dy = dc[r][j] * act_grad_t<ACT>(h);
            dg[j] += dy * xh;
            dbe[j] += dy;
            const float dxh = dy * gam;
            z[r][j] = xh;
            dc[r][j] = dxh;
            m1 += dxh;
            m2 += dxh * xh;
          }
        if (LN) {
          m1 = wave_sum(m1) * invH;
          m2 = wave_sum(m2) * invH;
        }
#pragma unroll
        for (int j = 0; j < L1_MAXJ; ++j)
          if (j < NJ) a.H[row * a.Hd + lane + 64 * j] = LN ? rstd * (dc[r][j] - m1 - z[r][j] * m2) : dc[r][j];
      }
    }
  }
}

// second layer of the same shape on the same input rows, launched as blockIdx.y == 1 (the recurrent policy's two observation
// encoders, ppo_lstm/flax_full_jit/policy.py:76-86): {W, b, g, be, H, ln_partials} of that layer
struct L1Twin { const float *W = nullptr, *b = nullptr, *g = nullptr, *be = nullptr; float *H = nullptr, *lnp = nullptr; };

template <bool BWD>
__global__ __launch_bounds__(L1_THREADS) void k_l1(const float* __restrict__ X, const float* __restrict__ W,
                                                   const float* __restrict__ b, const float* __restrict__ g,
                                                   const float* __restrict__ be, float* __restrict__ H,
                                                   float* __restrict__ ln_partials /*bwd+ln: [gridDim.x][2*Hd]*/,
                                                   int64_t M, int O, int Hd, int act, int ln,
                                                   const int32_t* __restrict__ m_dev /*optional device row count < M*/,
                                                   L1Twin tw = L1Twin()) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (m_dev && (int64_t)*m_dev < M) M = *m_dev;
  if (blockIdx.y) { W = tw.W; b = tw.b; g = tw.g; be = tw.be; H = tw.H; ln_partials = tw.lnp; }
  float* Ws = smem;            // [O][Hd]
  float* bs = Ws + O * Hd;     // [Hd]
  float* gs = bs + Hd;         // [Hd]
  float* bes = gs + Hd;        // [Hd]
  lds_stage<L1_THREADS>(Ws, W, O * Hd);
  for (int i = threadIdx.x; i < Hd; i += L1_THREADS) {
    bs[i] = b[i];
    gs[i] = ln ? g[i] : 1.f;
    bes[i] = ln ? be[i] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int NW = L1_THREADS / 64;
  float dg[L1_MAXJ], dbe[L1_MAXJ];
#pragma unroll
  for (int j = 0; j < L1_MAXJ; ++j) dg[j] = dbe[j] = 0.f;
  const L1Args a{X, H, M, O, Hd, Hd >> 6};
  const int64_t wave_g = (int64_t)blockIdx.x * NW + w, nwaves = (int64_t)gridDim.x * NW;
#define RLX_L1_BODY(ACT, LNB) \
  l1_rows<BWD, ACT, LNB, (BWD ? L1_R_BWD : L1_R_FWD)>(a, Ws, bs, gs, bes, dg, dbe, lane, wave_g, nwaves)
  if (ln) {
    if (act == RLX_ACT_ELU) RLX_L1_BODY(RLX_ACT_ELU, true);
    else if (act == RLX_ACT_TANH) RLX_L1_BODY(RLX_ACT_TANH, true);
    else RLX_L1_BODY(RLX_ACT_RELU, true);
  } else {
    if (act == RLX_ACT_TANH) RLX_L1_BODY(RLX_ACT_TANH, false);
    else if (act == RLX_ACT_ELU) RLX_L1_BODY(RLX_ACT_ELU, false);
    else RLX_L1_BODY(RLX_ACT_RELU, false);
  }
#undef RLX_L1_BODY
  if (BWD && ln) {
    // block partial of d(ln scale), d(ln bias): reduce the waves through LDS (reuse Ws)
    __syncthreads();
    float* red = smem;  // [NW][2*Hd]
#pragma unroll
    for (int j = 0; j < L1_MAXJ; ++j)
      if (j < (Hd >> 6)) {
        red[w * 2 * Hd + lane + 64 * j] = dg[j];
        red[w * 2 * Hd + Hd + lane + 64 * j] = dbe[j];
      }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Hd; i += L1_THREADS) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) v += red[q * 2 * Hd + i];
      ln_partials[(int64_t)blockIdx.x * 2 * Hd + i] = v;
    }
  }
}

static inline size_t l1_lds_bytes(int O, int Hd) {
  const size_t a = (size_t)(O + 3) * Hd, b = (size_t)(L1_THREADS / 64) * 2 * Hd;
  return (a > b ? a : b) * sizeof(float);
}

template <bool BWD>
static int launch_l1(const float* X, const float* W, const float* b, const float* g, const float* be, float* H,
                     float* ln_partials, int64_t M, int O, int Hd, int act, int ln, int grid, hipStream_t st,
                     const int32_t* m_dev = nullptr, const L1Twin* tw = nullptr) {
  hipLaunchKernelGGL(k_l1<BWD>, dim3(grid, tw ? 2 : 1), dim3(L1_THREADS), l1_lds_bytes(O, Hd), st, X, W, b, g, be, H, ln_partials,
                     M, O, Hd, act, ln, m_dev, tw ? *tw : L1Twin());
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

static inline int l1_grid(int64_t M, int num_cus) {
  int grid = div_up(M, (L1_THREADS / 64) * L1_R_FWD);
  const int cap = num_cus * 3;
  return grid > cap ? cap : grid;
}

// =======================================================================================
// NN: C[M,N] = act(A[M,K] @ W[K,N] + bias[N])          (forward hidden layer)
// =======================================================================================
template <int ACT>
__global__ __launch_bounds__(G_THREADS) void k_gemm_fwd(const float* __restrict__ A, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ C,
                                                        int64_t M, int N, int K, int lda, int ntn,
                                                        const int32_t* __restrict__ m_dev) {
  __shared__ __attribute__((aligned(16))) float As[G_LDS_A];
  __shared__ __attribute__((aligned(16))) float Bs[G_LDS_B];
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(tile / ntn) * G_BM;
  // m_dev (optional): the number of rows that really hold work lives on the device (compacted row lists whose length
  // the host never learns); the grid covers the capacity M and the row tiles beyond the count leave at once
  if (m_dev) {
    const int64_t mv = *m_dev;
    if (mv < M) M = mv;
    if (m0 >= M) return;
  }
  const int n0 = (tile % ntn) * G_BN;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
  const int a_r = t >> 3, a_c = (t & 7) * 4;   // A tile: 8 threads per 32-float row, 32 rows per pass
  const int b_r = t >> 5, b_c = (t & 31) * 4;  // B tile: 32 threads per 128-float row, 8 rows per pass
  f32x16 acc[2][2];
  zero_acc(acc);
  float4 ra[4], rb[4];
  const int nk = (K + G_BK - 1) / G_BK;
  float bv[2];  // loaded ahead of the main loop: the epilogue must not wait on memory between its stores
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + acc_col(wn, j, lane);
    bv[j] = col < N ? bias[col] : 0.f;
  }
// one K-tile step: stage the register tile, barrier, prefetch the next tile with LOAD(k0), MFMAs, barrier
#define RLX_FWD_KLOOP(LOAD)                                                                     \
  LOAD(0)                                                                                       \
  for (int kt = 0; kt < nk; ++kt) {                                                             \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                             \
      float* d = As + (a_r + 32 * p) * G_SA_ROW + a_c;                                          \
      d[0] = ra[p].x; d[1] = ra[p].y; d[2] = ra[p].z; d[3] = ra[p].w;                           \
      *reinterpret_cast<float4*>(Bs + (b_r + 8 * p) * G_SB + b_c) = rb[p];                      \
    }                                                                                           \
    __syncthreads();                                                                            \
    if (kt + 1 < nk) { LOAD((kt + 1) * G_BK) }                                                  \
    mma_ktile<G_SA_ROW, 1>(As, Bs, acc, wm, wn, lane);                                          \
    __syncthreads();                                                                            \
  }
  if (m0 + G_BM <= M && n0 + G_BN <= N && K % G_BK == 0) {
    // interior tile: plain loads off per-lane base pointers (the bounds checks of the guarded form cost
    // ~8 % of the loop: 64-bit compares and an exec-masked branch per load)
    const float* ap = A + (m0 + a_r) * lda + a_c;
    const float* wp = W + (int64_t)b_r * N + n0 + b_c;
#define RLX_LOAD_PLAIN(K0)                                                                      \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                               \
    ra[p] = *reinterpret_cast<const float4*>(ap + (int64_t)(32 * p) * lda + (K0));              \
    rb[p] = *reinterpret_cast<const float4*>(wp + (int64_t)((K0) + 8 * p) * N);                 \
  }
    RLX_FWD_KLOOP(RLX_LOAD_PLAIN)
#undef RLX_LOAD_PLAIN
  } else {
#define RLX_LOAD_GUARDED(K0)                                                                    \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                               \
    ra[p] = ld4(A, m0 + a_r + 32 * p, (K0) + a_c, M, lda, lda); /* A rows are zero-padded up to lda */ \
    rb[p] = ld4(W, (K0) + b_r + 8 * p, n0 + b_c, K, N, N);                                      \
  }
    RLX_FWD_KLOOP(RLX_LOAD_GUARDED)
#undef RLX_LOAD_GUARDED
  }
#undef RLX_FWD_KLOOP
  if (m0 + G_BM <= M && n0 + G_BN <= N) {
    // interior tile (uniform branch): 64 independent stores per lane off one per-lane base, no exec masking
    float* cb = C + (m0 + wm * 64 + 4 * (lane >> 5)) * N + n0 + wn * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          cb[(i * 32 + (r & 3) + 8 * (r >> 2)) * N + j * 32] = act_fwd_t<ACT>(acc[i][j][r] + bv[j]);
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + acc_col(wn, j, lane);
    if (col >= N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + acc_row(wm, i, r, lane);
        if (row < M) C[row * N + col] = act_fwd_t<ACT>(acc[i][j][r] + bv[j]);
      }
  }
}

// =======================================================================================
// NT: dZprev[M,Kd] = (dZ[M,N] @ W[Kd,N]^T) * act'(Hprev[M,Kd])   (input gradient)
// HD (the activation of the previous layer) is overwritten IN PLACE by dZprev; with
// apply_act = 0 the raw product is stored (the first-layer backward applies LN'/act').
// =======================================================================================
template <int ACT, bool APPLY>
__global__ __launch_bounds__(G_THREADS) void k_gemm_dx(const float* __restrict__ dZ, const float* __restrict__ W,
                                                       float* __restrict__ HD, int64_t M, int N, int Kd, int ldo, int ntn,
                                                       const float* __restrict__ Hsrc) {
  // Hsrc (optional, APPLY only): the previous layer's activations when the result must NOT overwrite them (out-of-place
  // backward: the weight-gradient kernel of this layer may then run later / concurrently); same shape and row stride as HD
  __shared__ __attribute__((aligned(16))) float As[G_LDS_A];
  __shared__ __attribute__((aligned(16))) float Bs[G_LDS_B];
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(tile / ntn) * G_BM;
  const int c0 = (tile % ntn) * G_BN;  // output column (= Kd index) base
  const float* __restrict__ Hin = Hsrc ? Hsrc : HD;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
  const int a_r = t >> 3, a_c = (t & 7) * 4;  // A (dZ) and W tiles: [128 rows][32 reduction cols]
  f32x16 acc[2][2];
  zero_acc(acc);
  float4 ra[4], rb[4];
  const int nk = (N + G_BK - 1) / G_BK;
#define RLX_DX_KLOOP(LOAD)                                                                      \
  LOAD(0)                                                                                       \
  for (int kt = 0; kt < nk; ++kt) {                                                             \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                             \
      float* d = As + (a_r + 32 * p) * G_SA_ROW + a_c;                                          \
      d[0] = ra[p].x; d[1] = ra[p].y; d[2] = ra[p].z; d[3] = ra[p].w;                           \
      float* e = Bs + a_c * G_SBT + (a_r + 32 * p); /* transpose: Bs[n][kd] */                  \
      e[0] = rb[p].x; e[G_SBT] = rb[p].y; e[2 * G_SBT] = rb[p].z; e[3 * G_SBT] = rb[p].w;       \
    }                                                                                           \
    __syncthreads();                                                                            \
    if (kt + 1 < nk) { LOAD((kt + 1) * G_BK) }                                                  \
    mma_ktile<G_SA_ROW, 1, G_SBT>(As, Bs, acc, wm, wn, lane);                                          \
    __syncthreads();                                                                            \
  }
  if (m0 + G_BM <= M && c0 + G_BN <= Kd && N % G_BK == 0) {
    const float* ap = dZ + (m0 + a_r) * N + a_c;
    const float* wp = W + (int64_t)(c0 + a_r) * N + a_c;
#define RLX_LOAD_PLAIN(K0)                                                                      \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                               \
    ra[p] = *reinterpret_cast<const float4*>(ap + (int64_t)(32 * p) * N + (K0));                \
    rb[p] = *reinterpret_cast<const float4*>(wp + (int64_t)(32 * p) * N + (K0));                \
  }
    RLX_DX_KLOOP(RLX_LOAD_PLAIN)
#undef RLX_LOAD_PLAIN
  } else {
#define RLX_LOAD_GUARDED(K0)                                                                    \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                               \
    ra[p] = ld4(dZ, m0 + a_r + 32 * p, (K0) + a_c, M, N, N);                                    \
    rb[p] = ld4(W, c0 + a_r + 32 * p, (K0) + a_c, Kd, N, N);                                    \
  }
    RLX_DX_KLOOP(RLX_LOAD_GUARDED)
#undef RLX_LOAD_GUARDED
  }
#undef RLX_DX_KLOOP
  if (m0 + G_BM <= M && c0 + G_BN <= Kd) {
    // interior tile: all 64 activation loads of a lane are issued before the first use, then 64 independent stores
    float* hb = HD + (m0 + wm * 64 + 4 * (lane >> 5)) * ldo + c0 + wn * 64 + (lane & 31);
    const float* hs = Hin + (m0 + wm * 64 + 4 * (lane >> 5)) * ldo + c0 + wn * 64 + (lane & 31);
    if (APPLY) {
      float h[2][2][16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) h[i][j][r] = hs[(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo + j * 32];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            hb[(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo + j * 32] = acc[i][j][r] * act_grad_t<ACT>(h[i][j][r]);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) hb[(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo + j * 32] = acc[i][j][r];
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = c0 + acc_col(wn, j, lane);
    if (col >= Kd) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + acc_row(wm, i, r, lane);
        if (row < M) {
          const int64_t o = row * ldo + col;
          float v = acc[i][j][r];
          if (APPLY) v *= act_grad_t<ACT>(Hin[o]);
          HD[o] = v;
        }
      }
  }
}

// runtime (act, apply) -> template instance
#define RLX_GEMM_FWD_LAUNCH(ACTV, GRID, ST, ...)                                                                  \
  switch (ACTV) {                                                                                                 \
    case RLX_ACT_TANH: RLX_PLAUNCH(k_gemm_fwd<RLX_ACT_TANH>, GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break; \
    case RLX_ACT_ELU: RLX_PLAUNCH(k_gemm_fwd<RLX_ACT_ELU>, GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;   \
    case RLX_ACT_RELU: RLX_PLAUNCH(k_gemm_fwd<RLX_ACT_RELU>, GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break; \
    default: RLX_PLAUNCH(k_gemm_fwd<RLX_ACT_NONE>, GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;           \
  }
#define RLX_GEMM_DX_LAUNCH(ACTV, APPLYV, GRID, ST, ...)                                                           \
  if (!(APPLYV)) RLX_PLAUNCH((k_gemm_dx<RLX_ACT_NONE, false>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__);  \
  else switch (ACTV) {                                                                                            \
    case RLX_ACT_TANH: RLX_PLAUNCH((k_gemm_dx<RLX_ACT_TANH, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break; \
    case RLX_ACT_ELU: RLX_PLAUNCH((k_gemm_dx<RLX_ACT_ELU, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;   \
    case RLX_ACT_RELU: RLX_PLAUNCH((k_gemm_dx<RLX_ACT_RELU, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break; \
    default: RLX_PLAUNCH((k_gemm_dx<RLX_ACT_NONE, false>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;          \
  }

// =======================================================================================
// TN: dW[Kd,N] (+)= Hprev[M,Kd]^T @ dZ[M,N], split over M: workgroup (tile, s) reduces rows
// [s*Mc, (s+1)*Mc) and writes a partial slab; bias gradient (column sums of dZ) rides along
// in the kd-tile-0 workgroups.  Slabs are summed in fixed order by k_reduce_segments
// (deterministic; no float atomics).
//   partials layout: [S][Kd*N] then db partials [S][N]
// =======================================================================================
__global__ __launch_bounds__(G_THREADS) void k_gemm_dw(const float* __restrict__ Hp, const float* __restrict__ dZ,
                                                       float* __restrict__ partW, float* __restrict__ partB,
                                                       int64_t M, int Kd, int ldh, int N, int64_t Mc, int ntk, int ntn) {
  __shared__ __attribute__((aligned(16))) float As[G_LDS_A];
  __shared__ __attribute__((aligned(16))) float Bs[G_LDS_B];
  const int ntiles = ntk * ntn;
  // all tiles of one M-slab read the same H / dZ rows: keep them on one XCD so its L2 serves the re-reads
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int s = lb / ntiles, tile = lb % ntiles;
  const int k0d = (tile / ntn) * G_BM;  // row (kd) base of the dW tile
  const int n0 = (tile % ntn) * G_BN;
  const int64_t mbeg = (int64_t)s * Mc;
  int64_t mend = mbeg + Mc;
  if (mend > M) mend = M;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
  const int b_r = t >> 5, b_c = (t & 31) * 4;  // both tiles: [32 m rows][128 cols]
  f32x16 acc[2][2];
  zero_acc(acc);
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
  float4 ra[4], rb[4];
  const int nk = (int)((mend - mbeg + G_BK - 1) / G_BK);
#define RLX_DW_KLOOP(LOAD)                                                                      \
  LOAD(0)                                                                                       \
  for (int kt = 0; kt < nk; ++kt) {                                                             \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                             \
      *reinterpret_cast<float4*>(As + (b_r + 8 * p) * G_SA_COL + b_c) = ra[p];                  \
      *reinterpret_cast<float4*>(Bs + (b_r + 8 * p) * G_SB + b_c) = rb[p];                      \
      colsum[0] += rb[p].x; colsum[1] += rb[p].y; colsum[2] += rb[p].z; colsum[3] += rb[p].w;   \
    }                                                                                           \
    __syncthreads();                                                                            \
    if (kt + 1 < nk) { LOAD((kt + 1) * G_BK) }                                                  \
    mma_ktile<1, G_SA_COL>(As, Bs, acc, wm, wn, lane);                                          \
    __syncthreads();                                                                            \
  }
  if (k0d + G_BM <= Kd && n0 + G_BN <= N && (mend - mbeg) % G_BK == 0) {
    const float* hp = Hp + (mbeg + b_r) * ldh + k0d + b_c;
    const float* zp = dZ + (mbeg + b_r) * N + n0 + b_c;
#define RLX_LOAD_PLAIN(M0)                                                                      \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                               \
    ra[p] = *reinterpret_cast<const float4*>(hp + (int64_t)((M0) + 8 * p) * ldh);               \
    rb[p] = *reinterpret_cast<const float4*>(zp + (int64_t)((M0) + 8 * p) * N);                 \
  }
    RLX_DW_KLOOP(RLX_LOAD_PLAIN)
#undef RLX_LOAD_PLAIN
  } else {
#define RLX_LOAD_GUARDED(M0)                                                                    \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                               \
    ra[p] = ld4(Hp, mbeg + (M0) + b_r + 8 * p, k0d + b_c, mend, ldh, ldh);                      \
    rb[p] = ld4(dZ, mbeg + (M0) + b_r + 8 * p, n0 + b_c, mend, N, N);                           \
  }
    RLX_DW_KLOOP(RLX_LOAD_GUARDED)
#undef RLX_LOAD_GUARDED
  }
#undef RLX_DW_KLOOP
  float* outW = partW + (int64_t)s * Kd * N;
  if (k0d + G_BM <= Kd && n0 + G_BN <= N) {  // interior tile: 64 independent stores off one per-lane base
    float* ob = outW + (int64_t)(k0d + wm * 64 + 4 * (lane >> 5)) * N + n0 + wn * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[(i * 32 + (r & 3) + 8 * (r >> 2)) * N + j * 32] = acc[i][j][r];
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + acc_col(wn, j, lane);
      if (col >= N) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = k0d + acc_row(wm, i, r, lane);
          if (row < Kd) outW[(int64_t)row * N + col] = acc[i][j][r];
        }
    }
  }
  if (k0d == 0 && partB) {
    // column sums: thread t holds 4 columns (b_c..b_c+3) over rows b_r + 8p; reduce the 8 row groups
    float* red = As;  // [8][128]
    __syncthreads();
    red[b_r * 128 + b_c + 0] = colsum[0];
    red[b_r * 128 + b_c + 1] = colsum[1];
    red[b_r * 128 + b_c + 2] = colsum[2];
    red[b_r * 128 + b_c + 3] = colsum[3];
    __syncthreads();
    if (t < 128 && n0 + t < N) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) v += red[q * 128 + t];
      partB[(int64_t)s * N + n0 + t] = v;
    }
  }
}

// =======================================================================================
// TN with a skinny Kd (first layer, Kd = in_dim <= 32): dW1[Kd,N] = X[M,Kd]^T @ dZ1[M,N].
// Output tile 32 x 128 (one 32x32 MFMA tile per wave); X rows are zero-padded to 32.
// =======================================================================================
__global__ __launch_bounds__(G_THREADS) void k_gemm_dw_skinny(const float* __restrict__ X,
                                                              const float* __restrict__ dZ,
                                                              float* __restrict__ partW, float* __restrict__ partB,
                                                              int64_t M, int Kd, int N, int64_t Mc, int ntn) {
  __shared__ __attribute__((aligned(16))) float As[G_BK * 33];    // As[m][kd], kd < 32
  __shared__ __attribute__((aligned(16))) float Bs[G_LDS_B];      // Bs[m][n]
  const int s = blockIdx.x / ntn, n0 = (blockIdx.x % ntn) * G_BN;
  const int64_t mbeg = (int64_t)s * Mc;
  int64_t mend = mbeg + Mc;
  if (mend > M) mend = M;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b_r = t >> 5, b_c = (t & 31) * 4;
  const int x_r = t >> 3, x_c = (t & 7) * 4;  // X tile [32 m][32 kd] scalar loads (Kd arbitrary)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
  const int nk = (int)((mend - mbeg + G_BK - 1) / G_BK);
  for (int kt = 0; kt < nk; ++kt) {
    const int64_t mm = mbeg + (int64_t)kt * G_BK;
    float xr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xr[q] = (mm + x_r < mend && x_c + q < Kd) ? X[(mm + x_r) * Kd + x_c + q] : 0.f;
    float4 rb[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) rb[p] = ld4(dZ, mm + b_r + 8 * p, n0 + b_c, mend, N, N);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) As[x_r * 33 + x_c + q] = xr[q];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<float4*>(Bs + (b_r + 8 * p) * G_SB + b_c) = rb[p];
      colsum[0] += rb[p].x; colsum[1] += rb[p].y; colsum[2] += rb[p].z; colsum[3] += rb[p].w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < G_BK; kk += 2) {
      const float a = As[(kk + lh) * 33 + li];
      const float b = Bs[(kk + lh) * G_SB + wv * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  float* outW = partW + (int64_t)s * Kd * N;
  const int col = n0 + wv * 32 + li;
  if (col < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < Kd) outW[(int64_t)row * N + col] = acc[r];
    }
  }
  if (partB) {
    __syncthreads();
    float* red = Bs;
    red[b_r * 128 + b_c + 0] = colsum[0];
    red[b_r * 128 + b_c + 1] = colsum[1];
    red[b_r * 128 + b_c + 2] = colsum[2];
    red[b_r * 128 + b_c + 3] = colsum[3];
    __syncthreads();
    if (t < 128 && n0 + t < N) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) v += red[q * 128 + t];
      partB[(int64_t)s * N + n0 + t] = v;
    }
  }
}

// =======================================================================================
// Head forward (tiny out_dim): out[M,A] = H[M,K] @ W[K,A] + b.  64 rows per block.
// =======================================================================================
// ROWS rows per workgroup (64 for large batches; 16 when M is small, so that SAC-sized batches still fill the chip)
// w_trans: W is stored [A][K] (a row block of a Dense kernel used as W^T: the input-gradient of a few input columns);
// ldo: row stride of out (>= A); b may be NULL.  K % 4 == 0 (hidden widths are multiples of 4), A <= 64.
// Thread (r = t % ROWS, g = t / ROWS) owns outputs (r, g + q * 256 / ROWS): one 16-B LDS read of the activation row feeds
// four k steps of every output column it owns; each output is one ascending-k fmaf chain.
// NQA: output columns per thread (compile time, >= ceil(A / (256 / ROWS))): the k loop is branch-free -- a thread whose q-th
// column does not exist reads a clamped one and drops the result -- so the LDS reads of a k step are all issued before its
// FMAs (with a per-column `if (a < A)` in the loop every column group waited out its own LDS latency: 15 us for a 34-wide head)
constexpr int head_fwd_lds_floats(int rows, int K, int A) { return rows * (K + 4) + K * A; }
template <int ROWS, int NQA>
__global__ __launch_bounds__(256) void k_head_fwd(const float* __restrict__ H, const float* __restrict__ W,
                                                  const float* __restrict__ b, float* __restrict__ out, int64_t M,
                                                  int K, int A, const int32_t* __restrict__ m_dev, int w_trans, int ldo,
                                                  Twin tw) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NG = 256 / ROWS;
  static_assert(NQA >= 1 && NQA * NG <= 64, "at most 64 output columns");
  if (blockIdx.y) {   // twin launch: {H, W, b, out} of the second problem
    H = static_cast<const float*>(tw.p[0]);
    W = static_cast<const float*>(tw.p[1]);
    b = static_cast<const float*>(tw.p[2]);
    out = const_cast<float*>(static_cast<const float*>(tw.p[3]));
  }
  const int HS = K + 4;               // 16-B aligned rows, 4 banks apart: the 16 rows of a b128 read cover all 64 banks
  float* Hs = smem;                   // [ROWS][K+4]
  float* Ws = Hs + ROWS * HS;         // [K][A]
  const int64_t r0 = (int64_t)blockIdx.x * ROWS;
  if (m_dev) {
    const int64_t mv = *m_dev;
    if (mv < M) M = mv;
    if (r0 >= M) return;
  }
  // ONE round trip to memory for the tile: the activation rows and the first 36 weight words per thread are all in flight
  // before the first LDS store (staged loop by loop, the 35 KB weight block of a 34-wide head cost five dependent round trips)
  const int K4 = K >> 2, KA = K * A;
  constexpr int HB = ROWS / 16;     // activation rows per thread (16 threads x 16 B per row pass)
  constexpr int WB = 36;            // weight words per thread per batch
  float4 hreg[HB][4];
  {
    const int c4 = threadIdx.x & 15;
#pragma unroll
    for (int q = 0; q < HB; ++q) {
      const int r = (threadIdx.x >> 4) + 16 * q;
      const float4* src = reinterpret_cast<const float4*>(H + (r0 + r) * K);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        hreg[q][j] = (r0 + r < M && c4 + 16 * j < K4) ? src[c4 + 16 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // (the head's offset inside the flat parameter vector is only 4-B aligned in general: scalar loads)
  float wreg[WB];
#pragma unroll
  for (int j = 0; j < WB; ++j) {
    const int i = threadIdx.x + 256 * j;
    wreg[j] = i < KA ? W[i] : 0.f;
  }
  {
    const int c4 = threadIdx.x & 15;
#pragma unroll
    for (int q = 0; q < HB; ++q) {
      const int r = (threadIdx.x >> 4) + 16 * q;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c4 + 16 * j < K4) *reinterpret_cast<float4*>(&Hs[r * HS + 4 * (c4 + 16 * j)]) = hreg[q][j];
    }
    for (int c = c4 + 64; c < K4; c += 16) {   // K > 256: the remaining 16-B columns
#pragma unroll
      for (int q = 0; q < HB; ++q) {
        const int r = (threadIdx.x >> 4) + 16 * q;
        *reinterpret_cast<float4*>(&Hs[r * HS + 4 * c]) =
            r0 + r < M ? reinterpret_cast<const float4*>(H + (r0 + r) * K)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  for (int base = 0; base < KA; base += 256 * WB) {
    if (base) {
#pragma unroll
      for (int j = 0; j < WB; ++j) {
        const int i = base + threadIdx.x + 256 * j;
        wreg[j] = i < KA ? W[i] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < WB; ++j) {
      const int i = base + threadIdx.x + 256 * j;
      if (i < KA) {
        if (w_trans) {
          const int a = i / K, k = i - a * K;      // coalesced reads of W[a][k]
          Ws[k * A + a] = wreg[j];
        } else {
          Ws[i] = wreg[j];
        }
      }
    }
  }
  __syncthreads();
  const int r = threadIdx.x % ROWS, g = threadIdx.x / ROWS;
  float acc[NQA];
  int ac[NQA];
#pragma unroll
  for (int q = 0; q < NQA; ++q) {
    acc[q] = 0.f;
    ac[q] = g + q * NG < A ? g + q * NG : A - 1;
  }
#pragma unroll 2
  for (int k = 0; k < K; k += 4) {
    const float4 h = *reinterpret_cast<const float4*>(&Hs[r * HS + k]);
    const float* wk = Ws + k * A;
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      acc[q] = fmaf(h.x, wk[ac[q]], acc[q]);
      acc[q] = fmaf(h.y, wk[A + ac[q]], acc[q]);
      acc[q] = fmaf(h.z, wk[2 * A + ac[q]], acc[q]);
      acc[q] = fmaf(h.w, wk[3 * A + ac[q]], acc[q]);
    }
  }
  if (r0 + r < M) {
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      const int a = g + q * NG;
      if (a < A) out[(r0 + r) * ldo + a] = acc[q] + (b ? b[a] : 0.f);
    }
  }
}

// instantiation for a head of A columns: the smallest NQA in {1, 2, 3, 4} (16 rows) / {2, 4, 8, 16} (64 rows) that covers it
// (tiles above the default 64 KB of dynamic LDS -- a 512-wide layer with more than 15 output columns -- need the attribute)
constexpr size_t HEAD_FWD_MAX_LDS = 128 * 1024;
static void head_fwd16_allow_large_lds() {
  static AttrOnce done;      // (per device: the attribute belongs to the device's copy of the function)
  if (done.done()) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_fwd<16, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_FWD_MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_fwd<16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_FWD_MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_fwd<16, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_FWD_MAX_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_fwd<16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_FWD_MAX_LDS);
  done.mark();
}
#define RLX_HEAD_FWD_16(A_, GRID, LDS, ST, ...)                                                                   \
  if ((LDS) > 64 * 1024) head_fwd16_allow_large_lds();                                                            \
  switch (div_up((A_), 16)) {                                                                                     \
    case 1: hipLaunchKernelGGL((k_head_fwd<16, 1>), GRID, dim3(256), LDS, ST, __VA_ARGS__); break;                \
    case 2: hipLaunchKernelGGL((k_head_fwd<16, 2>), GRID, dim3(256), LDS, ST, __VA_ARGS__); break;                \
    case 3: hipLaunchKernelGGL((k_head_fwd<16, 3>), GRID, dim3(256), LDS, ST, __VA_ARGS__); break;                \
    default: hipLaunchKernelGGL((k_head_fwd<16, 4>), GRID, dim3(256), LDS, ST, __VA_ARGS__); break;               \
  }
#define RLX_HEAD_FWD_64(A_, GRID, LDS, ST, ...)                                                                   \
  do {                                                                                                            \
    const int nq_ = div_up((A_), 4);                                                                              \
    if (nq_ <= 2) hipLaunchKernelGGL((k_head_fwd<64, 2>), GRID, dim3(256), LDS, ST, __VA_ARGS__);                 \
    else if (nq_ <= 4) hipLaunchKernelGGL((k_head_fwd<64, 4>), GRID, dim3(256), LDS, ST, __VA_ARGS__);            \
    else if (nq_ <= 8) hipLaunchKernelGGL((k_head_fwd<64, 8>), GRID, dim3(256), LDS, ST, __VA_ARGS__);            \
    else hipLaunchKernelGGL((k_head_fwd<64, 16>), GRID, dim3(256), LDS, ST, __VA_ARGS__);                         \
  } while (0)

// =======================================================================================
// Deterministic reduction of partial slabs into the flat gradient buffer (+ per-block sum
// of squares for the global norm).  One launch per network.
// =======================================================================================
__global__ __launch_bounds__(256) void k_reduce_segments(ReduceTable tab, float* __restrict__ sumsq_partials) {
  __shared__ float4 s_acc[4][64];
  __shared__ float s_buf[4];
  // locate this block's segment
  int seg = 0;
  int blk = blockIdx.x;
  while (seg < tab.n - 1 && blk >= tab.seg[seg].nblocks) {
    blk -= tab.seg[seg].nblocks;
    ++seg;
  }
  const ReduceSeg sg = tab.seg[seg];
  float sq = 0.f;
  if (sg.vec == 2) {
    // many-slab segments (the policy head's partials: one slab per 64 minibatch rows, S = 512 at mb 32768): 16 x 16 threads -- x = float4
    // column of 64 consecutive outputs, y = slab residue class mod 16 -- so a thread walks S / 16 slabs, sixteen loads in flight:
    // two round trips at S = 512.  (With the 64 x 4 arrangement below these few workgroups walked S / 4 slabs each -- eight dependent
    // rounds -- and were the tail of the policy's launch: 10.4 us alone against the critic's 7.4.)
    const int x = threadIdx.x & 15, y = threadIdx.x >> 4;
    const int64_t i = (int64_t)blk * 64 + 4 * x;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < sg.len) {
      const float* p = sg.src + i;
      int sidx = y;
      for (; sidx + 240 < sg.S; sidx += 256) {
        float4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const float4*>(p + (int64_t)(sidx + 16 * q) * sg.stride);
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
          a.x += (v[q].x + v[q + 1].x) + (v[q + 2].x + v[q + 3].x);
          a.y += (v[q].y + v[q + 1].y) + (v[q + 2].y + v[q + 3].y);
          a.z += (v[q].z + v[q + 1].z) + (v[q + 2].z + v[q + 3].z);
          a.w += (v[q].w + v[q + 1].w) + (v[q + 2].w + v[q + 3].w);
        }
      }
      for (; sidx < sg.S; sidx += 64) {      // the rest, four at a time (clamped index, value masked afterwards)
        float4 r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int si = sidx + 16 * q;
          r[q] = *reinterpret_cast<const float4*>(p + (int64_t)(si < sg.S ? si : sg.S - 1) * sg.stride);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (sidx + 16 * q < sg.S) { a.x += r[q].x; a.y += r[q].y; a.z += r[q].z; a.w += r[q].w; }
      }
    }
    float4* sred4 = &s_acc[0][0];      // [16 residue classes][16 columns]
    sred4[y * 16 + x] = a;
    __syncthreads();
    if (y == 0 && i < sg.len) {
      float4 v = a;
#pragma unroll
      for (int q = 1; q < 16; ++q) {
        const float4 b = sred4[q * 16 + x];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      v.x = v.x * sg.scale + sg.bias;
      v.y = v.y * sg.scale + sg.bias;
      v.z = v.z * sg.scale + sg.bias;
      v.w = v.w * sg.scale + sg.bias;
      sg.dst[i] = v.x; sg.dst[i + 1] = v.y; sg.dst[i + 2] = v.z; sg.dst[i + 3] = v.w;
      if (sg.in_norm) sq = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  } else if (sg.vec) {
    // 64 x 4 threads: x = float4 column of 256 consecutive outputs, y = slab residue class
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const int64_t i = (int64_t)blk * 256 + 4 * x;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < sg.len) {
      const float* p = sg.src + i;
      int sidx = y;
      // many-slab segments (the head partials: one slab per 64 minibatch rows, S = 512 at mb 32768) are a serial chain of
      // S / 16 dependent round trips for the few workgroups that own them -- the tail of the whole launch (23 us for the
      // policy, whose bulk needs 10).  Sixteen loads in flight per thread shorten that chain four times.
      for (; sidx + 60 < sg.S; sidx += 64) {
        float4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const float4*>(p + (int64_t)(sidx + 4 * q) * sg.stride);
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
          a.x += (v[q].x + v[q + 1].x) + (v[q + 2].x + v[q + 3].x);
          a.y += (v[q].y + v[q + 1].y) + (v[q + 2].y + v[q + 3].y);
          a.z += (v[q].z + v[q + 1].z) + (v[q + 2].z + v[q + 3].z);
          a.w += (v[q].w + v[q + 1].w) + (v[q + 2].w + v[q + 3].w);
        }
      }
      for (; sidx + 12 < sg.S; sidx += 16) {  // 4 independent 16-B loads in flight
        const float4 v0 = *reinterpret_cast<const float4*>(p + (int64_t)sidx * sg.stride);
        const float4 v1 = *reinterpret_cast<const float4*>(p + (int64_t)(sidx + 4) * sg.stride);
        const float4 v2 = *reinterpret_cast<const float4*>(p + (int64_t)(sidx + 8) * sg.stride);
        const float4 v3 = *reinterpret_cast<const float4*>(p + (int64_t)(sidx + 12) * sg.stride);
        a.x += (v0.x + v1.x) + (v2.x + v3.x);
        a.y += (v0.y + v1.y) + (v2.y + v3.y);
        a.z += (v0.z + v1.z) + (v2.z + v3.z);
        a.w += (v0.w + v1.w) + (v2.w + v3.w);
      }
      // at most three slabs are left for this residue class: requested TOGETHER (clamped index, value masked afterwards -- no branch
      // around the loads) and added in the order of the former one-slab-at-a-time loop, which was up to three dependent round trips
      // behind the block above (25 slabs at the bench shape: 4 round trips per launch, now 2)
      {
        float4 r[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int si = sidx + 4 * q;
          r[q] = *reinterpret_cast<const float4*>(p + (int64_t)(si < sg.S ? si : sg.S - 1) * sg.stride);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if (sidx + 4 * q < sg.S) { a.x += r[q].x; a.y += r[q].y; a.z += r[q].z; a.w += r[q].w; }
        }
      }
    }
    s_acc[y][x] = a;
    __syncthreads();
    if (y == 0 && i < sg.len) {
      const float4 b1 = s_acc[1][x], b2 = s_acc[2][x], b3 = s_acc[3][x];
      float4 v;
      v.x = ((a.x + b1.x) + (b2.x + b3.x)) * sg.scale + sg.bias;
      v.y = ((a.y + b1.y) + (b2.y + b3.y)) * sg.scale + sg.bias;
      v.z = ((a.z + b1.z) + (b2.z + b3.z)) * sg.scale + sg.bias;
      v.w = ((a.w + b1.w) + (b2.w + b3.w)) * sg.scale + sg.bias;
      // dst may be only 4-B aligned (second critic of a VectorCritic starts at an odd offset)
      sg.dst[i] = v.x; sg.dst[i + 1] = v.y; sg.dst[i + 2] = v.z; sg.dst[i + 3] = v.w;
      if (sg.in_norm) sq = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  } else {
    // narrow segments (biases, logstd, metric sums): 16 outputs x 16 slab groups per block
    const int x = threadIdx.x & 15, y = threadIdx.x >> 4;
    const int64_t i = (int64_t)blk * 16 + x;
    float a = 0.f;
    if (i < sg.len) {
      const float* p = sg.src + i;
      int sidx = y;
      for (; sidx + 48 < sg.S; sidx += 64) {
        const float v0 = p[(int64_t)sidx * sg.stride], v1 = p[(int64_t)(sidx + 16) * sg.stride];
        const float v2 = p[(int64_t)(sidx + 32) * sg.stride], v3 = p[(int64_t)(sidx + 48) * sg.stride];
        a += (v0 + v1) + (v2 + v3);
      }
      for (; sidx < sg.S; sidx += 16) a += p[(int64_t)sidx * sg.stride];
    }
    float* sred = reinterpret_cast<float*>(&s_acc[0][0]);  // [16][16]
    sred[y * 16 + x] = a;
    __syncthreads();
    if (y == 0 && i < sg.len) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v += sred[q * 16 + x];
      v = v * sg.scale + sg.bias;
      sg.dst[i] = v;
      if (sg.in_norm) sq = v * v;
    }
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0 && sumsq_partials) sumsq_partials[blockIdx.x] = s_buf[0] + s_buf[1] + s_buf[2] + s_buf[3];
}

// ---------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------
static int check_desc(const rlx_mlp_desc& d) {
  RLX_REQUIRE(d.n_hidden >= 1 && d.n_hidden <= 3, RLX_EUNSUP, "mlp: n_hidden must be 1..3");
  RLX_REQUIRE(d.in_dim >= 1 && d.in_dim <= 8192, RLX_EUNSUP, "mlp: in_dim must be 1..8192");
  // in_dim <= 32: fused Dense(+LN)+act kernels; wider inputs go through the MFMA GEMM, a LayerNorm after it through
  // k_ln_act (pre-LayerNorm values kept in acts[3] for the backward)
  RLX_REQUIRE(d.in_dim <= 32 || !d.ln_first || d.n_hidden >= 2, RLX_EUNSUP,
              "mlp: LayerNorm after a wide first layer needs at least two hidden layers");
  RLX_REQUIRE(d.hidden[0] % 64 == 0 && d.hidden[0] >= 64 && d.hidden[0] <= 512, RLX_EUNSUP,
              "mlp: hidden[0] must be a multiple of 64 in [64, 512]");
  for (int l = 1; l < d.n_hidden; ++l)
    RLX_REQUIRE(d.hidden[l] % 4 == 0 && d.hidden[l] >= 4, RLX_EUNSUP, "mlp: hidden dims must be multiples of 4");
  RLX_REQUIRE(d.out_dim >= 1 && d.out_dim <= 64, RLX_EUNSUP, "mlp: out_dim must be 1..64");
  RLX_REQUIRE(d.act >= 0 && d.act <= 2, RLX_EINVAL, "mlp: unknown activation");
  return RLX_OK;
}

int mlp_check_desc(const rlx_mlp_desc& d) { return check_desc(d); }

int launch_l1_fwd(const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x, float* h1,
                  int64_t M, int num_cus, hipStream_t st, bool allow_mfma, const int32_t* m_dev = nullptr, rlx_ctx* prof_ctx = nullptr) {
  if (allow_mfma && M >= 1024 && l1fwd_mfma_supported(d)) return launch_l1fwd_mfma(d, L, params, x, h1, M, num_cus, st, m_dev, prof_ctx);
  const LayerOff& o = L.layer[0];
  const int grid = l1_grid(M, num_cus);
  return launch_l1<false>(x, params + o.W, params + o.b, o.g >= 0 ? params + o.g : nullptr,
                          o.be >= 0 ? params + o.be : nullptr, h1, nullptr, M, o.in, o.out, d.act,
                          d.ln_first ? 1 : 0, grid, st, m_dev);
}

int launch_gemm_fwd(rlx_ctx* ctx, const float* A, const float* W, const float* bias, float* C, int64_t M, int N, int K,
                    int act, hipStream_t st, int lda, const int32_t* m_dev) {
  if (const void* img = bx_lookup(ctx, W, 0, K, N)) return bx_launch_fwd(ctx, A, img, bias, C, M, N, K, act, st, lda, m_dev);
  // (m_dev: the rows actually processed are a device-side count -- next-value reuse, gae.hip --: no algorithmic figure to report)
  ProfScope prof(m_dev ? nullptr : ctx, PK_GEMM_FWD, 2.0 * (double)M * N * K, st, gemm_bytes(M, N, K), M, N, K);
  const int ntn = div_up(N, G_BN);
  const int grid = div_up(M, G_BM) * ntn;
  RLX_GEMM_FWD_LAUNCH(act, dim3(grid), st, A, W, bias, C, M, N, K, lda > 0 ? lda : K, ntn, m_dev);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// dX[:, 0:nc] = dZ[M, K] @ Wblk[nc, K]^T  (Wblk: nc consecutive rows of a Dense kernel): the LDS-staged head kernel with the
// weight block read transposed; tw (optional): {dZ, Wblk, -, dX} of a second problem
// (launch_dx_cols below)

// narrow heads (A <= 4, e.g. the value / Q heads with A = 1): one wave per row, the K products spread over the 64 lanes
// (the generic kernel gives such a head one thread per row: 16 of 256 threads busy)
__global__ __launch_bounds__(256) void k_head_fwd_narrow(const float* __restrict__ H, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ out, int64_t M,
                                                         int K, int A, const int32_t* __restrict__ m_dev, Twin tw) {
  if (blockIdx.y) {   // twin launch: {H, W, b, out} of the second problem
    H = static_cast<const float*>(tw.p[0]);
    W = static_cast<const float*>(tw.p[1]);
    b = static_cast<const float*>(tw.p[2]);
    out = const_cast<float*>(static_cast<const float*>(tw.p[3]));
  }
  if (m_dev) {
    const int64_t mv = *m_dev;
    if (mv < M) M = mv;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < M; row += nw) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < K; k += 64) {
      const float h = H[row * K + k];
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (a < A) acc[a] = fmaf(h, W[k * A + a], acc[a]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
      if (a < A) {
        const float v = wave_sum(acc[a]);
        if (lane == 0) out[row * A + a] = v + b[a];
      }
  }
}

// tw (optional): {H, W, b, out} of a second head of the same shape, same launch (grid.y == 2)
int launch_head_fwd(const float* H, const float* W, const float* b, float* out, int64_t M, int K, int A,
                    hipStream_t st, const int32_t* m_dev, const Twin* tw) {
  const unsigned gy = tw ? 2 : 1;
  const Twin t2 = tw ? *tw : Twin{};
  static AttrOnce lds_opt_in;
  if (!lds_opt_in.done()) {
#define RLX_HEAD_ATTR(R, Q)                                                                                                     \
  RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_fwd<R, Q>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    RLX_HEAD_ATTR(16, 1); RLX_HEAD_ATTR(16, 2); RLX_HEAD_ATTR(16, 3); RLX_HEAD_ATTR(16, 4);
    RLX_HEAD_ATTR(64, 2); RLX_HEAD_ATTR(64, 4); RLX_HEAD_ATTR(64, 8); RLX_HEAD_ATTR(64, 16);
#undef RLX_HEAD_ATTR
    lds_opt_in.mark();
  }
  if (A <= 4) {
    int grid = div_up(M, 4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_head_fwd_narrow, dim3(grid, gy), dim3(256), 0, st, H, W, b, out, M, K, A, m_dev, t2);
  } else if (M < 65536) {
    const size_t lds = (size_t)head_fwd_lds_floats(16, K, A) * sizeof(float);
    RLX_REQUIRE(K % 4 == 0 && A <= 64 && lds <= 160 * 1024, RLX_EUNSUP, "head forward: K % 4 == 0, A <= 64, tile within the LDS");
    RLX_HEAD_FWD_16(A, dim3(div_up(M, 16), gy), lds, st, H, W, b, out, M, K, A, m_dev, 0, A, t2);
  } else {
    const size_t lds = (size_t)head_fwd_lds_floats(64, K, A) * sizeof(float);
    RLX_REQUIRE(K % 4 == 0 && A <= 64 && lds <= 160 * 1024, RLX_EUNSUP, "head forward: K % 4 == 0, A <= 64, tile within the LDS");
    RLX_HEAD_FWD_64(A, dim3(div_up(M, 64), gy), lds, st, H, W, b, out, M, K, A, m_dev, 0, A, t2);
  }
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// trunk forward: x -> acts[0..n_hidden-1] (acts[l] is [M, hidden[l]])
bool dx_cols_ok(int K, int nc) {
  return nc >= 1 && nc <= 64 && K % 4 == 0 && (size_t)head_fwd_lds_floats(16, K, nc) * sizeof(float) <= HEAD_FWD_MAX_LDS;
}

int launch_dx_cols(const float* dZ, const float* Wblk, float* dX, int64_t M, int K, int nc, int ldo, hipStream_t st,
                   const Twin* tw) {
  const size_t lds = (size_t)head_fwd_lds_floats(16, K, nc) * sizeof(float);
  RLX_REQUIRE(nc <= 64 && K % 4 == 0 && lds <= HEAD_FWD_MAX_LDS, RLX_EUNSUP, "launch_dx_cols: at most 64 columns, tile within 128 KB of LDS");
  RLX_HEAD_FWD_16(nc, dim3(div_up(M, 16), tw ? 2 : 1), lds, st, dZ, Wblk, (const float*)nullptr, dX, M, K, nc,
                  (const int32_t*)nullptr, 1, ldo, tw ? *tw : Twin{});
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int mlp_trunk_fwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x,
                  float* const* acts, int64_t M, hipStream_t st, int ldx, bool gemm_l0, const int32_t* m_dev, int n_layers) {
  int rc;
  const int nl = (n_layers >= 1 && n_layers <= d.n_hidden) ? n_layers : d.n_hidden;
  int l_next = 1;
  if (ctx->l12_fused && d.in_dim <= 32 && !gemm_l0 && !m_dev && nl >= 2 && M >= 4096 && (ldx <= 0 || ldx == d.in_dim) &&
      l12fwd_supported(d)) {
    // first + second layer in one launch (k_l12fwd) when both forward split images are registered (the update passes)
    const LayerOff &o0 = L.layer[0], &o1 = L.layer[1];
    const void* w1x = bx_lookup(ctx, params + o0.W, 0, o0.in, o0.out);
    const void* w2x = w1x ? bx_lookup(ctx, params + o1.W, 0, o1.in, o1.out) : nullptr;
    if (w1x && w2x) {
      rc = launch_l12fwd(ctx, d, L, params, x, acts[0], acts[1], w1x, w2x, M, st, nullptr, ctx->l12_stats);
      if (rc) return rc;
      l_next = 2;
      ctx->l12_ran = ctx->l12_stats != nullptr;
    }
  }
  if (l_next == 2) {
  } else if (d.in_dim <= 32 && !gemm_l0) {
    RLX_REQUIRE(ldx <= 0 || ldx == d.in_dim, RLX_EUNSUP, "mlp: padded input rows need in_dim > 32");
    rc = launch_l1_fwd(d, L, params, x, acts[0], M, ctx->num_cus, st, ctx->l1fwd_mfma, m_dev, ctx);
  } else {
    const LayerOff& o = L.layer[0];
    const int ld = ldx > 0 ? ldx : d.in_dim;
    RLX_REQUIRE(ld % 4 == 0 && ld >= d.in_dim, RLX_EUNSUP, "mlp: wide inputs need a row stride that is a multiple of 4");
    if (d.ln_first) {
      // Dense -> acts[3] (kept: the backward needs the pre-LayerNorm values), LayerNorm + activation -> acts[0]
      RLX_REQUIRE(acts[3] != nullptr && d.n_hidden >= 2, RLX_EUNSUP, "mlp: wide LayerNorm layer without its pre-activation buffer");
      rc = launch_gemm_fwd(ctx, x, params + o.W, params + o.b, acts[3], M, o.out, o.in, RLX_ACT_NONE, st, ld, m_dev);
      if (rc) return rc;
      int grid = div_up(M, 4);
      if (grid > ctx->num_cus * 8) grid = ctx->num_cus * 8;
      hipLaunchKernelGGL(k_ln_act<false>, dim3(grid), dim3(256), 0, st, acts[3], acts[0], params + o.g, params + o.be,
                         (float*)nullptr, M, o.out, d.act);
      RLX_LAUNCH_CHECK();
      rc = RLX_OK;
    } else {
      rc = launch_gemm_fwd(ctx, x, params + o.W, params + o.b, acts[0], M, o.out, o.in, d.act, st, ld, m_dev);
    }
  }
  if (rc) return rc;
  for (int l = l_next; l < nl; ++l) {
    const LayerOff& o = L.layer[l];
    rc = launch_gemm_fwd(ctx, acts[l - 1], params + o.W, params + o.b, acts[l], M, o.out, o.in, d.act, st, 0, m_dev);
    if (rc) return rc;
  }
  return RLX_OK;
}

// choose the M-split so the dW grid has ~2 workgroups per CU
int64_t choose_mc(int64_t M, int tiles, int num_cus, int* S_out) {
  int S = (num_cus + tiles - 1) / tiles;
  if (S < 1) S = 1;
  int64_t Mc = (M + S - 1) / S;
  Mc = ((Mc + G_BK - 1) / G_BK) * G_BK;
  if (Mc < G_BK) Mc = G_BK;
  S = (int)((M + Mc - 1) / Mc);
  if (S < 1) S = 1;
  *S_out = S;
  return Mc;
}

bool dw_merge_ok(const rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, int64_t M) {
  return ctx->dw_merge && d.n_hidden == 3 && bx_dw_usable(ctx, M, L.layer[2].in, L.layer[2].in, L.layer[2].out) &&
         bx_dw_usable(ctx, M, L.layer[1].in, L.layer[1].in, L.layer[1].out);
}

// M-split whose S * tiles workgroups FIT the CUs in one round (choose_mc rounds up: 26 x 10 = 260 workgroups on 256 CUs is a second
// round for four of them -- measured: the two-job launch took 77 us instead of 58)
int64_t choose_mc_fit(int64_t M, int tiles, int num_cus, int* S_out) {
  int S = num_cus / tiles;
  if (S < 1) S = 1;
  int64_t Mc = (M + S - 1) / S;
  Mc = ((Mc + G_BK - 1) / G_BK) * G_BK;
  if (Mc < G_BK) Mc = G_BK;
  S = (int)((M + Mc - 1) / Mc);
  *S_out = S < 1 ? 1 : S;
  return Mc;
}

// trunk backward.  On entry acts[last] holds dZ_last (head kernel wrote it in place);
// on exit acts[l] hold dZ_l.  Weight/bias/LN gradients are reduced into grads (flat).
// `extra` segments (head partials) are appended to the same reduction launch.
int mlp_trunk_bwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x,
                  float* const* acts, float* grads, int64_t M, const ReduceSeg* extra, int n_extra,
                  float* sumsq_partials, int* n_sumsq_blocks, hipStream_t st, const TrunkOpts* opt) {
  ReduceTable tab;
  tab.n = 0;
  const bool wide = d.in_dim > 32 || (opt && opt->gemm_l0);
  const bool wide_ln = wide && d.ln_first;
  RLX_REQUIRE(!wide_ln || (d.n_hidden >= 2 && acts[3] != nullptr), RLX_EUNSUP,
              "mlp: wide LayerNorm layer needs two hidden layers and its pre-activation buffer");
  const bool pgrads = grads != nullptr;   // nullptr: input-gradient only (parameters are stop_gradient'ed)
  const int ldx = (opt && opt->ldx > 0) ? opt->ldx : d.in_dim;
  // size the partial arena
  size_t need = 0;
  int S_l[4];
  int64_t Mc_l[4];
  for (int l = d.n_hidden - 1; l >= 0; --l) {
    const LayerOff& o = L.layer[l];
    const int tiles = (l == 0 && !wide) ? div_up(o.out, G_BN) : div_up(o.in, G_BM) * div_up(o.out, G_BN);
    Mc_l[l] = choose_mc(M, tiles, ctx->num_cus, &S_l[l]);
  }
  // The caller's tail kernel has produced dZ of BOTH upper layers (TrunkOpts::dz_below_last): their weight gradients go out as ONE
  // two-job launch (bx_launch_dw2) whose jobs share the CUs -- the same M-slabs for both, about half as many as alone, i.e.
  // half the slab bytes the reduction reads back.
  const bool dw_merge = pgrads && opt && opt->dz_below_last && dw_merge_ok(ctx, d, L, M);
  RLX_REQUIRE(dw_merge || !ctx->l12_ran, RLX_EUNSUP, "mlp bwd: the first-layer activations were not stored but cannot be recomputed here");
  if (dw_merge) {
    const int tiles = div_up(L.layer[2].in, G_BM) * div_up(L.layer[2].out, G_BN) + div_up(L.layer[1].in, G_BM) * div_up(L.layer[1].out, G_BN);
    Mc_l[2] = Mc_l[1] = choose_mc_fit(M, tiles, ctx->num_cus, &S_l[2]);
    S_l[1] = S_l[2];
  }
  for (int l = d.n_hidden - 1; l >= 0; --l) need += (size_t)S_l[l] * ((size_t)L.layer[l].in * L.layer[l].out + L.layer[l].out);
  const LayerOff& o0 = L.layer[0];
  const int l1_grid = rlx::l1_grid(M, ctx->num_cus);
  int ln_grid = div_up(M, 16);    // 16 rows per workgroup: S = M / 16 partial slabs for the reduction's chain (sac.hip: ln_bwd_grid)
  if (ln_grid > ctx->num_cus * 4) ln_grid = ctx->num_cus * 4;
  if (d.ln_first) need += (size_t)(wide_ln ? ln_grid : l1_grid) * 2 * o0.out;
  const bool fuse_l1 = pgrads && !wide && l1fused_supported(d) && !ctx->disable_l1fused;
  const int lf_grid = l1fused_grid(M, ctx->num_cus - ctx->lf_idle_cus > 8 ? ctx->num_cus - ctx->lf_idle_cus : 8);
  if (fuse_l1) need += l1fused_partial_floats(d, lf_grid);
  float* arena = (float*)scratch(ctx, SL_PARTIAL, need * sizeof(float));
  if (!arena) return RLX_ENOMEM;
  float* cur = arena;
  // gradient operand of layer l: acts[l] (written in place by the layer above), or the caller's buffer (TrunkOpts::dz_below_last)
  const float* dz[4] = {acts[0], acts[1], acts[2], acts[3]};
  const int pre = d.n_hidden - 2;
  const bool dz_ready = opt && opt->dz_below_last && pre >= 1;
  if (dz_ready) dz[pre] = opt->dz_below_last;

  // Plain two-hidden-layer net with a GEMM first layer (SAC: 256-256, observations wider than 32): dZ0 goes out of place (act' from
  // h1, which survives), and both weight gradients -- h1^T dZ1 and x^T dZ0 -- are ONE two-job launch after it instead of two launches
  // with the input gradient between them.
  const bool merge2 = pgrads && ctx->dw_merge && d.n_hidden == 2 && wide && !d.ln_first && !dz_ready && !fuse_l1 &&
                      bx_lookup(ctx, params + L.layer[1].W, 1, L.layer[1].out, L.layer[1].in) != nullptr &&
                      bx_dw_usable(ctx, M, L.layer[1].in, L.layer[1].in, L.layer[1].out) && bx_dw_usable(ctx, M, o0.in, ldx, o0.out);
  if (merge2) {
    const LayerOff& o1 = L.layer[1];
    float* dz0 = (float*)scratch(ctx, SL_DZ0, (size_t)M * o0.out * sizeof(float));
    if (!dz0) return RLX_ENOMEM;
    int rcm = bx_launch_dx(ctx, acts[1], bx_lookup(ctx, params + o1.W, 1, o1.out, o1.in), dz0, M, o1.out, o1.in, o1.in, d.act, 1, st,
                           nullptr, acts[0]);
    if (rcm) return rcm;
    const int tiles = div_up(o1.in, G_BM) * div_up(o1.out, G_BN) + div_up(o0.in, G_BM) * div_up(o0.out, G_BN);
    int Sm = 0;
    const int64_t Mcm = choose_mc_fit(M, tiles, ctx->num_cus, &Sm);
    RLX_REQUIRE(Sm <= S_l[0] && Sm <= S_l[1], RLX_EUNSUP, "mlp bwd: merged weight-gradient slabs exceed the arena");
    float* pW1 = cur; cur += (size_t)S_l[1] * o1.in * o1.out;      // (arena laid out for the unmerged slab counts: it only shrinks)
    float* pB1 = cur; cur += (size_t)S_l[1] * o1.out;
    float* pW0 = cur; cur += (size_t)S_l[0] * o0.in * o0.out;
    float* pB0 = cur; cur += (size_t)S_l[0] * o0.out;
    const BxDwJob j0{x, dz0, pW0, pB0, o0.in, ldx, o0.out, Mcm, Sm, div_up(o0.in, G_BM), div_up(o0.out, G_BN)};
    const BxDwJob j1{acts[0], acts[1], pW1, pB1, o1.in, o1.in, o1.out, Mcm, Sm, div_up(o1.in, G_BM), div_up(o1.out, G_BN)};
    rcm = bx_launch_dw2(ctx, j0, j1, M, st);
    if (rcm) return rcm;
    tab.seg[tab.n++] = ReduceSeg{pW1, grads + o1.W, (int64_t)o1.in * o1.out, (int64_t)o1.in * o1.out, Sm, 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{pB1, grads + o1.b, (int64_t)o1.out, (int64_t)o1.out, Sm, 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{pW0, grads + o0.W, (int64_t)o0.in * o0.out, (int64_t)o0.in * o0.out, Sm, 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{pB0, grads + o0.b, (int64_t)o0.out, (int64_t)o0.out, Sm, 0, 1.f, 0.f, 1};
    if (opt && opt->dx_out) RLX_REQUIRE(false, RLX_EUNSUP, "mlp bwd: input-gradient columns with parameter gradients");
    for (int e = 0; e < n_extra; ++e) tab.seg[tab.n++] = extra[e];
    return launch_reduce_segments(tab, sumsq_partials, n_sumsq_blocks, st, ctx);
  }
  BxDwJob dw_job3{};
  // The first-layer backward (k_dx_l1bwd) goes IN FRONT of the two-job weight-gradient launch: both read dZ2 and neither reads what
  // the other writes, but in the two-chain schedule the order decides which of the other network's kernels each of them shares the
  // chip with -- the HBM-bound weight gradient behind it no longer runs beside the other chain's k_l12fwd (107 MB of writes).
  // MEASURED (same box, two builds, tools/ab_lib.sh): 69.35 -> 68.49 ms per iteration.
  bool l1_done = false;
  if (dw_merge && fuse_l1 && dz_ready) {
    float* lf_arena = arena + need - l1fused_partial_floats(d, lf_grid);
    const int rcf = launch_l1fused(ctx, d, L, params, x, dz[1], lf_arena, lf_grid, grads, M, &tab, st);
    if (rcf) return rcf;
    l1_done = true;
  }
  for (int l = d.n_hidden - 1; l >= 1; --l) {
    const LayerOff& o = L.layer[l];
    float* pW = cur; cur += (size_t)S_l[l] * o.in * o.out;
    float* pB = cur; cur += (size_t)S_l[l] * o.out;
    const int ntk = div_up(o.in, G_BM), ntn = div_up(o.out, G_BN);
    if (dw_merge) {
      BxDwJob job{acts[l - 1], dz[l], pW, pB, o.in, o.in, o.out, Mc_l[l], S_l[l], ntk, ntn};
      if (l == 2) dw_job3 = job;
      else {
        BxDwRecompute rcd;
        if (ctx->l12_ran) {      // h1 was not stored (k_l12fwd left the rows' statistics): the job rebuilds its operand
          rcd.X = x; rcd.W1x = bx_lookup(ctx, params + o0.W, 0, o0.in, o0.out); rcd.b1 = params + o0.b; rcd.g = params + o0.g;
          rcd.be = params + o0.be; rcd.stats = ctx->l12_stats; rcd.xmax = ctx->l1_xmax; rcd.O = o0.in; rcd.NT1 = o0.out / 32;
          job.rc = &rcd;
        }
        const int rcw = bx_launch_dw2(ctx, job, dw_job3, M, st);      // (the larger job first: its blocks start first)
        if (rcw) return rcw;
      }
      tab.seg[tab.n++] = ReduceSeg{pW, grads + o.W, (int64_t)o.in * o.out, (int64_t)o.in * o.out, S_l[l], 0, 1.f, 0.f, 1};
      tab.seg[tab.n++] = ReduceSeg{pB, grads + o.b, (int64_t)o.out, (int64_t)o.out, S_l[l], 0, 1.f, 0.f, 1};
    } else if (pgrads) {
      hipStream_t sw = st;
      if (bx_dw_usable(ctx, M, o.in, o.in, o.out)) {
        const int rcw = bx_launch_dw(ctx, acts[l - 1], dz[l], pW, pB, M, o.in, o.in, o.out, Mc_l[l], S_l[l], ntk, ntn, sw);
        if (rcw) return rcw;
      } else {
        ProfScope prof(ctx, PK_GEMM_DW, 2.0 * (double)M * o.in * o.out, sw, gemm_bytes(o.in, o.out, M), o.in, o.out, (int)M);
        RLX_PLAUNCH(k_gemm_dw, dim3(S_l[l] * ntk * ntn), dim3(G_THREADS), 0, sw, (const float*)acts[l - 1], dz[l], pW, pB, M,
                           o.in, o.in, o.out, Mc_l[l], ntk, ntn);
      }
      RLX_LAUNCH_CHECK();
      tab.seg[tab.n++] = ReduceSeg{pW, grads + o.W, (int64_t)o.in * o.out, (int64_t)o.in * o.out, S_l[l], 0, 1.f, 0.f, 1};
      tab.seg[tab.n++] = ReduceSeg{pB, grads + o.b, (int64_t)o.out, (int64_t)o.out, S_l[l], 0, 1.f, 0.f, 1};
    }
    if (l == 1 && fuse_l1) continue;  // layer-1 input gradient is folded into launch_l1fused below
    if (dz_ready && l == d.n_hidden - 1) continue;   // dZ_{l-1} came with dZ_last from the caller's tail kernel
    // dZ_{l-1} = (dZ_l @ W_l^T) * act'(H_{l-1})   in place over acts[l-1]
    const int ntn2 = div_up(o.in, G_BN);
    const int apply = (l - 1 == 0 && (!wide || wide_ln)) ? 0 : 1;  // first layer with LayerNorm / narrow: act' and LN' applied later
    if (const void* img = bx_lookup(ctx, params + o.W, 1, o.out, o.in)) {
      const int rcx = bx_launch_dx(ctx, dz[l], img, acts[l - 1], M, o.out, o.in, o.in, d.act, apply, st);
      if (rcx) return rcx;
    } else {
      ProfScope prof(ctx, PK_GEMM_DX, 2.0 * (double)M * o.in * o.out, st, gemm_bytes(M, o.in, o.out, apply), M, o.in, o.out);
      RLX_GEMM_DX_LAUNCH(d.act, apply, dim3(div_up(M, G_BM) * ntn2), st, dz[l], params + o.W, acts[l - 1], M, o.out, o.in,
                         o.in, ntn2, (const float*)nullptr);
      RLX_LAUNCH_CHECK();
    }
  }
  if (fuse_l1 && !l1_done) {
    float* lf_arena = cur; cur += l1fused_partial_floats(d, lf_grid);
    const int rcf = launch_l1fused(ctx, d, L, params, x, dz[1], lf_arena, lf_grid, grads, M, &tab, st);
    if (rcf) return rcf;
  }
  if (wide_ln) {
    // acts[0] holds dL/dH0 (raw); LayerNorm' and act' from the kept pre-LayerNorm values -> dZ0 in place
    float* pLNw = cur; cur += (size_t)ln_grid * 2 * o0.out;
    hipLaunchKernelGGL(k_ln_act<true>, dim3(ln_grid), dim3(256), (size_t)8 * o0.out * sizeof(float), st, acts[3], acts[0],
                       params + o0.g, params + o0.be, pLNw, M, o0.out, d.act);
    RLX_LAUNCH_CHECK();
    if (pgrads) {
      tab.seg[tab.n++] = ReduceSeg{pLNw, grads + o0.g, (int64_t)o0.out, (int64_t)2 * o0.out, ln_grid, 0, 1.f, 0.f, 1};
      tab.seg[tab.n++] = ReduceSeg{pLNw + o0.out, grads + o0.be, (int64_t)o0.out, (int64_t)2 * o0.out, ln_grid, 0, 1.f, 0.f, 1};
    }
  }
  if (wide) {
    // generic first layer: acts[0] already holds dZ0 (or dZ_head if n_hidden == 1)
    if (pgrads) {
      float* pW = cur; cur += (size_t)S_l[0] * o0.in * o0.out;
      float* pB = cur; cur += (size_t)S_l[0] * o0.out;
      const int ntk = div_up(o0.in, G_BM), ntn = div_up(o0.out, G_BN);
      if (bx_dw_usable(ctx, M, o0.in, ldx, o0.out)) {
        const int rcw = bx_launch_dw(ctx, x, acts[0], pW, pB, M, o0.in, ldx, o0.out, Mc_l[0], S_l[0], ntk, ntn, st);
        if (rcw) return rcw;
      } else {
        ProfScope prof(ctx, PK_GEMM_DW, 2.0 * (double)M * o0.in * o0.out, st, gemm_bytes(o0.in, o0.out, M), o0.in, o0.out, (int)M);
        RLX_PLAUNCH(k_gemm_dw, dim3(S_l[0] * ntk * ntn), dim3(G_THREADS), 0, st, x, acts[0], pW, pB, M, o0.in, ldx,
                           o0.out, Mc_l[0], ntk, ntn);
      }
      RLX_LAUNCH_CHECK();
      tab.seg[tab.n++] = ReduceSeg{pW, grads + o0.W, (int64_t)o0.in * o0.out, (int64_t)o0.in * o0.out, S_l[0], 0, 1.f, 0.f, 1};
      tab.seg[tab.n++] = ReduceSeg{pB, grads + o0.b, (int64_t)o0.out, (int64_t)o0.out, S_l[0], 0, 1.f, 0.f, 1};
    }
    if (opt && opt->dx_out) {
      // dL/dx[:, c0 : c0+nc] = dZ0 @ W0[c0 : c0+nc, :]^T
      RLX_REQUIRE(opt->dx_nc > 0 && opt->dx_c0 >= 0 && opt->dx_c0 + opt->dx_nc <= o0.in && opt->dx_ld >= opt->dx_nc,
                  RLX_EINVAL, "mlp bwd: bad input-gradient column range");
      if (opt->dx_nc <= 64 && (size_t)head_fwd_lds_floats(16, o0.out, opt->dx_nc) * sizeof(float) <= HEAD_FWD_MAX_LDS) {
        // a handful of input columns (SAC: dQ/da, 17 of 393): a 128-column MFMA tile would be 87 % padding and its grid M / 128
        // workgroups; the LDS-staged head kernel with the weight block read transposed does it in M / 16 workgroups
        const size_t lds = (size_t)head_fwd_lds_floats(16, o0.out, opt->dx_nc) * sizeof(float);
        RLX_HEAD_FWD_16(opt->dx_nc, dim3(div_up(M, 16)), lds, st, (const float*)acts[0],
                        params + o0.W + (int64_t)opt->dx_c0 * o0.out, (const float*)nullptr, opt->dx_out, M, o0.out,
                        opt->dx_nc, (const int32_t*)nullptr, 1, opt->dx_ld, Twin{});
        RLX_LAUNCH_CHECK();
      } else {
      const int ntn2 = div_up(opt->dx_nc, G_BN);
      ProfScope prof(ctx, PK_GEMM_DX, 2.0 * (double)M * opt->dx_nc * o0.out, st, gemm_bytes(M, opt->dx_nc, o0.out), M, opt->dx_nc, o0.out);
      RLX_GEMM_DX_LAUNCH(d.act, 0, dim3(div_up(M, G_BM) * ntn2), st, acts[0],
                         params + o0.W + (int64_t)opt->dx_c0 * o0.out, opt->dx_out, M, o0.out, opt->dx_nc, opt->dx_ld, ntn2,
                         (const float*)nullptr);
      RLX_LAUNCH_CHECK();
      }
    }
  }
  // first layer: dH1 -> dZ1 (recompute forward), LN scale/bias partials
  float* pLN = nullptr;
  if (!fuse_l1 && !wide) {
    RLX_REQUIRE(pgrads, RLX_EUNSUP, "mlp bwd: input-gradient-only mode needs in_dim > 32");
    if (d.ln_first) { pLN = cur; cur += (size_t)l1_grid * 2 * o0.out; }
    int rc1 = launch_l1<true>(x, params + o0.W, params + o0.b, o0.g >= 0 ? params + o0.g : nullptr,
                              o0.be >= 0 ? params + o0.be : nullptr, acts[0], pLN, M, o0.in, o0.out, d.act,
                              d.ln_first ? 1 : 0, l1_grid, st);
    if (rc1) return rc1;
    if (d.ln_first) {
      tab.seg[tab.n++] = ReduceSeg{pLN, grads + o0.g, (int64_t)o0.out, (int64_t)2 * o0.out, l1_grid, 0, 1.f, 0.f, 1};
      tab.seg[tab.n++] = ReduceSeg{pLN + o0.out, grads + o0.be, (int64_t)o0.out, (int64_t)2 * o0.out, l1_grid, 0, 1.f, 0.f, 1};
    }
    float* pW = cur; cur += (size_t)S_l[0] * o0.in * o0.out;
    float* pB = cur; cur += (size_t)S_l[0] * o0.out;
    const int ntn = div_up(o0.out, G_BN);
    RLX_REQUIRE(o0.in <= 32, RLX_EUNSUP, "mlp bwd: in_dim > 32 not supported by the skinny dW1 kernel yet");
    hipLaunchKernelGGL(k_gemm_dw_skinny, dim3(S_l[0] * ntn), dim3(G_THREADS), 0, st, x, acts[0], pW, pB, M, o0.in,
                       o0.out, Mc_l[0], ntn);
    RLX_LAUNCH_CHECK();
    tab.seg[tab.n++] = ReduceSeg{pW, grads + o0.W, (int64_t)o0.in * o0.out, (int64_t)o0.in * o0.out, S_l[0], 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{pB, grads + o0.b, (int64_t)o0.out, (int64_t)o0.out, S_l[0], 0, 1.f, 0.f, 1};
  }
  if (!pgrads) {
    if (n_sumsq_blocks) *n_sumsq_blocks = 0;
    return RLX_OK;
  }
  for (int e = 0; e < n_extra; ++e) tab.seg[tab.n++] = extra[e];
  return launch_reduce_segments(tab, sumsq_partials, n_sumsq_blocks, st, ctx);
}

// one launch reducing every segment of `tab` (block counts and the vector-path flags are filled in here)
int launch_reduce_segments(ReduceTable& tab, float* sumsq_partials, int* n_blocks_out, hipStream_t st, rlx_ctx* prof_ctx) {
  int total_blocks = 0;
  double bytes = 0.0;
  int64_t outputs = 0;
  for (int i = 0; i < tab.n; ++i) {
    bytes += 4.0 * (double)tab.seg[i].len * (tab.seg[i].S + 1);     // every slab once in, the reduced values once out
    outputs += tab.seg[i].len;
  }
  ProfScope prof(prof_ctx, PK_REDUCE, 0.0, st, bytes, outputs, tab.n, 0, PROF_ENGINE_HBM);
  for (int i = 0; i < tab.n; ++i) {
    ReduceSeg& g = tab.seg[i];
    g.vec = (g.len % 4 == 0 && g.stride % 4 == 0 && (reinterpret_cast<uintptr_t>(g.src) & 15) == 0 && g.len >= 256)
                ? 1 : 0;
    if (g.vec && g.S >= 128) g.vec = 2;      // many slabs: 64 outputs per workgroup, 16 residue classes (k_reduce_segments)
    g.nblocks = g.vec == 2 ? div_up(g.len, 64) : (g.vec ? div_up(g.len, 256) : div_up(g.len, 16));
    total_blocks += g.nblocks;
  }
  RLX_REQUIRE(total_blocks <= REDUCE_MAX_BLOCKS, RLX_EUNSUP, "mlp bwd: too many reduction blocks");
  RLX_PLAUNCH(k_reduce_segments, dim3(total_blocks), dim3(256), 0, st, tab, sumsq_partials);
  RLX_LAUNCH_CHECK();
  if (n_blocks_out) *n_blocks_out = total_blocks;
  return RLX_OK;
}

// ---------------------------------------------------------------------------------------
// Stage helpers for composite models (ppo_lstm.hip): each launches the gradient kernel(s) of ONE layer,
// reduces the partial slabs straight into `g*` and appends its sum-of-squares partials at sumsq + *nsq.
// ---------------------------------------------------------------------------------------
static inline size_t align64(size_t n) { return (n + 63) & ~size_t(63); }

float* stage_alloc(rlx_ctx* ctx, size_t floats) {
  ReduceDefer* D = static_cast<ReduceDefer*>(ctx->defer);
  if (!D) return (float*)scratch(ctx, SL_STAGE, floats * sizeof(float));
  if (D->off + align64(floats) > D->cap) {
    set_error("stage_alloc: the deferred-reduction arena is too small (stage_*_floats out of step with the stages)");
    return nullptr;
  }
  float* p = D->base + D->off;
  D->off += align64(floats);
  return p;
}

size_t stage_dw_floats(const rlx_ctx* ctx, int64_t M, int Kd, int N) {
  int S = 1;
  (void)choose_mc(M, div_up(Kd, G_BM) * div_up(N, G_BN), ctx->num_cus, &S);
  return align64((size_t)S * Kd * N + (size_t)S * N);
}

size_t stage_l1_bwd_floats(const rlx_ctx* ctx, int64_t M, int O, int Hd) {
  int S = 1;
  (void)choose_mc(M, div_up(Hd, G_BN), ctx->num_cus, &S);
  return align64((size_t)l1_grid(M, ctx->num_cus) * 2 * Hd + (size_t)S * (O + 1) * Hd);
}

static int reduce_launch(rlx_ctx* ctx, ReduceTable& tab, float* sumsq, int* nsq, hipStream_t st);

int stage_reduce_flush(rlx_ctx* ctx, float* sumsq, int* nsq, hipStream_t st) {
  ReduceDefer* D = static_cast<ReduceDefer*>(ctx->defer);
  ctx->defer = nullptr;
  if (!D || D->tab.n == 0) return RLX_OK;
  return reduce_launch(ctx, D->tab, sumsq, nsq, st);
}

static int reduce_now(rlx_ctx* ctx, ReduceTable& tab, float* sumsq, int* nsq, hipStream_t st) {
  if (ReduceDefer* D = static_cast<ReduceDefer*>(ctx->defer)) {
    RLX_REQUIRE(D->tab.n + tab.n <= REDUCE_MAX_SEGS, RLX_EUNSUP, "deferred reduction: too many segments");
    for (int i = 0; i < tab.n; ++i) D->tab.seg[D->tab.n++] = tab.seg[i];
    return RLX_OK;
  }
  return reduce_launch(ctx, tab, sumsq, nsq, st);
}

static int reduce_launch(rlx_ctx* ctx, ReduceTable& tab, float* sumsq, int* nsq, hipStream_t st) {
  int total = 0;
  for (int i = 0; i < tab.n; ++i) {
    ReduceSeg& g = tab.seg[i];
    g.vec = (g.len % 4 == 0 && g.stride % 4 == 0 && (reinterpret_cast<uintptr_t>(g.src) & 15) == 0 && g.len >= 256) ? 1 : 0;
    if (g.vec && g.S >= 128) g.vec = 2;      // many slabs: 64 outputs per workgroup, 16 residue classes (k_reduce_segments)
    g.nblocks = g.vec == 2 ? div_up(g.len, 64) : (g.vec ? div_up(g.len, 256) : div_up(g.len, 16));
    total += g.nblocks;
  }
  RLX_REQUIRE(!nsq || *nsq + total <= REDUCE_MAX_BLOCKS, RLX_EUNSUP, "stage reduce: norm partial array exhausted");
  hipLaunchKernelGGL(k_reduce_segments, dim3(total), dim3(256), 0, st, tab, sumsq ? sumsq + (nsq ? *nsq : 0) : nullptr);
  RLX_LAUNCH_CHECK();
  if (nsq) *nsq += total;
  return RLX_OK;
}

int stage_reduce(rlx_ctx* ctx, ReduceTable& tab, float* sumsq, int* nsq, hipStream_t st) {
  return reduce_now(ctx, tab, sumsq, nsq, st);
}

// gW[Kd,N] = Hp[M,Kd(ld)]^T @ dZ[M,N];  gB[N] = colsum(dZ) (optional)
int stage_dw(rlx_ctx* ctx, const float* Hp, int ldh, const float* dZ, int64_t M, int Kd, int N, float* gW, float* gB,
             float* sumsq, int* nsq, hipStream_t st) {
  const int ntk = div_up(Kd, G_BM), ntn = div_up(N, G_BN);
  int S = 1;
  const int64_t Mc = choose_mc(M, ntk * ntn, ctx->num_cus, &S);
  float* pW = stage_alloc(ctx, (size_t)S * Kd * N + (size_t)S * N);
  if (!pW) return RLX_ENOMEM;
  float* pB = pW + (size_t)S * Kd * N;
  if (bx_dw_usable(ctx, M, Kd, ldh, N)) {
    const int rcw = bx_launch_dw(ctx, Hp, dZ, pW, pB, M, Kd, ldh, N, Mc, S, ntk, ntn, st);
    if (rcw) return rcw;
  } else {
    ProfScope prof(ctx, PK_GEMM_DW, 2.0 * (double)M * Kd * N, st, gemm_bytes(Kd, N, M), Kd, N, (int)M);
    RLX_PLAUNCH(k_gemm_dw, dim3(S * ntk * ntn), dim3(G_THREADS), 0, st, Hp, dZ, pW, pB, M, Kd, ldh, N, Mc, ntk, ntn);
  }
  RLX_LAUNCH_CHECK();
  ReduceTable tab;
  tab.n = 0;
  tab.seg[tab.n++] = ReduceSeg{pW, gW, (int64_t)Kd * N, (int64_t)Kd * N, S, 0, 1.f, 0.f, 1};
  if (gB) tab.seg[tab.n++] = ReduceSeg{pB, gB, (int64_t)N, (int64_t)N, S, 0, 1.f, 0.f, 1};
  return reduce_now(ctx, tab, sumsq, nsq, st);
}

// out[M, Kd(ldo)] = (dZ[M,N] @ W[Kd,N]^T) (* act'(out) when apply_act)
int stage_dx(rlx_ctx* ctx, const float* dZ, const float* W, float* out, int64_t M, int N, int Kd, int ldo, int act,
             int apply_act, hipStream_t st, const float* hsrc) {
  if (const void* img = bx_lookup(ctx, W, 1, N, Kd)) return bx_launch_dx(ctx, dZ, img, out, M, N, Kd, ldo, act, apply_act, st, nullptr, hsrc);
  const int ntn = div_up(Kd, G_BN);
  ProfScope prof(ctx, PK_GEMM_DX, 2.0 * (double)M * N * Kd, st, gemm_bytes(M, Kd, N, apply_act), M, Kd, N);
  RLX_GEMM_DX_LAUNCH(act, apply_act, dim3(div_up(M, G_BM) * ntn), st, dZ, W, out, M, N, Kd, ldo, ntn, hsrc);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// fused small-input layer (Dense + LN + act): forward
int stage_l1_fwd(rlx_ctx* ctx, const float* x, const float* W, const float* b, const float* g, const float* be, float* H,
                 int64_t M, int O, int Hd, int act, int ln, hipStream_t st) {
  return launch_l1<false>(x, W, b, g, be, H, nullptr, M, O, Hd, act, ln, l1_grid(M, ctx->num_cus), st);
}

// two layers of one shape on the same rows in ONE launch (grid.y = 2): W2 .. H2 = the second layer
int stage_l1_fwd2(rlx_ctx* ctx, const float* x, const float* W, const float* b, const float* g, const float* be, float* H,
                  const float* W2, const float* b2, const float* g2, const float* be2, float* H2, int64_t M, int O, int Hd,
                  int act, int ln, hipStream_t st) {
  L1Twin tw;
  tw.W = W2; tw.b = b2; tw.g = g2; tw.be = be2; tw.H = H2;
  return launch_l1<false>(x, W, b, g, be, H, nullptr, M, O, Hd, act, ln, l1_grid(M, ctx->num_cus), st, nullptr, &tw);
}

// ... and backward: H holds dL/dH on entry, dZ on exit; gradients reduced into gW, gb, gg, gbe
int stage_l1_bwd(rlx_ctx* ctx, const float* x, const float* W, const float* b, const float* g, const float* be, float* H,
                 int64_t M, int O, int Hd, int act, int ln, float* gW, float* gb, float* gg, float* gbe, float* sumsq,
                 int* nsq, hipStream_t st) {
  RLX_REQUIRE(O <= 32, RLX_EUNSUP, "stage_l1_bwd: in_dim must be <= 32");
  const int grid = l1_grid(M, ctx->num_cus);
  const int ntn = div_up(Hd, G_BN);
  int S = 1;
  const int64_t Mc = choose_mc(M, ntn, ctx->num_cus, &S);
  float* arena = stage_alloc(ctx, (size_t)grid * 2 * Hd + (size_t)S * (O + 1) * Hd);
  if (!arena) return RLX_ENOMEM;
  float* pLN = arena;
  float* pW = arena + (size_t)grid * 2 * Hd;
  float* pB = pW + (size_t)S * O * Hd;
  int rc = launch_l1<true>(x, W, b, g, be, H, ln ? pLN : nullptr, M, O, Hd, act, ln, grid, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_gemm_dw_skinny, dim3(S * ntn), dim3(G_THREADS), 0, st, x, H, pW, pB, M, O, Hd, Mc, ntn);
  RLX_LAUNCH_CHECK();
  ReduceTable tab;
  tab.n = 0;
  tab.seg[tab.n++] = ReduceSeg{pW, gW, (int64_t)O * Hd, (int64_t)O * Hd, S, 0, 1.f, 0.f, 1};
  tab.seg[tab.n++] = ReduceSeg{pB, gb, (int64_t)Hd, (int64_t)Hd, S, 0, 1.f, 0.f, 1};
  if (ln) {
    tab.seg[tab.n++] = ReduceSeg{pLN, gg, (int64_t)Hd, (int64_t)2 * Hd, grid, 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{pLN + Hd, gbe, (int64_t)Hd, (int64_t)2 * Hd, grid, 0, 1.f, 0.f, 1};
  }
  return reduce_now(ctx, tab, sumsq, nsq, st);
}

// the backward of stage_l1_fwd2: ONE k_l1<bwd> launch for both layers (each its own partial arena), then their skinny
// weight-gradient kernels; gradients of layer 2 into gW2 .. gbe2
int stage_l1_bwd2(rlx_ctx* ctx, const float* x, const float* W, const float* b, const float* g, const float* be, float* H,
                  const float* W2, const float* b2, const float* g2, const float* be2, float* H2, int64_t M, int O, int Hd, int act,
                  int ln, float* gW, float* gb, float* gg, float* gbe, float* gW2, float* gb2, float* gg2, float* gbe2, float* sumsq,
                  int* nsq, hipStream_t st) {
  RLX_REQUIRE(O <= 32, RLX_EUNSUP, "stage_l1_bwd2: in_dim must be <= 32");
  const int grid = l1_grid(M, ctx->num_cus);
  const int ntn = div_up(Hd, G_BN);
  int S = 1;
  const int64_t Mc = choose_mc(M, ntn, ctx->num_cus, &S);
  float* arena[2];
  for (int k = 0; k < 2; ++k) {
    arena[k] = stage_alloc(ctx, (size_t)grid * 2 * Hd + (size_t)S * (O + 1) * Hd);
    if (!arena[k]) return RLX_ENOMEM;
  }
  RLX_REQUIRE(ctx->defer || arena[0] != arena[1], RLX_EUNSUP, "stage_l1_bwd2 needs the deferred-reduction arena (two live partial sets)");
  L1Twin tw;
  tw.W = W2; tw.b = b2; tw.g = g2; tw.be = be2; tw.H = H2; tw.lnp = ln ? arena[1] : nullptr;
  int rc = launch_l1<true>(x, W, b, g, be, H, ln ? arena[0] : nullptr, M, O, Hd, act, ln, grid, st, nullptr, &tw);
  if (rc) return rc;
  ReduceTable tab;
  tab.n = 0;
  float* Hs[2] = {H, H2};
  float* gWs[2] = {gW, gW2};
  float* gbs[2] = {gb, gb2};
  float* ggs[2] = {gg, gg2};
  float* gbes[2] = {gbe, gbe2};
  for (int k = 0; k < 2; ++k) {
    float* pLN = arena[k];
    float* pW = arena[k] + (size_t)grid * 2 * Hd;
    float* pB = pW + (size_t)S * O * Hd;
    hipLaunchKernelGGL(k_gemm_dw_skinny, dim3(S * ntn), dim3(G_THREADS), 0, st, x, Hs[k], pW, pB, M, O, Hd, Mc, ntn);
    RLX_LAUNCH_CHECK();
    tab.seg[tab.n++] = ReduceSeg{pW, gWs[k], (int64_t)O * Hd, (int64_t)O * Hd, S, 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{pB, gbs[k], (int64_t)Hd, (int64_t)Hd, S, 0, 1.f, 0.f, 1};
    if (ln) {
      tab.seg[tab.n++] = ReduceSeg{pLN, ggs[k], (int64_t)Hd, (int64_t)2 * Hd, grid, 0, 1.f, 0.f, 1};
      tab.seg[tab.n++] = ReduceSeg{pLN + Hd, gbes[k], (int64_t)Hd, (int64_t)2 * Hd, grid, 0, 1.f, 0.f, 1};
    }
  }
  return reduce_now(ctx, tab, sumsq, nsq, st);
}

}  // namespace rlx

using namespace rlx;

// Debug / micro-benchmark hook (tests/test_gpu_gemm.py, tools/gemm_bench.py): run ONE of the MFMA
// GEMM kernels on caller buffers.
//   mode 0: C[M,N]  = act(A[M,K] @ B[K,N] + bias[N])                (k_gemm_fwd;  aux = bias)
//   mode 1: C[M,K] <- (A[M,N] @ B[K,N]^T) * act'(C[M,K]) in place    (k_gemm_dx;   aux unused, act applied iff act>=0)
//   mode 2: C[K,N]  = A[M,K]^T @ B[M,N], aux[N] = column sums of B   (k_gemm_dw + slab reduction)
extern "C" int rlx_dbg_gemm_f32(rlx_ctx* ctx, int mode, const float* A, const float* B, float* C, float* aux,
                                int64_t M, int N, int K, int act, void* stream) {
  RLX_REQUIRE(ctx && A && B && C && M > 0 && N > 0 && K > 0, RLX_EINVAL, "rlx_dbg_gemm_f32: bad args");
  RLX_REQUIRE(N % 4 == 0 && K % 4 == 0, RLX_EUNSUP, "rlx_dbg_gemm_f32: N and K must be multiples of 4");
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    RLX_REQUIRE(aux, RLX_EINVAL, "rlx_dbg_gemm_f32: mode 0 needs bias");
    return launch_gemm_fwd(ctx, A, B, aux, C, M, N, K, act, st, 0, nullptr);
  }
  if (mode == 1) {
    const int ntn = div_up(K, G_BN);
    ProfScope prof(ctx, PK_GEMM_DX, 2.0 * (double)M * N * K, st, gemm_bytes(M, K, N, act >= 0 ? 1 : 0));
    RLX_GEMM_DX_LAUNCH(act >= 0 ? act : 0, act >= 0 ? 1 : 0, dim3(div_up(M, G_BM) * ntn), st, A, B, C, M, N, K, K, ntn, (const float*)nullptr);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
  }
  if (mode == 2 || mode == 5) {   // 5: the split-fp16 weight-gradient kernel
    const int ntk = div_up(K, G_BM), ntn = div_up(N, G_BN);
    int S = 1;
    const int64_t Mc = choose_mc(M, ntk * ntn, ctx->num_cus, &S);
    float* pW = (float*)scratch(ctx, SL_PARTIAL, ((size_t)S * K * N + (size_t)S * N) * sizeof(float));
    if (!pW) return RLX_ENOMEM;
    float* pB = pW + (size_t)S * K * N;
    if (mode == 5) {
      const int rcw = bx_launch_dw(ctx, A, B, pW, pB, M, K, K, N, Mc, S, ntk, ntn, st);
      if (rcw) return rcw;
    } else {
      ProfScope prof(ctx, PK_GEMM_DW, 2.0 * (double)M * N * K, st, gemm_bytes(K, N, M));
      RLX_PLAUNCH(k_gemm_dw, dim3(S * ntk * ntn), dim3(G_THREADS), 0, st, A, B, pW, pB, M, K, K, N, Mc, ntk, ntn);
    }
    RLX_LAUNCH_CHECK();
    ReduceTable tab;
    tab.n = 0;
    tab.seg[tab.n++] = ReduceSeg{pW, C, (int64_t)K * N, (int64_t)K * N, S, 0, 1.f, 0.f, 0};
    if (aux) tab.seg[tab.n++] = ReduceSeg{pB, aux, (int64_t)N, (int64_t)N, S, 0, 1.f, 0.f, 0};
    int total = 0;
    for (int i = 0; i < tab.n; ++i) {
      ReduceSeg& g = tab.seg[i];
      g.vec = (g.len % 4 == 0 && g.stride % 4 == 0 && (reinterpret_cast<uintptr_t>(g.src) & 15) == 0 && g.len >= 256) ? 1 : 0;
      if (g.vec && g.S >= 128) g.vec = 2;      // many slabs: 64 outputs per workgroup, 16 residue classes (k_reduce_segments)
      g.nblocks = g.vec == 2 ? div_up(g.len, 64) : (g.vec ? div_up(g.len, 256) : div_up(g.len, 16));
      total += g.nblocks;
    }
    hipLaunchKernelGGL(k_reduce_segments, dim3(total), dim3(256), 0, st, tab, (float*)nullptr);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
  }
  if (mode == 3 || mode == 4) {
    // the split-fp16 forms of modes 0 / 1: lay out the weight image, run, drop it again
    rlx_mlp_desc d{};
    MlpLayout L{};
    d.n_hidden = 2;
    L.n_hidden = 2;
    LayerOff& o = L.layer[1];
    o.W = 0;
    o.in = K;   // B is [K rows, N cols] row-major in both modes
    o.out = N;
    const bool was = ctx->gemm_bx;
    ctx->gemm_bx = true;
    int rc = bx_prepare_mlp(ctx, d, L, B, mode == 4, st);
    if (!rc) {
      if (mode == 3) {
        RLX_REQUIRE(aux, RLX_EINVAL, "rlx_dbg_gemm_f32: mode 3 needs bias");
        const void* img = bx_lookup(ctx, B, 0, K, N);
        rc = img ? bx_launch_fwd(ctx, A, img, aux, C, M, N, K, act, st, 0, nullptr) : RLX_EUNSUP;
      } else {
        const void* img = bx_lookup(ctx, B, 1, N, K);
        rc = img ? bx_launch_dx(ctx, A, img, C, M, N, K, K, act >= 0 ? act : 0, act >= 0 ? 1 : 0, st) : RLX_EUNSUP;
      }
    }
    bx_release(ctx);
    ctx->gemm_bx = was;
    return rc;
  }
  RLX_REQUIRE(false, RLX_EINVAL, "rlx_dbg_gemm_f32: mode must be 0 .. 5");
}

// debug / micro-benchmark hook for the first-layer kernel (fwd: H out; bwd: H = dH in -> dZ1 out)
extern "C" int rlx_dbg_l1_f32(rlx_ctx* ctx, int bwd, const float* X, const float* W, const float* b, const float* g,
                              const float* be, float* H, float* ln_partials, int64_t M, int O, int Hd, int act, int ln,
                              int grid, void* stream) {
  RLX_REQUIRE(ctx && X && W && b && H && M > 0 && O >= 1 && O <= 32 && Hd % 64 == 0 && Hd <= 512 && grid > 0,
              RLX_EINVAL, "rlx_dbg_l1_f32: bad args");
  if (bwd) return launch_l1<true>(X, W, b, g, be, H, ln_partials, M, O, Hd, act, ln, grid, (hipStream_t)stream);
  return launch_l1<false>(X, W, b, g, be, H, nullptr, M, O, Hd, act, ln, grid, (hipStream_t)stream);
}

extern "C" int rlx_mlp_fwd_f32(rlx_ctx* ctx, const rlx_mlp_desc* desc, const float* params, const float* x, float* out,
                               int64_t n, void* stream) {
  RLX_REQUIRE(ctx && desc && params && x && out && n >= 0, RLX_EINVAL, "rlx_mlp_fwd_f32: bad args");
  if (n == 0) return RLX_OK;
  int rc = mlp_check_desc(*desc);
  if (rc) return rc;
  const MlpLayout L = make_layout(*desc);
  hipStream_t st = (hipStream_t)stream;
  int maxh = 0;
  for (int l = 0; l < desc->n_hidden; ++l) maxh = desc->hidden[l] > maxh ? desc->hidden[l] : maxh;
  float* bufA = (float*)scratch(ctx, SL_FWD_A, (size_t)n * maxh * sizeof(float));
  float* bufB = (float*)scratch(ctx, SL_FWD_B, (size_t)n * maxh * sizeof(float));
  if (!bufA || !bufB) return RLX_ENOMEM;
  float* acts[4] = {bufA, bufB, bufA, bufB};
  int ldx = 0;
  if (desc->in_dim > 32 && desc->in_dim % 4 != 0) {
    // the GEMM first layer loads 16-B vectors: re-pitch the rows to a multiple of 4 (pad columns meet guarded weight rows)
    ldx = (desc->in_dim + 3) & ~3;
    float* xp = (float*)scratch(ctx, SL_STAGE, (size_t)n * ldx * sizeof(float));
    if (!xp) return RLX_ENOMEM;
    RLX_HIP_TRY(hipMemcpy2DAsync(xp, (size_t)ldx * sizeof(float), x, (size_t)desc->in_dim * sizeof(float),
                                 (size_t)desc->in_dim * sizeof(float), (size_t)n, hipMemcpyDeviceToDevice, st));
    x = xp;
  }
  // large batches: the hidden-layer GEMMs on the fp16 pipe (images laid out for this call only, unless a caller's are registered)
  const bool own_images = n >= 4096 && ctx->gemm_bx && ctx->bx_n[0] == 0 && ctx->bx_n[1] == 0;
  struct BxOwn { rlx_ctx* c; bool on; ~BxOwn() { if (on) bx_release_all(c); } } bx_own{ctx, own_images};
  if (own_images) {
    const BxNetSpec net = {desc, params, false, desc->in_dim > 32};
    rc = bx_prepare_nets(ctx, &net, 1, st);
    if (rc) return rc;
  }
  {
    const void *w1x = nullptr, *w2x = nullptr;      // 256-256 nets with registered images: trunk + head in one launch, nothing stored
    if (fwd2h_supported(ctx, *desc, L, params, n, ldx, &w1x, &w2x))
      return launch_fwd2h(ctx, *desc, L, params, w1x, w2x, x, ldx, nullptr, nullptr, out, n, st);
    const void* wx[3];                              // 512-LayerNorm-256-128 nets with a wide input: k_fwd3h
    if (fwd3h_supported(ctx, *desc, L, params, n, ldx, wx))
      return launch_fwd3h(ctx, *desc, L, params, wx, x, ldx, nullptr, nullptr, nullptr, nullptr, out, n, st);
  }
  rc = mlp_trunk_fwd(ctx, *desc, L, params, x, acts, n, st, ldx);
  if (rc) return rc;
  return launch_head_fwd(acts[desc->n_hidden - 1], params + L.head.W, params + L.head.b, out, n, L.head.in,
                         L.head.out, st);
}
