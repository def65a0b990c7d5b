// l1fused.hip -- the whole first-layer backward folded into the second layer's input-gradient
// GEMM: per 32-row tile
//     dH1 = dZ2 @ W2^T                         (exact-fp32 MFMA, K = hidden[1])
//     z1  = X @ W1 + b1  (recomputed, K = obs)  -> LayerNorm statistics, xhat, h1 = act(...)
//     dZ1 = LN'( dH1 * act'(h1) )              (all in MFMA accumulator registers)
//     dW1 += X^T dZ1, db1 += colsum(dZ1), dgamma += colsum(dy*xhat), dbeta += colsum(dy)
// Nothing of shape [M, hidden[0]] is written to or re-read from HBM: this replaces k_gemm_dx
// (layer 1), k_l1<bwd> and k_gemm_dw_skinny and removes ~256 MB of traffic per network and
// update at mb = 32768 (the reverse-mode gradient of ppo/flax_full_jit/policy.py:31-34 /
// critic.py:22-25 inside jax.value_and_grad, rl_x/algorithms/ppo/flax/ppo.py:189).
//
// Workgroup = 32 rows x ALL hidden[0] columns (4 waves x NT 32-column MFMA tiles, hidden[0] =
// 128*NT), so LayerNorm row reductions stay on chip: DPP + v_permlane16_swap inside each
// 32-lane half (one accumulator register = one row per half), then a 4-wave LDS combine.
// W1 stays resident in LDS; W2^T (pre-transposed copy, [hidden[1]][hidden[0]]) streams through
// a register-prefetched LDS stage.  Workgroups are persistent over row tiles and keep the
// dW1/db1/dgamma/dbeta partial sums in registers; one slab per workgroup goes to the
// deterministic slab reduction.
#include "mlp.h"

namespace rlx {

constexpr int LF_ROWS = 32;
constexpr int LF_THREADS = 256;
constexpr int LF_XS = 33;  // Xs[row][k] stride

typedef float lf_v4 __attribute__((ext_vector_type(4)));

// sum over the 32 lanes of each half-wave (lanes 0-31 / 32-63 hold different rows): 4 DPP steps
// inside rows of 16 lanes, then v_permlane16_swap to fold the two 16-lane rows of each half.
__device__ __forceinline__ float half_sum(float v) {
  v += dpp_f(v, 0);
  v += dpp_f(v, 1);
  v += dpp_f(v, 2);
  v += dpp_f(v, 3);
  const unsigned u = (unsigned)__float_as_int(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
}

struct L1FusedArgs {
  const float* X;      // [M, O]
  const float* dZ2;    // [M, N2]
  const float* W2t;    // [N2, H1]  (transposed copy of W2[H1, N2])
  const float* W1;     // [O, H1]
  const float* b1;
  const float* g;      // LN scale or null
  const float* be;     // LN bias or null
  float* partials;     // [grid][(O + 3) * H1] : dW1 [O][H1], db1, dgamma, dbeta
  int64_t M;
  int O, H1, N2, act, ln;
};

template <int NT>
__global__ __launch_bounds__(LF_THREADS, 1) void k_dx_l1bwd(L1FusedArgs a) {
  constexpr int H1 = 128 * NT;
  constexpr int SB = H1 + 4;
  constexpr int PERB = H1 / 32;  // 16-B loads per thread for one [32][H1] weight tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int OP = (a.O + 1) & ~1;              // obs dim padded to the MFMA k-step
  float* Bs = smem;                           // [32][SB]   W2^T k-tile
  float* W1s = Bs + G_BK * SB;                // [OP][H1]
  float* As = W1s + OP * H1;                  // [32][33]   dZ2 k-tile
  float* Xs = As + LF_ROWS * LF_XS;           // [32][33]   X tile (cols >= O zero)
  float* red = Xs + LF_ROWS * LF_XS;          // [2][2][4][32]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  const int O = a.O, N2 = a.N2, act = a.act;
  const bool ln = a.ln != 0;

  for (int i = t; i < OP * H1; i += LF_THREADS) W1s[i] = (i < O * H1) ? a.W1[i] : 0.f;
  float bias[NT], gam[NT], bet[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = w * 32 * NT + 32 * j + li;
    bias[j] = a.b1[col];
    gam[j] = ln ? a.g[col] : 1.f;
    bet[j] = ln ? a.be[col] : 0.f;
  }
  f32x16 dW[NT];
  float dgam[NT], dbet[NT], db1[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    dgam[j] = dbet[j] = db1[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dW[j][r] = 0.f;
  }
  const float invH = 1.0f / (float)H1;
  const int nk = N2 / G_BK;
  const int f_row = t / (H1 / 4), f_col = (t % (H1 / 4)) * 4;
  constexpr int RPP = LF_THREADS / (H1 / 4);  // weight rows per pass
  const int a_r = t >> 3, a_c = (t & 7) * 4;  // dZ2 tile: 8 threads per 32-float row

  const int64_t ntiles = (a.M + LF_ROWS - 1) / LF_ROWS;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * LF_ROWS;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    lf_v4 rb[PERB], ra;
#define LF_LD(KT)                                                                                              \
  {                                                                                                            \
    _Pragma("unroll") for (int p = 0; p < PERB; ++p)                                                           \
        rb[p] = *reinterpret_cast<const lf_v4*>(a.W2t + (int64_t)((KT) * G_BK + f_row + RPP * p) * H1 + f_col); \
    const int64_t row = r0 + a_r;                                                                              \
    ra = row < a.M ? *reinterpret_cast<const lf_v4*>(a.dZ2 + row * N2 + (KT) * G_BK + a_c) : lf_v4{0.f, 0.f, 0.f, 0.f}; \
  }
    LF_LD(0)
    __syncthreads();  // previous tile's readers of Xs / As / Bs are done
    for (int i = t; i < LF_ROWS * 32; i += LF_THREADS) {
      const int r = i >> 5, k = i & 31;
      Xs[r * LF_XS + k] = (k < O && r0 + r < a.M) ? a.X[(r0 + r) * O + k] : 0.f;
    }
    // ---- main GEMM: dH1 tile
    for (int kt = 0; kt < nk; ++kt) {
      if (kt > 0) __syncthreads();
#pragma unroll
      for (int p = 0; p < PERB; ++p) *reinterpret_cast<lf_v4*>(Bs + (f_row + RPP * p) * SB + f_col) = rb[p];
      {
        float* d = As + a_r * LF_XS + a_c;
        d[0] = ra[0]; d[1] = ra[1]; d[2] = ra[2]; d[3] = ra[3];
      }
      __syncthreads();
      if (kt + 1 < nk) LF_LD(kt + 1)
      const float* a0 = As + li * LF_XS + lh;
      const float* b0 = Bs + lh * SB + w * 32 * NT + li;
#pragma unroll
      for (int kk = 0; kk < G_BK; kk += 2) {
        const float av = a0[kk];
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[kk * SB + 32 * j], acc[j], 0, 0, 0);
      }
    }
#undef LF_LD
    // ---- recompute z1 = X @ W1 + b1 in the same accumulator layout
    f32x16 z[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) z[j][r] = bias[j];
    {
      const float* x0 = Xs + li * LF_XS + lh;
      const float* w0 = W1s + lh * H1 + w * 32 * NT + li;
      for (int kk = 0; kk < OP; kk += 2) {
        const float av = x0[kk];
#pragma unroll
        for (int j = 0; j < NT; ++j)
          z[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w0[kk * H1 + 32 * j], z[j], 0, 0, 0);
      }
    }
    // accumulator register r of half lh is row rho = (r&3) + 8*(r>>2) + 4*lh of the tile
    float mean[16], rstd[16];
    if (ln) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) { s += z[j][r]; ss += z[j][r] * z[j][r]; }
        s = half_sum(s);
        ss = half_sum(ss);
        if (li == 0) {
          const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
          red[(0 * 4 + w) * 32 + rho] = s;
          red[(1 * 4 + w) * 32 + rho] = ss;
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float s = (red[0 * 32 + rho] + red[1 * 32 + rho]) + (red[2 * 32 + rho] + red[3 * 32 + rho]);
        const float ss = (red[4 * 32 + rho] + red[5 * 32 + rho]) + (red[6 * 32 + rho] + red[7 * 32 + rho]);
        mean[r] = s * invH;
        rstd[r] = rsqrtf(fmaxf(0.f, ss * invH - mean[r] * mean[r]) + 1e-6f);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { mean[r] = 0.f; rstd[r] = 1.f; }
    }
    // dy = dH1 * act'(h);  z <- xhat;  acc <- d xhat;  row sums m1, m2
    float m1[16], m2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float xh = (z[j][r] - mean[r]) * rstd[r];
        const float y = ln ? xh * gam[j] + bet[j] : z[j][r];
        const float h = act_fwd(y, act);
        const float dy = acc[j][r] * act_grad_from_out(h, act);
        dgam[j] += dy * xh;
        dbet[j] += dy;
        const float dxh = dy * gam[j];
        z[j][r] = xh;
        acc[j][r] = dxh;
        a1 += dxh;
        a2 += dxh * xh;
      }
      m1[r] = m2[r] = 0.f;
      if (ln) {
        a1 = half_sum(a1);
        a2 = half_sum(a2);
        if (li == 0) {
          const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
          red[256 + (0 * 4 + w) * 32 + rho] = a1;
          red[256 + (1 * 4 + w) * 32 + rho] = a2;
        }
      }
    }
    if (ln) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float* q = red + 256;
        m1[r] = ((q[0 * 32 + rho] + q[1 * 32 + rho]) + (q[2 * 32 + rho] + q[3 * 32 + rho])) * invH;
        m2[r] = ((q[4 * 32 + rho] + q[5 * 32 + rho]) + (q[6 * 32 + rho] + q[7 * 32 + rho])) * invH;
      }
    }
    // dZ1 (in acc), bias gradient, and dW1 += X^T dZ1 with the accumulator registers as the B operand:
    // MFMA step r contracts row rho(r,0) (lanes 0-31) and row rho(r,1) (lanes 32-63).
    const float* xt = Xs + li;  // A operand: A[i = obs index li][k = lh] = X[rho(r, lh)][li]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float av = xt[rho * LF_XS];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float dz = ln ? rstd[r] * (acc[j][r] - m1[r] - z[j][r] * m2[r]) : acc[j][r];
        db1[j] += dz;
        dW[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, dz, dW[j], 0, 0, 0);
      }
    }
  }
  // ---- one slab per workgroup
  float* out = a.partials + (int64_t)blockIdx.x * (O + 3) * H1;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = w * 32 * NT + 32 * j + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;  // obs index
      if (row < O) out[(int64_t)row * H1 + col] = dW[j][r];
    }
    // the two halves hold different rows of the same column: fold them
    float v0 = db1[j], v1 = dgam[j], v2 = dbet[j];
    {
      const unsigned u0 = (unsigned)__float_as_int(v0), u1 = (unsigned)__float_as_int(v1), u2 = (unsigned)__float_as_int(v2);
      const auto s0 = __builtin_amdgcn_permlane32_swap(u0, u0, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
      const auto s2 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
      v0 = __int_as_float((int)s0[0]) + __int_as_float((int)s0[1]);
      v1 = __int_as_float((int)s1[0]) + __int_as_float((int)s1[1]);
      v2 = __int_as_float((int)s2[0]) + __int_as_float((int)s2[1]);
    }
    if (lh == 0) {
      out[(int64_t)O * H1 + col] = v0;
      out[(int64_t)(O + 1) * H1 + col] = v1;
      out[(int64_t)(O + 2) * H1 + col] = v2;
    }
  }
}

// W[R, C] -> Wt[C, R]
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ W, float* __restrict__ Wt, int R, int C) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8)
    if (by + i < R && bx + tx < C) tile[i][tx] = W[(int64_t)(by + i) * C + bx + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (bx + i < C && by + tx < R) Wt[(int64_t)(bx + i) * R + by + tx] = tile[tx][i];
}

bool l1fused_supported(const rlx_mlp_desc& d) {
  return d.n_hidden >= 2 && d.in_dim <= 32 && (d.hidden[0] == 256 || d.hidden[0] == 512) && d.hidden[1] % G_BK == 0;
}

size_t l1fused_partial_floats(const rlx_mlp_desc& d, int grid) {
  return (size_t)grid * (d.in_dim + 3) * d.hidden[0] + (size_t)d.hidden[0] * d.hidden[1];  // slabs + W2^T
}

int l1fused_grid(int64_t M, int num_cus) {
  const int64_t nt = (M + LF_ROWS - 1) / LF_ROWS;
  return (int)(nt < num_cus ? nt : num_cus);
}

// arena: [grid][(O+3)*H1] slabs followed by the W2^T copy.  Emits the reduce segments.
int launch_l1fused(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x,
                   const float* dZ2, float* arena, int grid, float* grads, int64_t M, ReduceTable* tab, hipStream_t st) {
  const LayerOff& o0 = L.layer[0];
  const LayerOff& o1 = L.layer[1];
  const int H1 = o0.out, N2 = o1.out, O = o0.in;
  float* slabs = arena;
  float* W2t = arena + (size_t)grid * (O + 3) * H1;
  hipLaunchKernelGGL(k_transpose, dim3(div_up(N2, 32), div_up(H1, 32)), dim3(256), 0, st, params + o1.W, W2t, H1, N2);
  RLX_LAUNCH_CHECK();
  L1FusedArgs a;
  a.X = x; a.dZ2 = dZ2; a.W2t = W2t; a.W1 = params + o0.W; a.b1 = params + o0.b;
  a.g = o0.g >= 0 ? params + o0.g : nullptr;
  a.be = o0.be >= 0 ? params + o0.be : nullptr;
  a.partials = slabs; a.M = M; a.O = O; a.H1 = H1; a.N2 = N2; a.act = d.act; a.ln = d.ln_first ? 1 : 0;
  const int OP = (O + 1) & ~1;
  const size_t lds = ((size_t)G_BK * (H1 + 4) + (size_t)OP * H1 + 2 * LF_ROWS * LF_XS + 512) * sizeof(float);
  {
    // main GEMM + z recompute + dW1 on the matrix pipe
    ProfScope prof(ctx, PK_GEMM_DX, 2.0 * (double)M * H1 * (N2 + O), st);  // algorithmic: dX + dW1
    if (H1 == 512) {
      static bool set4 = false;
      if (!set4) {
        RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dx_l1bwd<4>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        set4 = true;
      }
      hipLaunchKernelGGL(k_dx_l1bwd<4>, dim3(grid), dim3(LF_THREADS), lds, st, a);
    } else {
      static bool set2 = false;
      if (!set2) {
        RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dx_l1bwd<2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        set2 = true;
      }
      hipLaunchKernelGGL(k_dx_l1bwd<2>, dim3(grid), dim3(LF_THREADS), lds, st, a);
    }
  }
  RLX_LAUNCH_CHECK();
  const int64_t PS = (int64_t)(O + 3) * H1;
  tab->seg[tab->n++] = ReduceSeg{slabs, grads + o0.W, (int64_t)O * H1, PS, grid, 0, 1.f, 0.f, 1};
  tab->seg[tab->n++] = ReduceSeg{slabs + (int64_t)O * H1, grads + o0.b, (int64_t)H1, PS, grid, 0, 1.f, 0.f, 1};
  if (d.ln_first) {
    tab->seg[tab->n++] = ReduceSeg{slabs + (int64_t)(O + 1) * H1, grads + o0.g, (int64_t)H1, PS, grid, 0, 1.f, 0.f, 1};
    tab->seg[tab->n++] = ReduceSeg{slabs + (int64_t)(O + 2) * H1, grads + o0.be, (int64_t)H1, PS, grid, 0, 1.f, 0.f, 1};
  }
  return RLX_OK;
}

}  // namespace rlx
