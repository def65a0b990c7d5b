// gae.hip -- `calculate_gae_advantages` (rl_x/algorithms/ppo/flax/ppo.py:122-135,
// full-jit twin rl_x/algorithms/ppo/flax_full_jit/ppo.py:161-175), minus the critic
// forward on next_states (that is rlx_mlp_fwd_f32).
//
// HBM-bound: 6 arrays x 4 B = 24 B per (t, env) element.  One lane per env, lanes
// along N so every load/store of a time row is one coalesced 256-B wave access; the T
// dependent steps are a register recurrence, loads are issued UNROLL rows ahead so the
// serial chain never waits on memory.
#include "mlp.h"

namespace rlx {

constexpr int GAE_UNROLL = 8;

__global__ __launch_bounds__(64) void k_gae(const float* __restrict__ rewards, const float* __restrict__ values,
                                            const float* __restrict__ next_values,
                                            const float* __restrict__ terminations, float* __restrict__ advantages,
                                            float* __restrict__ returns, int T, int N, float gamma, float lam) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float gl = gamma * lam;
  float adv = 0.f;  // A[t+1]; the t = T-1 step uses delta only (mask irrelevant: adv = 0)
  int t = T - 1;
  for (; t >= GAE_UNROLL - 1; t -= GAE_UNROLL) {
    float r[GAE_UNROLL], v[GAE_UNROLL], nv[GAE_UNROLL], tm[GAE_UNROLL];
#pragma unroll
    for (int u = 0; u < GAE_UNROLL; ++u) {
      const int64_t o = (int64_t)(t - u) * N + n;
      r[u] = rewards[o];
      v[u] = values[o];
      nv[u] = next_values[o];
      tm[u] = terminations[o];
    }
#pragma unroll
    for (int u = 0; u < GAE_UNROLL; ++u) {
      const int64_t o = (int64_t)(t - u) * N + n;
      const float nt = 1.0f - tm[u];
      const float delta = r[u] + gamma * nv[u] * nt - v[u];
      adv = delta + gl * nt * adv;
      advantages[o] = adv;
      returns[o] = adv + v[u];
    }
  }
  for (; t >= 0; --t) {
    const int64_t o = (int64_t)t * N + n;
    const float nt = 1.0f - terminations[o];
    const float vv = values[o];
    const float delta = rewards[o] + gamma * next_values[o] * nt - vv;
    adv = delta + gl * nt * adv;
    advantages[o] = adv;
    returns[o] = adv + vv;
  }
}

// ---------------------------------------------------------------------------------------
// next_values = critic(next_states) of calculate_gae_advantages (ppo.py:129) WITHOUT evaluating the critic on all T*N rows:
// the rollout already holds values[t+1] = critic(states[t+1]) for the same parameters, and next_states[t] equals
// states[t+1] bit for bit except where an episode ended (final-observation patch, ppo.py:277-286).  k_nv_prepare compares
// the two rows of every (t, n), copies values[t+1] where they agree and appends the others -- plus the whole last step --
// to a compacted row list whose length stays ON THE DEVICE; the critic then runs over the list's capacity (T*N rows)
// with the forward kernels leaving at once beyond the device-side count (mlp.h: m_dev), and k_nv_scatter puts the
// results back.  No host round trip, work proportional to the number of episode ends.  The slot a row gets in the list
// depends on the order of an integer atomic, its value does not: every row goes through the same fixed-order dot products.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nv_prepare(const float* __restrict__ states, const float* __restrict__ next_states,
                                                    const float* __restrict__ values, float* __restrict__ next_values,
                                                    int32_t* __restrict__ rows, int32_t* __restrict__ count, int T, int N,
                                                    int O) {
  // A workgroup compares ITS 256 rows element by element with COALESCED loads (lane <-> consecutive floats of the two [256, O]
  // blocks); a mismatch raises the row's flag in LDS.  (One thread per row walking its O floats with a short-circuit `&&` was 2 O
  // dependent, uncoalesced loads per thread: 90 us for the 71 MB of config 2.)
  __shared__ int s_diff[256];
  const int64_t row0 = (int64_t)blockIdx.x * 256;
  const int64_t BT = (int64_t)T * N, with_next = (int64_t)(T - 1) * N;      // rows [with_next, BT): the last step, no successor row
  s_diff[threadIdx.x] = 0;
  __syncthreads();
  const int64_t nrows = BT - row0 < 256 ? BT - row0 : 256;
  const int64_t cmp_rows = with_next - row0 < nrows ? (with_next - row0 > 0 ? with_next - row0 : 0) : nrows;   // rows of this block with a successor
  const float* a = next_states + row0 * O;
  const float* b = states + (row0 + N) * O;
  const int64_t ne = cmp_rows * O;
  for (int64_t e0 = threadIdx.x; e0 < ne; e0 += 256 * 4) {
    uint32_t va[4], vb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t e = e0 + 256 * q < ne ? e0 + 256 * q : ne - 1;
      va[q] = __float_as_uint(a[e]);
      vb[q] = __float_as_uint(b[e]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t e = e0 + 256 * q;
      if (e < ne && va[q] != vb[q]) s_diff[(int)(e / O)] = 1;
    }
  }
  __syncthreads();
  const int64_t i = row0 + threadIdx.x;   // flattened (t, n)
  if (i >= BT) return;
  const bool same = i < with_next && s_diff[threadIdx.x] == 0;
  if (same) {
    next_values[i] = values[i + N];
  } else {
    rows[atomicAdd(count, 1)] = (int32_t)i;
  }
}

// x: [count, ldx] rows of pitch ldx >= O (the GEMM first layer of wide observations loads 16-byte vectors: ldx = O rounded up
// to 4; the pad columns are zeroed once by the caller)
__global__ __launch_bounds__(256) void k_nv_gather(const float* __restrict__ next_states, const int32_t* __restrict__ rows,
                                                   const int32_t* __restrict__ count, float* __restrict__ x, int O, int ldx) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = e / O;
  if (r >= *count) return;
  const int d = (int)(e - r * O);
  x[r * ldx + d] = next_states[(int64_t)rows[r] * O + d];
}

// out[m, j] = x[m, cols[j]]: the `x[..., indices]` of the networks that read a subset of the observation
__global__ __launch_bounds__(256) void k_select_columns(const float* __restrict__ x, int ldx, const int32_t* __restrict__ cols,
                                                       int n_cols, float* __restrict__ out, int ldo, int64_t M) {
  const int64_t total = M * n_cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / n_cols;
    const int j = (int)(e - r * n_cols);
    out[r * ldo + j] = x[r * ldx + cols[j]];
  }
}

__global__ __launch_bounds__(256) void k_nv_scatter(const float* __restrict__ v, const int32_t* __restrict__ rows,
                                                    const int32_t* __restrict__ count, float* __restrict__ next_values) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < *count) next_values[rows[r]] = v[r];
}

// ---------------------------------------------------------------------------------------
// The iteration's logged scalars (ppo/flax/ppo.py:215-216, 226-230, 300-307) in two launches, no framework kernels:
//   k_metric_partials   per-block fp64 sums {sum r, sum r^2, sum d, sum d^2} of returns r and d = returns - values
//   k_metric_finalize   out[0..9] = column means of the [n_upd, 10] per-update metric rows; out[10] = explained variance
//                       1 - var(d) / (var(r) + 1e-8) (population variances); out[11] = mean(exp(logstd)) (0 without logstd)
// Fixed summation order (block partials folded in index order by one workgroup): reproducible bit for bit.
// ---------------------------------------------------------------------------------------
constexpr int MET_BLOCKS = 256;

__global__ __launch_bounds__(256) void k_metric_partials(const float* __restrict__ returns, const float* __restrict__ values,
                                                         int64_t n, double* __restrict__ part) {
  __shared__ double s_red[16];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double r = (double)returns[i], d = r - (double)values[i];
    s[0] += r; s[1] += r * r; s[2] += d; s[3] += d * d;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) s_red[q * 4 + w] = s[q];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    part[(int64_t)blockIdx.x * 4 + q] = (s_red[q * 4] + s_red[q * 4 + 1]) + (s_red[q * 4 + 2] + s_red[q * 4 + 3]);
  }
}

__global__ __launch_bounds__(64) void k_metric_finalize(const float* __restrict__ metrics, int n_upd, const double* __restrict__ part,
                                                        int n_part, int64_t n, const float* __restrict__ logstd, int A,
                                                        float* __restrict__ out) {
  const int t = threadIdx.x;
  if (t < 10) {
    // sixteen rows requested at a time, added in row order (the same sum as a one-row-at-a-time loop, which was n_upd dependent
    // round trips for ten lanes of one wave: 56 us at 160 updates, 174 us at 1280)
    double acc = 0.0;
    int u = 0;
    for (; u + 16 <= n_upd; u += 16) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = metrics[(int64_t)(u + q) * 10 + t];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += (double)v[q];
    }
    for (; u < n_upd; ++u) acc += (double)metrics[(int64_t)u * 10 + t];
    out[t] = (float)(acc / (double)(n_upd > 0 ? n_upd : 1));
  } else if (t == 10) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    int b = 0;
    for (; b + 4 <= n_part; b += 4) {           // sixteen doubles in flight, added in block order
      double v[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[i][q] = part[(int64_t)(b + i) * 4 + q];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] += v[i][q];
    }
    for (; b < n_part; ++b)
      for (int q = 0; q < 4; ++q) s[q] += part[(int64_t)b * 4 + q];
    const double inv = 1.0 / (double)(n > 0 ? n : 1);
    const double mr = s[0] * inv, md = s[2] * inv;
    double vr = s[1] * inv - mr * mr, vd = s[3] * inv - md * md;
    if (vr < 0.0) vr = 0.0;
    if (vd < 0.0) vd = 0.0;
    out[10] = (float)(1.0 - vd / (vr + 1e-8));
  } else if (t == 11) {
    float acc = 0.f;
    for (int a = 0; a < A; ++a) acc += expf(logstd[a]);
    out[11] = (logstd && A > 0) ? acc / (float)A : 0.f;
  }
}

}  // namespace rlx

extern "C" int rlx_ppo_next_values_f32(rlx_ctx* ctx, const rlx_mlp_desc* cdesc, const float* cparams, const float* states,
                                       const float* next_states, const float* values, float* next_values, int T, int N,
                                       void* stream) {
  using namespace rlx;
  RLX_REQUIRE(ctx && cdesc && cparams && states && next_states && values && next_values, RLX_EINVAL,
              "rlx_ppo_next_values_f32: NULL pointer");
  RLX_REQUIRE(T >= 0 && N >= 0 && (int64_t)T * N < (1ll << 31), RLX_EINVAL, "rlx_ppo_next_values_f32: bad sizes");
  if (T == 0 || N == 0) return RLX_OK;
  int rc = mlp_check_desc(*cdesc);
  if (rc) return rc;
  RLX_REQUIRE(cdesc->out_dim == 1, RLX_EINVAL, "rlx_ppo_next_values_f32: the critic has one output");
  hipStream_t st = (hipStream_t)stream;
  const int O = cdesc->in_dim;
  const int64_t B = (int64_t)T * N;
  int maxh = 0;
  for (int l = 0; l < cdesc->n_hidden; ++l) maxh = cdesc->hidden[l] > maxh ? cdesc->hidden[l] : maxh;
  const int ldx = (O > 32 && O % 4 != 0) ? ((O + 3) & ~3) : O;   // the GEMM first layer loads 16-B vectors
  // row list + count, gathered rows, values; activations ping-pong in the forward arenas
  int32_t* rows = (int32_t*)scratch(ctx, SL_NV_ROWS, (size_t)(B + 4) * sizeof(int32_t));
  float* xg = (float*)scratch(ctx, SL_STAGE, (size_t)B * ldx * sizeof(float));
  float* vg = (float*)scratch(ctx, SL_VALUE, (size_t)B * sizeof(float));
  float* bufA = (float*)scratch(ctx, SL_FWD_A, (size_t)B * maxh * sizeof(float));
  float* bufB = (float*)scratch(ctx, SL_FWD_B, (size_t)B * maxh * sizeof(float));
  if (!rows || !xg || !vg || !bufA || !bufB) return RLX_ENOMEM;
  int32_t* count = rows + B;
  RLX_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t), st));
  if (ldx != O) RLX_HIP_TRY(hipMemsetAsync(xg, 0, (size_t)B * ldx * sizeof(float), st));   // zero pad columns (never hot: O % 4 != 0 and O > 32)
  hipLaunchKernelGGL(k_nv_prepare, dim3(div_up(B, 256)), dim3(256), 0, st, states, next_states, values, next_values, rows,
                     count, T, N, O);
  RLX_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_nv_gather, dim3(div_up(B * O, 256)), dim3(256), 0, st, next_states, rows, count, xg, O, ldx);
  RLX_LAUNCH_CHECK();
  const MlpLayout L = make_layout(*cdesc);
  float* acts[4] = {bufA, bufB, bufA, bufB};   // ([3]: pre-LayerNorm values of a wide first layer; forward only, so it may alias [1])
  rc = mlp_trunk_fwd(ctx, *cdesc, L, cparams, xg, acts, B, st, ldx != O ? ldx : 0, false, count);
  if (rc) return rc;
  rc = launch_head_fwd(acts[cdesc->n_hidden - 1], cparams + L.head.W, cparams + L.head.b, vg, B, L.head.in, 1, st, count);
  if (rc) return rc;
  hipLaunchKernelGGL(k_nv_scatter, dim3(div_up(B, 256)), dim3(256), 0, st, vg, rows, count, next_values);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

extern "C" int rlx_select_columns_f32(rlx_ctx* ctx, const float* x, int ldx, const int32_t* cols, int n_cols, float* out, int ldo,
                                      int64_t M, void* stream) {
  RLX_REQUIRE(ctx && x && cols && out, RLX_EINVAL, "rlx_select_columns_f32: NULL pointer");
  RLX_REQUIRE(ldx > 0 && n_cols > 0 && ldo >= n_cols && M >= 0, RLX_EINVAL, "rlx_select_columns_f32: bad sizes");
  if (M == 0) return RLX_OK;
  int grid = rlx::div_up(M * n_cols, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(rlx::k_select_columns, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, cols, n_cols, out, ldo, M);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

extern "C" int rlx_gae_f32(rlx_ctx* ctx, const float* rewards, const float* values, const float* next_values,
                           const float* terminations, float* advantages, float* returns, int T, int N, float gamma,
                           float gae_lambda, void* stream) {
  RLX_REQUIRE(ctx && rewards && values && next_values && terminations && advantages && returns, RLX_EINVAL,
              "rlx_gae_f32: NULL pointer");
  RLX_REQUIRE(T >= 0 && N >= 0, RLX_EINVAL, "rlx_gae_f32: negative size");
  if (T == 0 || N == 0) return RLX_OK;
  // 64-thread blocks: N=4096 -> 64 workgroups spread over the XCDs (one wave each)
  hipLaunchKernelGGL(rlx::k_gae, dim3(rlx::div_up(N, 64)), dim3(64), 0, (hipStream_t)stream, rewards, values,
                     next_values, terminations, advantages, returns, T, N, gamma, gae_lambda);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

extern "C" int rlx_ppo_reduce_metrics_f32(rlx_ctx* ctx, const float* metrics, int n_updates, const float* returns, const float* values,
                                          int64_t n, const float* logstd, int act_dim, float* out12, void* stream) {
  RLX_REQUIRE(ctx && metrics && returns && values && out12 && n_updates > 0 && n > 0 && act_dim >= 0, RLX_EINVAL,
              "rlx_ppo_reduce_metrics_f32: bad args");
  double* part = (double*)rlx::scratch(ctx, rlx::SL_STAT_PART, (size_t)rlx::MET_BLOCKS * 4 * sizeof(double));
  if (!part) return RLX_ENOMEM;
  hipStream_t st = (hipStream_t)stream;
  int grid = rlx::div_up(n, 256 * 8);
  if (grid > rlx::MET_BLOCKS) grid = rlx::MET_BLOCKS;
  hipLaunchKernelGGL(rlx::k_metric_partials, dim3(grid), dim3(256), 0, st, returns, values, n, part);
  RLX_LAUNCH_CHECK();
  hipLaunchKernelGGL(rlx::k_metric_finalize, dim3(1), dim3(64), 0, st, metrics, n_updates, part, grid, n, logstd, logstd ? act_dim : 0,
                     out12);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}
