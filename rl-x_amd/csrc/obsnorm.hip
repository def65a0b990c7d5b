// obsnorm.hip -- running observation statistics and normalisation of FastSAC
// (rl_x/algorithms/fastsac/pytorch/observation_normalizer.py:11-53).
//   update(obs [B, O]):  batch mean / population variance per column, merged into the running mean / variance by the
//                        parallel-variance formula AS THE REFERENCE WRITES IT: the squared-difference term uses
//                        delta2 = batch_mean - running_mean taken AFTER running_mean was updated (observation_normalizer.py:44-49);
//   normalize(obs):      (obs - running_mean) / (running_std_dev + eps).
// The column sums are accumulated in fp64 in a fixed order (bit-reproducible): chunks of 512 rows x 32 columns per workgroup (8 row
// lanes, every row access one coalesced 128-byte piece, the 8 partials folded in order), then the chunks added in chunk order by one
// thread per column (one workgroup per 32 columns over ALL rows took 2.2 ms at 65536 x 48).  State: running_mean [O],
// running_var [O], running_std_dev [O] fp32 and count int64[1], all on the device -- nothing returns to the host.
#include "common.h"
#include "dist.h"

namespace rlx {

// stage 1: fp64 column sums of one chunk of OBS_CHUNK rows (blockIdx.y), 32 columns per workgroup (blockIdx.x): 8 row lanes in
// fixed row order, folded in order -> part[chunk][2][O]
constexpr int OBS_CHUNK = 512;
__global__ __launch_bounds__(256) void k_obs_norm_partial(const float* __restrict__ obs, int64_t B, int O, double* __restrict__ part) {
  __shared__ double s1[8][32], s2[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
  const int64_t r0 = (int64_t)blockIdx.y * OBS_CHUNK;
  const int64_t r1 = r0 + OBS_CHUNK < B ? r0 + OBS_CHUNK : B;
  double a1 = 0.0, a2 = 0.0;
  if (c < O)
    for (int64_t r = r0 + rl; r < r1; r += 32) {   // four loads in flight, added in row order
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = r + 8 * u < r1 ? obs[(r + 8 * u) * O + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a1 += (double)v[u];
        a2 += (double)v[u] * (double)v[u];
      }
    }
  s1[rl][threadIdx.x & 31] = a1;
  s2[rl][threadIdx.x & 31] = a2;
  __syncthreads();
  if (rl == 0 && c < O) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { t1 += s1[q][threadIdx.x]; t2 += s2[q][threadIdx.x]; }
    part[((int64_t)blockIdx.y * 2 + 0) * O + c] = t1;
    part[((int64_t)blockIdx.y * 2 + 1) * O + c] = t2;
  }
}

// stage 2: the chunks added in chunk order (one thread per column) -> tot[2][O] fp64 + tot[2 * O] = row count; under data
// parallelism the library all-reduces tot over the ranks (every rank then merges the same GLOBAL batch: replicated statistics)
__global__ __launch_bounds__(256) void k_obs_norm_fold(const double* __restrict__ part, int nchunks, int64_t B, int O,
                                                      double* __restrict__ tot) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0) tot[2 * O] = (double)B;
  if (c >= O) return;
  double t1 = 0.0, t2 = 0.0;
  for (int q = 0; q < nchunks; ++q) {
    t1 += part[((int64_t)q * 2 + 0) * O + c];
    t2 += part[((int64_t)q * 2 + 1) * O + c];
  }
  tot[c] = t1;
  tot[O + c] = t2;
}

// stage 3: the reference's merge of the batch statistics into the running ones; the count advances behind it (k_obs_norm_count)
__global__ __launch_bounds__(256) void k_obs_norm_update(const double* __restrict__ tot, int O, float* __restrict__ mean,
                                                        float* __restrict__ var, float* __restrict__ stdv,
                                                        const int64_t* __restrict__ count) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= O) return;
  const double nb = tot[2 * O], n0 = (double)*count, n1 = n0 + nb;
  const double bm = tot[c] / nb;
  double bv = tot[O + c] / nb - bm * bm;
  if (bv < 0.0) bv = 0.0;
  const double m0 = (double)mean[c], v0 = (double)var[c];
  const double m1 = m0 + (bm - m0) * nb / n1;                 // running_mean + delta * batch_count / new_count
  const double d2 = bm - m1;                                  // the reference's delta2: against the UPDATED mean
  const double M2 = v0 * n0 + bv * nb + d2 * d2 * n0 * nb / n1;
  const double v1 = M2 / n1;
  mean[c] = (float)m1;
  var[c] = (float)v1;
  stdv[c] = sqrtf((float)v1);
}

__global__ void k_obs_norm_count(int64_t* __restrict__ count, const double* __restrict__ tot, int O) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *count += (int64_t)tot[2 * O];
}

__global__ __launch_bounds__(256) void k_obs_norm_apply(const float* __restrict__ obs, const float* __restrict__ mean,
                                                       const float* __restrict__ stdv, float eps, float* __restrict__ out,
                                                       int64_t n, int O) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % O);
    out[e] = (obs[e] - mean[c]) / (stdv[c] + eps);
  }
}

}  // namespace rlx

extern "C" {

int rlx_obs_norm_update_f32(rlx_ctx* ctx, const float* obs, int64_t B, int O, float* running_mean, float* running_var,
                            float* running_std_dev, int64_t* count, void* stream) {
  RLX_REQUIRE(ctx && obs && running_mean && running_var && running_std_dev && count, RLX_EINVAL, "rlx_obs_norm_update_f32: NULL pointer");
  RLX_REQUIRE(B > 0 && O > 0, RLX_EINVAL, "rlx_obs_norm_update_f32: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int nchunks = rlx::div_up(B, rlx::OBS_CHUNK);
  double* part = (double*)rlx::scratch(ctx, rlx::SL_STAT_PART, ((size_t)nchunks * 2 * O + 2 * O + 8) * sizeof(double));
  if (!part) return RLX_ENOMEM;
  double* tot = part + (size_t)nchunks * 2 * O;
  hipLaunchKernelGGL(rlx::k_obs_norm_partial, dim3(rlx::div_up(O, 32), nchunks), dim3(256), 0, st, obs, B, O, part);
  RLX_LAUNCH_CHECK();
  hipLaunchKernelGGL(rlx::k_obs_norm_fold, dim3(rlx::div_up(O, 256)), dim3(256), 0, st, (const double*)part, nchunks, B, O, tot);
  RLX_LAUNCH_CHECK();
  if (rlx::dist_active(ctx)) {   // data parallel: sums and row count of ALL ranks' batches -- the statistics stay replicated
    const int rc = rlx::dist_allreduce(ctx, tot, 2 * (int64_t)O + 1, 1, st);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(rlx::k_obs_norm_update, dim3(rlx::div_up(O, 256)), dim3(256), 0, st, (const double*)tot, O, running_mean, running_var,
                     running_std_dev, count);
  RLX_LAUNCH_CHECK();
  hipLaunchKernelGGL(rlx::k_obs_norm_count, dim3(1), dim3(64), 0, st, count, (const double*)tot, O);   // after every column has read the old count
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_obs_norm_apply_f32(rlx_ctx* ctx, const float* obs, int64_t B, int O, const float* running_mean,
                           const float* running_std_dev, float epsilon, float* out, void* stream) {
  RLX_REQUIRE(ctx && obs && running_mean && running_std_dev && out, RLX_EINVAL, "rlx_obs_norm_apply_f32: NULL pointer");
  RLX_REQUIRE(B >= 0 && O > 0, RLX_EINVAL, "rlx_obs_norm_apply_f32: bad sizes");
  if (B == 0) return RLX_OK;
  int grid = rlx::div_up(B * O, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(rlx::k_obs_norm_apply, dim3(grid), dim3(256), 0, (hipStream_t)stream, obs, running_mean, running_std_dev,
                     epsilon, out, B * (int64_t)O, O);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

}  // extern "C"
