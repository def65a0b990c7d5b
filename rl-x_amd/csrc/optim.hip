// optim.hip -- optax.chain(clip_by_global_norm(c), inject_hyperparams(adam)(lr))
// as used at rl_x/algorithms/ppo/flax/ppo.py:84-100 and applied at :212-213
// (optax>=0.2.6 is third-party, restated; CPU twin: oracle/ppo.py adam_step).
//
// Two launches per network: (1) per-block partial sums of g^2 (deterministic order),
// (2) every block re-reduces the <=1024 partials to the global norm, then clip + Adam (adam_body.h).
// HBM/L2 traffic: 28 B per parameter (read p,g,m,v; write p,m,v) -- 1.4 MB fits L2.
#include "mlp.h"
#include "gemm_bx.h"
#include "adam_body.h"

namespace rlx {

constexpr int OPT_BLOCK = 256;
constexpr int OPT_VEC = 4;
constexpr int OPT_MAX_PARTIALS = REDUCE_MAX_BLOCKS;

__device__ __forceinline__ float block_sum(float v, float* s_buf /*[OPT_BLOCK/64]*/) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) s_buf[w] = v;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < OPT_BLOCK / 64; ++i) tot += s_buf[i];
  __syncthreads();
  return tot;
}

__global__ __launch_bounds__(OPT_BLOCK) void k_sumsq_partials(const float* __restrict__ g, int64_t n,
                                                              float* __restrict__ partials) {
  __shared__ float s_buf[OPT_BLOCK / 64];
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * OPT_BLOCK * OPT_VEC;
  for (int64_t base = ((int64_t)blockIdx.x * OPT_BLOCK + threadIdx.x) * OPT_VEC; base < n; base += stride) {
    if (base + OPT_VEC <= n && ((reinterpret_cast<uintptr_t>(g + base) & 15) == 0)) {
      const float4 x = *reinterpret_cast<const float4*>(g + base);
      acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    } else {
      for (int j = 0; j < OPT_VEC && base + j < n; ++j) acc += g[base + j] * g[base + j];
    }
  }
  const float tot = block_sum(acc, s_buf);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void k_norm_from_partials(const float* __restrict__ partials, int n_partials, float* __restrict__ out) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_partials; i += 64) acc += partials[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[0] = sqrtf(acc);
}

// One parameter set per job (adam_body.h: matrices with registered weight images are stepped tile by tile and their image
// fragments rewritten whole; everything else one element per thread).
__global__ __launch_bounds__(OPT_BLOCK) void k_clip_adam(AdamJob J, float b1, float b2, float eps) {
  __shared__ float s_buf[OPT_BLOCK / 64];
  __shared__ __attribute__((aligned(16))) uint16_t s_tile[ADAM_LDS_HALVES];
  clip_adam_job(J, (int)blockIdx.x, b1, b2, eps, s_buf, s_tile);
}

// Two independent optimizers in ONE launch (ppo.py:212-213 steps the policy's and the critic's TrainState one after the other on
// independent gradients): blocks [0, nb0) run job 0, the rest job 1.  Same arithmetic as two k_clip_adam launches.
__global__ __launch_bounds__(OPT_BLOCK) void k_clip_adam2(AdamJob j0, AdamJob j1, int nb0, float b1, float b2, float eps) {
  __shared__ float s_buf[OPT_BLOCK / 64];
  __shared__ __attribute__((aligned(16))) uint16_t s_tile[ADAM_LDS_HALVES];
  if ((int)blockIdx.x < nb0) clip_adam_job(j0, (int)blockIdx.x, b1, b2, eps, s_buf, s_tile);
  else clip_adam_job(j1, (int)blockIdx.x - nb0, b1, b2, eps, s_buf, s_tile);
}

// sum-of-squares partials of TWO gradient vectors in one launch (the data-parallel twin update: after the all-reduce)
__global__ __launch_bounds__(OPT_BLOCK) void k_sumsq_partials2(const float* __restrict__ g0, int64_t n0, float* __restrict__ part0,
                                                               int nb0, const float* __restrict__ g1, int64_t n1,
                                                               float* __restrict__ part1, int nb1) {
  __shared__ float s_buf[OPT_BLOCK / 64];
  const bool second = (int)blockIdx.x >= nb0;
  const float* g = second ? g1 : g0;
  const int64_t n = second ? n1 : n0;
  const int bid = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x, nblk = second ? nb1 : nb0;
  float acc = 0.f;
  const int64_t stride = (int64_t)nblk * OPT_BLOCK * OPT_VEC;
  for (int64_t base = ((int64_t)bid * OPT_BLOCK + threadIdx.x) * OPT_VEC; base < n; base += stride) {
    if (base + OPT_VEC <= n && ((reinterpret_cast<uintptr_t>(g + base) & 15) == 0)) {
      const float4 x = *reinterpret_cast<const float4*>(g + base);
      acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    } else {
      for (int j = 0; j < OPT_VEC && base + j < n; ++j) acc += g[base + j] * g[base + j];
    }
  }
  const float tot = block_sum(acc, s_buf);
  if (threadIdx.x == 0) (second ? part1 : part0)[bid] = tot;
}

inline int partial_grid(int64_t n) {
  int g = div_up(n, (int64_t)OPT_BLOCK * OPT_VEC);
  return g > 1024 ? 1024 : (g < 1 ? 1 : g);
}

// per-block partial sums of g^2 for launch_clip_adam; returns the number of partials written (<= REDUCE_MAX_BLOCKS)
int launch_sumsq_partials(const float* g, int64_t n, float* partials, hipStream_t st) {
  const int grid = partial_grid(n);
  hipLaunchKernelGGL(k_sumsq_partials, dim3(grid), dim3(OPT_BLOCK), 0, st, g, n, partials);
  return grid;
}

void adam_schedule_entry(float* out3, int64_t step, float lr, float b1, float b2) {
  out3[0] = lr;
  out3[1] = (float)(1.0 - pow((double)b1, (double)step));
  out3[2] = (float)(1.0 - pow((double)b2, (double)step));
}

// sched_dev (optional, DEVICE {lr, 1 - b1^step, 1 - b2^step}): the per-update values come from a device table instead of the
// launch arguments (the SAC update keeps them next to its key: one small upload per call)
static AdamJob make_adam_job(float* params, const float* grads, float* m, float* v, int64_t n, const float* partials, int n_partials,
                             int64_t step, float lr, float max_norm, float b1, float b2, float* norm_out, const float* sched_dev,
                             const BxEmit* emit, float* polyak_target, float tau, float weight_decay, int clip_mode) {
  AdamJob J{params, grads, m, v, n, partials, n_partials, max_norm, norm_out, sched_dev, polyak_target, tau, weight_decay};
  J.clip_mode = clip_mode;
  J.lr = lr;
  J.bc1 = (float)(1.0 - pow((double)b1, (double)step));
  J.bc2 = (float)(1.0 - pow((double)b2, (double)step));
  BxEmit em;
  em.n = 0;
  if (emit) em = *emit;
  BxEmit none;
  none.n = 0;
  adam_job_plan(J, em, none);
  return J;
}

int launch_clip_adam(float* params, const float* grads, float* m, float* v, int64_t n, const float* sumsq_partials,
                     int n_partials, int64_t step, float lr, float max_norm, float b1, float b2, float eps,
                     float* norm_out, hipStream_t st, const float* sched_dev, const BxEmit* emit, float* polyak_target,
                     float tau, float weight_decay, int clip_mode) {
  const AdamJob J = make_adam_job(params, grads, m, v, n, sumsq_partials, n_partials, step, lr, max_norm, b1, b2, norm_out, sched_dev,
                                  emit, polyak_target, tau, weight_decay, clip_mode);
  hipLaunchKernelGGL(k_clip_adam, dim3(J.n_tile_blocks + J.n_rest_blocks), dim3(OPT_BLOCK), 0, st, J, b1, b2, eps);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// both networks' clip + Adam in one launch (k_clip_adam2); g-norm partials per network as launch_reduce_segments left them
int launch_clip_adam2(float* p0, const float* g0, float* m0, float* v0, int64_t n0, const float* part0, int np0, float* norm0,
                      const BxEmit* e0, float* p1, const float* g1, float* m1, float* v1, int64_t n1, const float* part1, int np1,
                      float* norm1, const BxEmit* e1, int64_t step, float lr, float max_norm, float b1, float b2, float eps,
                      hipStream_t st, const float* sched_dev) {
  const AdamJob j0 = make_adam_job(p0, g0, m0, v0, n0, part0, np0, step, lr, max_norm, b1, b2, norm0, sched_dev, e0, nullptr, 0.f, 0.f, 0);
  const AdamJob j1 = make_adam_job(p1, g1, m1, v1, n1, part1, np1, step, lr, max_norm, b1, b2, norm1, sched_dev, e1, nullptr, 0.f, 0.f, 0);
  const int nb0 = j0.n_tile_blocks + j0.n_rest_blocks, nb1 = j1.n_tile_blocks + j1.n_rest_blocks;
  hipLaunchKernelGGL(k_clip_adam2, dim3(nb0 + nb1), dim3(OPT_BLOCK), 0, st, j0, j1, nb0, b1, b2, eps);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// per-block sums of g^2 of two gradient vectors in one launch; *np0 / *np1 = partials written
int launch_sumsq_partials2(const float* g0, int64_t n0, float* part0, int* np0, const float* g1, int64_t n1, float* part1, int* np1,
                           hipStream_t st) {
  const int nb0 = partial_grid(n0), nb1 = partial_grid(n1);
  hipLaunchKernelGGL(k_sumsq_partials2, dim3(nb0 + nb1), dim3(OPT_BLOCK), 0, st, g0, n0, part0, nb0, g1, n1, part1, nb1);
  RLX_LAUNCH_CHECK();
  *np0 = nb0;
  *np1 = nb1;
  return RLX_OK;
}

int clip_adam_step(rlx_ctx* ctx, float* params, const float* grads, float* m, float* v, int64_t n_params, int64_t step, float lr,
                   float max_grad_norm, float b1, float b2, float eps, float* grad_norm_out, hipStream_t st, const BxEmit* emit) {
  if (ctx) ctx->ro_img.valid = false;
  RLX_REQUIRE(ctx && params && grads && m && v, RLX_EINVAL, "rlx_clip_adam_step_f32: NULL pointer");
  RLX_REQUIRE(n_params > 0 && step >= 1, RLX_EINVAL, "rlx_clip_adam_step_f32: n_params>0 and step>=1 (1-based) required");
  // two alternating buffers: consecutive calls (policy / critic) may be in flight on different streams
  float* partials = (float*)scratch(ctx, (ctx->opt_flip ^= 1) ? SL_OPT_A : SL_OPT_B, OPT_MAX_PARTIALS * sizeof(float));
  if (!partials) return RLX_ENOMEM;
  const int grid = partial_grid(n_params);
  hipLaunchKernelGGL(k_sumsq_partials, dim3(grid), dim3(OPT_BLOCK), 0, st, grads, n_params, partials);
  RLX_LAUNCH_CHECK();
  return launch_clip_adam(params, grads, m, v, n_params, partials, grid, step, lr, max_grad_norm, b1, b2, eps, grad_norm_out, st,
                          nullptr, emit);
}

}  // namespace rlx

using namespace rlx;

extern "C" {

int rlx_grad_global_norm_f32(rlx_ctx* ctx, const float* grads, int64_t n, float* norm_out, void* stream) {
  RLX_REQUIRE(ctx && grads && norm_out && n > 0, RLX_EINVAL, "rlx_grad_global_norm_f32: bad args");
  float* partials = (float*)scratch(ctx, SL_NORM, OPT_MAX_PARTIALS * sizeof(float));
  if (!partials) return RLX_ENOMEM;
  const int grid = partial_grid(n);
  hipLaunchKernelGGL(k_sumsq_partials, dim3(grid), dim3(OPT_BLOCK), 0, (hipStream_t)stream, grads, n, partials);
  RLX_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_norm_from_partials, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, grid, norm_out);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_clip_adam_step_f32(rlx_ctx* ctx, float* params, const float* grads, float* m, float* v, int64_t n_params,
                           int64_t step, float lr, float max_grad_norm, float b1, float b2, float eps,
                           float* grad_norm_out, void* stream) {
  if (ctx && params && n_params > 0) ctx->sac_img.written(params, n_params);
  return clip_adam_step(ctx, params, grads, m, v, n_params, step, lr, max_grad_norm, b1, b2, eps, grad_norm_out,
                        (hipStream_t)stream, nullptr);
}

}  // extern "C"
