// optim.hip -- optax.chain(clip_by_global_norm(c), inject_hyperparams(adam)(lr))
// as used at rl_x/algorithms/ppo/flax/ppo.py:84-100 and applied at :212-213
// (optax>=0.2.6 is third-party, restated; CPU twin: oracle/ppo.py adam_step).
//
// Two launches per network: (1) per-block partial sums of g^2 (deterministic order),
// (2) every block re-reduces the <=1024 partials to the global norm, then clip + Adam.
// HBM/L2 traffic: 28 B per parameter (read p,g,m,v; write p,m,v) -- 1.4 MB fits L2.
#include "mlp.h"
#include "gemm_bx.h"

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

// one parameter set: blocks [0, nblk) of OPT_BLOCK threads cover it, bid = this block's index among them
__device__ __forceinline__ void clip_adam_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, int64_t n, const float* __restrict__ partials,
                                               int n_partials, float lr, float max_norm, float b1, float b2, float eps,
                                               float bc1, float bc2, float* __restrict__ norm_out, const BxEmit& emit,
                                               float* __restrict__ polyak_target, float tau, float weight_decay, int clip_mode,
                                               int bid, int nblk, float* s_buf) {
  // (four loads in flight; added in the same order as one by one -- an absent element adds +0 to a non-negative sum)
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_partials; i += OPT_BLOCK * 4) {
    float pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pv[u] = i + OPT_BLOCK * u < n_partials ? partials[i + OPT_BLOCK * u] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += pv[u];
  }
  const float norm = sqrtf(block_sum(acc, s_buf));
  if (bid == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
  // A non-finite gradient norm (an operand left the split-fp16 window of gemm_bx.h, or the loss itself overflowed) must not
  // reach the parameters or the Adam moments: the step is SKIPPED (every block sees the same norm), the norm is still
  // reported, and the plugins' per-iteration finite check raises with the last good parameters intact.
  if (!(norm < INFINITY)) return;
  // clip_mode 0: optax.clip_by_global_norm -- g if norm < c else (g / norm) * c
  // clip_mode 1: torch.nn.utils.clip_grad_norm_ -- g * min(1, c / (norm + 1e-6))   (fastsac/pytorch/fastsac.py:129-130,218-219)
  const bool clip = (max_norm > 0.f) && (clip_mode == 1 ? (max_norm / (norm + 1e-6f) < 1.0f) : !(norm < max_norm));
  const float coef = max_norm / (norm + 1e-6f);
  const int64_t stride = (int64_t)nblk * OPT_BLOCK;
  for (int64_t i = (int64_t)bid * OPT_BLOCK + threadIdx.x; i < n; i += stride) {
    float gi = g[i];
    if (clip) gi = clip_mode == 1 ? gi * coef : (gi / norm) * max_norm;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float mhat = mi / bc1;
    const float vhat = vi / bc2;
    // weight_decay != 0: torch.optim.AdamW's decoupled decay, p *= 1 - lr * wd in front of the Adam step (fastsac.py:88-91)
    const float pn = p[i] * (1.0f - lr * weight_decay) - lr * (mhat / (sqrtf(vhat) + eps));
    p[i] = pn;
    // SAC target critics: target = tau * params + (1 - tau) * target with the parameters just written (sac.py:208)
    if (polyak_target) polyak_target[i] = tau * pn + (1.f - tau) * polyak_target[i];
    // hidden-layer weights: rewrite their split image entries (same arithmetic as bx_split2 in k_bx_wfrag, element by element)
    for (int q = 0; q < emit.n; ++q) {
      const BxEmitLayer& e = emit.l[q];
      const int64_t r = i - e.w_off;
      if (r < 0 || r >= (int64_t)e.in * e.out) continue;
      const int k = (int)(r / e.out), j = (int)(r - (int64_t)k * e.out);
      uint16_t h[X_NP];
      {
        uint32_t p0, p1;
        bx_split2(pn * X_WSCALE, 0.f, p0, p1);
        h[0] = (uint16_t)(p0 & 0xffffu);
        h[1] = (uint16_t)(p1 & 0xffffu);
      }
      if (e.nn) {   // B(k, j) = W[k][j]: 16-k block k >> 4, half (k >> 3) & 1, element k & 7; column tile j >> 5, lane j & 31
        uint16_t* img = reinterpret_cast<uint16_t*>(e.nn);
        const int64_t base = ((int64_t)((k >> 4) * e.nt_nn + (j >> 5)) * X_NP) * 64 + ((k >> 3) & 1) * 32 + (j & 31);
#pragma unroll
        for (int pl = 0; pl < X_NP; ++pl) img[(base + pl * 64) * 8 + (k & 7)] = h[pl];
      }
      if (e.tt) {   // B(k', j') = W[j'][k'] with k' = j, j' = k
        uint16_t* img = reinterpret_cast<uint16_t*>(e.tt);
        const int64_t base = ((int64_t)((j >> 4) * e.nt_tt + (k >> 5)) * X_NP) * 64 + ((j >> 3) & 1) * 32 + (k & 31);
#pragma unroll
        for (int pl = 0; pl < X_NP; ++pl) img[(base + pl * 64) * 8 + (j & 7)] = h[pl];
      }
    }
  }
}

__global__ __launch_bounds__(OPT_BLOCK) void k_clip_adam(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                         const float* __restrict__ partials, int n_partials, float lr,
                                                         float max_norm, float b1, float b2, float eps, float bc1,
                                                         float bc2, float* __restrict__ norm_out,
                                                         const float* __restrict__ sched, BxEmit emit,
                                                         float* __restrict__ polyak_target, float tau, float weight_decay,
                                                         int clip_mode) {
  // sched (optional, DEVICE {lr, 1 - b1^step, 1 - b2^step}): the per-update values come from a device table instead of the
  // launch arguments (the SAC update keeps them next to its key: one small upload per call)
  if (sched) { lr = sched[0]; bc1 = sched[1]; bc2 = sched[2]; }
  __shared__ float s_buf[OPT_BLOCK / 64];
  clip_adam_body(p, g, m, v, n, partials, n_partials, lr, max_norm, b1, b2, eps, bc1, bc2, norm_out, emit, polyak_target, tau,
                 weight_decay, clip_mode, (int)blockIdx.x, (int)gridDim.x, s_buf);
}

// Two independent optimizers in ONE launch (ppo.py:212-213 steps the policy's and the critic's TrainState one after the other on
// independent gradients): blocks [0, nb0) run job 0, blocks [nb0, nb0 + nb1) job 1.  Same arithmetic as two k_clip_adam launches.
struct Adam2Job {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  const float* partials;
  int n_partials;
  float* norm_out;
};
__global__ __launch_bounds__(OPT_BLOCK) void k_clip_adam2(Adam2Job j0, Adam2Job j1, int nb0, int nb1, float lr, float max_norm,
                                                          float b1, float b2, float eps, float bc1, float bc2,
                                                          const float* __restrict__ sched, BxEmit e0, BxEmit e1) {
  if (sched) { lr = sched[0]; bc1 = sched[1]; bc2 = sched[2]; }
  __shared__ float s_buf[OPT_BLOCK / 64];
  if ((int)blockIdx.x < nb0)
    clip_adam_body(j0.p, j0.g, j0.m, j0.v, j0.n, j0.partials, j0.n_partials, lr, max_norm, b1, b2, eps, bc1, bc2, j0.norm_out, e0,
                   nullptr, 0.f, 0.f, 0, (int)blockIdx.x, nb0, s_buf);
  else
    clip_adam_body(j1.p, j1.g, j1.m, j1.v, j1.n, j1.partials, j1.n_partials, lr, max_norm, b1, b2, eps, bc1, bc2, j1.norm_out, e1,
                   nullptr, 0.f, 0.f, 0, (int)blockIdx.x - nb0, nb1, s_buf);
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

int launch_clip_adam(float* params, const float* grads, float* m, float* v, int64_t n, const float* sumsq_partials,
                     int n_partials, int64_t step, float lr, float max_norm, float b1, float b2, float eps,
                     float* norm_out, hipStream_t st, const float* sched_dev, const BxEmit* emit, float* polyak_target,
                     float tau, float weight_decay, int clip_mode) {
  BxEmit em;
  em.n = 0;
  if (emit) em = *emit;
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
  const float bc2 = (float)(1.0 - pow((double)b2, (double)step));
  const int agrid = div_up(n, OPT_BLOCK) > 2048 ? 2048 : div_up(n, OPT_BLOCK);
  hipLaunchKernelGGL(k_clip_adam, dim3(agrid), dim3(OPT_BLOCK), 0, st, params, grads, m, v, n, sumsq_partials,
                     n_partials, lr, max_norm, b1, b2, eps, bc1, bc2, norm_out, sched_dev, em, polyak_target, tau, weight_decay, clip_mode);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// both networks' clip + Adam in one launch (k_clip_adam2); g-norm partials per network as launch_reduce_segments left them
int launch_clip_adam2(float* p0, const float* g0, float* m0, float* v0, int64_t n0, const float* part0, int np0, float* norm0,
                      const BxEmit* e0, float* p1, const float* g1, float* m1, float* v1, int64_t n1, const float* part1, int np1,
                      float* norm1, const BxEmit* e1, int64_t step, float lr, float max_norm, float b1, float b2, float eps,
                      hipStream_t st, const float* sched_dev) {
  BxEmit em0, em1;
  em0.n = em1.n = 0;
  if (e0) em0 = *e0;
  if (e1) em1 = *e1;
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
  const float bc2 = (float)(1.0 - pow((double)b2, (double)step));
  const int nb0 = div_up(n0, OPT_BLOCK) > 1024 ? 1024 : div_up(n0, OPT_BLOCK);
  const int nb1 = div_up(n1, OPT_BLOCK) > 1024 ? 1024 : div_up(n1, OPT_BLOCK);
  const Adam2Job j0{p0, g0, m0, v0, n0, part0, np0, norm0}, j1{p1, g1, m1, v1, n1, part1, np1, norm1};
  hipLaunchKernelGGL(k_clip_adam2, dim3(nb0 + nb1), dim3(OPT_BLOCK), 0, st, j0, j1, nb0, nb1, lr, max_norm, b1, b2, eps, bc1, bc2,
                     sched_dev, em0, em1);
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
