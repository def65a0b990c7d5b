// adam_body.h -- the clip + Adam arithmetic of k_clip_adam (optim.hip) for ONE parameter set as a device function, so that a
// caller's kernel can run several optimizers in one launch (sac.hip: entropy coefficient + policy + critics, three optimizers
// that sac/flax/sac.py:95,102,108 steps one after the other on independent gradients).  Same expressions, same order.
#pragma once
#include "common.h"
#include "gemm_bx.h"
#include "mlp.h"

namespace rlx {

// weight matrices inside a flat parameter vector whose split-fp16 images (gemm_bx.h) are rewritten from the values the optimizer has
// just stored (SAC: the images persist from one update call to the next -- no k_bx_wfrag launch per call, rlx_sac_hparams::keep_images)
struct BxEmitN {
  int n = 0;
  BxEmitLayer l[8];
};

// element i of the flat vector now holds `val`: rewrite its entries of the forward / transposed images (the arithmetic of k_bx_wfrag
// and of clip_adam_body in optim.hip, element by element)
__device__ __forceinline__ void bx_emit_value(const BxEmitN& emit, int64_t i, float val) {
  for (int q = 0; q < emit.n; ++q) {
    const BxEmitLayer& e = emit.l[q];
    const int64_t r = i - e.w_off;
    if (r < 0 || r >= (int64_t)e.in * e.out) continue;
    const int k = (int)(r / e.out), j = (int)(r - (int64_t)k * e.out);
    uint32_t p0, p1;
    bx_split2(val * X_WSCALE, 0.f, p0, p1);
    const uint16_t h[X_NP] = {(uint16_t)(p0 & 0xffffu), (uint16_t)(p1 & 0xffffu)};
    if (e.nn) {
      uint16_t* img = reinterpret_cast<uint16_t*>(e.nn);
      const int64_t base = ((int64_t)((k >> 4) * e.nt_nn + (j >> 5)) * X_NP) * 64 + ((k >> 3) & 1) * 32 + (j & 31);
#pragma unroll
      for (int pl = 0; pl < X_NP; ++pl) img[(base + pl * 64) * 8 + (k & 7)] = h[pl];
    }
    if (e.tt) {
      uint16_t* img = reinterpret_cast<uint16_t*>(e.tt);
      const int64_t base = ((int64_t)((j >> 4) * e.nt_tt + (k >> 5)) * X_NP) * 64 + ((j >> 3) & 1) * 32 + (k & 31);
#pragma unroll
      for (int pl = 0; pl < X_NP; ++pl) img[(base + pl * 64) * 8 + (j & 7)] = h[pl];
    }
  }
}

struct AdamJob {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  const float* partials;   // per-block sums of g^2 (launch_sumsq_partials / the weight-gradient kernels)
  int n_partials;
  float max_norm;          // <= 0: no clipping
  float* norm_out;         // optional: the gradient norm
  const float* sched;      // DEVICE {lr, 1 - b1^step, 1 - b2^step}
  float* polyak_target;    // optional: target = tau * p_new + (1 - tau) * target
  float tau, weight_decay;
  BxEmitN emit;            // images of matrices inside p
  BxEmitN emit_t;          // images of matrices inside polyak_target
};

// blocks [0, nblk) of 256 threads cover the job; bid = this block's index within it; s_buf: 4 floats of LDS
__device__ __forceinline__ void clip_adam_job(const AdamJob& J, int bid, int nblk, float b1, float b2, float eps, float* s_buf) {
  const float lr = J.sched[0], bc1 = J.sched[1], bc2 = J.sched[2];
  float acc = 0.f;
  for (int i = threadIdx.x; i < J.n_partials; i += 256 * 4) {   // four loads in flight, added in index order
    float pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pv[u] = i + 256 * u < J.n_partials ? J.partials[i + 256 * u] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += pv[u];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = acc;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) tot += s_buf[i];
  __syncthreads();
  const float norm = sqrtf(tot);
  if (bid == 0 && threadIdx.x == 0 && J.norm_out) J.norm_out[0] = norm;
  if (!(norm < INFINITY)) return;   // non-finite gradients never reach the parameters / moments (optim.hip: clip_adam_body)
  const bool clip = (J.max_norm > 0.f) && !(norm < J.max_norm);
  const int64_t stride = (int64_t)nblk * 256;
  for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < J.n; i += stride) {
    float gi = J.g[i];
    if (clip) gi = (gi / norm) * J.max_norm;
    const float mi = b1 * J.m[i] + (1.f - b1) * gi;
    const float vi = b2 * J.v[i] + (1.f - b2) * gi * gi;
    J.m[i] = mi;
    J.v[i] = vi;
    const float mhat = mi / bc1;
    const float vhat = vi / bc2;
    const float pn = J.p[i] * (1.0f - lr * J.weight_decay) - lr * (mhat / (sqrtf(vhat) + eps));
    J.p[i] = pn;
    bx_emit_value(J.emit, i, pn);
    if (J.polyak_target) {
      const float tn = J.tau * pn + (1.f - J.tau) * J.polyak_target[i];
      J.polyak_target[i] = tn;
      bx_emit_value(J.emit_t, i, tn);
    }
  }
}

}  // namespace rlx
