// adam_body.h -- the clip + Adam arithmetic for ONE parameter set as a device function: k_clip_adam / k_clip_adam2 (optim.hip) and the
// kernels that run several optimizers in one launch (sac.hip: entropy coefficient + policy + critics, three optimizers that
// sac/flax/sac.py:95,102,108 steps one after the other on independent gradients) are thin wrappers around clip_adam_job.
#pragma once
#include "common.h"
#include "gemm_bx.h"
#include "mlp.h"

namespace rlx {

// Weight matrices inside a flat parameter vector whose split-fp16 images (gemm_bx.h) are rewritten from the values the optimizer has
// just stored (SAC: the images persist from one update call to the next -- no k_bx_wfrag launch per call, rlx_sac_hparams::keep_images).
// A matrix with images is stepped TILE BY TILE: one workgroup owns 32 x 32 elements of W[in, out], runs the Adam arithmetic on them
// (four per thread, coalesced 128-B row segments), parks the two fp16 planes of every new value in LDS and then stores whole 1-KiB
// MFMA fragments -- two of the forward image (16 k x 32 j each), two of the transposed one (16 j x 32 k), per plane, and the forward
// ones of the Polyak target -- as one 16-B store per lane.  (Rounds 4-5 stepped element by element and wrote each element's 2-byte
// image entries where they fall: 4-12 two-byte stores per element, 16 B apart across a wave -- 25 us for the 1.1 M parameters of the
// configs[3] nets, bound by the L2's partial-sector writes.)  Everything that is not such a matrix (biases, LayerNorm, the heads) is
// the `rest`: one element per thread.  Image entries: the arithmetic of k_bx_wfrag (gemm_bx.hip), element by element.
struct BxEmitN {
  int n = 0;
  BxEmitLayer l[8];
};

constexpr int ADAM_TILE = 32;
constexpr int ADAM_MAX_LAYERS = 8;
struct AdamTileLayer {
  int64_t w_off;           // offset of W[in, out] inside the flat vector
  int in, out;
  u32x4 *nn, *tt, *nn_t;   // forward / transposed image of the parameters, forward image of the Polyak target (each may be null)
  int nt_nn, nt_tt;        // 32-column tiles per 16-k block of the images
  int first_tile, tiles_j; // index of the layer's first tile among the job's tile blocks; tiles along `out`
};
struct AdamRest {
  int64_t off;             // a run of elements outside every tiled matrix ...
  int64_t first;           // ... and the index of its first element among all rest elements
};

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
  int clip_mode = 0;       // 0: optax.clip_by_global_norm, 1: torch.nn.utils.clip_grad_norm_ (see clip_adam_job)
  float lr = 0.f, bc1 = 1.f, bc2 = 1.f;   // by-value schedule entry, used when sched == nullptr
  // the plan (adam_job_plan): blocks [0, n_tile_blocks) step the tiled matrices, the next n_rest_blocks the rest
  int n_layers = 0, n_rest = 0, n_tile_blocks = 0, n_rest_blocks = 0;
  int64_t rest_total = 0;
  AdamTileLayer layer[ADAM_MAX_LAYERS];
  AdamRest rest[ADAM_MAX_LAYERS + 2];      // rest[n_rest] = {n, rest_total}: the sentinel
};

// host: lay out the blocks of a job.  emit / emit_t: the image tables of the matrices inside p / inside polyak_target, in ascending
// offset order (emit_t entries pair with the emit entry of the same offset)
template <class TA, class TB>
inline void adam_job_plan(AdamJob& J, const TA& emit, const TB& emit_t) {
  J.n_layers = 0;
  int tiles = 0;
  int qa = 0, qb = 0;      // merge of the two tables by offset
  while ((qa < emit.n || qb < emit_t.n) && J.n_layers < ADAM_MAX_LAYERS) {
    const BxEmitLayer* a = qa < emit.n ? &emit.l[qa] : nullptr;
    const BxEmitLayer* b = qb < emit_t.n ? &emit_t.l[qb] : nullptr;
    const bool take_a = a && (!b || a->w_off <= b->w_off), take_b = b && (!a || b->w_off <= a->w_off);
    const BxEmitLayer& e = take_a ? *a : *b;
    AdamTileLayer& L = J.layer[J.n_layers++];
    L.w_off = e.w_off; L.in = e.in; L.out = e.out;
    L.nn = take_a ? static_cast<u32x4*>(a->nn) : nullptr;
    L.tt = take_a ? static_cast<u32x4*>(a->tt) : nullptr;
    L.nn_t = take_b ? static_cast<u32x4*>(b->nn) : nullptr;
    L.nt_nn = e.nt_nn; L.nt_tt = e.nt_tt;
    qa += take_a; qb += take_b;
    L.first_tile = tiles;
    L.tiles_j = (e.out + ADAM_TILE - 1) / ADAM_TILE;
    tiles += L.tiles_j * ((e.in + ADAM_TILE - 1) / ADAM_TILE);
  }
  J.n_tile_blocks = tiles;
  J.n_rest = 0;
  int64_t at = 0, cnt = 0;
  for (int q = 0; q <= J.n_layers; ++q) {
    const int64_t end = q < J.n_layers ? J.layer[q].w_off : J.n;
    if (end > at) { J.rest[J.n_rest].off = at; J.rest[J.n_rest].first = cnt; ++J.n_rest; cnt += end - at; }
    if (q < J.n_layers) at = J.layer[q].w_off + (int64_t)J.layer[q].in * J.layer[q].out;
  }
  J.rest[J.n_rest].off = J.n; J.rest[J.n_rest].first = cnt;
  J.rest_total = cnt;
  const int64_t nb = (cnt + 255) / 256;
  J.n_rest_blocks = (int)(nb > 2048 ? 2048 : nb);      // (strided beyond that: every block re-reduces the norm's partials)
}

// one element's fp16 planes of val * X_WSCALE (bx_split2 in k_bx_wfrag, element by element)
__device__ __forceinline__ void adam_split(float val, uint16_t& h0, uint16_t& h1) {
  uint32_t p0, p1;
  bx_split2(val * X_WSCALE, 0.f, p0, p1);
  h0 = (uint16_t)(p0 & 0xffffu);
  h1 = (uint16_t)(p1 & 0xffffu);
}

// the gradient norm from the per-block partial sums (every block computes it; s_buf: 4 floats of LDS)
__device__ __forceinline__ float adam_job_norm(const AdamJob& J, float* s_buf) {
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
  return sqrtf(tot);
}

struct AdamConsts {
  float lr, bc1, bc2, b1, b2, eps, norm, max_norm, wd, tau, coef;
  bool clip;
  int clip_mode;
};
// clip_mode 0: optax.clip_by_global_norm -- g if norm < c else (g / norm) * c
// clip_mode 1: torch.nn.utils.clip_grad_norm_ -- g * min(1, c / (norm + 1e-6))   (fastsac/pytorch/fastsac.py:129-130,218-219)
__device__ __forceinline__ void adam_set_norm(AdamConsts& c, float norm) {
  c.norm = norm;
  c.clip = (c.max_norm > 0.f) && (c.clip_mode == 1 ? (c.max_norm / (norm + 1e-6f) < 1.0f) : !(norm < c.max_norm));
  c.coef = c.max_norm / (norm + 1e-6f);
}
// The arithmetic of one element, with every multiply-add fusion spelled out.  The function is inlined into several kernels and
// into two paths of each (tile / rest); the library is built with -ffp-contract=fast, under which the backend fuses `a * b + c * d` one
// way or the other depending on the code around it (a `#pragma clang fp contract(off)` does not stop it) -- the SAC update with kept
// images (tile path) and with fresh ones (rest path) then differed in the last bit of a third of the Polyak targets.  Written as
// explicit fmaf chains whose remaining products feed only fma addends, there is nothing left to fuse differently.
__device__ __forceinline__ void adam_element(const AdamConsts& c, float gi, float m0, float v0, float p0, float& mi, float& vi,
                                             float& pn) {
  if (c.clip) gi = c.clip_mode == 1 ? gi * c.coef : (gi / c.norm) * c.max_norm;
  mi = fmaf(c.b1, m0, (1.f - c.b1) * gi);                       // b1 m + (1 - b1) g
  vi = fmaf(c.b2, v0, ((1.f - c.b2) * gi) * gi);                // b2 v + (1 - b2) g^2
  const float mhat = mi / c.bc1;
  const float vhat = vi / c.bc2;
  // wd != 0: torch.optim.AdamW's decoupled decay, p *= 1 - lr * wd in front of the Adam step (fastsac.py:88-91)
  const float decay = fmaf(-c.lr, c.wd, 1.0f);
  pn = fmaf(-c.lr, mhat / (sqrtf(vhat) + c.eps), p0 * decay);   // p (1 - lr wd) - lr mhat / (sqrt(vhat) + eps)
}
// SAC target critics: target = tau * params + (1 - tau) * target with the parameters just written (sac.py:208)
__device__ __forceinline__ float adam_polyak(const AdamConsts& c, float pn, float tg) {
  return fmaf(c.tau, pn, (1.f - c.tau) * tg);
}

constexpr int ADAM_LDS_PITCH = 40;                                         // halfwords per tile row: 16-B aligned rows, 80 B apart
constexpr int ADAM_LDS_HALVES = 2 * X_NP * ADAM_TILE * ADAM_LDS_PITCH;      // [p | target][plane][k][j]

// bid = this block's index within the job (256 threads); s_buf: 4 floats, s_tile: ADAM_LDS_HALVES halfwords of LDS
__device__ __forceinline__ void clip_adam_job(const AdamJob& J, int bid, float b1, float b2, float eps, float* s_buf,
                                              uint16_t* s_tile) {
  static_assert(X_NP == 2, "two fp16 planes per image");
  const int t = threadIdx.x;
  AdamConsts c;
  if (J.sched) { c.lr = J.sched[0]; c.bc1 = J.sched[1]; c.bc2 = J.sched[2]; }
  else { c.lr = J.lr; c.bc1 = J.bc1; c.bc2 = J.bc2; }
  c.clip_mode = J.clip_mode;
  c.b1 = b1; c.b2 = b2; c.eps = eps; c.max_norm = J.max_norm; c.wd = J.weight_decay; c.tau = J.tau;
  if (bid < J.n_tile_blocks) {
    int li = 0;
#pragma unroll
    for (int q = 1; q < ADAM_MAX_LAYERS; ++q)
      if (q < J.n_layers && bid >= J.layer[q].first_tile) li = q;
    const AdamTileLayer& L = J.layer[li];
    const int tile = bid - L.first_tile, tk = tile / L.tiles_j, tj = tile - tk * L.tiles_j;
    const int k0 = tk * ADAM_TILE, j0 = tj * ADAM_TILE;
    const int jj = t & 31, kq = t >> 5, j = j0 + jj;
    const bool has_t = J.polyak_target != nullptr;
    // every load of the tile in flight before the norm's barrier (clamped addresses, masked afterwards)
    float g[4], m0[4], v0[4], p0[4], tg[4];
    int64_t idx[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + kq + 8 * u;
      ok[u] = k < L.in && j < L.out;
      idx[u] = ok[u] ? L.w_off + (int64_t)k * L.out + j : L.w_off;
      g[u] = J.g[idx[u]];
      m0[u] = J.m[idx[u]];
      v0[u] = J.v[idx[u]];
      p0[u] = J.p[idx[u]];
      tg[u] = has_t ? J.polyak_target[idx[u]] : 0.f;
    }
    adam_set_norm(c, adam_job_norm(J, s_buf));
    if (bid == 0 && t == 0 && J.norm_out) J.norm_out[0] = c.norm;
    // A non-finite gradient norm (an operand left the split-fp16 window of gemm_bx.h, or the loss itself overflowed) must not
    // reach the parameters or the Adam moments: the step is SKIPPED (every block sees the same norm), the norm is still
    // reported, and the plugins' per-iteration finite check raises with the last good parameters intact.
    if (!(c.norm < INFINITY)) return;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float mi, vi, pn;
      adam_element(c, g[u], m0[u], v0[u], p0[u], mi, vi, pn);
      uint16_t h0 = 0, h1 = 0, q0 = 0, q1 = 0;
      if (ok[u]) {
        J.m[idx[u]] = mi;
        J.v[idx[u]] = vi;
        J.p[idx[u]] = pn;
        adam_split(pn, h0, h1);
        if (has_t) {
          const float tn = adam_polyak(c, pn, tg[u]);
          J.polyak_target[idx[u]] = tn;
          adam_split(tn, q0, q1);
        }
      }
      const int at = (kq + 8 * u) * ADAM_LDS_PITCH + jj;
      s_tile[at] = h0;
      s_tile[ADAM_TILE * ADAM_LDS_PITCH + at] = h1;
      if (has_t) {
        s_tile[2 * ADAM_TILE * ADAM_LDS_PITCH + at] = q0;
        s_tile[3 * ADAM_TILE * ADAM_LDS_PITCH + at] = q1;
      }
    }
    __syncthreads();
    // whole fragments: wave w stores fragment f = w >> 1 of plane w & 1
    const int l = t & 63, f = t >> 7, pl = (t >> 6) & 1;
#pragma unroll
    for (int set = 0; set < 2; ++set) {      // forward image of the parameters / of the target: 16 k x 32 j, lane = (k >> 3 & 1) * 32 + j, element k & 7
      u32x4* img = set ? L.nn_t : L.nn;
      if (!img || k0 + 16 * f >= L.in) continue;
      const uint16_t* src = s_tile + (2 * set + pl) * ADAM_TILE * ADAM_LDS_PITCH + (16 * f + 8 * (l >> 5)) * ADAM_LDS_PITCH + (l & 31);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = (uint32_t)src[(2 * e) * ADAM_LDS_PITCH] | ((uint32_t)src[(2 * e + 1) * ADAM_LDS_PITCH] << 16);
      img[((int64_t)(((k0 >> 4) + f) * L.nt_nn + (j0 >> 5)) * X_NP + pl) * 64 + l] = o;
    }
    if (L.tt && j0 + 16 * f < L.out) {       // transposed image: 16 j x 32 k, lane = (j >> 3 & 1) * 32 + k, element j & 7
      const u32x4 o = *reinterpret_cast<const u32x4*>(s_tile + pl * ADAM_TILE * ADAM_LDS_PITCH + (l & 31) * ADAM_LDS_PITCH + 16 * f + 8 * (l >> 5));
      L.tt[((int64_t)(((j0 >> 4) + f) * L.nt_tt + (k0 >> 5)) * X_NP + pl) * 64 + l] = o;
    }
    return;
  }
  // the rest: one element per thread and pass
  adam_set_norm(c, adam_job_norm(J, s_buf));
  if (bid == 0 && t == 0 && J.norm_out) J.norm_out[0] = c.norm;
  if (!(c.norm < INFINITY)) return;
  const int64_t stride = (int64_t)J.n_rest_blocks * 256;
  for (int64_t r = (int64_t)(bid - J.n_tile_blocks) * 256 + t; r < J.rest_total; r += stride) {
    int s = 0;
#pragma unroll
    for (int q = 1; q < ADAM_MAX_LAYERS + 1; ++q)
      if (q < J.n_rest && r >= J.rest[q].first) s = q;
    const int64_t i = J.rest[s].off + (r - J.rest[s].first);
    float mi, vi, pn;
    adam_element(c, J.g[i], J.m[i], J.v[i], J.p[i], mi, vi, pn);
    J.m[i] = mi;
    J.v[i] = vi;
    J.p[i] = pn;
    if (J.polyak_target) J.polyak_target[i] = adam_polyak(c, pn, J.polyak_target[i]);
  }
}

}  // namespace rlx
