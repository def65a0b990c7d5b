// fastsac.hip -- FastSAC update steps (rl_x/algorithms/fastsac/pytorch/fastsac.py:105-241, :323-329): distributional (C51) twin
// critics, tanh-Gaussian policy with a tanh-mapped log-std and an action scale, AdamW, Polyak targets.
//
// Networks (policy.py:46-57, q_network.py:27-38): every hidden layer is Dense -> LayerNorm (torch: eps 1e-5) -> SiLU, then a
// Dense head -- rlx_lnmlp_desc.  They are composed from the library's GEMM stages (launch_gemm_fwd / stage_dx / stage_dw: the
// exact-fp32 MFMA engine for the forward and input-gradient products -- no weight images are registered here --, the
// split-operand weight-gradient kernel for batches >= 4096 rows) and the row-wise LayerNorm + activation kernels of ln_kernels.h
// (k_ln_act_wide: widths up to 768, eps argument, SiLU' from the recomputed pre-activation).  CPU twin: oracle/fastsac.py,
// pinned by outputs of the reference's own modules and closures (tests/golden/reference_fastsac.npz).
//
// Noise: the reference draws with torch's CUDA generator (Normal.rsample), which no other implementation reproduces; the
// library uses its counter RNG (threefry, the key split per call like rlx_sac_*), and rlx_dbg_set_sac_noise injects a given
// eps for parity tests.
#include "gemm_bx.h"
#include "ln_kernels.h"
#include "mlp.h"

extern "C" int rlx_c51_critic_loss_f32(rlx_ctx* ctx, const float* q1_logits, const float* q2_logits, const float* q1_next_logits,
                                       const float* q2_next_logits, const float* rewards, const float* dones,
                                       const float* truncations, const float* effective_n_steps, const float* next_log_probs,
                                       const float* log_alpha, int64_t B, int nr_atoms, float gamma, float v_min, float v_max,
                                       int clipped_double_q, float* d_q1_logits, float* d_q2_logits, float* out4, void* stream);

namespace rlx {

constexpr float FS_LN_EPS = 1e-5f;          // torch.nn.LayerNorm default
constexpr float FS_LOG_SQRT_2PI = 0.91893853320467274178f;

struct LnLayer { int in, out; int64_t W, b, g, be; };
struct LnLayout {
  int n_hidden;
  LnLayer layer[4];
  int head_in, head_out;
  int64_t hW, hb, n_params;
};

static int ln_check(const rlx_lnmlp_desc& d) {
  RLX_REQUIRE(d.n_hidden >= 1 && d.n_hidden <= 4 && d.in_dim > 0 && d.out_dim > 0, RLX_EINVAL, "rlx_lnmlp_desc: 1..4 hidden layers, positive widths");
  for (int l = 0; l < d.n_hidden; ++l)
    RLX_REQUIRE(d.hidden[l] > 0 && d.hidden[l] % 64 == 0 && d.hidden[l] <= 768, RLX_EUNSUP,
                "rlx_lnmlp_desc: hidden widths must be multiples of 64, at most 768 (the LayerNorm kernel holds a row in one wave)");
  return RLX_OK;
}

static LnLayout ln_layout(const rlx_lnmlp_desc& d) {
  LnLayout L{};
  L.n_hidden = d.n_hidden;
  int64_t off = 0;
  int in = d.in_dim;
  for (int l = 0; l < d.n_hidden; ++l) {
    LnLayer& o = L.layer[l];
    o.in = in; o.out = d.hidden[l];
    o.W = off; off += (int64_t)in * o.out;
    o.b = off; off += o.out;
    o.g = off; off += o.out;
    o.be = off; off += o.out;
    in = o.out;
  }
  L.head_in = in; L.head_out = d.out_dim;
  L.hW = off; off += (int64_t)in * d.out_dim;
  L.hb = off; off += d.out_dim;
  L.n_params = off;
  return L;
}

struct LnBufs { float* Z[4]; float* H[4]; };   // [M, out_l]: pre-LayerNorm values, activations (the backward reuses H_l for dH_l / dZ_l)

static size_t ln_buf_floats(const LnLayout& L, int64_t M) {
  size_t n = 0;
  for (int l = 0; l < L.n_hidden; ++l) n += 2 * (((size_t)M * L.layer[l].out + 63) & ~size_t(63));
  return n;
}
static void ln_carve(const LnLayout& L, int64_t M, float*& cur, LnBufs* b) {
  for (int l = 0; l < L.n_hidden; ++l) {
    const size_t n = ((size_t)M * L.layer[l].out + 63) & ~size_t(63);
    b->Z[l] = cur; cur += n;
    b->H[l] = cur; cur += n;
  }
}

static inline int ln_rows_grid(const rlx_ctx* ctx, int64_t M) {
  int grid = div_up(M, 4);
  if (grid > ctx->num_cus * 8) grid = ctx->num_cus * 8;
  return grid;
}

static inline int ln_bwd_rows_grid(const rlx_ctx* ctx, int64_t M) {   // 16 rows per workgroup: fewer partial slabs (sac.hip: ln_bwd_grid)
  int grid = div_up(M, 16);
  if (grid > ctx->num_cus * 4) grid = ctx->num_cus * 4;
  return grid < 1 ? 1 : grid;
}

int fs_head_fwd(const float* H, const float* W, const float* b, float* out, int64_t M, int K, int N, hipStream_t st);
int fs_concat(const float* obs, int Oc, const float* act, int A, float* out, int ld, int64_t M, hipStream_t st);
int fs_head_bwd(rlx_ctx* ctx, float* H_dH, const float* W, const float* d, float* gW, float* gb, int64_t M, int K, int N, hipStream_t st);

// Split-operand weight images for the trunk GEMMs of a pass with >= 4096 rows (gemm_bx.h): launch_gemm_fwd / stage_dx pick them up
// by weight pointer.  nets[i]: parameter vector, layout, whether the pass needs the transposed images (input gradients).
struct FsNet { const float* p; const LnLayout* L; bool bwd; };
static int fs_images(rlx_ctx* ctx, const FsNet* nets, int n, int64_t M, hipStream_t st) {
  if (M < 4096 || !ctx->gemm_bx) return RLX_OK;
  BxMat mats[BX_MAX_JOBS];
  int k = 0;
  for (int i = 0; i < n; ++i)
    for (int l = 0; l < nets[i].L->n_hidden; ++l) {
      const LnLayer& o = nets[i].L->layer[l];
      if (o.in % 4 != 0 || k >= BX_MAX_JOBS / 2) continue;      // (a ragged first layer stays on the exact engine)
      mats[k++] = BxMat{nets[i].p + o.W, o.in, o.out, true, nets[i].bwd && l > 0};
    }
  if (!k) return RLX_OK;
  const int rc = bx_prepare_mats(ctx, mats, k, st);
  if (rc) return rc;
  // the second critic's passes run on the side stream under scratch bank 1 (FsFork): the same images serve both banks
  for (int i = 0; i < ctx->bx_n[0]; ++i) ctx->bx_img[1][i] = ctx->bx_img[0][i];
  ctx->bx_n[1] = ctx->bx_n[0];
  return RLX_OK;
}

// Fork / join of the update's two independent halves (critic 1 || critic 2; target passes || online passes): the side half runs
// on ctx->side under scratch bank 1.  With option two_streams = 0 both halves stay on the caller's stream.
struct FsFork {
  rlx_ctx* c;
  hipStream_t main_st, side_st;
  bool on;
  int next_ev = 0;
  FsFork(rlx_ctx* ctx, hipStream_t st) : c(ctx), main_st(st), side_st(st), on(false) {}
  int begin() {
    if (!c->two_streams) return RLX_OK;
    const int rc = ctx_sac_streams(c);
    if (rc) return rc;
    side_st = c->side;
    on = side_st != main_st;
    return RLX_OK;
  }
  int fork() {   // the side stream sees everything issued on the main stream so far
    if (!on) return RLX_OK;
    hipEvent_t e = c->sac_ev[next_ev++ % 6];
    RLX_HIP_TRY(hipEventRecord(e, main_st));
    RLX_HIP_TRY(hipStreamWaitEvent(side_st, e, 0));
    return RLX_OK;
  }
  int join() {   // the main stream waits for the side stream
    c->bank = 0;
    if (!on) return RLX_OK;
    hipEvent_t e = c->sac_ev[next_ev++ % 6];
    RLX_HIP_TRY(hipEventRecord(e, side_st));
    RLX_HIP_TRY(hipStreamWaitEvent(main_st, e, 0));
    return RLX_OK;
  }
  hipStream_t side() { c->bank = on ? 1 : 0; return side_st; }   // (sets the scratch bank the following launches use)
  hipStream_t main() { c->bank = 0; return main_st; }
  ~FsFork() { c->bank = 0; }
};

// forward through all hidden layers and the head; x: [M, in] with row stride ldx (a multiple of four, zero padded)
static int ln_fwd(rlx_ctx* ctx, const LnLayout& L, const float* p, const float* x, int ldx, const LnBufs& b, float* head_out, int64_t M,
                  hipStream_t st) {
  const float* h = x;
  int ld = ldx;
  for (int l = 0; l < L.n_hidden; ++l) {
    const LnLayer& o = L.layer[l];
    int rc = launch_gemm_fwd(ctx, h, p + o.W, p + o.b, b.Z[l], M, o.out, o.in, RLX_ACT_NONE, st, ld, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ln_act_wide<false>, dim3(ln_rows_grid(ctx, M)), dim3(256), 0, st, (const float*)b.Z[l], b.H[l], p + o.g, p + o.be,
                       (float*)nullptr, M, o.out, RLX_ACT_SILU, FS_LN_EPS);
    RLX_LAUNCH_CHECK();
    h = b.H[l];
    ld = o.out;
  }
  return fs_head_fwd(h, p + L.hW, p + L.hb, head_out, M, L.head_in, L.head_out, st);
}

// floats the partial-sum buffers of one ln_bwd take from the deferred-reduction arena (stage_alloc rounds each to 64)
static size_t ln_bwd_stage_floats(const rlx_ctx* ctx, const LnLayout& L, int64_t M, bool grads) {
  auto a64 = [](size_t n) { return (n + 63) & ~size_t(63); };
  size_t n = 0;
  if (grads) n += a64((size_t)div_up(M, 32) * (((size_t)L.head_in * L.head_out + 3 & ~size_t(3)) + ((L.head_out + 3) & ~3)));
  for (int l = 0; l < L.n_hidden; ++l) {
    n += a64((size_t)ln_bwd_rows_grid(ctx, M) * 2 * L.layer[l].out);
    if (grads) n += a64(stage_dw_floats(ctx, M, L.layer[l].in, L.layer[l].out));
  }
  return n;
}
// One reduction launch for ALL the parameter-gradient partials of an update's backward passes (head slabs, LayerNorm scale /
// bias partials, weight-gradient slabs of every layer of every network): 14 launches of ~12 us per network otherwise.
struct FsDefer {
  rlx_ctx* c;
  ReduceDefer d;
  explicit FsDefer(rlx_ctx* ctx) : c(ctx) {}
  int begin(size_t floats) {
    d.base = (float*)scratch(c, SL_STAGE, floats * sizeof(float));
    if (!d.base) return RLX_ENOMEM;
    d.cap = floats;
    d.off = 0;
    d.tab.n = 0;
    c->defer = &d;
    return RLX_OK;
  }
  ~FsDefer() { if (c->defer == &d) c->defer = nullptr; }
};

// backward from d_head [M, head_out].  grads != NULL: parameter gradients (flat layout); dx != NULL: input gradient [M, in] (row
// stride lddx).  The activation buffers are consumed (dH_l / dZ_l overwrite H_l).
// dx_nc > 0: only the input columns [dx_c0, dx_c0 + dx_nc) (the policy loss wants dQ/da, 12 of 60 columns: the column-restricted
// product instead of an exact-fp32 GEMM over all of them -- 98 us x 4 per vector step were the one exact-engine GEMM of the step)
static int ln_bwd(rlx_ctx* ctx, const LnLayout& L, const float* p, const float* x, int ldx, const LnBufs& b, const float* d_head,
                  float* grads, float* dx, int lddx, int64_t M, hipStream_t st, int dx_c0 = 0, int dx_nc = 0) {
  const int last = L.n_hidden - 1;
  int rc;
  // head: weight / bias gradients from H_last, then dH_last over it
  rc = fs_head_bwd(ctx, b.H[last], p + L.hW, d_head, grads ? grads + L.hW : nullptr, grads ? grads + L.hb : nullptr, M, L.head_in,
                   L.head_out, st);
  if (rc) return rc;
  for (int l = last; l >= 0; --l) {
    const LnLayer& o = L.layer[l];
    const int grid = ln_bwd_rows_grid(ctx, M);
    float* part = stage_alloc(ctx, (size_t)grid * 2 * o.out);
    if (!part) return RLX_ENOMEM;
    hipLaunchKernelGGL(k_ln_act_wide<true>, dim3(grid), dim3(256), (size_t)8 * o.out * sizeof(float), st, (const float*)b.Z[l], b.H[l],
                       p + o.g, p + o.be, part, M, o.out, RLX_ACT_SILU, FS_LN_EPS);
    RLX_LAUNCH_CHECK();
    if (grads) {
      ReduceTable tab;
      tab.n = 0;
      tab.seg[tab.n++] = ReduceSeg{part, grads + o.g, (int64_t)o.out, (int64_t)2 * o.out, grid, 0, 1.f, 0.f, 1};
      tab.seg[tab.n++] = ReduceSeg{part + o.out, grads + o.be, (int64_t)o.out, (int64_t)2 * o.out, grid, 0, 1.f, 0.f, 1};
      rc = stage_reduce(ctx, tab, nullptr, nullptr, st);
      if (rc) return rc;
      rc = stage_dw(ctx, l == 0 ? x : b.H[l - 1], l == 0 ? ldx : o.in, b.H[l], M, o.in, o.out, grads + o.W, grads + o.b, nullptr, nullptr, st);
      if (rc) return rc;
    }
    if (l > 0) rc = stage_dx(ctx, b.H[l], p + o.W, b.H[l - 1], M, o.out, o.in, o.in, RLX_ACT_NONE, 0, st, nullptr);
    else if (dx && dx_nc > 0 && dx_cols_ok(o.out, dx_nc))
      rc = launch_dx_cols(b.H[0], p + o.W + (int64_t)dx_c0 * o.out, dx + dx_c0, M, o.out, dx_nc, lddx, st);
    else if (dx) rc = stage_dx(ctx, b.H[0], p + o.W, dx, M, o.out, o.in, lddx, RLX_ACT_NONE, 0, st, nullptr);
    if (rc) return rc;
  }
  return RLX_OK;
}

// ---------------------------------------------------------------------------------------------------------------- kernels
// Dense heads (nr_atoms = 101 or 2 * act_dim outputs: widths the float4-tiled GEMM stages do not take -- their contraction and
// leading dimensions have to be multiples of four).  K <= 768 inputs, any N; small next to the trunk, so plain kernels.
// out[M, N] = H[M, K] @ W[K, N] + b: a workgroup of 128 threads per 8 rows x 128 columns, thread <-> output column.  The H values
// of a row are the same for every lane: their addresses are wave-uniform, so they arrive through the scalar cache as SGPR operands
// of the FMAs -- no LDS, no barrier, eight W loads in flight per thread.  (With H in LDS the kernel was LDS-issue bound: eight
// broadcast reads per k and wave; 22 -> 34 us when the W tiles went through LDS as well.)  One ascending fmaf chain per output.
__global__ __launch_bounds__(128) void k_fs_head_fwd(const float* __restrict__ H, const float* __restrict__ W, const float* __restrict__ b,
                                                     float* __restrict__ out, int64_t M, int K, int N) {
  const int64_t r0 = (int64_t)blockIdx.x * 8;
  const int n = blockIdx.y * 128 + threadIdx.x;
  const int nc = n < N ? n : N - 1;                      // (idle lanes compute a copy of the last column)
  const float* hr[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) hr[r] = H + (r0 + r < M ? r0 + r : M - 1) * K;
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
  int k = 0;
  for (; k + 8 <= K; k += 8) {
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = W[(int64_t)(k + u) * N + nc];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[r] = fmaf(hr[r][k + u], w[u], acc[r]);
    }
  }
  for (; k < K; ++k) {
    const float w = W[(int64_t)k * N + nc];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = fmaf(hr[r][k], w, acc[r]);
  }
  if (n < N) {
    const float bv = b[n];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r0 + r < M) out[(r0 + r) * N + n] = acc[r] + bv;
  }
}
// dH[M, K] = d[M, N] @ W[K, N]^T: a workgroup per 8 rows, d rows in LDS, thread <-> input column k
__global__ __launch_bounds__(256) void k_fs_head_dx(const float* __restrict__ d, const float* __restrict__ W, float* __restrict__ dH,
                                                    int64_t M, int K, int N) {
  extern __shared__ float s_d[];   // [8][N]
  const int64_t r0 = (int64_t)blockIdx.x * 8;
  {
    const int64_t left = (M - r0) * N;
    lds_stage<256, float>(s_d, d + r0 * N, 8 * N, left < 8 * N ? (int)left : 8 * N, 0.f);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) {
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    const float* wr = W + (int64_t)k * N;
    for (int n = 0; n < N; ++n) {
      const float w = wr[n];
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = fmaf(s_d[r * N + n], w, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r0 + r < M) dH[(r0 + r) * K + k] = acc[r];
  }
}
// partial[s][K * N + N]: dW = H^T d and db = column sums of d over the rows [s * rows, (s + 1) * rows) -- summed in row order;
// the slabs are added in slab order by the reduction kernel.  blockIdx.y = slab, thread <-> (k, n) pairs.
__global__ __launch_bounds__(256) void k_fs_head_dw(const float* __restrict__ H, const float* __restrict__ d, float* __restrict__ partial,
                                                    int64_t M, int K, int N, int rows, int64_t PS, int boff) {
  const int64_t r0 = (int64_t)blockIdx.y * rows;
  const int64_t r1 = r0 + rows < M ? r0 + rows : M;
  float* out = partial + (int64_t)blockIdx.y * PS;
  const int total = K * N + N;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    float acc = 0.f;
    if (e < K * N) {
      const int k = e / N, n = e - k * N;
      for (int64_t r = r0; r < r1; ++r) acc = fmaf(H[r * K + k], d[r * N + n], acc);
    } else {
      const int n = e - K * N;
      for (int64_t r = r0; r < r1; ++r) acc += d[r * N + n];
    }
    out[e < K * N ? e : boff + (e - K * N)] = acc;
  }
}

// The same partials from a register tile: workgroup = slab, thread = 8 k x NJ n outputs (k = 8 tk .. 8 tk + 7, n = tn + TN j),
// 16-row chunks of H and d staged in LDS; every output is one ascending fmaf chain over the slab's rows like above.  (The
// per-output loop above re-reads H and d from L2 for every output: 134 us at [8192, 192] x [8192, 101]; this one 10.)
// Slab layout: [K * N] dW, then db at float `boff` (both 16-byte aligned when the caller pads: vector path of the reduction).
template <int NJ>
__global__ __launch_bounds__(256) void k_fs_head_dw_tiled(const float* __restrict__ H, const float* __restrict__ d,
                                                          float* __restrict__ partial, int64_t M, int K, int N, int rows, int TK, int TN,
                                                          int64_t PS, int boff) {
  extern __shared__ __attribute__((aligned(16))) float s_hd[];
  constexpr int RC = 16;
  float* Hs = s_hd;              // [RC][K]
  float* Ds = s_hd + RC * K;     // [RC][N]
  const int tk = threadIdx.x / TN, tn = threadIdx.x - tk * TN;
  const bool on = tk < TK;
  float acc[8][NJ], bsum[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    bsum[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i][j] = 0.f;
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows;
  const int64_t r1 = r0 + rows < M ? r0 + rows : M;
  for (int64_t c0 = r0; c0 < r1; c0 += RC) {
    const int nr = (int)(r1 - c0 < RC ? r1 - c0 : RC);
    __syncthreads();
    lds_stage<256, float>(Hs, H + c0 * K, RC * K, nr * K, 0.f);
    lds_stage<256, float>(Ds, d + c0 * N, RC * N, nr * N, 0.f);
    __syncthreads();
    if (on) {
#pragma unroll 2
      for (int r = 0; r < RC; ++r) {
        const float4 h0 = *reinterpret_cast<const float4*>(Hs + r * K + 8 * tk);
        const float4 h1 = *reinterpret_cast<const float4*>(Hs + r * K + 8 * tk + 4);
        const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int n = tn + TN * j;
          const float dv = n < N ? Ds[r * N + n] : 0.f;
          bsum[j] += dv;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i][j] = fmaf(hv[i], dv, acc[i][j]);
        }
      }
    }
  }
  if (!on) return;
  float* out = partial + (int64_t)blockIdx.x * PS;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = tn + TN * j;
    if (n >= N) continue;
#pragma unroll
    for (int i = 0; i < 8; ++i) out[(int64_t)(8 * tk + i) * N + n] = acc[i][j];
    if (tk == 0) out[boff + n] = bsum[j];
  }
}

// [obs | action] rows of a critic input (row stride ld; the action columns may be filled later by k_fs_sample)
__global__ __launch_bounds__(256) void k_fs_concat(const float* __restrict__ obs, int Oc, const float* __restrict__ act, int A,
                                                   float* __restrict__ out, int ld, int64_t M) {
  const int64_t n = M * ld;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / ld;
    const int c = (int)(i - r * ld);
    out[i] = c < Oc ? obs[r * Oc + c] : (act && c < Oc + A ? act[r * A + (c - Oc)] : 0.f);
  }
}

// policy.get_action_and_log_prob (policy.py:75-90) from the head output [M, 2A] = [mean | raw log-std]: one thread per
// (row, action dim); the log-prob terms of a row are added in index order by one lane.  act_out: row stride ld, first column c0.
__global__ __launch_bounds__(256) void k_fs_sample(const float* __restrict__ head, const float* __restrict__ scale, uint32_t k0,
                                                   uint32_t k1, int scheme, const float* __restrict__ eps_inject,
                                                   float* __restrict__ act_out, int ld, int c0, float* __restrict__ logp, int64_t M,
                                                   int A, float ls_min, float ls_max, int deterministic, int64_t row_off,
                                                   int64_t M_global) {
  extern __shared__ float s_term[];   // [rows per block][A]
  const int rpb = 256 / A > 0 ? 256 / A : 1;
  const int rl = threadIdx.x / A, j = threadIdx.x - rl * A;
  const int64_t i = (int64_t)blockIdx.x * rpb + rl;
  const bool on = rl < rpb && i < M;
  if (on) {
    const float mean = head[i * 2 * A + j];
    const float ls = ls_min + 0.5f * (ls_max - ls_min) * (tanhf(head[i * 2 * A + A + j]) + 1.0f);
    const float sd = expf(ls);
    float eps = 0.f;
    if (!deterministic) {
      eps = eps_inject ? eps_inject[i * A + j]
                       : normal_from_bits(random_bits_at(k0, k1, (uint64_t)(i + row_off) * A + j, (uint64_t)M_global * A, scheme));
    }
    const float raw = mean + sd * eps;
    const float t = tanhf(raw);
    const float sc = scale[j];
    act_out[i * ld + c0 + j] = t * sc;
    const float d = raw - mean;
    s_term[rl * A + j] = -(d * d) / (2.0f * sd * sd) - ls - FS_LOG_SQRT_2PI - logf((1.0f - t * t) + 1e-6f) - logf(sc + 1e-6f);
  }
  if (!logp) return;
  __syncthreads();
  if (on && j == 0) {
    float lp = 0.f;
    for (int q = 0; q < A; ++q) lp += s_term[rl * A + q];
    logp[i] = lp;
  }
}

// policy loss seeds (fastsac.py:109-127): expected values q_k = sum_j softmax(l_k)_j z_j, q = (q1 + q2) / 2 or min(q1, q2),
// loss_b = alpha log_prob_b - q_b; d loss / d l_kj = -(w_k / B) p_kj (z_j - q_k).  One wave per row; partial[block] = sum of loss_b.
__global__ __launch_bounds__(256) void k_fs_policy_seed(const float* __restrict__ l1, const float* __restrict__ l2,
                                                        const float* __restrict__ logp, const float* __restrict__ log_alpha,
                                                        float* __restrict__ d1, float* __restrict__ d2, float* __restrict__ partial,
                                                        int64_t M, int NA, float v_min, float v_max, int clipped, float inv_b) {
  __shared__ float s_loss[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + w;
  const float dz = (v_max - v_min) / (float)(NA - 1);
  float loss = 0.f;
  if (row < M) {
    float q[2], p[2][4];
    const float* lg[2] = {l1 + row * NA, l2 + row * NA};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float v[4], mx = -3.4e38f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = lane + 64 * u;
        v[u] = a < NA ? lg[k][a] : -3.4e38f;
        mx = fmaxf(mx, v[u]);
      }
      mx = wave_max(mx);
      float se = 0.f, sq = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = lane + 64 * u;
        p[k][u] = a < NA ? expf(v[u] - mx) : 0.f;
        se += p[k][u];
        sq += p[k][u] * (v_min + dz * (float)a);
      }
      se = wave_sum(se);
      sq = wave_sum(sq);
      q[k] = sq / se;
#pragma unroll
      for (int u = 0; u < 4; ++u) p[k][u] /= se;
    }
    float w1 = 0.5f, w2 = 0.5f, qv = (q[0] + q[1]) * 0.5f;
    if (clipped) {
      w1 = q[0] <= q[1] ? 1.f : 0.f;      // torch.minimum: the gradient goes to the smaller one (to the first on a tie)
      w2 = 1.f - w1;
      qv = fminf(q[0], q[1]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int a = lane + 64 * u;
      if (a < NA) {
        const float z = v_min + dz * (float)a;
        d1[row * NA + a] = -(w1 * inv_b) * p[0][u] * (z - q[0]);
        d2[row * NA + a] = -(w2 * inv_b) * p[1][u] * (z - q[1]);
      }
    }
    loss = expf(log_alpha[0]) * logp[row] - qv;
  }
  if (lane == 0) s_loss[w] = loss;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
}

// d loss / d head [M, 2A] of the policy (chain through the tanh squashing, the action scale and the tanh-mapped log-std):
// da1 + da2 = d loss / d action from the two critics' input gradients (row stride lda, first column c0); the log-prob term
// carries alpha / B.
__global__ __launch_bounds__(256) void k_fs_policy_grad(const float* __restrict__ head, const float* __restrict__ scale,
                                                        const float* __restrict__ eps_inject, uint32_t k0, uint32_t k1, int scheme,
                                                        const float* __restrict__ da1, const float* __restrict__ da2, int lda, int c0,
                                                        const float* __restrict__ log_alpha, float* __restrict__ dhead, int64_t M, int A,
                                                        float ls_min, float ls_max, float inv_b, int64_t row_off, int64_t M_global) {
  const int64_t n = M * A;
  const float ab = expf(log_alpha[0]) * inv_b;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / A;
    const int j = (int)(e - i * A);
    const float mean = head[i * 2 * A + j];
    const float th = tanhf(head[i * 2 * A + A + j]);
    const float half = 0.5f * (ls_max - ls_min);
    const float ls = ls_min + half * (th + 1.0f);
    const float sd = expf(ls);
    const float eps = eps_inject ? eps_inject[i * A + j]
                                 : normal_from_bits(random_bits_at(k0, k1, (uint64_t)(i + row_off) * A + j, (uint64_t)M_global * A, scheme));
    const float t = tanhf(mean + sd * eps);
    const float om = 1.0f - t * t;
    // d loss / d t: through the action, and through -log(1 - t^2 + 1e-6) of the log-prob
    const float dt = (da1[i * lda + c0 + j] + da2[i * lda + c0 + j]) * scale[j] + ab * (2.0f * t / (om + 1e-6f));
    const float draw = dt * om;
    // the Gaussian term -(raw - mean)^2 / (2 sd^2) equals -eps^2 / 2: no dependence on mean or log-std
    dhead[i * 2 * A + j] = draw;
    const float dls = draw * eps * sd - ab;
    dhead[i * 2 * A + A + j] = dls * half * (1.0f - th * th);
  }
}

// entropy-coefficient step (fastsac.py:226-239, entropy_coefficient.py:25-30) + the scalars of the critic step:
// entropy = -mean(next_log_prob); loss = exp(log_alpha) (entropy - target); AdamW on log_alpha.
// metrics: [0] q_loss [1] entropy_loss [2] q_min [3] q_max [4] entropy [5] (critic grad norm, by the Adam launch) [6] entropy grad norm ^ 2 [7] alpha
__global__ __launch_bounds__(256) void k_fs_alpha_step(const float* __restrict__ logp, int64_t M, float* __restrict__ log_alpha,
                                                       float* __restrict__ am, float* __restrict__ av, const float* __restrict__ c51_out,
                                                       float target_entropy, float lr, float wd, float b1, float b2, float eps, float bc1,
                                                       float bc2, float* __restrict__ metrics) {
  __shared__ float s_buf[4];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < M; i += 256) acc += logp[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float entropy = -((s_buf[0] + s_buf[1]) + (s_buf[2] + s_buf[3])) / (float)M;
    const float la = log_alpha[0], alpha = expf(la);
    const float g = alpha * (entropy - target_entropy);
    metrics[0] = c51_out[0];
    metrics[1] = g;                 // the loss has the same value as its derivative with respect to log_alpha
    metrics[2] = c51_out[1];
    metrics[3] = c51_out[2];
    metrics[4] = entropy;
    metrics[6] = g * g;
    metrics[7] = alpha;
    const float mi = b1 * am[0] + (1.f - b1) * g, vi = b2 * av[0] + (1.f - b2) * g * g;
    am[0] = mi;
    av[0] = vi;
    log_alpha[0] = la * (1.0f - lr * wd) - lr * ((mi / bc1) / (sqrtf(vi / bc2) + eps));
  }
}

__global__ __launch_bounds__(256) void k_fs_policy_metrics(const float* __restrict__ partial, int n, const float* __restrict__ log_alpha,
                                                           float inv_b, float* __restrict__ metrics) {
  __shared__ float s_buf[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    metrics[0] = ((s_buf[0] + s_buf[1]) + (s_buf[2] + s_buf[3])) * inv_b;   // policy loss
    metrics[1] = expf(log_alpha[0]);                                          // alpha used by this step
  }
}

int fs_head_fwd(const float* H, const float* W, const float* b, float* out, int64_t M, int K, int N, hipStream_t st) {
  hipLaunchKernelGGL(k_fs_head_fwd, dim3(div_up(M, 8), div_up(N, 128)), dim3(128), 0, st, H, W, b, out, M, K, N);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int fs_head_bwd(rlx_ctx* ctx, float* H_dH, const float* W, const float* d, float* gW, float* gb, int64_t M, int K, int N, hipStream_t st) {
  if (gW) {
    const int boff = (K * N + 3) & ~3;
    const int64_t PS = boff + ((N + 3) & ~3);
    const int TK = K / 8, TN = TK > 0 && TK <= 256 ? 256 / TK : 0;
    const int nj = TN ? div_up(N, TN) : 99;
    const bool tiled = K % 8 == 0 && nj <= 12 && (size_t)16 * (K + N) * sizeof(float) <= 48 * 1024;
    const int rows = tiled ? (M >= 8192 ? 32 : 64) : 128, S = div_up(M, rows);   // (>= 256 slabs: one per CU)
    float* part = stage_alloc(ctx, (size_t)S * PS);
    if (!part) return RLX_ENOMEM;
    if (tiled) {
      const size_t lds = (size_t)16 * (K + N) * sizeof(float);
#define FS_DW_TILED(NJ) hipLaunchKernelGGL(k_fs_head_dw_tiled<NJ>, dim3(S), dim3(256), lds, st, (const float*)H_dH, d, part, M, K, N, rows, TK, TN, PS, boff)
      if (nj <= 2) FS_DW_TILED(2);
      else if (nj <= 4) FS_DW_TILED(4);
      else if (nj <= 8) FS_DW_TILED(8);
      else FS_DW_TILED(12);
#undef FS_DW_TILED
    } else {
      int gx = div_up(K * N + N, 256);
      if (gx > 64) gx = 64;
      hipLaunchKernelGGL(k_fs_head_dw, dim3(gx, S), dim3(256), 0, st, (const float*)H_dH, d, part, M, K, N, rows, PS, boff);
    }
    RLX_LAUNCH_CHECK();
    ReduceTable tab;
    tab.n = 0;
    tab.seg[tab.n++] = ReduceSeg{part, gW, (int64_t)K * N, PS, S, 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{part + boff, gb, (int64_t)N, PS, S, 0, 1.f, 0.f, 1};
    const int rc = stage_reduce(ctx, tab, nullptr, nullptr, st);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_fs_head_dx, dim3(div_up(M, 8)), dim3(256), (size_t)8 * N * sizeof(float), st, d, W, H_dH, M, K, N);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

static int fs_sample(const float* head, const float* scale, const uint32_t ks[2], int scheme, const float* inject, float* act_out, int ld,
                     int c0, float* logp, int64_t M, int A, const rlx_fastsac_hparams& hp, int deterministic, int64_t row_off,
                     int64_t M_global, hipStream_t st) {
  RLX_REQUIRE(A >= 1 && A <= 256, RLX_EUNSUP, "fastsac: 1 <= act_dim <= 256");
  const int rpb = 256 / A;
  hipLaunchKernelGGL(k_fs_sample, dim3(div_up(M, rpb)), dim3(256), (size_t)rpb * A * sizeof(float), st, head, scale, ks[0], ks[1], scheme,
                     inject, act_out, ld, c0, logp, M, A, hp.log_std_min, hp.log_std_max, deterministic, row_off, M_global);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int fs_concat(const float* obs, int Oc, const float* act, int A, float* out, int ld, int64_t M, hipStream_t st) {
  int grid = div_up(M * ld, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_fs_concat, dim3(grid), dim3(256), 0, st, obs, Oc, act, A, out, ld, M);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// ReplayBuffer.sample (replay_buffer.py:34-96) for given start rows idx_t and env columns idx_e: the n-step return over the
// steps up to the first done, gamma^k discounts, the number of steps that counted, next state / done / truncation of the step
// where the window ends (first done or first truncation, else the last step); with a full ring the newest row counts as
// truncated unless it is done (:49-55).  One wave per sample: lane 0 scans the window, all lanes copy the rows.
__global__ __launch_bounds__(256) void k_fs_nstep_sample(const float* __restrict__ rs, const float* __restrict__ rns,
                                                         const float* __restrict__ ra, const float* __restrict__ rr,
                                                         const float* __restrict__ rd, const float* __restrict__ rt, int capacity,
                                                         int nr_envs, int O, int A, int n_steps, float gamma, int last_idx,
                                                         const int32_t* __restrict__ idx_t, const int32_t* __restrict__ idx_e, int64_t B,
                                                         float* __restrict__ os, float* __restrict__ ons, float* __restrict__ oa,
                                                         float* __restrict__ orw, float* __restrict__ od, float* __restrict__ otr,
                                                         float* __restrict__ on) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const int t0 = idx_t[i], e = idx_e[i];
  int ft = t0;
  if (lane == 0) {
    auto trunc_at = [&](int t) {
      const float tr = rt[(int64_t)t * nr_envs + e];
      return (n_steps > 1 && t == last_idx) ? (rd[(int64_t)t * nr_envs + e] > 0.f ? tr : 1.0f) : tr;
    };
    float rew, eff;
    if (n_steps == 1) {
      rew = rr[(int64_t)t0 * nr_envs + e];
      eff = 1.0f;
    } else {
      float mask = 1.0f;
      rew = 0.f;
      eff = 0.f;
      int first_done = n_steps - 1, first_trunc = n_steps - 1;
      bool fd = false, ftr = false;
      for (int k = 0; k < n_steps; ++k) {
        const int t = (t0 + k) % capacity;
        const float d = rd[(int64_t)t * nr_envs + e];
        rew += rr[(int64_t)t * nr_envs + e] * mask * powf(gamma, (float)k);
        eff += mask;
        if (!fd && d > 0.f) { first_done = k; fd = true; }
        if (!ftr && trunc_at(t) > 0.f) { first_trunc = k; ftr = true; }
        mask *= 1.0f - d;
      }
      ft = (t0 + (first_done < first_trunc ? first_done : first_trunc)) % capacity;
    }
    orw[i] = rew;
    on[i] = eff;
    od[i] = rd[(int64_t)ft * nr_envs + e];
    otr[i] = trunc_at(ft);
  }
  ft = __shfl(ft, 0, 64);
  const float* s0 = rs + ((int64_t)t0 * nr_envs + e) * O;
  const float* s1 = rns + ((int64_t)ft * nr_envs + e) * O;
  for (int c = lane; c < O; c += 64) {
    os[i * O + c] = s0[c];
    ons[i * O + c] = s1[c];
  }
  const float* a0 = ra + ((int64_t)t0 * nr_envs + e) * A;
  for (int c = lane; c < A; c += 64) oa[i * A + c] = a0[c];
}

static int fs_check(const rlx_lnmlp_desc& pd, const rlx_lnmlp_desc& qd, const rlx_fastsac_hparams& hp, int* A_out, int* Oc_out) {
  int rc = ln_check(pd);
  if (rc) return rc;
  rc = ln_check(qd);
  if (rc) return rc;
  RLX_REQUIRE(pd.out_dim % 2 == 0, RLX_EINVAL, "fastsac: policy out_dim must be 2 * act_dim ([mean | log_std])");
  const int A = pd.out_dim / 2;
  RLX_REQUIRE(qd.in_dim > A && qd.out_dim == hp.nr_atoms && hp.nr_atoms >= 2 && hp.nr_atoms <= 128, RLX_EINVAL,
              "fastsac: critic in_dim = critic obs + act, out_dim = nr_atoms (2..128)");
  RLX_REQUIRE(hp.v_max > hp.v_min && hp.log_std_max > hp.log_std_min, RLX_EINVAL, "fastsac: v_max > v_min and log_std_max > log_std_min");
  *A_out = A;
  *Oc_out = qd.in_dim - A;
  return RLX_OK;
}

}  // namespace rlx

using namespace rlx;

extern "C" {

int64_t rlx_lnmlp_param_count(const rlx_lnmlp_desc* d) {
  if (!d || d->n_hidden < 1 || d->n_hidden > 4) return -1;
  return ln_layout(*d).n_params;
}

int rlx_lnmlp_fwd_f32(rlx_ctx* ctx, const rlx_lnmlp_desc* d, const float* params, const float* x, int ldx, float* out, int64_t M,
                      void* stream) {
  RLX_REQUIRE(ctx && d && params && x && out && M > 0 && ldx >= (d ? d->in_dim : 1), RLX_EINVAL, "rlx_lnmlp_fwd_f32: bad args");
  int rc = ln_check(*d);
  if (rc) return rc;
  bx_release_all(ctx);
  const LnLayout L = ln_layout(*d);
  const int ldp = (d->in_dim + 3) & ~3;
  float* base = (float*)scratch(ctx, SL_SAC, (ln_buf_floats(L, M) + (size_t)M * ldp + 64) * sizeof(float));
  if (!base) return RLX_ENOMEM;
  LnBufs b;
  float* cur = base;
  ln_carve(L, M, cur, &b);
  if (ldx % 4 != 0 || ldx < ldp) {   // the GEMM stages read 16-byte pieces: rows at a pitch that is a multiple of four, zero padded
    RLX_REQUIRE(ldx == d->in_dim, RLX_EUNSUP, "rlx_lnmlp_fwd_f32: a row stride that is not a multiple of 4 must equal in_dim");
    rc = fs_concat(x, d->in_dim, nullptr, 0, cur, ldp, M, (hipStream_t)stream);
    if (rc) return rc;
    x = cur;
    ldx = ldp;
  }
  return ln_fwd(ctx, L, params, x, ldx, b, out, M, (hipStream_t)stream);
}

int rlx_fastsac_replay_sample_f32(rlx_ctx* ctx, const float* ring_states, const float* ring_next_states, const float* ring_actions,
                                  const float* ring_rewards, const float* ring_dones, const float* ring_truncations, int capacity,
                                  int nr_envs, int obs_dim, int act_dim, int n_steps, float gamma, int pos, int size,
                                  const int32_t* idx_t, const int32_t* idx_e, int64_t B, float* states, float* next_states,
                                  float* actions, float* rewards, float* dones, float* truncations, float* effective_n_steps,
                                  void* stream) {
  RLX_REQUIRE(ctx && ring_states && ring_next_states && ring_actions && ring_rewards && ring_dones && ring_truncations && idx_t && idx_e &&
                  states && next_states && actions && rewards && dones && truncations && effective_n_steps,
              RLX_EINVAL, "rlx_fastsac_replay_sample_f32: NULL pointer");
  RLX_REQUIRE(B > 0 && capacity > 0 && nr_envs > 0 && obs_dim > 0 && act_dim > 0 && n_steps >= 1 && n_steps <= capacity && size >= 1 &&
                  size <= capacity && pos >= 0 && pos < capacity,
              RLX_EINVAL, "rlx_fastsac_replay_sample_f32: bad sizes");
  const int last_idx = size >= capacity ? (pos + capacity - 1) % capacity : -1;
  hipLaunchKernelGGL(k_fs_nstep_sample, dim3(div_up(B, 4)), dim3(256), 0, (hipStream_t)stream, ring_states, ring_next_states, ring_actions,
                     ring_rewards, ring_dones, ring_truncations, capacity, nr_envs, obs_dim, act_dim, n_steps, gamma, last_idx, idx_t, idx_e, B,
                     states, next_states, actions, rewards, dones, truncations, effective_n_steps);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_fastsac_act_f32(rlx_ctx* ctx, const rlx_lnmlp_desc* pdesc, const float* pparams, const float* obs, const float* action_scale,
                        uint32_t key_io[2], int scheme, float* action, int N, int deterministic, int row_offset, int N_global,
                        const rlx_fastsac_hparams* hp, void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && obs && action_scale && key_io && action && hp && N > 0 && N_global >= N, RLX_EINVAL,
              "rlx_fastsac_act_f32: bad args");
  int rc = ln_check(*pdesc);
  if (rc) return rc;
  RLX_REQUIRE(pdesc->out_dim % 2 == 0, RLX_EINVAL, "rlx_fastsac_act_f32: policy out_dim must be 2 * act_dim");
  const int A = pdesc->out_dim / 2;
  hipStream_t st = (hipStream_t)stream;
  bx_release_all(ctx);
  const LnLayout L = ln_layout(*pdesc);
  const size_t nb = ln_buf_floats(L, N);
  const int ldp = (pdesc->in_dim + 3) & ~3;
  float* base = (float*)scratch(ctx, SL_SAC, (nb + (size_t)N * 2 * A + (size_t)N * ldp + 128) * sizeof(float));
  if (!base) return RLX_ENOMEM;
  LnBufs b;
  float* cur = base;
  ln_carve(L, N, cur, &b);
  float* head = cur;
  float* xs = head + (((size_t)N * 2 * A + 63) & ~size_t(63));
  rc = fs_concat(obs, pdesc->in_dim, nullptr, 0, xs, ldp, N, st);
  if (!rc) rc = ln_fwd(ctx, L, pparams, xs, ldp, b, head, N, st);
  if (rc) return rc;
  uint32_t ks[4] = {key_io[0], key_io[1], 0, 0};
  if (!deterministic) {
    split_host(key_io, ks, 2, scheme);      // key, subkey = split(key)
    key_io[0] = ks[0];
    key_io[1] = ks[1];
  }
  return fs_sample(head, action_scale, ks + 2, scheme, ctx->dbg_sac_eps[0], action, A, 0, nullptr, N, A, *hp, deterministic, row_offset,
                   N_global, st);
}

int rlx_fastsac_critic_update_f32(rlx_ctx* ctx, const rlx_lnmlp_desc* pdesc, const float* pparams, const rlx_lnmlp_desc* qdesc,
                                  float* qparams, float* qm, float* qv, float* qtarget, float* log_alpha, float* am, float* av,
                                  const float* states, const float* next_states, const float* critic_states,
                                  const float* critic_next_states, const float* actions, const float* rewards, const float* dones,
                                  const float* truncations, const float* effective_n_steps, const float* action_scale, int64_t B,
                                  uint32_t key_io[2], int scheme, int64_t* opt_count_io, const rlx_fastsac_hparams* hp,
                                  float* metrics_out, void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && qdesc && qparams && qm && qv && qtarget && log_alpha && am && av && states && next_states &&
                  actions && rewards && dones && truncations && effective_n_steps && action_scale && key_io && opt_count_io && hp &&
                  metrics_out && B > 0,
              RLX_EINVAL, "rlx_fastsac_critic_update_f32: bad args");
  int A, Oc;
  int rc = fs_check(*pdesc, *qdesc, *hp, &A, &Oc);
  if (rc) return rc;
  RLX_REQUIRE((critic_states != nullptr) == (critic_next_states != nullptr) && (critic_states || Oc == pdesc->in_dim), RLX_EINVAL,
              "rlx_fastsac_critic_update_f32: critic obs width != policy obs width needs critic_states AND critic_next_states");
  const float* cs = critic_states ? critic_states : states;
  const float* cn = critic_next_states ? critic_next_states : next_states;
  hipStream_t st = (hipStream_t)stream;
  bx_release_all(ctx);
  const LnLayout LP = ln_layout(*pdesc), LQ = ln_layout(*qdesc);
  const int64_t nq = LQ.n_params;
  const int NA = hp->nr_atoms, ldc = (Oc + A + 3) & ~3;
  // ---- arena: policy activations, one set for the two target passes (inference), two sets for the online critics
  auto a64 = [](size_t n) { return (n + 63) & ~size_t(63); };
  const size_t np_b = ln_buf_floats(LP, B), nq_b = ln_buf_floats(LQ, B);
  const size_t n_x = a64((size_t)B * ldc), n_log = a64((size_t)B * NA);
  const int ldp = (pdesc->in_dim + 3) & ~3;
  const size_t total = np_b + 3 * nq_b + 2 * n_x + a64((size_t)B * 2 * A) + 6 * n_log + a64(B) + a64(2 * nq) + a64((size_t)B * ldp) + 256;
  float* base = (float*)scratch(ctx, SL_SAC, total * sizeof(float));
  float* sq = (float*)scratch(ctx, SL_NORM, REDUCE_MAX_BLOCKS * sizeof(float));
  if (!base || !sq) return RLX_ENOMEM;
  float* cur = base;
  LnBufs bp, bt, b1, b2;
  ln_carve(LP, B, cur, &bp);
  ln_carve(LQ, B, cur, &bt);
  ln_carve(LQ, B, cur, &b1);
  ln_carve(LQ, B, cur, &b2);
  float* xc = cur; cur += n_x;
  float* xn = cur; cur += n_x;
  float* head = cur; cur += a64((size_t)B * 2 * A);
  float *lt1 = cur, *lt2 = cur + n_log, *l1 = cur + 2 * n_log, *l2 = cur + 3 * n_log, *d1 = cur + 4 * n_log, *d2 = cur + 5 * n_log;
  cur += 6 * n_log;
  float* lpn = cur; cur += a64(B);
  float* gq = cur; cur += a64(2 * nq);
  float* xs = cur; cur += a64((size_t)B * ldp);                 // policy observations at a 16-byte row pitch
  float* c51o = cur;                                            // 4 floats
  // key, subkey = split(key)
  uint32_t ks[4];
  split_host(key_io, ks, 2, scheme);
  key_io[0] = ks[0];
  key_io[1] = ks[1];
  struct BxAll { rlx_ctx* c; ~BxAll() { bx_release_all(c); } } bx_all{ctx};
  {
    const FsNet nets[5] = {{pparams, &LP, false}, {qtarget, &LQ, false}, {qtarget + nq, &LQ, false}, {qparams, &LQ, true}, {qparams + nq, &LQ, true}};
    rc = fs_images(ctx, nets, 5, B, st);
    if (rc) return rc;
  }
  // Two streams: the policy on s' + both target critics on (s', a') on the caller's stream, both online critics on (s, a) on the
  // side stream; after the C51 loss one critic's backward on each.  (Every pass is a chain of one-wave launches -- 8192 rows are
  // 128 row tiles -- so the halves overlap almost for free.)
  FsFork fk(ctx, st);
  rc = fk.begin();
  if (rc) return rc;
  rc = fs_concat(cs, Oc, actions, A, xc, ldc, B, st);
  if (!rc) rc = fk.fork();
  // ---- online critics on (s, a)
  if (!rc) rc = ln_fwd(ctx, LQ, qparams, xc, ldc, b1, l1, B, fk.side());
  if (!rc) rc = ln_fwd(ctx, LQ, qparams + nq, xc, ldc, b2, l2, B, fk.side());
  // ---- next action and log-prob from the policy (no gradient), target critics on (s', a')
  if (!rc) rc = fs_concat(cn, Oc, nullptr, A, xn, ldc, B, fk.main());
  if (!rc) rc = fs_concat(next_states, pdesc->in_dim, nullptr, 0, xs, ldp, B, st);
  if (!rc) rc = ln_fwd(ctx, LP, pparams, xs, ldp, bp, head, B, st);
  if (!rc) rc = fs_sample(head, action_scale, ks + 2, scheme, ctx->dbg_sac_eps[0], xn, ldc, Oc, lpn, B, A, *hp, 0, 0, B, st);
  if (!rc) rc = ln_fwd(ctx, LQ, qtarget, xn, ldc, bt, lt1, B, st);
  if (!rc) rc = ln_fwd(ctx, LQ, qtarget + nq, xn, ldc, bt, lt2, B, st);
  if (!rc) rc = fk.join();
  // ---- the C51 loss and its logit gradients
  if (!rc) rc = rlx_c51_critic_loss_f32(ctx, l1, l2, lt1, lt2, rewards, dones, truncations, effective_n_steps, lpn, log_alpha, B, NA, hp->gamma,
                                        hp->v_min, hp->v_max, hp->clipped_double_q, d1, d2, c51o, stream);
  if (rc) return rc;
  {
    GradScaleScope gscope(ctx, bx_grad_scale(B));   // d logits ~ 1 / B
    FsDefer defer(ctx);
    rc = defer.begin(2 * ln_bwd_stage_floats(ctx, LQ, B, true));
    if (!rc) rc = fk.fork();
    if (!rc) rc = ln_bwd(ctx, LQ, qparams + nq, xc, ldc, b2, d2, gq + nq, nullptr, 0, B, fk.side());
    if (!rc) rc = ln_bwd(ctx, LQ, qparams, xc, ldc, b1, d1, gq, nullptr, 0, B, fk.main());
    if (!rc) rc = fk.join();
    if (!rc) rc = stage_reduce_flush(ctx, nullptr, nullptr, st);
    if (rc) return rc;
  }
  // ---- entropy coefficient (uses alpha BEFORE its own step inside the C51 target: the launch order above), then AdamW + Polyak
  const int64_t step = *opt_count_io + 1;
  const float bc1 = (float)(1.0 - pow((double)hp->adam_b1, (double)step)), bc2 = (float)(1.0 - pow((double)hp->adam_b2, (double)step));
  hipLaunchKernelGGL(k_fs_alpha_step, dim3(1), dim3(256), 0, st, (const float*)lpn, B, log_alpha, am, av, (const float*)c51o,
                     hp->target_entropy, hp->lr_alpha, hp->weight_decay, hp->adam_b1, hp->adam_b2, hp->adam_eps, bc1, bc2, metrics_out);
  RLX_LAUNCH_CHECK();
  const int nsq = launch_sumsq_partials(gq, 2 * nq, sq, st);
  RLX_LAUNCH_CHECK();
  rc = launch_clip_adam(qparams, gq, qm, qv, 2 * nq, sq, nsq, step, hp->lr_critic, hp->max_grad_norm > 0.f ? hp->max_grad_norm : -1.f,
                        hp->adam_b1, hp->adam_b2, hp->adam_eps, metrics_out + 5, st, nullptr, nullptr, qtarget, hp->tau, hp->weight_decay,
                        /*clip_mode: torch clip_grad_norm_*/ 1);
  if (rc) return rc;
  *opt_count_io += 1;
  return RLX_OK;
}

int rlx_fastsac_policy_update_f32(rlx_ctx* ctx, const rlx_lnmlp_desc* pdesc, float* pparams, float* pm, float* pv,
                                  const rlx_lnmlp_desc* qdesc, const float* qparams, const float* log_alpha, const float* states,
                                  const float* critic_states, const float* action_scale, int64_t B, uint32_t key_io[2], int scheme,
                                  int64_t* opt_count_io, const rlx_fastsac_hparams* hp, float* metrics_out, void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && pm && pv && qdesc && qparams && log_alpha && states && action_scale && key_io && opt_count_io &&
                  hp && metrics_out && B > 0,
              RLX_EINVAL, "rlx_fastsac_policy_update_f32: bad args");
  int A, Oc;
  int rc = fs_check(*pdesc, *qdesc, *hp, &A, &Oc);
  if (rc) return rc;
  RLX_REQUIRE(critic_states || Oc == pdesc->in_dim, RLX_EINVAL,
              "rlx_fastsac_policy_update_f32: critic obs width != policy obs width needs critic_states");
  const float* cs = critic_states ? critic_states : states;
  hipStream_t st = (hipStream_t)stream;
  bx_release_all(ctx);
  const LnLayout LP = ln_layout(*pdesc), LQ = ln_layout(*qdesc);
  const int64_t np_ = LP.n_params, nq = LQ.n_params;
  const int NA = hp->nr_atoms, ldc = (Oc + A + 3) & ~3;
  auto a64 = [](size_t n) { return (n + 63) & ~size_t(63); };
  const size_t np_b = ln_buf_floats(LP, B), nq_b = ln_buf_floats(LQ, B);
  const size_t n_x = a64((size_t)B * ldc), n_log = a64((size_t)B * NA), n_hd = a64((size_t)B * 2 * A);
  const int nblk = div_up(B, 4);
  const int ldp = (pdesc->in_dim + 3) & ~3;
  const size_t total = np_b + 2 * nq_b + 3 * n_x + 2 * n_hd + 4 * n_log + a64(B) + a64(nblk) + a64(np_) + a64((size_t)B * ldp) + 256;
  float* base = (float*)scratch(ctx, SL_SAC, total * sizeof(float));
  float* sq = (float*)scratch(ctx, SL_NORM, REDUCE_MAX_BLOCKS * sizeof(float));
  if (!base || !sq) return RLX_ENOMEM;
  float* cur = base;
  LnBufs bp, b1, b2;
  ln_carve(LP, B, cur, &bp);
  ln_carve(LQ, B, cur, &b1);
  ln_carve(LQ, B, cur, &b2);
  float* xp = cur; cur += n_x;
  float* dx1 = cur; cur += n_x;
  float* dx2 = cur; cur += n_x;
  float* head = cur; cur += n_hd;
  float* dhead = cur; cur += n_hd;
  float *l1 = cur, *l2 = cur + n_log, *d1 = cur + 2 * n_log, *d2 = cur + 3 * n_log;
  cur += 4 * n_log;
  float* lp = cur; cur += a64(B);
  float* part = cur; cur += a64(nblk);
  float* gp = cur; cur += a64(np_);
  float* xs = cur;                                              // policy observations at a 16-byte row pitch
  uint32_t ks[4];
  split_host(key_io, ks, 2, scheme);
  key_io[0] = ks[0];
  key_io[1] = ks[1];
  const float inv_b = 1.0f / (float)B;
  struct BxAll { rlx_ctx* c; ~BxAll() { bx_release_all(c); } } bx_all{ctx};
  {
    const FsNet nets[3] = {{pparams, &LP, true}, {qparams, &LQ, true}, {qparams + nq, &LQ, true}};
    rc = fs_images(ctx, nets, 3, B, st);
    if (rc) return rc;
  }
  // policy on s, sampled action into the critics' input rows, both critics, seeds
  rc = fs_concat(cs, Oc, nullptr, A, xp, ldc, B, st);
  if (!rc) rc = fs_concat(states, pdesc->in_dim, nullptr, 0, xs, ldp, B, st);
  if (!rc) rc = ln_fwd(ctx, LP, pparams, xs, ldp, bp, head, B, st);
  if (!rc) rc = fs_sample(head, action_scale, ks + 2, scheme, ctx->dbg_sac_eps[1], xp, ldc, Oc, lp, B, A, *hp, 0, 0, B, st);
  FsFork fk(ctx, st);
  if (!rc) rc = fk.begin();
  if (!rc) rc = fk.fork();
  if (!rc) rc = ln_fwd(ctx, LQ, qparams + nq, xp, ldc, b2, l2, B, fk.side());
  if (!rc) rc = ln_fwd(ctx, LQ, qparams, xp, ldc, b1, l1, B, fk.main());
  if (!rc) rc = fk.join();
  if (rc) return rc;
  hipLaunchKernelGGL(k_fs_policy_seed, dim3(nblk), dim3(256), 0, st, (const float*)l1, (const float*)l2, (const float*)lp, log_alpha, d1, d2,
                     part, B, NA, hp->v_min, hp->v_max, hp->clipped_double_q, inv_b);
  RLX_LAUNCH_CHECK();
  {
    GradScaleScope gscope(ctx, bx_grad_scale(B));
    FsDefer defer(ctx);
    rc = defer.begin(2 * ln_bwd_stage_floats(ctx, LQ, B, false) + ln_bwd_stage_floats(ctx, LP, B, true));
    if (rc) return rc;
    // the critics' input gradients (no parameter gradients; one critic per stream), then the policy's backward
    rc = fk.fork();
    if (!rc) rc = ln_bwd(ctx, LQ, qparams + nq, xp, ldc, b2, d2, nullptr, dx2, ldc, B, fk.side(), Oc, A);
    if (!rc) rc = ln_bwd(ctx, LQ, qparams, xp, ldc, b1, d1, nullptr, dx1, ldc, B, fk.main(), Oc, A);
    if (!rc) rc = fk.join();
    if (rc) return rc;
    int grid = div_up(B * A, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_fs_policy_grad, dim3(grid), dim3(256), 0, st, (const float*)head, action_scale, ctx->dbg_sac_eps[1], ks[2], ks[3],
                       scheme, (const float*)dx1, (const float*)dx2, ldc, Oc, log_alpha, dhead, B, A, hp->log_std_min, hp->log_std_max, inv_b,
                       (int64_t)0, B);
    RLX_LAUNCH_CHECK();
    rc = ln_bwd(ctx, LP, pparams, xs, ldp, bp, dhead, gp, nullptr, 0, B, st);
    if (!rc) rc = stage_reduce_flush(ctx, nullptr, nullptr, st);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_fs_policy_metrics, dim3(1), dim3(256), 0, st, (const float*)part, nblk, log_alpha, inv_b, metrics_out);
  RLX_LAUNCH_CHECK();
  const int64_t step = *opt_count_io + 1;
  const int nsq = launch_sumsq_partials(gp, np_, sq, st);
  RLX_LAUNCH_CHECK();
  rc = launch_clip_adam(pparams, gp, pm, pv, np_, sq, nsq, step, hp->lr_policy, hp->max_grad_norm > 0.f ? hp->max_grad_norm : -1.f,
                        hp->adam_b1, hp->adam_b2, hp->adam_eps, metrics_out + 2, st, nullptr, nullptr, nullptr, 0.f, hp->weight_decay,
                        /*clip_mode: torch clip_grad_norm_*/ 1);
  if (rc) return rc;
  *opt_count_io += 1;
  return RLX_OK;
}

}  // extern "C"
