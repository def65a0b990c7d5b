// ppo_lstm.hip -- PPO with a recurrent (LSTM) policy for gfx950: acting step, one sequence
// minibatch (loss + BPTT gradients) and the whole update.  Replaces the XLA fusions of
//   Policy.apply_one_step / forward_sequence  rl_x/algorithms/ppo_lstm/flax_full_jit/policy.py:121-142
//   single_rollout                             rl_x/algorithms/ppo_lstm/flax_full_jit/ppo_lstm.py:134-161
//   loss_fn / minibatch_update / env-index permutation            ppo_lstm.py:181-263
// The critic is the feed-forward PPO critic (ppo_lstm/flax_full_jit/critic.py:18-33), "recurrent GAE" is
// the same GAE.  CPU twin: oracle/ppo_lstm.py (torch autograd).
//
// Decomposition (rows are time-major: row = t * n + env):
//   encoders (Dense+LN+ELU, K = obs)            k_l1  (fused VALU kernel of mlp.hip)
//   x-projection Gx = E_l @ Wi for ALL t        k_gemm_fwd (exact-fp32 MFMA)
//   recurrence over t, 32 envs per workgroup    k_lstm_seq_fwd / k_lstm_seq_bwd (lstm_kernels.h)
//   LN+ELU of h, concat, torso (LN after the 192-wide first layer), head + PPO loss
//   backward: the same GEMM kernels (dX / dW) + BPTT, every layer's slabs reduced at once
#include "dist.h"
#include "lstm_kernels.h"
#include "ppo_internal.h"
#include "gemm_bx.h"

namespace rlx {

struct LstmLayout {
  int O, A, E, H, D1, D2, D3, share, gru, film, K1;   // K1: width of the torso input (E + H, or E with FiLM)
  int64_t fm_W, fm_b;                    // FiLM: W[H, 2E] (gamma | beta kernels), b[2E]
  int64_t el_W, el_b, el_g, el_be, eo_W, eo_b, eo_g, eo_be;
  int64_t Wi, Wh, bh, ln_g, ln_be;       // LSTM: Wi[E,4H], Wh[H,4H], bh[4H]
  int64_t g_bi, g_Whrz, g_Whn, g_bhn;    // GRU:  Wi[E,3H], bi[3H], Wh_rz[H,2H], Wh_n[H,H], bhn[H]
  int64_t t1_W, t1_b, t1_g, t1_be, t2_W, t2_b, t3_W, t3_b, hd_W, hd_b, logstd, n_params;
};

static LstmLayout lstm_layout(const rlx_lstm_policy_desc& d) {
  LstmLayout L{};
  L.O = d.obs_dim; L.A = d.act_dim; L.E = d.enc_dim; L.H = d.lstm_hidden;
  L.D1 = d.torso[0]; L.D2 = d.torso[1]; L.D3 = d.torso[2]; L.share = d.share_encoder;
  L.gru = d.cell == RLX_CELL_GRU;
  L.film = d.combine == RLX_COMBINE_FILM;
  L.K1 = L.film ? L.E : L.E + L.H;
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off += n; return o; };
  L.el_W = take((int64_t)L.O * L.E); L.el_b = take(L.E); L.el_g = take(L.E); L.el_be = take(L.E);
  if (!L.share) { L.eo_W = take((int64_t)L.O * L.E); L.eo_b = take(L.E); L.eo_g = take(L.E); L.eo_be = take(L.E); }
  else { L.eo_W = L.el_W; L.eo_b = L.el_b; L.eo_g = L.el_g; L.eo_be = L.el_be; }
  if (!L.gru) {
    L.Wi = take((int64_t)L.E * 4 * L.H); L.Wh = take((int64_t)L.H * 4 * L.H); L.bh = take(4 * L.H);
  } else {
    L.Wi = take((int64_t)L.E * 3 * L.H); L.g_bi = take(3 * L.H); L.g_Whrz = take((int64_t)L.H * 2 * L.H);
    L.g_Whn = take((int64_t)L.H * L.H); L.g_bhn = take(L.H);
  }
  L.ln_g = take(L.H); L.ln_be = take(L.H);
  L.fm_W = L.fm_b = -1;
  if (L.film) { L.fm_W = take((int64_t)L.H * 2 * L.E); L.fm_b = take(2 * L.E); }
  L.t1_W = take((int64_t)L.K1 * L.D1); L.t1_b = take(L.D1); L.t1_g = take(L.D1); L.t1_be = take(L.D1);
  L.t2_W = take((int64_t)L.D1 * L.D2); L.t2_b = take(L.D2);
  L.t3_W = take((int64_t)L.D2 * L.D3); L.t3_b = take(L.D3);
  L.hd_W = take((int64_t)L.D3 * L.A); L.hd_b = take(L.A);
  L.logstd = take(L.A);
  L.n_params = off;
  return L;
}

static int check_lstm_desc(const rlx_lstm_policy_desc& d) {
  RLX_REQUIRE(d.obs_dim >= 1 && d.obs_dim <= 32, RLX_EUNSUP, "ppo_lstm: obs_dim must be 1..32 (encoders use the small-K fused layer)");
  RLX_REQUIRE(d.lstm_hidden == LSTM_H, RLX_EUNSUP, "ppo_lstm: lstm_hidden_dim must be 64 in this build");
  RLX_REQUIRE(d.cell == RLX_CELL_LSTM || d.cell == RLX_CELL_GRU, RLX_EINVAL, "ppo_lstm: cell must be RLX_CELL_LSTM or RLX_CELL_GRU");
  RLX_REQUIRE(d.combine == RLX_COMBINE_CONCAT || d.combine == RLX_COMBINE_FILM, RLX_EINVAL,
              "ppo_lstm: combine must be RLX_COMBINE_CONCAT or RLX_COMBINE_FILM");
  RLX_REQUIRE(d.enc_dim % 64 == 0 && d.enc_dim >= 64 && d.enc_dim <= 512, RLX_EUNSUP, "ppo_lstm: obs_encoding_dim must be a multiple of 64");
  RLX_REQUIRE(d.torso[0] % 64 == 0 && d.torso[0] <= 512 && d.torso[1] % 4 == 0 && d.torso[2] % 4 == 0 && d.torso[2] >= 4, RLX_EUNSUP,
              "ppo_lstm: torso widths unsupported");
  RLX_REQUIRE(d.act_dim >= 1 && d.act_dim <= 64, RLX_EUNSUP, "ppo_lstm: act_dim must be 1..64");
  RLX_REQUIRE((d.enc_dim + d.lstm_hidden) % 4 == 0, RLX_EUNSUP, "ppo_lstm: enc_dim + lstm_hidden must be a multiple of 4");
  return RLX_OK;
}

struct LstmBufs {
  float *El, *Eo, *GA, *hout, *cout, *hin, *cin, *Lat, *Xc, *Z1, *H1, *H2, *H3, *done, *c0, *h0;
  float* GB;                // FiLM: [gamma | beta] [M, 2E] (backward: their gradients)
  float *GX, *dGRZ, *dHN;   // GRU: x-projection [M,3H] (backward: d x-projection), (dr_pre|dz_pre) [M,2H], d hnp [M,H]
  float *D2, *D1, *DX;      // backward of the torso, OUT OF PLACE: dZ2 [M,D2], dH1 -> dZ1 [M,D1], d(concat input) [M,E+H]
  int32_t* idx_flat;
};

static int lstm_bufs(rlx_ctx* ctx, const LstmLayout& L, int64_t M, int64_t ne, LstmBufs* b) {
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 63) & ~size_t(63); return o; };
  const size_t oEl = take(M * L.E), oEo = take(M * L.E), oGA = take(M * 4 * L.H), oho = take(M * L.H), oco = take(M * L.H),
               ohi = take(M * L.H), oci = take(M * L.H), oLat = take(M * L.H), oXc = take(M * (L.E + L.H)),
               oZ1 = take(M * L.D1), oH1 = take(M * L.D1), oH2 = take(M * L.D2), oH3 = take(M * L.D3), odn = take(M),
               oc0 = take(ne * L.H), oh0 = take(ne * L.H), oGX = take(L.gru ? M * 3 * L.H : 0),
               oRZ = take(L.gru ? M * 2 * L.H : 0), oHN = take(L.gru ? M * L.H : 0), oGB = take(L.film ? M * 2 * L.E : 0),
               oD2 = take(M * L.D2), oD1 = take(M * L.D1), oDX = take(M * (L.E + L.H));
  float* base = (float*)scratch(ctx, SL_LSTM, off * sizeof(float));
  b->idx_flat = (int32_t*)scratch(ctx, SL_LSTM_IDX, (size_t)M * sizeof(int32_t));
  if (!base || !b->idx_flat) return RLX_ENOMEM;
  b->El = base + oEl; b->Eo = L.share ? b->El : base + oEo; b->GA = base + oGA; b->hout = base + oho; b->cout = base + oco;
  b->hin = base + ohi; b->cin = base + oci; b->Lat = base + oLat; b->Xc = base + oXc; b->Z1 = base + oZ1;
  b->H1 = base + oH1; b->H2 = base + oH2; b->H3 = base + oH3; b->done = base + odn; b->c0 = base + oc0; b->h0 = base + oh0;
  b->GX = base + oGX; b->dGRZ = base + oRZ; b->dHN = base + oHN; b->GB = base + oGB;
  b->D2 = base + oD2; b->D1 = base + oD1; b->DX = base + oDX;
  return RLX_OK;
}

__global__ void k_add_inplace(float* __restrict__ a, const float* __restrict__ b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] += b[i];
}

__global__ void k_mask_carry(float* __restrict__ c, float* __restrict__ h, const float* __restrict__ term,
                             const float* __restrict__ trunc, float* __restrict__ done_out, int N, int H) {
  const int64_t total = (int64_t)N * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / H);
    const float d = (term[r] != 0.f || (trunc && trunc[r] != 0.f)) ? 1.f : 0.f;
    c[i] *= 1.f - d;
    h[i] *= 1.f - d;
    if (done_out && i % H == 0) done_out[r] = d;
  }
}

// FiLM (policy.py:97-100): x = obs_latent * gamma + beta with GB = [gamma | beta]
__global__ void k_film_fwd(const float* __restrict__ Eo, const float* __restrict__ GB, float* __restrict__ X, int64_t M, int E) {
  const int64_t total = M * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / E;
    const int e = (int)(i - r * E);
    X[i] = fmaf(Eo[i], GB[r * 2 * E + e], GB[r * 2 * E + E + e]);
  }
}
// dX [M, E] -> d obs_latent = dX * gamma (dEo, may alias Eo), GB <- [dX * obs_latent | dX]
__global__ void k_film_bwd(const float* __restrict__ dX, const float* __restrict__ Eo, float* __restrict__ GB,
                           float* __restrict__ dEo, int64_t M, int E) {
  const int64_t total = M * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / E;
    const int e = (int)(i - r * E);
    const float d = dX[i], eo = Eo[i], gam = GB[r * 2 * E + e];
    dEo[i] = d * gam;
    GB[r * 2 * E + e] = d * eo;
    GB[r * 2 * E + E + e] = d;
  }
}

static inline int ew_grid(int64_t n) {
  int g = div_up(n, 256);
  return g > 4096 ? 4096 : (g < 1 ? 1 : g);
}

// policy forward for T x n rows (time-major).  On exit b.H3 = last torso activation [M, D3].
static int lstm_policy_fwd(rlx_ctx* ctx, const LstmLayout& L, const float* p, const float* obs, const LstmBufs& b, int T,
                           int n, float* cT, float* hT, int mask_final, hipStream_t st, bool stop_at_cell = false) {
  const int64_t M = (int64_t)T * n;
  const int E = L.E, H = L.H;
  int rc;
  if (!L.share) {   // both observation encoders in one launch (same rows, same shape)
    rc = stage_l1_fwd2(ctx, obs, p + L.el_W, p + L.el_b, p + L.el_g, p + L.el_be, b.El, p + L.eo_W, p + L.eo_b, p + L.eo_g,
                       p + L.eo_be, b.Eo, M, L.O, E, RLX_ACT_ELU, 1, st);
    if (rc) return rc;
  } else {
    rc = stage_l1_fwd(ctx, obs, p + L.el_W, p + L.el_b, p + L.el_g, p + L.el_be, b.El, M, L.O, E, RLX_ACT_ELU, 1, st);
    if (rc) return rc;
  }
  if (L.gru) {
    // Gx = E_l @ Wi + bi (flax GRUCell: biased input projections), then the recurrence
    rc = launch_gemm_fwd(ctx, b.El, p + L.Wi, p + L.g_bi, b.GX, M, 3 * H, E, RLX_ACT_NONE, st, 0);
    if (rc) return rc;
    if (n % LSTM_ROWS == 0)
      hipLaunchKernelGGL(k_gru_seq_fwd<true>, dim3(n / LSTM_ROWS), dim3(256), 0, st, b.GX, b.GA, p + L.g_Whrz, p + L.g_Whn,
                         p + L.g_bhn, b.h0, b.done, b.hout, b.hin, hT, T, n, mask_final);
    else
      hipLaunchKernelGGL(k_gru_seq_fwd<false>, dim3(div_up(n, LSTM_ROWS)), dim3(256), 0, st, b.GX, b.GA, p + L.g_Whrz,
                         p + L.g_Whn, p + L.g_bhn, b.h0, b.done, b.hout, b.hin, hT, T, n, mask_final);
    RLX_LAUNCH_CHECK();
  } else {
  // Gx = E_l @ Wi (no bias: flax OptimizedLSTMCell puts the bias on the recurrent kernels) -- bias pointer = zeros
  const float* zeros = zeros_f32(ctx, 4 * LSTM_H);
  if (!zeros) return RLX_ENOMEM;
  rc = launch_gemm_fwd(ctx, b.El, p + L.Wi, zeros, b.GA, M, 4 * H, E, RLX_ACT_NONE, st, 0);
  if (rc) return rc;
  {
    // the recurrent product on the fp16 pipe with split operands (lstm_kernels.h) unless the exact-fp32 engine is selected
    const bool bf = ctx->gemm_bx && !(ctx->bx_debug & 256);
#define RLX_LSTM_FWD(FULLV, BFV, GRID)                                                                                   \
  hipLaunchKernelGGL((k_lstm_seq_fwd<FULLV, BFV>), dim3(GRID), dim3(256), 0, st, b.GA, p + L.Wh, p + L.bh, b.c0, b.h0, \
                     b.done, b.hout, b.cout, b.hin, b.cin, cT, hT, T, n, mask_final)
    if (n % LSTM_ROWS == 0) {
      if (bf) RLX_LSTM_FWD(true, true, n / LSTM_ROWS);
      else RLX_LSTM_FWD(true, false, n / LSTM_ROWS);
    } else {
      if (bf) RLX_LSTM_FWD(false, true, div_up(n, LSTM_ROWS));
      else RLX_LSTM_FWD(false, false, div_up(n, LSTM_ROWS));
    }
#undef RLX_LSTM_FWD
    RLX_LAUNCH_CHECK();
  }
  }
  if (stop_at_cell) return RLX_OK;   // acting: latent LayerNorm, concat, torso, head and sampling run in one fused launch
  {
    int grid = div_up(M, 4);
    if (grid > ctx->num_cus * 8) grid = ctx->num_cus * 8;
    hipLaunchKernelGGL(k_ln_act<false>, dim3(grid), dim3(256), 0, st, b.hout, b.Lat, p + L.ln_g, p + L.ln_be, (float*)nullptr, M, H,
                       RLX_ACT_ELU);
    RLX_LAUNCH_CHECK();
  }
  if (L.film) {
    rc = launch_gemm_fwd(ctx, b.Lat, p + L.fm_W, p + L.fm_b, b.GB, M, 2 * E, H, RLX_ACT_NONE, st, 0);   // [gamma | beta]
    if (rc) return rc;
    hipLaunchKernelGGL(k_film_fwd, dim3(ew_grid(M * E)), dim3(256), 0, st, b.Eo, b.GB, b.Xc, M, E);
  } else {
    hipLaunchKernelGGL(k_concat2, dim3(ew_grid(M * (E + H))), dim3(256), 0, st, b.Eo, b.Lat, b.Xc, M, E, H);
  }
  RLX_LAUNCH_CHECK();
  rc = launch_gemm_fwd(ctx, b.Xc, p + L.t1_W, p + L.t1_b, b.Z1, M, L.D1, L.K1, RLX_ACT_NONE, st, 0);
  if (rc) return rc;
  {
    int grid = div_up(M, 4);
    if (grid > ctx->num_cus * 8) grid = ctx->num_cus * 8;
    hipLaunchKernelGGL(k_ln_act<false>, dim3(grid), dim3(256), 0, st, b.Z1, b.H1, p + L.t1_g, p + L.t1_be, (float*)nullptr, M, L.D1,
                       RLX_ACT_ELU);
    RLX_LAUNCH_CHECK();
  }
  rc = launch_gemm_fwd(ctx, b.H1, p + L.t2_W, p + L.t2_b, b.H2, M, L.D2, L.D1, RLX_ACT_ELU, st, 0);
  if (rc) return rc;
  return launch_gemm_fwd(ctx, b.H2, p + L.t3_W, p + L.t3_b, b.H3, M, L.D3, L.D2, RLX_ACT_ELU, st, 0);
}

// policy backward; b.H3 holds dZ3 on entry (head kernel).  Gradients land in g (flat, policy layout).
// The torso's backward is OUT OF PLACE: the input-gradient chain dZ3 -> D2 -> D1 -> DX leaves the forward activations H2, H1,
// Xc untouched, so the three weight gradients dW_l = H_(l-1)^T dZ_l no longer have to run in front of the kernel that used to
// overwrite H_(l-1).  With sw != st they go to the second stream (the critic's: its chain finishes long before) and run
// UNDER the BPTT kernel, which occupies n / 16 workgroups for ~300 us (they are held back until it starts: issued next to the
// input-gradient chain they only slowed it down); the chain reaches the recurrence three GEMM launches earlier.  (Round 3 tried the same overlap with the in-place kernels and 128 MB of activation copies:
// slower.  Without the copies it is a pure win -- 288 GB of HBM make the three extra buffers free.)
struct TorsoDw {
  hipStream_t sw = nullptr;                                  // stream of the weight gradients (nullptr: same stream, same order)
  hipEvent_t e_ready = nullptr, e_done = nullptr;            // recorded in front of the BPTT kernel / behind the last weight gradient
};

static int lstm_policy_bwd(rlx_ctx* ctx, const LstmLayout& L, const float* p, float* g, const float* obs, const LstmBufs& b,
                           int T, int n, float* sumsq, int* nsq, hipStream_t st, const TorsoDw& td = TorsoDw()) {
  const int64_t M = (int64_t)T * n;
  const int E = L.E, H = L.H;
  int rc;
  const bool side = td.sw != nullptr && td.sw != st;
  hipStream_t sw = td.sw;
  // the three weight gradients of the torso: in line (one stream), or all of them behind e_ready on the second stream
  auto torso_dw = [&](hipStream_t s_) -> int {
    int r = stage_dw(ctx, b.H2, L.D2, b.H3, M, L.D2, L.D3, g + L.t3_W, g + L.t3_b, sumsq, nsq, s_);
    if (!r) r = stage_dw(ctx, b.H1, L.D1, b.D2, M, L.D1, L.D2, g + L.t2_W, g + L.t2_b, sumsq, nsq, s_);
    if (!r) r = stage_dw(ctx, b.Xc, L.K1, b.D1, M, L.K1, L.D1, g + L.t1_W, g + L.t1_b, sumsq, nsq, s_);
    return r;
  };
  // torso 3, 2 (input gradients only; see above)
  rc = stage_dx(ctx, b.H3, p + L.t3_W, b.D2, M, L.D3, L.D2, L.D2, RLX_ACT_ELU, 1, st, b.H2); if (rc) return rc;
  rc = stage_dx(ctx, b.D2, p + L.t2_W, b.D1, M, L.D2, L.D1, L.D1, RLX_ACT_ELU, 0, st); if (rc) return rc;
  // torso 1: LayerNorm + ELU backward (D1 = dH1 -> dZ1), then dW1 and the gradient of the concat input
  {
    int grid = div_up(M, 4);
    if (grid > ctx->num_cus * 4) grid = ctx->num_cus * 4;
    float* part = stage_alloc(ctx, (size_t)grid * 2 * L.D1);
    if (!part) return RLX_ENOMEM;
    hipLaunchKernelGGL(k_ln_act<true>, dim3(grid), dim3(256), (size_t)8 * L.D1 * sizeof(float), st, b.Z1, b.D1, p + L.t1_g,
                       p + L.t1_be, part, M, L.D1, RLX_ACT_ELU);
    RLX_LAUNCH_CHECK();
    ReduceTable tab;
    tab.n = 0;
    tab.seg[tab.n++] = ReduceSeg{part, g + L.t1_g, (int64_t)L.D1, (int64_t)2 * L.D1, grid, 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{part + L.D1, g + L.t1_be, (int64_t)L.D1, (int64_t)2 * L.D1, grid, 0, 1.f, 0.f, 1};
    rc = stage_reduce(ctx, tab, sumsq, nsq, st);
    if (rc) return rc;
  }
  rc = stage_dx(ctx, b.D1, p + L.t1_W, b.DX, M, L.D1, L.K1, L.K1, RLX_ACT_NONE, 0, st); if (rc) return rc;
  if (!side) { rc = torso_dw(st); if (rc) return rc; }
  // d[obs_latent], d[cell latent]; with a shared encoder dE_o is added to dE_l further down
  float* dEo = L.share ? b.Z1 : b.Eo;  // Z1 is free now ([M, D1] >= [M, E])
  if (L.film) {
    // x = obs_latent * gamma + beta: d obs_latent = dx * gamma; d gamma = dx * obs_latent; d beta = dx; then the two Dense
    // layers on the cell latent as ONE [H, 2E] GEMM pair (weight gradient, input gradient -> d cell latent)
    hipLaunchKernelGGL(k_film_bwd, dim3(ew_grid(M * E)), dim3(256), 0, st, b.DX, b.Eo, b.GB, dEo, M, E);
    RLX_LAUNCH_CHECK();
    rc = stage_dw(ctx, b.Lat, H, b.GB, M, H, 2 * E, g + L.fm_W, g + L.fm_b, sumsq, nsq, st); if (rc) return rc;
    rc = stage_dx(ctx, b.GB, p + L.fm_W, b.Lat, M, 2 * E, H, H, RLX_ACT_NONE, 0, st); if (rc) return rc;
  } else {
    hipLaunchKernelGGL(k_split2, dim3(ew_grid(M * (E + H))), dim3(256), 0, st, b.DX, E + H, dEo, b.Lat, M, E, H);
    RLX_LAUNCH_CHECK();
  }
  // LN + ELU on the LSTM output: Lat = dLat -> dh_ext
  {
    int grid = div_up(M, 4);
    if (grid > ctx->num_cus * 4) grid = ctx->num_cus * 4;
    float* part = stage_alloc(ctx, (size_t)grid * 2 * H);
    if (!part) return RLX_ENOMEM;
    hipLaunchKernelGGL(k_ln_act<true>, dim3(grid), dim3(256), (size_t)8 * H * sizeof(float), st, b.hout, b.Lat, p + L.ln_g,
                       p + L.ln_be, part, M, H, RLX_ACT_ELU);
    RLX_LAUNCH_CHECK();
    ReduceTable tab;
    tab.n = 0;
    tab.seg[tab.n++] = ReduceSeg{part, g + L.ln_g, (int64_t)H, (int64_t)2 * H, grid, 0, 1.f, 0.f, 1};
    tab.seg[tab.n++] = ReduceSeg{part + H, g + L.ln_be, (int64_t)H, (int64_t)2 * H, grid, 0, 1.f, 0.f, 1};
    rc = stage_reduce(ctx, tab, sumsq, nsq, st);
    if (rc) return rc;
  }
  if (side) {
    // everything the torso's weight gradients read is final; they are issued now, on the other stream, and run while the
    // recurrence below holds n / 16 workgroups for T dependent steps
    RLX_HIP_TRY(hipEventRecord(td.e_ready, st));
    RLX_HIP_TRY(hipStreamWaitEvent(sw, td.e_ready, 0));
    rc = torso_dw(sw); if (rc) return rc;
    RLX_HIP_TRY(hipEventRecord(td.e_done, sw));
  }
  if (L.gru) {
    // BPTT of the GRU, then dWh_rz / dWh_n / dbhn from the carry fed to each step, dWi / dbi and dE_l from d x-projection
    if (n % LSTM_ROWS == 0)
      hipLaunchKernelGGL(k_gru_seq_bwd<true>, dim3(n / LSTM_ROWS), dim3(256), 0, st, b.GA, p + L.g_Whrz, p + L.g_Whn, b.hin,
                         b.done, b.Lat, b.GX, b.dGRZ, b.dHN, T, n);
    else
      hipLaunchKernelGGL(k_gru_seq_bwd<false>, dim3(div_up(n, LSTM_ROWS)), dim3(256), 0, st, b.GA, p + L.g_Whrz, p + L.g_Whn,
                         b.hin, b.done, b.Lat, b.GX, b.dGRZ, b.dHN, T, n);
    RLX_LAUNCH_CHECK();
    rc = stage_dw(ctx, b.hin, H, b.dGRZ, M, H, 2 * H, g + L.g_Whrz, nullptr, sumsq, nsq, st); if (rc) return rc;
    rc = stage_dw(ctx, b.hin, H, b.dHN, M, H, H, g + L.g_Whn, g + L.g_bhn, sumsq, nsq, st); if (rc) return rc;
    rc = stage_dw(ctx, b.El, E, b.GX, M, E, 3 * H, g + L.Wi, g + L.g_bi, sumsq, nsq, st); if (rc) return rc;
    rc = stage_dx(ctx, b.GX, p + L.Wi, b.El, M, 3 * H, E, E, RLX_ACT_NONE, 0, st); if (rc) return rc;
  } else {
  // BPTT: GA (activated gates) -> dG (pre-activation gate gradients)
  {
    // the recurrent product dG @ Wh^T on the fp16 pipe with split operands unless the exact-fp32 engine is selected
    const bool bf = ctx->gemm_bx && !(ctx->bx_debug & 256);
    const float gs = ctx->bx_gscale;
#define RLX_LSTM_BWD(FULLV, BFV, GRID)                                                                                   \
  hipLaunchKernelGGL((k_lstm_seq_bwd<FULLV, BFV>), dim3(GRID), dim3(256), 0, st, b.GA, p + L.Wh, b.cout, b.cin, b.done, \
                     b.Lat, T, n, gs)
    if (n % LSTM_ROWS == 0) {
      if (bf) RLX_LSTM_BWD(true, true, n / LSTM_ROWS);
      else RLX_LSTM_BWD(true, false, n / LSTM_ROWS);
    } else {
      if (bf) RLX_LSTM_BWD(false, true, div_up(n, LSTM_ROWS));
      else RLX_LSTM_BWD(false, false, div_up(n, LSTM_ROWS));
    }
#undef RLX_LSTM_BWD
    RLX_LAUNCH_CHECK();
  }
  rc = stage_dw(ctx, b.hin, H, b.GA, M, H, 4 * H, g + L.Wh, g + L.bh, sumsq, nsq, st); if (rc) return rc;   // dWh, dbh
  rc = stage_dw(ctx, b.El, E, b.GA, M, E, 4 * H, g + L.Wi, nullptr, sumsq, nsq, st); if (rc) return rc;     // dWi
  rc = stage_dx(ctx, b.GA, p + L.Wi, b.El, M, 4 * H, E, E, RLX_ACT_NONE, 0, st); if (rc) return rc;         // dE_l
  }
  if (L.share) {  // one encoder feeds both branches: its output gradient is the sum
    hipLaunchKernelGGL(k_add_inplace, dim3(ew_grid(M * E)), dim3(256), 0, st, b.El, dEo, M * E);
    RLX_LAUNCH_CHECK();
    return stage_l1_bwd(ctx, obs, p + L.el_W, p + L.el_b, p + L.el_g, p + L.el_be, b.El, M, L.O, E, RLX_ACT_ELU, 1, g + L.el_W,
                        g + L.el_b, g + L.el_g, g + L.el_be, sumsq, nsq, st);
  }
  if (ctx->defer)    // (two live partial sets: needs the deferred-reduction arena)
    return stage_l1_bwd2(ctx, obs, p + L.el_W, p + L.el_b, p + L.el_g, p + L.el_be, b.El, p + L.eo_W, p + L.eo_b, p + L.eo_g,
                         p + L.eo_be, b.Eo, M, L.O, E, RLX_ACT_ELU, 1, g + L.el_W, g + L.el_b, g + L.el_g, g + L.el_be, g + L.eo_W,
                         g + L.eo_b, g + L.eo_g, g + L.eo_be, sumsq, nsq, st);
  rc = stage_l1_bwd(ctx, obs, p + L.el_W, p + L.el_b, p + L.el_g, p + L.el_be, b.El, M, L.O, E, RLX_ACT_ELU, 1, g + L.el_W,
                    g + L.el_b, g + L.el_g, g + L.el_be, sumsq, nsq, st);
  if (rc) return rc;
  return stage_l1_bwd(ctx, obs, p + L.eo_W, p + L.eo_b, p + L.eo_g, p + L.eo_be, b.Eo, M, L.O, E, RLX_ACT_ELU, 1, g + L.eo_W,
                      g + L.eo_b, g + L.eo_g, g + L.eo_be, sumsq, nsq, st);
}

}  // namespace rlx

using namespace rlx;

extern "C" {

int64_t rlx_lstm_policy_param_count(const rlx_lstm_policy_desc* desc) {
  if (!desc) return -1;
  return lstm_layout(*desc).n_params;
}

// see include/rlx_hip.h: weight images of the fused decoder's hidden layers (policy torso layers 2 and 3, critic layers 1
// and 2) for the acting steps of one rollout
int rlx_ppo_lstm_rollout_begin(rlx_ctx* ctx, const rlx_lstm_policy_desc* desc, const float* pparams, const rlx_mlp_desc* cdesc,
                               const float* cparams, void* stream) {
  RLX_REQUIRE(ctx && desc && pparams && cdesc && cparams, RLX_EINVAL, "rlx_ppo_lstm_rollout_begin: NULL pointer");
  ctx->ro_img = rlx_ctx::RoImages();
  if (!ctx->gemm_bx || !ctx->fused_recurrent_act) return RLX_OK;
  int rc = check_lstm_desc(*desc);
  if (rc) return rc;
  const LstmLayout L = lstm_layout(*desc);
  const MlpLayout LC = make_layout(*cdesc);
  struct Mat { const float* W; int K, N; } mats[2][3] = {};
  mats[0][1] = {pparams + L.t2_W, L.D1, L.D2};
  mats[0][2] = {pparams + L.t3_W, L.D2, L.D3};
  for (int l = 1; l < cdesc->n_hidden && l < 3; ++l) mats[1][l] = {cparams + LC.layer[l].W, LC.layer[l].in, LC.layer[l].out};
  BxJobs jobs;
  jobs.n = 0;
  int blocks = 0;
  int64_t entries = 0, off[2][3] = {};
  for (int n = 0; n < 2; ++n)
    for (int l = 1; l < 3; ++l) {
      const Mat& m = mats[n][l];
      if (!m.W || m.K % 64 != 0 || (m.N != 128 && m.N != 256)) continue;
      BxJob& j = jobs.job[jobs.n++];
      j.W = m.W; j.ldw = m.N; j.K = m.K; j.N = m.N; j.trans = 0;
      j.KB = 2 * div_up(m.K, X_BK); j.NT = 4 * div_up(m.N, G_BN);
      j.first_block = blocks;
      off[n][l] = entries;
      ctx->ro_img.nt[n][l] = j.NT;
      blocks += div_up(j.KB * j.NT * 64, 256);
      entries += (int64_t)j.KB * j.NT * X_NP * 64;
    }
  if (jobs.n == 0) return RLX_OK;
  u32x4* arena = (u32x4*)scratch(ctx, SL_WFRAG_RO, (size_t)entries * sizeof(u32x4));
  if (!arena) return RLX_ENOMEM;
  int q = 0;
  for (int n = 0; n < 2; ++n)
    for (int l = 1; l < 3; ++l)
      if (ctx->ro_img.nt[n][l]) {
        jobs.job[q++].out = arena + off[n][l];
        ctx->ro_img.img[n][l] = arena + off[n][l];
      }
  bx_launch_wfrag(jobs, blocks, (hipStream_t)stream);
  RLX_LAUNCH_CHECK();
  ctx->ro_img.params[0] = pparams;
  ctx->ro_img.params[1] = cparams;
  ctx->ro_img.valid = true;
  return RLX_OK;
}

int rlx_ppo_lstm_act_f32(rlx_ctx* ctx, const rlx_lstm_policy_desc* desc, const float* pparams, const rlx_mlp_desc* cdesc,
                         const float* cparams, const float* obs, const float* critic_obs, float* c_io, float* h_io,
                         uint32_t key_io[2], int scheme,
                         float* action, float* processed, float* value, float* logp, int N, int clip_and_rescale,
                         const float* act_low, const float* act_high, int noise_row_offset, int N_global, int deterministic,
                         void* stream) {
  RLX_REQUIRE(ctx && desc && pparams && cdesc && cparams && obs && c_io && h_io && key_io && action && value && logp, RLX_EINVAL,
              "rlx_ppo_lstm_act_f32: NULL pointer");
  RLX_REQUIRE(N > 0 && N_global >= N, RLX_EINVAL, "rlx_ppo_lstm_act_f32: bad sizes");
  int rc = check_lstm_desc(*desc);
  if (rc) return rc;
  // the critic's own observation columns (critic_observation_indices, ppo_lstm/flax_full_jit/critic.py:12,23), width cdesc->in_dim
  RLX_REQUIRE(critic_obs || cdesc->in_dim == desc->obs_dim, RLX_EINVAL,
              "rlx_ppo_lstm_act_f32: critic in_dim != policy obs_dim needs critic_obs");
  const float* cobs = critic_obs ? critic_obs : obs;
  hipStream_t st = (hipStream_t)stream;
  const LstmLayout L = lstm_layout(*desc);
  LstmBufs b;
  rc = lstm_bufs(ctx, L, N, N, &b);
  if (rc) return rc;
  // (T = 1 with mask_final = 0 never uses the done flags: no need to clear them)
  b.c0 = c_io;
  b.h0 = h_io;
  uint32_t ks[4] = {0, 0, 0, 0};
  if (!deterministic) {
    split_host(key_io, ks, 2, scheme);
    key_io[0] = ks[0];
    key_io[1] = ks[1];
  }
  RolloutDecoder dec;
  dec.params = pparams; dec.x = b.Eo; dec.xb = b.hout; dec.xb_g = L.ln_g; dec.xb_be = L.ln_be; dec.K0 = L.E + L.H;
  dec.hidden[0] = L.D1; dec.hidden[1] = L.D2; dec.hidden[2] = L.D3; dec.out_dim = L.A; dec.act = RLX_ACT_ELU;
  dec.W[0] = L.t1_W; dec.W[1] = L.t2_W; dec.W[2] = L.t3_W; dec.b[0] = L.t1_b; dec.b[1] = L.t2_b; dec.b[2] = L.t3_b;
  dec.g0 = L.t1_g; dec.be0 = L.t1_be; dec.headW = L.hd_W; dec.headb = L.hd_b; dec.logstd = L.logstd;
  if (ctx->fused_recurrent_act && !L.film && rollout_decoder_supported(dec, *cdesc)) {
    // encoders + recurrent cell, then ONE launch for latent LayerNorm + torso + head + sampling (policy) and the critic
    rc = lstm_policy_fwd(ctx, L, pparams, obs, b, 1, N, c_io, h_io, 0, st, true);
    if (rc) return rc;
    return launch_rollout_decoder(ctx, dec, *cdesc, cparams, cobs, cdesc->in_dim, ks[2], ks[3], scheme, action, processed, value, logp, N,
                                  clip_and_rescale, act_low, act_high, noise_row_offset, N_global, deterministic, st);
  }
  // generic path: the critic is independent of the recurrent policy and runs on the side stream (scratch bank 1) under it
  hipStream_t st_c = st;
  if (ctx->two_streams) {
    rc = ctx_side_stream(ctx);
    if (rc) return rc;
    st_c = ctx->side;
    RLX_HIP_TRY(hipEventRecord(ctx->ev_fork, st));
    RLX_HIP_TRY(hipStreamWaitEvent(st_c, ctx->ev_fork, 0));
    ctx->bank = 1;
    rc = rlx_mlp_fwd_f32(ctx, cdesc, cparams, cobs, value, N, st_c);
    ctx->bank = 0;
    if (rc) return rc;
    RLX_HIP_TRY(hipEventRecord(ctx->ev_join, st_c));
  }
  rc = lstm_policy_fwd(ctx, L, pparams, obs, b, 1, N, c_io, h_io, 0, st);
  if (rc) return rc;
  float* mean = (float*)scratch(ctx, SL_MEAN, (size_t)N * L.A * sizeof(float));
  if (!mean) return RLX_ENOMEM;
  rc = launch_head_fwd(b.H3, pparams + L.hd_W, pparams + L.hd_b, mean, N, L.D3, L.A, st);
  if (rc) return rc;
  if (st_c == st) {
    rc = rlx_mlp_fwd_f32(ctx, cdesc, cparams, cobs, value, N, stream);
    if (rc) return rc;
  } else {
    RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
  }
  return ppo_sample(mean, pparams + L.logstd, ks[2], ks[3], scheme, action, processed, logp, nullptr, nullptr, N, L.A, L.O,
                    clip_and_rescale, act_low, act_high, noise_row_offset, N_global, st, deterministic);
}

int rlx_lstm_mask_carry_f32(rlx_ctx* ctx, float* c_io, float* h_io, const float* terminated, const float* truncated,
                            float* done_out, int N, int H, void* stream) {
  RLX_REQUIRE(ctx && c_io && h_io && terminated && N > 0 && H > 0, RLX_EINVAL, "rlx_lstm_mask_carry_f32: bad args");
  hipLaunchKernelGGL(k_mask_carry, dim3(ew_grid((int64_t)N * H)), dim3(256), 0, (hipStream_t)stream, c_io, h_io, terminated,
                     truncated, done_out, N, H);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

static int lstm_minibatch(rlx_ctx* ctx, const rlx_lstm_policy_desc& d, const LstmLayout& L, const float* pparams, float* pgrads,
                          const rlx_mlp_desc& cd, const float* cparams, float* cgrads, float* metrics, const float* states,
                          const float* actions, const float* log_probs, const float* returns, const float* advantages,
                          const float* dones, const float* c0, const float* h0, const int32_t* env_idx, int ne, int T, int N,
                          const rlx_ppo_hparams& hp, float* psq, int* npsq, float* csq, int* ncsq, hipStream_t st,
                          hipStream_t st_c, const double* stats_pre = nullptr, int64_t mb_global = 0) {
  // stats_pre / mb_global (data parallel): the all-reduced advantage sums of the GLOBAL minibatch and its size -- this rank's
  // T * ne rows are a shard of it, every mean of the loss divides by mb_global
  const int64_t M = (int64_t)T * ne;
  const int Mg = (int)(mb_global > 0 ? mb_global : M);
  LstmBufs b;
  int rc = lstm_bufs(ctx, L, M, ne, &b);
  if (rc) return rc;
  MbScratch s;
  const bool crows = hp.critic_states != nullptr;   // the critic's own observation columns [T, N, cd.in_dim]
  rc = ppo_mb_scratch(ctx, L.O, L.A, cd, L.D3, M, &s, crows);
  if (rc) return rc;
  hipLaunchKernelGGL(k_seq_index, dim3(ew_grid(M)), dim3(256), 0, st, env_idx, b.idx_flat, T, ne, N);
  RLX_LAUNCH_CHECK();
  if (stats_pre) s.stats = const_cast<double*>(stats_pre);
  rc = ppo_gather(ctx, states, actions, log_probs, returns, advantages, b.idx_flat, M, L.O, L.A, s, st, hp.critic_states,
                  crows ? cd.in_dim : 0, stats_pre == nullptr);
  if (rc) return rc;
  hipLaunchKernelGGL(k_gather_seq_aux, dim3(ew_grid(M + (int64_t)ne * L.H)), dim3(256), 0, st, dones, c0, h0, env_idx, b.done, b.c0,
                     b.h0, T, ne, N);
  RLX_LAUNCH_CHECK();
  RLX_HIP_TRY(hipMemsetAsync(metrics, 0, 8 * sizeof(float), st));
  RLX_HIP_TRY(hipMemsetAsync(pgrads, 0, (size_t)L.n_params * sizeof(float), st));
  if (st_c != st) {
    // the feed-forward critic is independent of the recurrent policy once the rows are gathered: it runs on the
    // side stream (own arenas: scratch bank 1) under the recurrence, which occupies only ne/16 workgroups
    RLX_HIP_TRY(hipEventRecord(ctx->ev_fork, st));
    RLX_HIP_TRY(hipStreamWaitEvent(st_c, ctx->ev_fork, 0));
    MbScratch s2 = s, tmp;
    ctx->bank = 1;
    rc = ppo_mb_scratch(ctx, L.O, L.A, cd, L.D3, M, &tmp);
    if (!rc) {
      for (int l = 0; l < 4; ++l) s2.acts[l] = tmp.acts[l];
      s2.head_part = tmp.head_part;
      rc = ppo_critic_fwd_bwd(ctx, cd, cparams, cgrads, metrics, s2, M, Mg, hp, csq, ncsq, st_c);
    }
    ctx->bank = 0;
    if (rc) return rc;
  }
  // the torso / gate-input GEMMs of the sequence pass (T * ne rows) on the fp16 pipe: one image launch for the policy's dense
  // matrices (scratch bank 0; the critic registers its own in bank 1)
  struct BxScope { rlx_ctx* c; ~BxScope() { const int b_ = c->bank; c->bank = 0; bx_release(c); c->bank = b_; } } bx_scope{ctx};
  if (M >= 4096) {
    const int gates = L.gru ? 3 : 4;
    const BxMat mats[4] = {{pparams + L.t1_W, L.K1, L.D1, true, true}, {pparams + L.t2_W, L.D1, L.D2, true, true},
                           {pparams + L.t3_W, L.D2, L.D3, true, true}, {pparams + L.Wi, L.E, gates * L.H, true, true}};
    rc = bx_prepare_mats(ctx, mats, 4, st);
    if (rc) return rc;
  }
  rc = lstm_policy_fwd(ctx, L, pparams, s.mb_x, b, T, ne, nullptr, nullptr, 0, st);
  if (rc) return rc;
  *npsq = 0;
  // ONE slab reduction for the whole policy: the head, the three torso layers, both LayerNorms, the recurrent weights and the
  // encoders reduce their partial slabs in a single launch at the end of the backward pass (11 launches of ~10 us + their
  // dependency gaps otherwise).  The slabs live in one arena sized from the same formulas the stages use.
  ReduceDefer defer;
  {
    int lgrid = div_up(M, 4);
    if (lgrid > ctx->num_cus * 4) lgrid = ctx->num_cus * 4;
    const int64_t E = L.E, H = L.H;
    auto a64 = [](size_t n) { return (n + 63) & ~size_t(63); };
    size_t need = stage_dw_floats(ctx, M, L.D2, L.D3) + stage_dw_floats(ctx, M, L.D1, L.D2) + a64((size_t)lgrid * 2 * L.D1) +
                  stage_dw_floats(ctx, M, L.K1, L.D1) + a64((size_t)lgrid * 2 * H) +
                  (L.film ? stage_dw_floats(ctx, M, (int)H, (int)(2 * E)) : 0) +
                  (L.gru ? stage_dw_floats(ctx, M, (int)H, (int)(2 * H)) + stage_dw_floats(ctx, M, (int)H, (int)H) +
                               stage_dw_floats(ctx, M, (int)E, (int)(3 * H))
                         : stage_dw_floats(ctx, M, (int)H, (int)(4 * H)) + stage_dw_floats(ctx, M, (int)E, (int)(4 * H))) +
                  (L.share ? 1 : 2) * stage_l1_bwd_floats(ctx, M, L.O, (int)E);
    defer.base = (float*)scratch(ctx, SL_STAGE, need * sizeof(float));
    if (!defer.base) return RLX_ENOMEM;
    defer.cap = need;
    defer.tab.n = 0;
    ctx->defer = &defer;
  }
  struct DeferGuard { rlx_ctx* c; ~DeferGuard() { c->defer = nullptr; } } defer_guard{ctx};
  rc = ppo_policy_head_loss(ctx, b.H3, pparams + L.hd_W, pparams + L.hd_b, pparams + L.logstd, s, metrics, M, Mg, L.D3, L.A,
                            RLX_ACT_ELU, hp, pgrads + L.hd_W, pgrads + L.hd_b, pgrads + L.logstd, psq, npsq, st);
  if (rc) return rc;
  // the torso's weight gradients on the critic's stream (its chain is short and was issued first), under the recurrence
  TorsoDw td;
  if (st_c != st) {
    rc = ctx_sac_streams(ctx);       // (events)
    if (rc) return rc;
    td.sw = st_c;
    td.e_ready = ctx->sac_ev[0]; td.e_done = ctx->sac_ev[1];
  }
  GradScaleScope gscope(ctx, bx_grad_scale(Mg));   // dZ ~ 1 / (global minibatch rows)
  rc = lstm_policy_bwd(ctx, L, pparams, pgrads, s.mb_x, b, T, ne, psq, npsq, st, td);
  if (rc) return rc;
  if (td.sw) RLX_HIP_TRY(hipStreamWaitEvent(st, td.e_done, 0));     // the torso's slabs are written
  rc = stage_reduce_flush(ctx, psq, npsq, st);
  if (rc || st_c != st) return rc;
  return ppo_critic_fwd_bwd(ctx, cd, cparams, cgrads, metrics, s, M, Mg, hp, csq, ncsq, st);
}

// stats[u] = {sum adv, sum adv^2, T * ne, 0} over sequence minibatch u (envs perm[u * ne ..], all T steps): one workgroup each,
// fp64, fixed order (thread t owns rows t, t + 256, ...; butterfly; the four waves in order) -- like k_mb_adv_sums (dist.hip)
__global__ __launch_bounds__(256) void k_seq_adv_sums(const float* __restrict__ adv, const int32_t* __restrict__ perm, int T, int ne,
                                                      int N, double* __restrict__ stats) {
  __shared__ double s_red[8];
  const int u = blockIdx.x, cnt = T * ne;
  const int32_t* env = perm + (int64_t)u * ne;
  double s1 = 0.0, s2 = 0.0;
  for (int r = threadIdx.x; r < cnt; r += 256) {
    const double a = (double)adv[(int64_t)(r / ne) * N + env[r % ne]];
    s1 += a;
    s2 += a * a;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_red[w] = s1; s_red[4 + w] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[4 * u + 0] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    stats[4 * u + 1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    stats[4 * u + 2] = (double)cnt;
    stats[4 * u + 3] = 0.0;
  }
}

int rlx_ppo_lstm_minibatch_fwd_bwd_f32(rlx_ctx* ctx, const rlx_lstm_policy_desc* desc, const float* pparams, float* pgrads,
                                       const rlx_mlp_desc* cdesc, const float* cparams, float* cgrads, float* metrics,
                                       const float* states, const float* actions, const float* log_probs, const float* returns,
                                       const float* advantages, const float* dones, const float* c0, const float* h0,
                                       const int32_t* env_idx, int nr_minibatch_envs, int T, int N, const rlx_ppo_hparams* hp,
                                       void* stream) {
  RLX_REQUIRE(ctx && desc && pparams && pgrads && cdesc && cparams && cgrads && metrics && states && actions && log_probs &&
                  returns && advantages && dones && c0 && h0 && env_idx && hp,
              RLX_EINVAL, "rlx_ppo_lstm_minibatch_fwd_bwd_f32: NULL pointer");
  RLX_REQUIRE(nr_minibatch_envs > 0 && T > 0 && N >= nr_minibatch_envs, RLX_EINVAL, "rlx_ppo_lstm_minibatch_fwd_bwd_f32: bad sizes");
  int rc = check_lstm_desc(*desc);
  if (rc) return rc;
  rc = mlp_check_desc(*cdesc);
  if (rc) return rc;
  float* psq = (float*)scratch(ctx, SL_NORM, REDUCE_MAX_BLOCKS * sizeof(float));
  float* csq = (float*)scratch(ctx, SL_NORM2, REDUCE_MAX_BLOCKS * sizeof(float));
  if (!psq || !csq) return RLX_ENOMEM;
  int np = 0, nc = 0;
  const LstmLayout L = lstm_layout(*desc);
  return lstm_minibatch(ctx, *desc, L, pparams, pgrads, *cdesc, cparams, cgrads, metrics, states, actions, log_probs, returns,
                        advantages, dones, c0, h0, env_idx, nr_minibatch_envs, T, N, *hp, psq, &np, csq, &nc, (hipStream_t)stream,
                        (hipStream_t)stream);
}

int rlx_ppo_lstm_update_f32(rlx_ctx* ctx, const rlx_lstm_policy_desc* desc, float* pparams, float* pm, float* pv,
                            const rlx_mlp_desc* cdesc, float* cparams, float* cm, float* cv, const float* states,
                            const float* actions, const float* log_probs, const float* returns, const float* advantages,
                            const float* dones, const float* c0, const float* h0, int T, int N, int nr_epochs, int minibatch_size,
                            uint32_t key_io[2], int scheme, int64_t* opt_count_io, const float* lr_schedule,
                            const rlx_ppo_hparams* hp, float* metrics_out, void* stream) {
  if (ctx) ctx->ro_img.valid = false;   // the acting nets' weight images go stale with this call
  RLX_REQUIRE(ctx && desc && pparams && pm && pv && cdesc && cparams && cm && cv && states && actions && log_probs && returns &&
                  advantages && dones && c0 && h0 && key_io && opt_count_io && lr_schedule && hp && metrics_out,
              RLX_EINVAL, "rlx_ppo_lstm_update_f32: NULL pointer");
  RLX_REQUIRE(T > 0 && N > 0 && nr_epochs > 0 && minibatch_size % T == 0 && minibatch_size >= T, RLX_EINVAL,
              "rlx_ppo_lstm_update_f32: minibatch_size must be a multiple of nr_steps (ppo_lstm.py:62-63)");
  // Data parallel (context with a communicator / hook): N = THIS RANK's envs, minibatch_size = the GLOBAL
  // minibatch.  Every rank permutes its local env indices with the same replicated key and takes minibatch_size / (T * world)
  // of its envs per minibatch: the global minibatch is the union over the ranks (each env once per epoch), its advantage
  // statistics come from ONE batched all-reduce in front of the updates, the losses are scaled by 1 / minibatch_size, and each
  // network's gradient is all-reduced once per minibatch on that network's stream before its (redundant) clip + Adam step.
  const bool collective = dist_active(ctx);
  const int world = collective ? (ctx->world > 1 ? ctx->world : 1) : 1;
  RLX_REQUIRE((minibatch_size / T) % world == 0, RLX_EINVAL,
              "rlx_ppo_lstm_update_f32: minibatch_size / nr_steps must be divisible by the number of ranks");
  const int ne = minibatch_size / T / world;  // nr_minibatch_envs (ppo_lstm.py:59), this rank's share
  RLX_REQUIRE(ne > 0 && N % ne == 0, RLX_EINVAL, "rlx_ppo_lstm_update_f32: nr_envs must be a multiple of minibatch_size / nr_steps");
  int rc = check_lstm_desc(*desc);
  if (rc) return rc;
  rc = mlp_check_desc(*cdesc);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int Mn = N / ne;
  const LstmLayout L = lstm_layout(*desc);
  const int64_t nc_ = rlx_mlp_param_count(cdesc);
  int32_t* perm = (int32_t*)scratch(ctx, SL_PERM, (size_t)nr_epochs * N * sizeof(int32_t));
  float* pg = (float*)scratch(ctx, SL_GRAD_P, (size_t)L.n_params * sizeof(float));
  float* cg = (float*)scratch(ctx, SL_GRAD_C, (size_t)nc_ * sizeof(float));
  float* psq = (float*)scratch(ctx, SL_NORM, REDUCE_MAX_BLOCKS * sizeof(float));
  float* csq = (float*)scratch(ctx, SL_NORM2, REDUCE_MAX_BLOCKS * sizeof(float));
  if (!perm || !pg || !cg || !psq || !csq) return RLX_ENOMEM;
  // key, sub = split(key); permutation(sub, tile(arange(N), (E,1)), axis=1, independent=True)   (ppo_lstm.py:226-229)
  rc = rlx_permutation_i32(ctx, key_io, perm, nr_epochs, N, scheme, stream);
  if (rc) return rc;
  hipStream_t st_c = st;
  if (ctx->two_streams) {
    rc = ctx_side_stream(ctx);
    if (rc) return rc;
    st_c = ctx->side;
  }
  const int n_upd = nr_epochs * Mn;
  double* stats_all = nullptr;
  // (always: the fp64 advantage sums of ALL minibatches in one launch up front -- one workgroup each -- instead of a 27 us
  //  single-workgroup launch in front of every minibatch)
  {
    stats_all = (double*)scratch(ctx, SL_STATS_ALL, (size_t)n_upd * 4 * sizeof(double));
    if (!stats_all) return RLX_ENOMEM;
    hipLaunchKernelGGL(k_seq_adv_sums, dim3(n_upd), dim3(256), 0, st, advantages, perm, T, ne, N, stats_all);
    RLX_LAUNCH_CHECK();
    if (collective) {
      rc = dist_allreduce(ctx, stats_all, (int64_t)n_upd * 4, 1, st);
      if (rc) return rc;
    }
  }
  for (int u = 0; u < n_upd; ++u) {
    float* met = metrics_out + (int64_t)u * 10;
    int npb = 0, ncb = 0;
    rc = lstm_minibatch(ctx, *desc, L, pparams, pg, *cdesc, cparams, cg, met, states, actions, log_probs, returns, advantages,
                        dones, c0, h0, perm + (int64_t)u * ne, ne, T, N, *hp, psq, &npb, csq, &ncb, st, st_c,
                        stats_all + 4 * u, collective ? minibatch_size : 0);
    if (rc) return rc;
    const int64_t step = *opt_count_io + u + 1;
    if (collective) {
      rc = dist_allreduce(ctx, cg, nc_, 0, st_c);
      if (rc) return rc;
      rc = clip_adam_step(ctx, cparams, cg, cm, cv, nc_, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1, hp->adam_b2,
                          hp->adam_eps, met + 9, st_c, nullptr);
      if (rc) return rc;
      rc = dist_allreduce(ctx, pg, L.n_params, 0, st);
      if (rc) return rc;
      rc = clip_adam_step(ctx, pparams, pg, pm, pv, L.n_params, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1, hp->adam_b2,
                          hp->adam_eps, met + 8, st, nullptr);
      if (rc) return rc;
    } else {
    rc = launch_clip_adam(cparams, cg, cm, cv, nc_, csq, ncb, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1, hp->adam_b2,
                          hp->adam_eps, met + 9, st_c);
    if (rc) return rc;
    rc = launch_clip_adam(pparams, pg, pm, pv, L.n_params, psq, npb, step, lr_schedule[u], hp->max_grad_norm, hp->adam_b1,
                          hp->adam_b2, hp->adam_eps, met + 8, st);
    if (rc) return rc;
    }
    if (st_c != st) {  // the next gather overwrites the rows the critic reads
      RLX_HIP_TRY(hipEventRecord(ctx->ev_join, st_c));
      RLX_HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
    }
  }
  if (collective) {   // per-update metrics: partial sums over this rank's rows -> ONE all-reduce per iteration
    rc = dist_mask_metrics(metrics_out, n_upd, ctx->rank, 0, st);
    if (rc) return rc;
    rc = dist_allreduce(ctx, metrics_out, (int64_t)n_upd * 10, 0, st);
    if (rc) return rc;
  }
  *opt_count_io += (int64_t)nr_epochs * Mn;
  return RLX_OK;
}

}  // extern "C"
