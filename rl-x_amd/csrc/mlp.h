// mlp.h -- internal interface of mlp.hip (layer kernels + deterministic slab reduction).
#pragma once
#include "gemm.h"

namespace rlx {

constexpr int REDUCE_MAX_SEGS = 48;   // (deferred reduction of the recurrent policy: ~30 parameter blocks in one launch)
constexpr int REDUCE_MAX_BLOCKS = 4096;  // == capacity of the SL_NORM sum-of-squares partial array

// dst[i] = scale * sum_{s<S} src[s*stride + i] + bias   for i < len
struct ReduceSeg {
  const float* src;
  float* dst;
  int64_t len;
  int64_t stride;
  int S;
  int nblocks;   // filled by the launcher
  float scale;
  float bias;
  int in_norm;   // contributes to the gradient global norm
  int vec;       // filled by the launcher: 16-B vector path usable
};
struct ReduceTable {
  int n;
  ReduceSeg seg[REDUCE_MAX_SEGS];
};

// options of the generic trunk passes (SAC: wide inputs, stop-gradient passes, action gradients)
struct TrunkOpts {
  int ldx = 0;             // row stride of x (0: in_dim); wide inputs (in_dim > 32) may be zero padded to a multiple of 4
  bool gemm_l0 = false;    // run the first layer on the GEMM kernels even when in_dim <= 32 (needs ldx % 4 == 0, no LN):
                           // the small-input fused kernel has no input-gradient output
  float* dx_out = nullptr; // optional dL/dx[:, dx_c0 : dx_c0 + dx_nc] -> [M, dx_ld]   (wide inputs only)
  int dx_c0 = 0, dx_nc = 0, dx_ld = 0;
  // backward only: dZ of the second-to-last hidden layer is ALREADY in this buffer (the caller's row-tile-local tail kernel wrote
  // it next to dZ_last; acts[n_hidden - 2] still holds that layer's forward activations): the last layer's input-gradient launch
  // is skipped and the buffer stands in for acts[n_hidden - 2] as the gradient operand of everything below
  const float* dz_below_last = nullptr;
};

int mlp_check_desc(const rlx_mlp_desc& d);
// m_dev (optional, all forward launchers): DEVICE int32 holding the number of rows that really carry work (<= M); the
// launch covers M rows (a capacity the host knows) and the row tiles beyond *m_dev leave at once
int launch_gemm_fwd(rlx_ctx* ctx, const float* A, const float* W, const float* bias, float* C, int64_t M, int N, int K,
                    int act, hipStream_t st, int lda, const int32_t* m_dev = nullptr);
int launch_head_fwd(const float* H, const float* W, const float* b, float* out, int64_t M, int K, int A,
                    hipStream_t st, const int32_t* m_dev = nullptr, const Twin* tw = nullptr);
bool dx_cols_ok(int K, int nc);      // launch_dx_cols takes the shape (at most 64 columns, K % 4 == 0, tile within the LDS budget)
int launch_dx_cols(const float* dZ, const float* Wblk, float* dX, int64_t M, int K, int nc, int ldo, hipStream_t st,
                   const Twin* tw = nullptr);
int64_t choose_mc(int64_t M, int tiles, int num_cus, int* S_out);   // M-split of the weight-gradient kernels: slab rows, *S_out slabs
int64_t choose_mc_fit(int64_t M, int tiles, int num_cus, int* S_out);   // the same, never more than num_cus workgroups (one round)
int launch_reduce_segments(ReduceTable& tab, float* sumsq_partials, int* n_blocks_out, hipStream_t st, rlx_ctx* prof_ctx = nullptr);
int mlp_trunk_fwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x,
                  float* const* acts, int64_t M, hipStream_t st, int ldx = 0, bool gemm_l0 = false,
                  const int32_t* m_dev = nullptr, int n_layers = -1);   // n_layers: hidden layers to run (-1: all)
// grads == nullptr: input-gradient-only pass (parameters are stop_gradient'ed; no dW kernels, no reduction)
int mlp_trunk_bwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x,
                  float* const* acts, float* grads, int64_t M, const ReduceSeg* extra, int n_extra,
                  float* sumsq_partials, int* n_sumsq_blocks, hipStream_t st, const TrunkOpts* opt = nullptr);

// stage helpers for composite models (each reduces its slabs immediately; sumsq partials appended at sumsq + *nsq)
int stage_reduce(rlx_ctx* ctx, ReduceTable& tab, float* sumsq, int* nsq, hipStream_t st);
// two first layers of one shape on the same input rows in one launch (grid.y = 2) -- the recurrent policy's two encoders
int stage_l1_fwd2(rlx_ctx* ctx, const float* x, const float* W, const float* b, const float* g, const float* be, float* H,
                  const float* W2, const float* b2, const float* g2, const float* be2, float* H2, int64_t M, int O, int Hd,
                  int act, int ln, hipStream_t st);
int stage_l1_bwd2(rlx_ctx* ctx, const float* x, const float* W, const float* b, const float* g, const float* be, float* H,
                  const float* W2, const float* b2, const float* g2, const float* be2, float* H2, int64_t M, int O, int Hd, int act,
                  int ln, float* gW, float* gb, float* gg, float* gbe, float* gW2, float* gb2, float* gg2, float* gbe2, float* sumsq,
                  int* nsq, hipStream_t st);
// Deferred slab reduction of a composite backward pass: while ctx->defer points to one of these, the stage_* helpers take
// their partial-slab buffers from ITS arena (instead of the shared SL_STAGE slot, which the next stage would overwrite) and
// append their segments to ITS table; stage_reduce_flush reduces everything in ONE launch.  (ppo_lstm.hip: 11 reductions of
// ~10 us + their dependency gaps per sequence minibatch -> 1.)
struct ReduceDefer {
  ReduceTable tab;
  float* base = nullptr;
  size_t cap = 0, off = 0;     // floats
};
float* stage_alloc(rlx_ctx* ctx, size_t floats);                  // arena of the active ReduceDefer, else the SL_STAGE slot
size_t stage_dw_floats(const rlx_ctx* ctx, int64_t M, int Kd, int N);          // what stage_dw / stage_l1_bwd will take
size_t stage_l1_bwd_floats(const rlx_ctx* ctx, int64_t M, int O, int Hd);
int stage_reduce_flush(rlx_ctx* ctx, float* sumsq, int* nsq, hipStream_t st);   // ends the deferral (ctx->defer = nullptr)
int stage_dw(rlx_ctx* ctx, const float* Hp, int ldh, const float* dZ, int64_t M, int Kd, int N, float* gW, float* gB,
             float* sumsq, int* nsq, hipStream_t st);
// hsrc (optional, apply_act only): act' is taken from hsrc[M, Kd(ldo)] instead of `out` -- out-of-place backward
int stage_dx(rlx_ctx* ctx, const float* dZ, const float* W, float* out, int64_t M, int N, int Kd, int ldo, int act,
             int apply_act, hipStream_t st, const float* hsrc = nullptr);
int stage_l1_fwd(rlx_ctx* ctx, const float* x, const float* W, const float* b, const float* g, const float* be, float* H,
                 int64_t M, int O, int Hd, int act, int ln, hipStream_t st);
int stage_l1_bwd(rlx_ctx* ctx, const float* x, const float* W, const float* b, const float* g, const float* be, float* H,
                 int64_t M, int O, int Hd, int act, int ln, float* gW, float* gb, float* gg, float* gbe, float* sumsq,
                 int* nsq, hipStream_t st);

// l1fused.hip: second-layer input gradient + whole first-layer backward in one kernel
bool l1fused_supported(const rlx_mlp_desc& d);
// first-layer forward on the matrix pipe (512-wide LayerNorm + ELU shape)
bool l1fwd_mfma_supported(const rlx_mlp_desc& d);
int launch_l1fwd_mfma(const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x, float* h1, int64_t M,
                      int num_cus, hipStream_t st, const int32_t* m_dev = nullptr, rlx_ctx* prof_ctx = nullptr,
                      const float* params1 = nullptr, float* h1_1 = nullptr);   // params1 / h1_1: twin launch (second network)
// first and second layer forward in one launch (l1fused.hip: k_l12fwd); needs the forward split images of both layers
struct L12Twin {
  const float* params;
  float *h1, *h2;
  const void *w1x, *w2x;
  float* stats = nullptr;
};
bool l12fwd_supported(const rlx_mlp_desc& d);
// stats (optional, [2][M]; twin: tw->stats too): the rows' LayerNorm mean and 1 / std go there and h1 is NOT stored -- the layer-2
// weight gradient then rebuilds it (BxDwRecompute below)
int launch_l12fwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x, float* h1, float* h2,
                  const void* w1x, const void* w2x, int64_t M, hipStream_t st, const L12Twin* tw = nullptr, float* stats = nullptr);
// whole forward of a 256-256 network incl. its head in one launch (fwd2h.hip); needs the forward split images of both layers
struct Fwd2hTwin {
  const float* params;
  const void *w1x, *w2x;
  float *h1, *h2, *out;
};
bool fwd2h_supported(const rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, int64_t M, int ldx,
                     const void** w1x, const void** w2x);
int launch_fwd2h(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const void* w1x, const void* w2x,
                 const float* x, int ldx, float* h1, float* h2, float* out, int64_t M, hipStream_t st, const Fwd2hTwin* tw = nullptr);
// whole forward of a 512-LayerNorm-256-128 network (the full-jit flavour's nets) incl. its head in one launch (fwd2h.hip: k_fwd3h);
// needs the forward split images of all three layers (wx[0..2])
struct Fwd3hTwin {
  const float* params;
  const void* wx[3];
  float *z1, *h1, *h2, *h3, *out;
};
bool fwd3h_supported(const rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, int64_t M, int ldx, const void** wx);
int launch_fwd3h(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const void* const* wx, const float* x, int ldx,
                 float* z1, float* h1, float* h2, float* h3, float* out, int64_t M, hipStream_t st, const Fwd3hTwin* tw = nullptr);
// dQ/da of a 256-256 critic (head backward + layer-2 input gradient + the first layer's product restricted to nc input columns) in one
// launch (fwd2h.hip: k_dxa2h); the activations are only read
struct Dxa2hTwin {
  const float* params;
  const void* w2t;
  const float *h1, *h2, *dq;
  float* da;
};
bool dxa2h_supported(const rlx_ctx* ctx, const rlx_mlp_desc& d, int64_t M, int nc);
int launch_dxa2h(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const void* w2t, const float* h1,
                 const float* h2, const float* dq, float* da, int c0, int nc, int ld_da, int64_t M, hipStream_t st,
                 const Dxa2hTwin* tw = nullptr);
// both upper layers' weight gradients as one two-job launch (mlp_trunk_bwd with TrunkOpts::dz_below_last): usable?
bool dw_merge_ok(const rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, int64_t M);
size_t l1fused_partial_floats(const rlx_mlp_desc& d, int grid);
int l1fused_grid(int64_t M, int num_cus);
// the second network of a twin launch of the fused first-layer backward (same shapes, same rows x)
struct L1FusedTwin {
  const float* params;
  const float* dZ2;
  float* arena;            // its slabs [grid][(O + 3) * H1]
  float* grads;
  const void *w2x, *w1x;   // its split images (l1fused_bx_images under the bank they were registered in)
  ReduceTable* tab;        // receives its reduce segments
};
bool l1fused_bx_images(const rlx_ctx* ctx, const MlpLayout& L, const float* params, const void** w2x, const void** w1x);
int launch_l1fused(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x,
                   const float* dZ2, float* arena, int grid, float* grads, int64_t M, ReduceTable* tab, hipStream_t st,
                   const L1FusedTwin* tw = nullptr);

// gemm_bx.hip: the same three GEMMs on the fp16 matrix pipe with split-fp32 operands.  bx_prepare_mlp lays out the weight
// images of the hidden layers l >= 1 of one network (forward, and with_bwd the transposed ones of the input gradients) in
// ONE launch and registers them for the current scratch bank; launch_gemm_fwd / the input-gradient launches pick them up by
// weight pointer until bx_release.  Without a registered image every caller runs the exact-fp32 engine.
int bx_prepare_mlp(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, bool with_bwd, hipStream_t st);
void bx_release(rlx_ctx* ctx);
// several networks at once (SAC: policy, twin critics, twin targets), first layers included when they run on the GEMM kernels;
// ONE launch, images registered for BOTH scratch banks (the update's two chains share the networks); bx_release_all drops them
struct BxNetSpec {
  const rlx_mlp_desc* d;
  const float* params;
  bool with_bwd;    // also the transposed images of the hidden layers l >= 1 (input gradients)
  bool first_layer; // include layer 0 (wide / GEMM first layers only)
};
// slot: the arena; launch = false only REGISTERS the images an earlier call with the same list laid out there (they were kept current)
int bx_prepare_nets(rlx_ctx* ctx, const BxNetSpec* nets, int n_nets, hipStream_t st, ScratchSlot slot = SL_WFRAG, bool launch = true);
// an explicit list of weight matrices W[K, N] (row-major, row stride N) for models that are not an rlx_mlp_desc (the recurrent
// policy's torso): forward and / or transposed images, registered for the CURRENT scratch bank until bx_release
struct BxMat {
  const float* W;
  int K, N;
  bool fwd, trans;
};
int bx_prepare_mats(rlx_ctx* ctx, const BxMat* mats, int n, hipStream_t st);
void bx_release_all(rlx_ctx* ctx);
const void* bx_lookup(const rlx_ctx* ctx, const float* W, int trans, int K, int N);
// tw (optional, bx_twin_usable): the second of two equally shaped problems in the same launch (see Twin, common.h)
bool bx_twin_usable(const rlx_ctx* ctx, int64_t M, int N);
int bx_launch_fwd(rlx_ctx* ctx, const float* A, const void* img, const float* bias, float* C, int64_t M, int N, int K, int act,
                  hipStream_t st, int lda, const int32_t* m_dev, const Twin* tw = nullptr);
int bx_launch_dx(rlx_ctx* ctx, const float* dZ, const void* img, float* HD, int64_t M, int N, int Kd, int ldo, int act, int apply,
                 hipStream_t st, const Twin* tw = nullptr, const float* hsrc = nullptr);
bool bx_dw_usable(const rlx_ctx* ctx, int64_t M, int Kd, int ldh, int N);
int bx_launch_dw(rlx_ctx* ctx, const float* Hp, const float* dZ, float* pW, float* pB, int64_t M, int Kd, int ldh, int N,
                 int64_t Mc, int S, int ntk, int ntn, hipStream_t st, const Twin* tw = nullptr);
// Hprev of a weight-gradient job REBUILT on the fly instead of read: Hprev = ELU(LayerNorm(X @ W1 + b1)), the 512-wide first layer
// (gemm_bx.hip: the recomputed-operand producers of k_gemm_dw_bx).  stats: [2][M] LayerNorm mean / (1 / std) of every row as the
// forward kernel (k_l12fwd) wrote them.  *1: the second network of a twin launch (parameters pdelta1 floats behind the first's).
struct BxDwRecompute {
  const float* X = nullptr;     // [M, O]
  const void* W1x = nullptr;    // forward split image of W1 (K = O padded to 32, N = 512)
  const float* b1 = nullptr;
  const float* g = nullptr;
  const float* be = nullptr;
  const float* stats = nullptr;
  const uint32_t* xmax = nullptr;
  const void* W1x1 = nullptr;
  const float* stats1 = nullptr;
  int64_t pdelta1 = 0;
  int O = 0, NT1 = 16;          // NT1: 32-column tiles of the W1 image (512 / 32)
};
// two weight-gradient problems over the same M rows in one launch (they share the CUs: each is split into about half the slabs)
struct BxDwJob {
  const float* Hp;
  const float* dZ;
  float *pW, *pB;
  int Kd, ldh, N;
  int64_t Mc;
  int S, ntk, ntn;
  const BxDwRecompute* rc = nullptr;   // job 0 only: its Hprev operand is recomputed (Hp unused)
};
int bx_launch_dw2(rlx_ctx* ctx, const BxDwJob& j0, const BxDwJob& j1, int64_t M, hipStream_t st, const Twin* tw0 = nullptr,
                  const Twin* tw1 = nullptr);


// optim.hip: clip + Adam consuming precomputed sum-of-squares partials
// sched_dev (optional): DEVICE {lr, 1 - b1^step, 1 - b2^step} overriding the by-value step / lr (graph-captured updates)
// emit (optional): hidden-layer weight matrices whose split-fp16 images (forward and transposed, gemm_bx.h) the kernel rewrites
// from the parameters it has just updated -- bit-identical to what k_bx_wfrag would lay out from them
struct BxEmitLayer {
  int64_t w_off;      // offset of W[in, out] inside the flat parameter vector
  int in, out;
  void* nn;           // forward image (B(k, j) = W[k][j]) or nullptr
  void* tt;           // transposed image (B(k, j) = W[j][k]) or nullptr
  int nt_nn, nt_tt;   // 32-column tiles per 16-k block of each image
};
struct BxEmit {
  int n;
  BxEmitLayer l[3];
};
BxEmit bx_emit_table(const rlx_ctx* ctx, const rlx_mlp_desc& d, const float* params);   // n = 0: nothing to emit
int launch_clip_adam(float* params, const float* grads, float* m, float* v, int64_t n, const float* sumsq_partials,
                     int n_partials, int64_t step, float lr, float max_norm, float b1, float b2, float eps,
                     float* norm_out, hipStream_t st, const float* sched_dev = nullptr, const BxEmit* emit = nullptr,
                     float* polyak_target = nullptr, float tau = 0.f,    // polyak_target: target <- tau p + (1 - tau) target, fused
                     float weight_decay = 0.f,                           // != 0: AdamW (p *= 1 - lr * wd in front of the step)
                     int clip_mode = 0);                                 // 0: optax.clip_by_global_norm; 1: torch clip_grad_norm_
int launch_clip_adam2(float* p0, const float* g0, float* m0, float* v0, int64_t n0, const float* part0, int np0, float* norm0,
                      const BxEmit* e0, float* p1, const float* g1, float* m1, float* v1, int64_t n1, const float* part1, int np1,
                      float* norm1, const BxEmit* e1, int64_t step, float lr, float max_norm, float b1, float b2, float eps,
                      hipStream_t st, const float* sched_dev = nullptr);
int launch_sumsq_partials2(const float* g0, int64_t n0, float* part0, int* np0, const float* g1, int64_t n1, float* part1, int* np1,
                           hipStream_t st);
int clip_adam_step(rlx_ctx* ctx, float* params, const float* grads, float* m, float* v, int64_t n_params, int64_t step, float lr,
                   float max_grad_norm, float b1, float b2, float eps, float* grad_norm_out, hipStream_t st, const BxEmit* emit);
void adam_schedule_entry(float* out3, int64_t step, float lr, float b1, float b2);
int launch_sumsq_partials(const float* g, int64_t n, float* partials, hipStream_t st);

}  // namespace rlx
