// fwd2h.hip -- a whole two-hidden-layer network forward in ONE launch per 32-row tile: out = head(act(act(X W1 + b1) W2 + b2)).
//
// The SAC step at BASELINE.json configs[3] (4096-row batches, 256-256 nets, sac/flax/policy.py:22-41, critic.py:17-41) is a chain
// of ~26 dependent 4096-row kernels that each sit at their launch + prologue + epilogue floor (10-14 us whatever they compute;
// profiles/r05_sac_timeline.txt): a network forward was three of them (k_gemm_bx, k_gemm_bx, k_head_fwd = 34 us).  Here the tile's
// observation rows become fp16 planes in LDS once, both hidden layers run on the fp16 pipe against the forward split images
// (gemm_bx.h) streaming from L2 -- the barrier-free K loops of k_l12fwd (l1fused.hip) --, h1 stays in LDS as the second layer's A
// operand, h2 stays in LDS as fp32 for the head, which is exact-fp32 VALU like k_head_fwd.  Activations go to HBM only when the
// caller's backward pass needs them.  TWIN: grid.y == 2, blockIdx.y == 1 takes the second argument set (the vmapped twin
// critics: same rows, same shapes).
//
// 8 waves; wave w owns columns [32 w, 32 w + 32) of both hidden layers (accumulator lane = column li, register r <-> row
// (r & 3) + 8 (r >> 2) + 4 lh).  LDS rows are padded so that the 16 rows of a ds_read_b128 service group start 4 banks apart
// (stride = 128 m + 16 bytes): every fragment / store address is one per-lane base plus an immediate.
#include "gemm_bx.h"
#include "mlp.h"

namespace rlx {

typedef float hl_f4 __attribute__((ext_vector_type(4)));
constexpr int F2_ROWS = 32, F2_H = 256, F2_THREADS = 512, F2_NW = 8;
constexpr int F2_HROW = 2 * F2_H + 16;        // bytes per row of one plane of the h1 image
constexpr int F2_HPL = F2_ROWS * F2_HROW;     // one plane
constexpr int F2_H2S = F2_H + 4;              // floats per row of the fp32 h2 tile
constexpr int F2_PF = 4;                      // 16-k blocks of weight fragments in flight per wave (NB16 % F2_PF == 0)
constexpr int F2_MAXV = 8;                    // float4 loads per thread and tile (K1 <= 512)

struct Fwd2hArgs {
  const float* X;      // [M, ldx]; columns [0, K1) are the input
  const void* W1x;     // forward split image of W1 [K1, 256] (K padded to 32)
  const float* b1;
  const void* W2x;     // forward split image of W2 [256, 256]
  const float* b2;
  const float* Wh;     // [256, OD] row-major
  const float* bh;     // [OD]
  float* H1;           // optional [M, 256]
  float* H2;           // optional [M, 256]
  float* OUT;          // [M, OD]
};

// NTH: 32-column tiles of the head on the fp32 matrix pipe (out_dim <= 32 NTH: 1 or 2); 0: out_dim == 1 (critics), a dot product per row
template <int ACT, bool TWIN, int NTH>
__global__ __launch_bounds__(F2_THREADS, 2) void k_fwd2h(Fwd2hArgs a, Fwd2hArgs a2, int64_t M, int ldx, int K1, int OD, int xrow,
                                                         int hoff, int woff, unsigned long long* dbg) {
#define F2_STAMP(I) if (dbg && t == 0 && blockIdx.x == 0 && blockIdx.y == 0) dbg[I] = clock64();
  if (TWIN && blockIdx.y) a = a2;
  extern __shared__ __attribute__((aligned(16))) char f2_smem[];
  const int KB1 = 2 * ((K1 + 31) >> 5);          // 16-k blocks of the first layer (the image's padding)
  const int XPL = F2_ROWS * xrow;                // bytes per observation plane
  char* Xs = f2_smem;                            // 2 planes [32][xrow]; after the first layer: h2 as fp32 [32][F2_H2S]
  char* Himg = f2_smem + hoff;                   // 2 planes [32][F2_HROW]
  float* Whs = reinterpret_cast<float*>(f2_smem + woff);   // out_dim == 1 only: the 256 head weights + partial sums [16][32]
  float* H2s = reinterpret_cast<float*>(Xs);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  F2_STAMP(0)
  const int col = w * 32 + li;
  const float b1v = a.b1[col], b2v = a.b2[col];
  // head weights.  out_dim == 1: the 256 weights to LDS.  Otherwise wave w keeps rows [32 w, 32 w + 32) of Wh as the B operands of
  // its sixteen v_mfma_f32_32x32x2_f32 steps (lane (li, lh) of step s: Wh[32 w + 2 s + lh][32 j + li]) -- loaded once per workgroup
  float whr[NTH > 0 ? NTH : 1][16];
  if (NTH == 0) {
    if (t < F2_H) Whs[t] = a.Wh[t];
  } else {
#pragma unroll
    for (int j = 0; j < NTH; ++j)
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) {      // (unconditional loads, masked afterwards: no branch per element)
        const int col = 32 * j + li;
        const float v = a.Wh[(32 * w + 2 * s_ + lh) * OD + (col < OD ? col : 0)];
        whr[j][s_] = col < OD ? v : 0.f;
      }
  }
  // head biases of this thread's output columns (t >> 5) + 16 i: once per workgroup (loaded where they are used, behind the head's
  // barrier, each of the three column passes opened with its own L2 round trip)
  float bhv[3] = {0.f, 0.f, 0.f};
  if (NTH != 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = (t >> 5) + 16 * i;
      bhv[i] = a.bh[c < OD ? c : 0];
    }
  }
  constexpr int w_step = (F2_H / 32) * X_NP * 64;                        // u32x4 entries per 16-k block of an N = 256 image
  const u32x4* __restrict__ W1x = reinterpret_cast<const u32x4*>(a.W1x) + (int64_t)w * X_NP * 64 + lane;
  const u32x4* __restrict__ W2x = reinterpret_cast<const u32x4*>(a.W2x) + (int64_t)w * X_NP * 64 + lane;
  const int nv_row = KB1 * 4;                    // float4 slots per row (covers the padded K; <= 128)
  const int xr_ = t >> 7, xc_ = t & 127;         // thread <-> float4 slot xc_ of rows xr_ + 4 c
  const bool vec = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;
  const float so = X_AINV * X_WINV;
  const int64_t ntiles = (M + F2_ROWS - 1) / F2_ROWS;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * F2_ROWS;
    __syncthreads();      // the previous tile's head has read h2 (aliases the observation planes)
    F2_STAMP(1)
    // ---- observation rows -> two fp16 planes (all loads of the tile in flight before the first LDS store)
    {
      hl_f4 xv[F2_MAXV];
      // Interior tiles of 16-byte-pitched rows: every load unconditional (a slot past the row's K reads slot 0 and is zeroed
      // afterwards), so all of the tile's loads really are in flight together.  With the nested conditions below hipcc gave each of
      // the eight row passes its own basic block with its own s_waitcnt vmcnt(0): eight dependent round trips per tile.
      const bool x_fast = vec && r0 + F2_ROWS <= M && ((K1 + 3) & ~3) <= ldx;
      if (x_fast) {
        const bool live = xc_ < nv_row && xc_ * 4 < K1;
        const float* src0 = a.X + (r0 + xr_) * ldx + (live ? xc_ * 4 : 0);
#pragma unroll
        for (int c = 0; c < F2_MAXV; ++c) xv[c] = *reinterpret_cast<const hl_f4*>(src0 + (int64_t)(4 * c) * ldx);
#pragma unroll
        for (int c = 0; c < F2_MAXV; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[c][e] = (live && xc_ * 4 + e < K1) ? xv[c][e] : 0.f;
      } else
#pragma unroll
      for (int c = 0; c < F2_MAXV; ++c) {
        xv[c] = hl_f4{0.f, 0.f, 0.f, 0.f};
        if (xc_ < nv_row) {
          const int r = xr_ + 4 * c, k = xc_ * 4;
          if (r0 + r < M && k < K1) {
            const float* src = a.X + (r0 + r) * ldx + k;
            if (vec && k + 4 <= ldx) {
              xv[c] = *reinterpret_cast<const hl_f4*>(src);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) xv[c][e] = k + e < K1 ? src[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[c][e] = k + e < K1 ? xv[c][e] : 0.f;      // pitch padding is not input
          }
        }
      }
#pragma unroll
      for (int c = 0; c < F2_MAXV; ++c) {
        if (xc_ < nv_row) {
          const int r = xr_ + 4 * c, k = xc_ * 4;
          uint32_t a0, a1, c0, c1;
          bx_split2(xv[c][0] * X_ASCALE, xv[c][1] * X_ASCALE, a0, a1);
          bx_split2(xv[c][2] * X_ASCALE, xv[c][3] * X_ASCALE, c0, c1);
          char* d = Xs + r * xrow + k * 2;
          *reinterpret_cast<u32x2*>(d) = u32x2{a0, c0};
          *reinterpret_cast<u32x2*>(d + XPL) = u32x2{a1, c1};
        }
      }
    }
    __syncthreads();
    F2_STAMP(2)
    // ---- first layer: z1 = X @ W1 (KB1 16-k blocks, weight fragments two blocks ahead)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
      const char* ard = Xs + li * xrow + lh * 16;
      u32x4 bx[F2_PF][X_NP];         // weight fragments F2_PF 16-k blocks ahead: two waves per SIMD do not hide an L2 round trip
#pragma unroll
      for (int u = 0; u < F2_PF; ++u)
#pragma unroll
        for (int p = 0; p < X_NP; ++p) bx[u][p] = W1x[(int64_t)(u < KB1 ? u : 0) * w_step + p * 64];      // (unconditional: a block that does not exist re-reads block 0 and is never used)
      // One 16-k block: A fragment from the observation planes, three plane products, and (REFILL) the fragments F2_PF blocks ahead
      // into the slot just consumed.
#define F2_L1_BLOCK(QU, REFILL)                                                                                         \
  {                                                                                                                     \
    u32x4 av[X_NP];                                                                                                     \
    _Pragma("unroll") for (int p = 0; p < X_NP; ++p) av[p] = *reinterpret_cast<const u32x4*>(ard + (QU) * 32 + p * XPL); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][1]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[1]), __builtin_bit_cast(f16x8, bx[u][0]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][0]), acc, 0, 0, 0); \
    if (REFILL) {                                                                                                       \
      _Pragma("unroll") for (int p = 0; p < X_NP; ++p) bx[u][p] = W1x[(int64_t)((QU) + F2_PF) * w_step + p * 64];        \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);   /* keep each block's refill behind ITS products: hipcc otherwise gathers the group's */ \
  }                                      /* eight loads at the end of the group and waits for all of them at its top       */
      // Main loop: groups of F2_PF blocks that ALL exist and ALL refill -- no condition inside, so hipcc counts the loads in flight
      // (s_waitcnt vmcnt(6): the oldest block's pair of 2 x F2_PF).  With the refill behind `if (q + u + F2_PF < KB1)` every block
      // opened with s_waitcnt vmcnt(0): the fragments requested for the blocks ahead were drained each time, ONE block was ever
      // in flight and the layer ran at an L2 round trip per block (625 clocks; phase stamps, round 6).
      int q = 0;
#pragma unroll 1
      for (; q + 2 * F2_PF <= KB1; q += F2_PF) {
#pragma unroll
        for (int u = 0; u < F2_PF; ++u) F2_L1_BLOCK(q + u, true)
      }
      // remainder (fewer than 2 F2_PF blocks): guarded
      for (; q < KB1; q += F2_PF) {
#pragma unroll
        for (int u = 0; u < F2_PF; ++u) {
          if (q + u < KB1) F2_L1_BLOCK(q + u, q + u + F2_PF < KB1)
        }
      }
#undef F2_L1_BLOCK
    }
    F2_STAMP(3)
    // ---- h1 = act(z1 + b1): to HBM when the backward needs it, and as fp16 planes into the second layer's A image
    {
      float* hb = a.H1 ? a.H1 + (r0 + 4 * lh) * F2_H + col : nullptr;
      char* awr = Himg + 4 * lh * F2_HROW + col * 2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2);
        const bool inb = r0 + rho + 4 * lh < M;
        const float h = act_fwd_t<ACT>(fmaf(acc[r], so, b1v));
        if (inb && hb) hb[(int64_t)rho * F2_H] = h;
        uint32_t p0, p1;
        bx_split2((inb ? h : 0.f) * X_ASCALE, 0.f, p0, p1);
        *reinterpret_cast<uint16_t*>(awr + rho * F2_HROW) = (uint16_t)p0;
        *reinterpret_cast<uint16_t*>(awr + rho * F2_HROW + F2_HPL) = (uint16_t)p1;
      }
    }
    __syncthreads();      // the h1 image is complete; nobody reads the observation planes any more
    F2_STAMP(4)
    // ---- second layer
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
      const char* ard = Himg + li * F2_HROW + lh * 16;
      constexpr int NB16 = F2_H / 16;
      u32x4 bx[F2_PF][X_NP];
#pragma unroll
      for (int u = 0; u < F2_PF; ++u)
#pragma unroll
        for (int p = 0; p < X_NP; ++p) bx[u][p] = W2x[(int64_t)u * w_step + p * 64];
#define F2_L2_BLOCK(QU, REFILL)                                                                                         \
  {                                                                                                                     \
    u32x4 av[X_NP];                                                                                                     \
    _Pragma("unroll") for (int p = 0; p < X_NP; ++p) av[p] = *reinterpret_cast<const u32x4*>(ard + (QU) * 32 + p * F2_HPL); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][1]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[1]), __builtin_bit_cast(f16x8, bx[u][0]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][0]), acc, 0, 0, 0); \
    if (REFILL) {                                                                                                       \
      _Pragma("unroll") for (int p = 0; p < X_NP; ++p) bx[u][p] = W2x[(int64_t)((QU) + F2_PF) * w_step + p * 64];        \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }
      // (last group peeled: unconditional refills in the loop -> counted waits; see the first layer)
#pragma unroll 1
      for (int q = 0; q < NB16 - F2_PF; q += F2_PF) {
#pragma unroll
        for (int u = 0; u < F2_PF; ++u) F2_L2_BLOCK(q + u, true)
      }
#pragma unroll
      for (int u = 0; u < F2_PF; ++u) F2_L2_BLOCK(NB16 - F2_PF + u, false)
#undef F2_L2_BLOCK
    }
    F2_STAMP(5)
    // ---- h2 = act(z2 + b2): to HBM when wanted, and as fp32 into LDS for the head
    {
      float* hb = a.H2 ? a.H2 + (r0 + 4 * lh) * F2_H + col : nullptr;
      float* hs = H2s + 4 * lh * F2_H2S + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2);
        const float h = act_fwd_t<ACT>(fmaf(acc[r], so, b2v));
        if (hb && r0 + rho + 4 * lh < M) hb[(int64_t)rho * F2_H] = h;
        hs[rho * F2_H2S] = h;
      }
    }
    __syncthreads();
    F2_STAMP(6)
    // ---- head (exact fp32): out[r][o] = bh[o] + sum_k h2[r][k] Wh[k][o]
    if (NTH == 0) {
      // 16 threads per row, 16 k each; the sixteen partial sums of a row are added in slice order
      float* part = Whs + F2_H;                       // [16][32]
      const int r = t & 31, g = t >> 5;
      const hl_f4* h4 = reinterpret_cast<const hl_f4*>(H2s + r * F2_H2S + 16 * g);
      const hl_f4* w4 = reinterpret_cast<const hl_f4*>(Whs + 16 * g);
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const hl_f4 hv = h4[q], wv = w4[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) s = fmaf(hv[e], wv[e], s);
      }
      part[g * 32 + r] = s;
      __syncthreads();
      if (t < 32 && r0 + t < M) {
        float o = a.bh[0];
#pragma unroll
        for (int q = 0; q < 16; ++q) o += part[q * 32 + t];
        a.OUT[r0 + t] = o;
      }
    } else {
      // wave w multiplies its 32-k slice of the h2 tile (fp32 from LDS) with its rows of Wh on the fp32 matrix pipe; the eight
      // partial tiles meet in LDS (the h1 image's space; columns 32 .. 47 behind the h2 tile) and are added in wave order.
      // (A VALU head -- 4 outputs per thread, Wh rows as broadcast LDS reads -- took 10 of the launch's 27 us: LDS-bandwidth bound.)
      constexpr int NH = NTH > 0 ? NTH : 1;
      f32x16 ah[NH];
#pragma unroll
      for (int j = 0; j < NTH; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ah[j][r] = 0.f;
      const float* hrd = H2s + li * F2_H2S + 32 * w + lh;
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) {
        const float av = hrd[2 * s_];
#pragma unroll
        for (int j = 0; j < NTH; ++j) ah[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, whr[j][s_], ah[j], 0, 0, 0);
      }
      float* p0 = reinterpret_cast<float*>(Himg);                  // [8][32][33]
      float* p1 = H2s + F2_ROWS * F2_H2S;                          // [8][32][17]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        p0[(w * 32 + row) * 33 + li] = ah[0][r];
        if (NTH == 2 && li < 16) p1[(w * 32 + row) * 17 + li] = ah[NH - 1][r];
      }
      __syncthreads();
      const int r = t & 31;
#pragma unroll
      for (int i = 0; i < 3; ++i) {            // OD <= 48: columns (t >> 5) + 16 i; their biases were requested once, up front
        const int c = (t >> 5) + 16 * i;
        if (c < OD) {
          float o = bhv[i];
#pragma unroll
          for (int q = 0; q < F2_NW; ++q) o += c < 32 ? p0[(q * 32 + r) * 33 + c] : p1[(q * 32 + r) * 17 + c - 32];
          if (r0 + r < M) a.OUT[(r0 + r) * OD + c] = o;
        }
      }
    }
    F2_STAMP(7)
  }
}

bool fwd2h_supported(const rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, int64_t M, int ldx,
                     const void** w1x, const void** w2x) {
  if (!ctx->fwd2h || !ctx->gemm_bx || d.n_hidden != 2 || d.hidden[0] != F2_H || d.hidden[1] != F2_H || d.ln_first) return false;
  if (d.act != RLX_ACT_RELU && d.act != RLX_ACT_TANH) return false;
  if (d.in_dim > 4 * F2_MAXV * F2_THREADS / F2_ROWS || d.out_dim > 48 || M < 1024) return false;
  if (ldx > 0 && ldx < d.in_dim) return false;
  *w1x = bx_lookup(ctx, params + L.layer[0].W, 0, L.layer[0].in, L.layer[0].out);
  *w2x = *w1x ? bx_lookup(ctx, params + L.layer[1].W, 0, L.layer[1].in, L.layer[1].out) : nullptr;
  return *w1x && *w2x;
}

// h1 / h2 may be NULL (forward-only pass).  tw (optional): the second net of a twin launch -- same x, same shapes.
int launch_fwd2h(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const void* w1x, const void* w2x,
                 const float* x, int ldx, float* h1, float* h2, float* out, int64_t M, hipStream_t st, const Fwd2hTwin* tw) {
  const LayerOff &o0 = L.layer[0], &o1 = L.layer[1];
  const int K1 = o0.in, OD = L.head.out, ld = ldx > 0 ? ldx : K1;
  Fwd2hArgs a;
  a.X = x; a.W1x = w1x; a.b1 = params + o0.b; a.W2x = w2x; a.b2 = params + o1.b; a.Wh = params + L.head.W; a.bh = params + L.head.b;
  a.H1 = h1; a.H2 = h2; a.OUT = out;
  Fwd2hArgs a2 = a;
  if (tw) {
    a2.W1x = tw->w1x; a2.b1 = tw->params + o0.b; a2.W2x = tw->w2x; a2.b2 = tw->params + o1.b; a2.Wh = tw->params + L.head.W;
    a2.bh = tw->params + L.head.b; a2.H1 = tw->h1; a2.H2 = tw->h2; a2.OUT = tw->out;
  }
  const int KB1 = 2 * div_up(K1, 32);
  const int xrow = 128 * div_up(KB1, 4) + 16;
  const int nth = OD == 1 ? 0 : (OD <= 32 ? 1 : 2);
  size_t xbytes = (size_t)2 * F2_ROWS * xrow;
  const size_t h2bytes = (size_t)F2_ROWS * F2_H2S * sizeof(float) + (nth == 2 ? (size_t)F2_NW * 32 * 17 * sizeof(float) : 0);
  const int hoff = (int)((xbytes > h2bytes ? xbytes : h2bytes) + 15) & ~15;
  const int woff = hoff + 2 * F2_HPL;
  const size_t lds = (size_t)woff + (nth == 0 ? (size_t)(F2_H + 16 * 32) * sizeof(float) : 0);
  RLX_REQUIRE(lds <= 160 * 1024, RLX_EUNSUP, "fwd2h: tile exceeds the LDS");
  const double nets = tw ? 2.0 : 1.0;
  // algorithmic: both hidden products + the head; rows in, activations out when stored, the weights
  ProfScope prof(ctx, PK_FWD2H, nets * 2.0 * (double)M * ((double)K1 * F2_H + (double)F2_H * F2_H + (double)F2_H * OD), st,
                 nets * 4.0 * ((double)M * (K1 / nets + OD + (h1 ? F2_H : 0) + (h2 ? F2_H : 0)) + (double)F2_H * (K1 + F2_H + OD)),
                 M, F2_H, K1, 1);
  const int64_t nt = (M + F2_ROWS - 1) / F2_ROWS;
  const int grid = (int)(nt < ctx->num_cus ? nt : ctx->num_cus);
#define RLX_F2_ATTR(KERNEL)                                                                                                \
  {                                                                                                                        \
    static AttrOnce attr_set;                                                                                                \
    if (!attr_set.done()) {                                                                                                       \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      attr_set.mark();                                                                                                       \
    }                                                                                                                      \
  }
#define RLX_F2_GO(KERNEL, GRID)                                                                                            \
  {                                                                                                                        \
    RLX_F2_ATTR((KERNEL))                                                                                                  \
    RLX_PLAUNCH((KERNEL), GRID, dim3(F2_THREADS), lds, st, a, a2, M, ld, K1, OD, xrow, hoff, woff, (unsigned long long*)ctx->dbg_stamps); \
  }
#define RLX_F2_LAUNCH(ACTV, NTHV)                                                                                          \
  {                                                                                                                        \
    if (tw) RLX_F2_GO((k_fwd2h<ACTV, true, NTHV>), dim3(grid, 2))                                                          \
    else RLX_F2_GO((k_fwd2h<ACTV, false, NTHV>), dim3(grid))                                                               \
  }
#define RLX_F2_ACT(NTHV)                                                                                                   \
  if (d.act == RLX_ACT_RELU) RLX_F2_LAUNCH(RLX_ACT_RELU, NTHV)                                                             \
  else RLX_F2_LAUNCH(RLX_ACT_TANH, NTHV)
  if (nth == 0) { RLX_F2_ACT(0) }
  else if (nth == 1) { RLX_F2_ACT(1) }
  else { RLX_F2_ACT(2) }
#undef RLX_F2_GO
#undef RLX_F2_ATTR
#undef RLX_F2_ACT
#undef RLX_F2_LAUNCH
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// ---- input gradient of a 256-256 critic w.r.t. a few input columns, in ONE launch per 32-row tile ----------------------------
// da[r][j] = sum_k dZ1[r][k] W1[c0 + j][k],  dZ1 = (dZ2 @ W2^T) * act'(h1),  dZ2[r][k] = dq[r] Wh[k] act'(h2[r][k])
// -- the dQ/da chain of the SAC policy loss (sac/flax/sac.py:163-176: the critics' parameters are constants there).  It was three
// dependent launches per update (k_head_bwd, k_gemm_bx<1>, the column-restricted first-layer product: 35 us of the policy chain).
// dZ2 is built from the stored h2 tile and split into fp16 planes (times the pass's gradient scale) as the A operand; the product
// streams the TRANSPOSED split image of W2 like k_fwd2h's layers; dZ1 stays in LDS as fp32 and meets the nc <= 32 rows of W1 on
// the fp32 matrix pipe (wave w: k-slice [32 w, 32 w + 32), partial tiles added in wave order).  Activations are only read.
struct Dxa2hArgs {
  const float* dq;     // [M] dL/dQ
  const float* Wh;     // [256] head weights
  const float* H1;     // [M, 256]
  const float* H2;     // [M, 256]
  const void* W2t;     // transposed split image of W2: B(k = layer-2 unit, j = layer-1 unit)
  const float* W1a;    // W1 + c0 * 256: rows [c0, c0 + nc) of W1 [in, 256]
  float* DA;           // [M, ld_da]
};

template <int ACT, bool TWIN>
__global__ __launch_bounds__(F2_THREADS, 2) void k_dxa2h(Dxa2hArgs a, Dxa2hArgs a2, int64_t M, int nc, int ld_da, float gs, float so) {
  if (TWIN && blockIdx.y) a = a2;
  extern __shared__ __attribute__((aligned(16))) char d2_smem[];
  char* Zimg = d2_smem;                                            // 2 planes [32][F2_HROW] of dZ2; later the partial tiles [8][32][33]
  float* Z1s = reinterpret_cast<float*>(d2_smem + 2 * F2_HPL);     // dZ1 as fp32 [32][F2_H2S]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  const int col = w * 32 + li;
  constexpr int w_step = (F2_H / 32) * X_NP * 64;
  const u32x4* __restrict__ W2t = reinterpret_cast<const u32x4*>(a.W2t) + (int64_t)w * X_NP * 64 + lane;
  // the wave's rows of W1 for the last product (lane (li, lh) of step s: W1[c0 + li][32 w + 2 s + lh]); once per workgroup
  float w1r[16];
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) w1r[s_] = li < nc ? a.W1a[(int64_t)li * F2_H + 32 * w + 2 * s_ + lh] : 0.f;
  const int64_t ntiles = (M + F2_ROWS - 1) / F2_ROWS;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * F2_ROWS;
    __syncthreads();      // the previous tile's reduction has read the partial tiles
    // ---- this wave's h1 values for the epilogue (in flight under the staging and the product)
    float h1v[16];
    {
      const float* hb = a.H1 + (r0 + 4 * lh) * F2_H + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2);
        h1v[r] = r0 + rho + 4 * lh < M ? hb[(int64_t)rho * F2_H] : 0.f;
      }
    }
    // ---- dZ2 tile -> fp16 planes: thread <-> (rows rr + 8 c, float4 slot cc of the row)
    {
      const int rr = t >> 6, cc = t & 63;
      const hl_f4 wh = *reinterpret_cast<const hl_f4*>(a.Wh + 4 * cc);
      hl_f4 hv[4];
      float dqv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = rr + 8 * c;
        const bool inb = r0 + r < M;
        hv[c] = inb ? *reinterpret_cast<const hl_f4*>(a.H2 + (r0 + r) * F2_H + 4 * cc) : hl_f4{0.f, 0.f, 0.f, 0.f};
        dqv[c] = inb ? a.dq[r0 + r] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = rr + 8 * c;
        float z[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = dqv[c] * wh[e] * act_grad_t<ACT>(hv[c][e]) * gs;
        uint32_t p0, p1, q0, q1;
        bx_split2(z[0], z[1], p0, p1);
        bx_split2(z[2], z[3], q0, q1);
        char* d = Zimg + r * F2_HROW + cc * 8;
        *reinterpret_cast<u32x2*>(d) = u32x2{p0, q0};
        *reinterpret_cast<u32x2*>(d + F2_HPL) = u32x2{p1, q1};
      }
    }
    __syncthreads();
    // ---- dH1 = dZ2 @ W2^T
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
      const char* ard = Zimg + li * F2_HROW + lh * 16;
      constexpr int NB16 = F2_H / 16;
      u32x4 bx[F2_PF][X_NP];
#pragma unroll
      for (int u = 0; u < F2_PF; ++u)
#pragma unroll
        for (int p = 0; p < X_NP; ++p) bx[u][p] = W2t[(int64_t)u * w_step + p * 64];
      // (unconditional refills in the loop, the last group peeled: counted waits -- see k_fwd2h's first layer)
#define F2_DX_BLOCK(QU, REFILL)                                                                                         \
  {                                                                                                                     \
    u32x4 av[X_NP];                                                                                                     \
    _Pragma("unroll") for (int p = 0; p < X_NP; ++p) av[p] = *reinterpret_cast<const u32x4*>(ard + (QU) * 32 + p * F2_HPL); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][1]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[1]), __builtin_bit_cast(f16x8, bx[u][0]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][0]), acc, 0, 0, 0); \
    if (REFILL) {                                                                                                       \
      _Pragma("unroll") for (int p = 0; p < X_NP; ++p) bx[u][p] = W2t[(int64_t)((QU) + F2_PF) * w_step + p * 64];        \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }
#pragma unroll 1
      for (int q = 0; q < NB16 - F2_PF; q += F2_PF) {
#pragma unroll
        for (int u = 0; u < F2_PF; ++u) F2_DX_BLOCK(q + u, true)
      }
#pragma unroll
      for (int u = 0; u < F2_PF; ++u) F2_DX_BLOCK(NB16 - F2_PF + u, false)
#undef F2_DX_BLOCK
    }
    // ---- dZ1 = dH1 * act'(h1) -> LDS (fp32)
    {
      float* zs = Z1s + 4 * lh * F2_H2S + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2);
        zs[rho * F2_H2S] = acc[r] * so * act_grad_t<ACT>(h1v[r]);
      }
    }
    __syncthreads();      // dZ1 complete; nobody reads the dZ2 planes any more
    // ---- da partial of this wave's k-slice on the fp32 matrix pipe
    {
      f32x16 ah;
#pragma unroll
      for (int r = 0; r < 16; ++r) ah[r] = 0.f;
      const float* zrd = Z1s + li * F2_H2S + 32 * w + lh;
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) ah = __builtin_amdgcn_mfma_f32_32x32x2f32(zrd[2 * s_], w1r[s_], ah, 0, 0, 0);
      float* p0 = reinterpret_cast<float*>(Zimg);
#pragma unroll
      for (int r = 0; r < 16; ++r) p0[(w * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = ah[r];
    }
    __syncthreads();
    {
      const float* p0 = reinterpret_cast<const float*>(Zimg);
      const int r = t & 31;
      for (int c = t >> 5; c < nc; c += 16) {
        float o = 0.f;
#pragma unroll
        for (int q = 0; q < F2_NW; ++q) o += p0[(q * 32 + r) * 33 + c];
        if (r0 + r < M) a.DA[(r0 + r) * ld_da + c] = o;
      }
    }
  }
}

bool dxa2h_supported(const rlx_ctx* ctx, const rlx_mlp_desc& d, int64_t M, int nc) {
  return ctx->fwd2h && ctx->gemm_bx && d.n_hidden == 2 && d.hidden[0] == F2_H && d.hidden[1] == F2_H && !d.ln_first && d.out_dim == 1 &&
         (d.act == RLX_ACT_RELU || d.act == RLX_ACT_TANH) && nc >= 1 && nc <= 32 && M >= 1024;
}

// h1 / h2: the activations the forward pass stored; w2t: the transposed split image of W2; tw: the second critic of a twin launch
int launch_dxa2h(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const void* w2t, const float* h1,
                 const float* h2, const float* dq, float* da, int c0, int nc, int ld_da, int64_t M, hipStream_t st, const Dxa2hTwin* tw) {
  Dxa2hArgs a;
  a.dq = dq; a.Wh = params + L.head.W; a.H1 = h1; a.H2 = h2; a.W2t = w2t; a.W1a = params + L.layer[0].W + (int64_t)c0 * F2_H; a.DA = da;
  Dxa2hArgs a2 = a;
  if (tw) {
    a2.dq = tw->dq; a2.Wh = tw->params + L.head.W; a2.H1 = tw->h1; a2.H2 = tw->h2; a2.W2t = tw->w2t;
    a2.W1a = tw->params + L.layer[0].W + (int64_t)c0 * F2_H; a2.DA = tw->da;
  }
  const float gs = ctx->bx_gscale;
  const double nets = tw ? 2.0 : 1.0;
  // (profiler row of the input-gradient kind: the layer-2 product dominates)
  ProfScope prof(ctx, PK_GEMM_DX, nets * 2.0 * (double)M * F2_H * (F2_H + nc + 1), st,
                 nets * 4.0 * ((double)M * (2 * F2_H + 1 + nc) + (double)F2_H * (F2_H + nc + 1)), M, F2_H, F2_H, 1);
  const size_t lds = (size_t)2 * F2_HPL + (size_t)F2_ROWS * F2_H2S * sizeof(float);
  const int64_t nt = (M + F2_ROWS - 1) / F2_ROWS;
  const int grid = (int)(nt < ctx->num_cus ? nt : ctx->num_cus);
#define RLX_D2_LAUNCH(ACTV)                                                                                                \
  {                                                                                                                        \
    static AttrOnce attr_set;                                                                                                \
    if (!attr_set.done()) {                                                                                                       \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dxa2h<ACTV, false>),                                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                            \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dxa2h<ACTV, true>),                                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                            \
      attr_set.mark();                                                                                                       \
    }                                                                                                                      \
    if (tw) { RLX_PLAUNCH((k_dxa2h<ACTV, true>), dim3(grid, 2), dim3(F2_THREADS), lds, st, a, a2, M, nc, ld_da, gs, X_WINV / gs); } \
    else { RLX_PLAUNCH((k_dxa2h<ACTV, false>), dim3(grid), dim3(F2_THREADS), lds, st, a, a2, M, nc, ld_da, gs, X_WINV / gs); }      \
  }
  if (d.act == RLX_ACT_RELU) RLX_D2_LAUNCH(RLX_ACT_RELU)
  else RLX_D2_LAUNCH(RLX_ACT_TANH)
#undef RLX_D2_LAUNCH
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// ---- the full-jit flavour's networks (sac/flax_full_jit/policy.py:21-45, critic.py:16-47): Dense(512) -> LayerNorm -> act ->
// Dense(256) -> act -> Dense(128) -> act -> head, the whole forward in ONE launch per 32-row tile -----------------------------------
// Same machinery as k_fwd2h: observation planes in LDS, every hidden layer on the fp16 pipe against its forward split image streaming
// from L2, activations handed from layer to layer as fp16 planes in LDS.  Wave w owns columns [64 w, 64 w + 64) of the first layer
// (LayerNorm statistics over the 512 columns: half_sum4 + one exchange through LDS, as k_l12fwd), [32 w, 32 w + 32) of the second,
// and column tile w & 3 of the third over the k-half w >> 2 (the two halves meet in LDS).  Were five launches per pass (k_gemm_bx,
// k_ln_act, k_gemm_bx, k_gemm_bx, k_head_fwd: ~70 us at 4096 rows).  The backward reads z1 (pre-LayerNorm), h1, h2, h3 from HBM: each
// is stored when its pointer is given.
constexpr int F3_H1 = 512, F3_H2 = 256, F3_H3 = 128;
constexpr int F3_A1ROW = 2 * F3_H1 + 16, F3_A1PL = F2_ROWS * F3_A1ROW;      // h1 image (one plane)
constexpr int F3_A2ROW = 2 * F3_H2 + 16, F3_A2PL = F2_ROWS * F3_A2ROW;      // h2 image
constexpr int F3_H3S = F3_H3 + 4;                                           // floats per row of the fp32 h3 tile

struct Fwd3hArgs {
  const float* X;
  const void *W1x, *W2x, *W3x;
  const float *b1, *g, *be, *b2, *b3, *Wh, *bh;
  float *Z1, *H1, *H2, *H3;      // optional stores (the backward's operands)
  float* OUT;
};

template <int ACT, bool TWIN, int NTH>
__global__ __launch_bounds__(F2_THREADS, 2) void k_fwd3h(Fwd3hArgs a, Fwd3hArgs a2, int64_t M, int ldx, int K1, int OD, int xrow, int a1off,
                                                         int soff) {
  if (TWIN && blockIdx.y) a = a2;
  extern __shared__ __attribute__((aligned(16))) char f3_smem[];
  const int KB1 = 2 * ((K1 + 31) >> 5);
  const int XPL = F2_ROWS * xrow;
  // region X: observation planes; after the first layer: the h2 image [2 planes] + the third layer's k-half partials [4][16][64]
  char* Xs = f3_smem;
  char* A2img = f3_smem;
  float* part3 = reinterpret_cast<float*>(f3_smem + 2 * F3_A2PL);
  // region A1: the h1 image [2 planes]; after the second layer: h3 as fp32 [32][F3_H3S] + the head's partial tiles [8][32][33]
  char* A1img = f3_smem + a1off;
  float* H3s = reinterpret_cast<float*>(A1img);
  float* hp0 = H3s + F2_ROWS * F3_H3S;
  float* hp1 = reinterpret_cast<float*>(A2img);                        // (NTH == 2) columns 32 .. 47 of the head partials [8][32][17]
  float* redA = reinterpret_cast<float*>(f3_smem + soff);             // LayerNorm exchange [2][8][32] + per-wave folded [8][64]; OD == 1: head weights + partials
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  const bool lb0 = (lane & 1) != 0, lb1 = (lane & 2) != 0;
  float* totA = redA + 2 * F2_NW * 32 + w * 64;
  float* Whs = redA + 2 * F2_NW * 32 + F2_NW * 64;                    // [128] + [16][32]   (OD == 1)
  float b1v[2], gv[2], bev[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = w * 64 + 32 * j + li;
    b1v[j] = a.b1[c];
    gv[j] = a.g[c];
    bev[j] = a.be[c];
  }
  const float b2v = a.b2[w * 32 + li], b3v = a.b3[(w & 3) * 32 + li];
  float whr[NTH > 0 ? NTH : 1][8];
  if (NTH == 0) {
    if (t < F3_H3) Whs[t] = a.Wh[t];
  } else {
#pragma unroll
    for (int j = 0; j < NTH; ++j)
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) whr[j][s_] = 32 * j + li < OD ? a.Wh[(16 * w + 2 * s_ + lh) * OD + 32 * j + li] : 0.f;
  }
  constexpr int w1_step = (F3_H1 / 32) * X_NP * 64, w2_step = (F3_H2 / 32) * X_NP * 64, w3_step = (F3_H3 / 32) * X_NP * 64;
  const u32x4* __restrict__ W1x = reinterpret_cast<const u32x4*>(a.W1x) + (int64_t)(2 * w) * X_NP * 64 + lane;
  const u32x4* __restrict__ W2x = reinterpret_cast<const u32x4*>(a.W2x) + (int64_t)w * X_NP * 64 + lane;
  const u32x4* __restrict__ W3x = reinterpret_cast<const u32x4*>(a.W3x) + (int64_t)(w & 3) * X_NP * 64 + lane;
  const int nv_row = KB1 * 4;
  const int xr_ = t >> 7, xc_ = t & 127;
  const bool vec = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;
  const float so = X_AINV * X_WINV, invH = 1.0f / (float)F3_H1;
  const int64_t ntiles = (M + F2_ROWS - 1) / F2_ROWS;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * F2_ROWS;
    __syncthreads();      // the previous tile's head has read its LDS operands
    // ---- observation rows -> two fp16 planes
    {
      hl_f4 xv[F2_MAXV];
      // Interior tiles of 16-byte-pitched rows: every load unconditional (a slot past the row's K reads slot 0 and is zeroed
      // afterwards), so all of the tile's loads really are in flight together.  With the nested conditions below hipcc gave each of
      // the eight row passes its own basic block with its own s_waitcnt vmcnt(0): eight dependent round trips per tile.
      const bool x_fast = vec && r0 + F2_ROWS <= M && ((K1 + 3) & ~3) <= ldx;
      if (x_fast) {
        const bool live = xc_ < nv_row && xc_ * 4 < K1;
        const float* src0 = a.X + (r0 + xr_) * ldx + (live ? xc_ * 4 : 0);
#pragma unroll
        for (int c = 0; c < F2_MAXV; ++c) xv[c] = *reinterpret_cast<const hl_f4*>(src0 + (int64_t)(4 * c) * ldx);
#pragma unroll
        for (int c = 0; c < F2_MAXV; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[c][e] = (live && xc_ * 4 + e < K1) ? xv[c][e] : 0.f;
      } else
#pragma unroll
      for (int c = 0; c < F2_MAXV; ++c) {
        xv[c] = hl_f4{0.f, 0.f, 0.f, 0.f};
        if (xc_ < nv_row) {
          const int r = xr_ + 4 * c, k = xc_ * 4;
          if (r0 + r < M && k < K1) {
            const float* src = a.X + (r0 + r) * ldx + k;
            if (vec && k + 4 <= ldx) {
              xv[c] = *reinterpret_cast<const hl_f4*>(src);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) xv[c][e] = k + e < K1 ? src[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[c][e] = k + e < K1 ? xv[c][e] : 0.f;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < F2_MAXV; ++c) {
        if (xc_ < nv_row) {
          const int r = xr_ + 4 * c, k = xc_ * 4;
          uint32_t a0, a1, c0, c1;
          bx_split2(xv[c][0] * X_ASCALE, xv[c][1] * X_ASCALE, a0, a1);
          bx_split2(xv[c][2] * X_ASCALE, xv[c][3] * X_ASCALE, c0, c1);
          char* d = Xs + r * xrow + k * 2;
          *reinterpret_cast<u32x2*>(d) = u32x2{a0, c0};
          *reinterpret_cast<u32x2*>(d + XPL) = u32x2{a1, c1};
        }
      }
    }
    __syncthreads();
    // ---- first layer: z1 = X @ W1 + b1 (two column tiles per wave)
    f32x16 z[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) z[j][r] = 0.f;
    {
      const char* ard = Xs + li * xrow + lh * 16;
      u32x4 bx[F2_PF][2][X_NP];
#pragma unroll
      for (int u = 0; u < F2_PF; ++u)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < X_NP; ++p) bx[u][j][p] = W1x[(int64_t)(u < KB1 ? u : 0) * w1_step + (j * X_NP + p) * 64];   // (unconditional, see k_fwd2h)
#define F3_L1_BLOCK(QU, REFILL)                                                                                         \
  {                                                                                                                     \
    u32x4 av[X_NP];                                                                                                     \
    _Pragma("unroll") for (int p = 0; p < X_NP; ++p) av[p] = *reinterpret_cast<const u32x4*>(ard + (QU) * 32 + p * XPL); \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                     \
      z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][j][1]), z[j], 0, 0, 0); \
      z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[1]), __builtin_bit_cast(f16x8, bx[u][j][0]), z[j], 0, 0, 0); \
      z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][j][0]), z[j], 0, 0, 0); \
    }                                                                                                                   \
    if (REFILL) {                                                                                                       \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int p = 0; p < X_NP; ++p)                     \
          bx[u][j][p] = W1x[(int64_t)((QU) + F2_PF) * w1_step + (j * X_NP + p) * 64];                                   \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }
      // (groups that all exist and all refill run without a condition: counted waits -- see k_fwd2h's first layer)
      int q = 0;
#pragma unroll 1
      for (; q + 2 * F2_PF <= KB1; q += F2_PF) {
#pragma unroll
        for (int u = 0; u < F2_PF; ++u) F3_L1_BLOCK(q + u, true)
      }
      for (; q < KB1; q += F2_PF) {
#pragma unroll
        for (int u = 0; u < F2_PF; ++u) {
          if (q + u < KB1) F3_L1_BLOCK(q + u, q + u + F2_PF < KB1)
        }
      }
#undef F3_L1_BLOCK
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) z[j][r] = fmaf(z[j][r], so, b1v[j]);
    // ---- LayerNorm row statistics (k_l12fwd's scheme)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      float sv[4], ssv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * gq + e;
        sv[e] = z[0][r] + z[1][r];
        ssv[e] = z[0][r] * z[0][r] + z[1][r] * z[1][r];
      }
      const float st_ = half_sum4(sv[0], sv[1], sv[2], sv[3], lb0, lb1);
      const float sst = half_sum4(ssv[0], ssv[1], ssv[2], ssv[3], lb0, lb1);
      if (li < 4) {
        redA[(0 * F2_NW + w) * 32 + 8 * gq + 4 * lh + li] = st_;
        redA[(1 * F2_NW + w) * 32 + 8 * gq + 4 * lh + li] = sst;
      }
    }
    __syncthreads();      // (also: nobody reads the observation planes any more)
    {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < F2_NW; ++q) v += redA[((lane >> 5) * F2_NW + q) * 32 + (lane & 31)];
      totA[lane] = v;
    }
    // ---- z1 -> HBM (pre-LayerNorm, the backward's operand); h1 = act(LN(z1)) -> HBM and fp16 planes
    {
      float* zb = a.Z1 ? a.Z1 + (r0 + 4 * lh) * F3_H1 + w * 64 + li : nullptr;
      float* hb = a.H1 ? a.H1 + (r0 + 4 * lh) * F3_H1 + w * 64 + li : nullptr;
      char* awr = A1img + 4 * lh * F3_A1ROW + (w * 64 + li) * 2;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const hl_f4 sv = *reinterpret_cast<const hl_f4*>(totA + 8 * gq + 4 * lh);
        const hl_f4 ssv = *reinterpret_cast<const hl_f4*>(totA + 32 + 8 * gq + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * gq + e, rho = 8 * gq + e;
          const float mean = sv[e] * invH;
          const float rs = rsqrtf(fmaxf(0.f, ssv[e] * invH - mean * mean) + 1e-6f);
          const bool inb = r0 + rho + 4 * lh < M;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float h = act_fwd_t<ACT>((z[j][r] - mean) * rs * gv[j] + bev[j]);
            if (inb && zb) zb[(int64_t)rho * F3_H1 + 32 * j] = z[j][r];
            if (inb && hb) hb[(int64_t)rho * F3_H1 + 32 * j] = h;
            uint32_t p0, p1;
            bx_split2((inb ? h : 0.f) * X_ASCALE, 0.f, p0, p1);
            char* d = awr + rho * F3_A1ROW + j * 64;
            *reinterpret_cast<uint16_t*>(d) = (uint16_t)p0;
            *reinterpret_cast<uint16_t*>(d + F3_A1PL) = (uint16_t)p1;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __syncthreads();      // the h1 image is complete
    // ---- second layer (K = 512)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
      const char* ard = A1img + li * F3_A1ROW + lh * 16;
      constexpr int NB16 = F3_H1 / 16;
      u32x4 bx[F2_PF][X_NP];
#pragma unroll
      for (int u = 0; u < F2_PF; ++u)
#pragma unroll
        for (int p = 0; p < X_NP; ++p) bx[u][p] = W2x[(int64_t)u * w2_step + p * 64];
#define F3_L2_BLOCK(QU, REFILL)                                                                                         \
  {                                                                                                                     \
    u32x4 av[X_NP];                                                                                                     \
    _Pragma("unroll") for (int p = 0; p < X_NP; ++p) av[p] = *reinterpret_cast<const u32x4*>(ard + (QU) * 32 + p * F3_A1PL); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][1]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[1]), __builtin_bit_cast(f16x8, bx[u][0]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][0]), acc, 0, 0, 0); \
    if (REFILL) {                                                                                                       \
      _Pragma("unroll") for (int p = 0; p < X_NP; ++p) bx[u][p] = W2x[(int64_t)((QU) + F2_PF) * w2_step + p * 64];       \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }
#pragma unroll 1
      for (int q = 0; q < NB16 - F2_PF; q += F2_PF) {
#pragma unroll
        for (int u = 0; u < F2_PF; ++u) F3_L2_BLOCK(q + u, true)
      }
#pragma unroll
      for (int u = 0; u < F2_PF; ++u) F3_L2_BLOCK(NB16 - F2_PF + u, false)
#undef F3_L2_BLOCK
    }
    {
      float* hb = a.H2 ? a.H2 + (r0 + 4 * lh) * F3_H2 + w * 32 + li : nullptr;
      char* awr = A2img + 4 * lh * F3_A2ROW + (w * 32 + li) * 2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2);
        const bool inb = r0 + rho + 4 * lh < M;
        const float h = act_fwd_t<ACT>(fmaf(acc[r], so, b2v));
        if (inb && hb) hb[(int64_t)rho * F3_H2] = h;
        uint32_t p0, p1;
        bx_split2((inb ? h : 0.f) * X_ASCALE, 0.f, p0, p1);
        *reinterpret_cast<uint16_t*>(awr + rho * F3_A2ROW) = (uint16_t)p0;
        *reinterpret_cast<uint16_t*>(awr + rho * F3_A2ROW + F3_A2PL) = (uint16_t)p1;
      }
    }
    __syncthreads();      // the h2 image is complete; nobody reads the h1 image any more
    // ---- third layer (K = 256, N = 128): column tile w & 3, k-half w >> 2
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
      const int kh = w >> 2;
      const char* ard = A2img + li * F3_A2ROW + lh * 16 + kh * 8 * 32;
      u32x4 bx[F2_PF][X_NP];
#pragma unroll
      for (int u = 0; u < F2_PF; ++u)
#pragma unroll
        for (int p = 0; p < X_NP; ++p) bx[u][p] = W3x[(int64_t)(8 * kh + u) * w3_step + p * 64];
#define F3_L3_BLOCK(QU, REFILL)                                                                                         \
  {                                                                                                                     \
    u32x4 av[X_NP];                                                                                                     \
    _Pragma("unroll") for (int p = 0; p < X_NP; ++p) av[p] = *reinterpret_cast<const u32x4*>(ard + (QU) * 32 + p * F3_A2PL); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][1]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[1]), __builtin_bit_cast(f16x8, bx[u][0]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[0]), __builtin_bit_cast(f16x8, bx[u][0]), acc, 0, 0, 0); \
    if (REFILL) {                                                                                                       \
      _Pragma("unroll") for (int p = 0; p < X_NP; ++p) bx[u][p] = W3x[(int64_t)(8 * kh + (QU) + F2_PF) * w3_step + p * 64]; \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }
      static_assert(F2_PF == 4, "the third layer's eight blocks are two groups of F2_PF");
#pragma unroll
      for (int u = 0; u < F2_PF; ++u) F3_L3_BLOCK(u, true)
#pragma unroll
      for (int u = 0; u < F2_PF; ++u) F3_L3_BLOCK(F2_PF + u, false)
#undef F3_L3_BLOCK
      if (kh) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part3[((w & 3) * 16 + r) * 64 + lane] = acc[r];
      }
    }
    __syncthreads();
    if (w < 4) {
      float* hb = a.H3 ? a.H3 + (r0 + 4 * lh) * F3_H3 + w * 32 + li : nullptr;
      float* hs = H3s + 4 * lh * F3_H3S + w * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2);
        const float h = act_fwd_t<ACT>(fmaf(acc[r] + part3[(w * 16 + r) * 64 + lane], so, b3v));
        if (hb && r0 + rho + 4 * lh < M) hb[(int64_t)rho * F3_H3] = h;
        hs[rho * F3_H3S] = h;
      }
    }
    __syncthreads();      // h3 (fp32) is complete; the h2 image and the partials are free
    // ---- head (exact fp32): out[r][o] = bh[o] + sum_k h3[r][k] Wh[k][o]
    if (NTH == 0) {
      float* part = Whs + F3_H3;                      // [16][32]
      const int r = t & 31, g = t >> 5;               // 16 threads per row, 8 k each
      const hl_f4* h4 = reinterpret_cast<const hl_f4*>(H3s + r * F3_H3S + 8 * g);
      const hl_f4* w4 = reinterpret_cast<const hl_f4*>(Whs + 8 * g);
      float s_ = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const hl_f4 hv = h4[q], wv = w4[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) s_ = fmaf(hv[e], wv[e], s_);
      }
      part[g * 32 + r] = s_;
      __syncthreads();
      if (t < 32 && r0 + t < M) {
        float o = a.bh[0];
#pragma unroll
        for (int q = 0; q < 16; ++q) o += part[q * 32 + t];
        a.OUT[r0 + t] = o;
      }
    } else {
      constexpr int NH = NTH > 0 ? NTH : 1;
      f32x16 ah[NH];
#pragma unroll
      for (int j = 0; j < NH; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ah[j][r] = 0.f;
      const float* hrd = H3s + li * F3_H3S + 16 * w + lh;
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const float av = hrd[2 * s_];
#pragma unroll
        for (int j = 0; j < NH; ++j) ah[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, whr[j][s_], ah[j], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        hp0[(w * 32 + row) * 33 + li] = ah[0][r];
        if (NTH == 2 && li < 16) hp1[(w * 32 + row) * 17 + li] = ah[NH - 1][r];
      }
      __syncthreads();
      const int r = t & 31;
      for (int c = t >> 5; c < OD; c += 16) {
        float o = a.bh[c];
#pragma unroll
        for (int q = 0; q < F2_NW; ++q) o += c < 32 ? hp0[(q * 32 + r) * 33 + c] : hp1[(q * 32 + r) * 17 + c - 32];
        if (r0 + r < M) a.OUT[(r0 + r) * OD + c] = o;
      }
    }
  }
}

bool fwd3h_supported(const rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, int64_t M, int ldx,
                     const void** wx) {
  if (!ctx->fwd2h || !ctx->gemm_bx || d.n_hidden != 3 || d.hidden[0] != F3_H1 || d.hidden[1] != F3_H2 || d.hidden[2] != F3_H3 || !d.ln_first) return false;
  if (d.act != RLX_ACT_ELU || d.in_dim <= 32 || d.in_dim > 4 * F2_MAXV * F2_THREADS / F2_ROWS || d.out_dim > 48 || M < 1024) return false;
  if (ldx > 0 && ldx < d.in_dim) return false;
  for (int l = 0; l < 3; ++l) {
    wx[l] = bx_lookup(ctx, params + L.layer[l].W, 0, L.layer[l].in, L.layer[l].out);
    if (!wx[l]) return false;
  }
  return true;
}

// z1 / h1 / h2 / h3 may be NULL (forward-only pass).  tw (optional): the second net of a twin launch -- same x, same shapes.
int launch_fwd3h(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const void* const* wx, const float* x, int ldx,
                 float* z1, float* h1, float* h2, float* h3, float* out, int64_t M, hipStream_t st, const Fwd3hTwin* tw) {
  const LayerOff &o0 = L.layer[0], &o1 = L.layer[1], &o2 = L.layer[2];
  const int K1 = o0.in, OD = L.head.out, ld = ldx > 0 ? ldx : K1;
  Fwd3hArgs a;
  a.X = x; a.W1x = wx[0]; a.W2x = wx[1]; a.W3x = wx[2]; a.b1 = params + o0.b; a.g = params + o0.g; a.be = params + o0.be;
  a.b2 = params + o1.b; a.b3 = params + o2.b; a.Wh = params + L.head.W; a.bh = params + L.head.b;
  a.Z1 = z1; a.H1 = h1; a.H2 = h2; a.H3 = h3; a.OUT = out;
  Fwd3hArgs a2 = a;
  if (tw) {
    const float* p = tw->params;
    a2.W1x = tw->wx[0]; a2.W2x = tw->wx[1]; a2.W3x = tw->wx[2]; a2.b1 = p + o0.b; a2.g = p + o0.g; a2.be = p + o0.be; a2.b2 = p + o1.b;
    a2.b3 = p + o2.b; a2.Wh = p + L.head.W; a2.bh = p + L.head.b; a2.Z1 = tw->z1; a2.H1 = tw->h1; a2.H2 = tw->h2; a2.H3 = tw->h3; a2.OUT = tw->out;
  }
  const int KB1 = 2 * div_up(K1, 32);
  const int xrow = 128 * div_up(KB1, 4) + 16;
  const int nth = OD == 1 ? 0 : (OD <= 32 ? 1 : 2);
  const size_t xbytes = (size_t)2 * F2_ROWS * xrow, x2bytes = (size_t)2 * F3_A2PL + (size_t)4 * 16 * 64 * sizeof(float);
  const int a1off = (int)((xbytes > x2bytes ? xbytes : x2bytes) + 15) & ~15;
  const size_t a1bytes = (size_t)2 * F3_A1PL, h3bytes = (size_t)F2_ROWS * F3_H3S * sizeof(float) + (size_t)F2_NW * 32 * 33 * sizeof(float);
  const int soff = a1off + (int)((a1bytes > h3bytes ? a1bytes : h3bytes) + 15 & ~(size_t)15);
  const size_t lds = (size_t)soff + ((size_t)2 * F2_NW * 32 + F2_NW * 64 + F3_H3 + 16 * 32) * sizeof(float);
  RLX_REQUIRE(lds <= 160 * 1024, RLX_EUNSUP, "fwd3h: tile exceeds the LDS");
  const double nets = tw ? 2.0 : 1.0;
  ProfScope prof(ctx, PK_FWD2H, nets * 2.0 * (double)M * ((double)K1 * F3_H1 + (double)F3_H1 * F3_H2 + (double)F3_H2 * F3_H3 + (double)F3_H3 * OD), st,
                 nets * 4.0 * ((double)M * (K1 / nets + OD + (z1 ? F3_H1 : 0) + (h1 ? F3_H1 : 0) + (h2 ? F3_H2 : 0) + (h3 ? F3_H3 : 0)) +
                               (double)K1 * F3_H1 + (double)F3_H1 * F3_H2 + (double)F3_H2 * F3_H3 + (double)F3_H3 * OD),
                 M, F3_H1, K1, 1);
  const int64_t nt = (M + F2_ROWS - 1) / F2_ROWS;
  const int grid = (int)(nt < ctx->num_cus ? nt : ctx->num_cus);
#define RLX_F3_GO(KERNEL, GRID)                                                                                            \
  {                                                                                                                        \
    static AttrOnce attr_set;                                                                                                \
    if (!attr_set.done()) {                                                                                                       \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      attr_set.mark();                                                                                                       \
    }                                                                                                                      \
    RLX_PLAUNCH((KERNEL), GRID, dim3(F2_THREADS), lds, st, a, a2, M, ld, K1, OD, xrow, a1off, soff);                       \
  }
#define RLX_F3_LAUNCH(NTHV)                                                                                                \
  {                                                                                                                        \
    if (tw) RLX_F3_GO((k_fwd3h<RLX_ACT_ELU, true, NTHV>), dim3(grid, 2))                                                   \
    else RLX_F3_GO((k_fwd3h<RLX_ACT_ELU, false, NTHV>), dim3(grid))                                                        \
  }
  if (nth == 0) RLX_F3_LAUNCH(0)
  else if (nth == 1) RLX_F3_LAUNCH(1)
  else RLX_F3_LAUNCH(2)
#undef RLX_F3_LAUNCH
#undef RLX_F3_GO
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

}  // namespace rlx
