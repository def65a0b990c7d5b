// l1fused.hip -- the whole first-layer backward folded into the second layer's input-gradient
// GEMM: per 32-row tile
//     dH1 = dZ2 @ W2^T                         (exact-fp32 MFMA, K = hidden[1])
//     z1  = X @ W1 + b1  (recomputed, K = obs)  -> LayerNorm statistics, xhat, h1 = act(...)
//     dZ1 = LN'( dH1 * act'(h1) )              (all in MFMA accumulator registers)
//     dW1 += X^T dZ1, db1 += colsum(dZ1), dgamma += colsum(dy*xhat), dbeta += colsum(dy)
// Nothing of shape [M, hidden[0]] is written to or re-read from HBM: this replaces k_gemm_dx
// (layer 1), k_l1<bwd> and k_gemm_dw_skinny and removes ~256 MB of traffic per network and
// update at mb = 32768 (the reverse-mode gradient of ppo/flax_full_jit/policy.py:31-34 /
// critic.py:22-25 inside jax.value_and_grad, rl_x/algorithms/ppo/flax/ppo.py:189).
//
// Workgroup = 32 rows x ALL hidden[0] columns (4 waves x NT 32-column MFMA tiles, hidden[0] =
// 128*NT), so LayerNorm row reductions stay on chip: DPP + v_permlane16_swap inside each
// 32-lane half (one accumulator register = one row per half), then a 4-wave LDS combine.
// W1 stays resident in LDS.  W2 is re-laid out once per update in MFMA-B-fragment order
// (k_frag_reorder) so every lane streams its B operand straight from L2 with fully coalesced 16-B
// loads -- no LDS stage and no barrier inside the K loop (at a 32-row tile the LDS write
// bandwidth of a staged B tile, 64 KB per 4096 MFMA cycles, was the bottleneck).  The
// contraction order inside a K-group of 8 is permuted (half-wave lh takes k = 8q + 4 lh + i);
// fp32 addition order only.  Workgroups are persistent over row tiles and keep the
// dW1/db1/dgamma/dbeta partial sums in registers; one slab per workgroup goes to the
// deterministic slab reduction.
#include <type_traits>

#include "mlp.h"
#include "gemm_bx.h"

namespace rlx {

#ifndef RLX_LF_PFX
#define RLX_LF_PFX 2   // must divide the number of 16-k blocks (N2 / 16); MEASURED 4: 67.8 vs 65.8 us (27 spilled registers)
#endif
#ifndef RLX_LF_DWSCALE
#define RLX_LF_DWSCALE 1   // 1: the dW1 operand dZ1 is scaled by gs / (the row tile's largest 1 / std(z1), as a power of two): dZ1 = LN'(.)
                           // carries 1 / std(z1) and shrinks with the observations' magnitude; 0: fixed gradient scale (A/B)
#endif
#ifndef RLX_LF_ABL
#define RLX_LF_ABL 0   // timing ablation of k_dx_l1bwd (round-4 one-off scripts, git history): 1 no main product, 2 nothing after it, 4 no dW1 MFMAs, 8 no z1 MFMAs
#endif
constexpr int LF_ROWS = 32;
constexpr int LF_THREADS = 256;
constexpr int LF_XS = 33;  // Xs[row][k] stride
constexpr int LFP_N2 = 256;  // hidden[1] of the specialised (double-buffered / pipelined) paths

typedef float lf_v4 __attribute__((ext_vector_type(4)));

// sum over the 32 lanes of each half-wave (lanes 0-31 / 32-63 hold different rows): 4 DPP steps
// inside rows of 16 lanes, then v_permlane16_swap to fold the two 16-lane rows of each half.
__device__ __forceinline__ float half_sum(float v) {
  v += dpp_f(v, 0);
  v += dpp_f(v, 1);
  v += dpp_f(v, 2);
  v += dpp_f(v, 3);
  const unsigned u = (unsigned)__float_as_int(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
}

struct L1FusedArgs {
  const float* X;      // [M, O]
  const float* dZ2;    // [M, N2]
  const float* W2t;    // [N2, H1]  (transposed copy of W2[H1, N2])
  const float* W1;     // [O, H1]
  const float* b1;
  const float* g;      // LN scale or null
  const float* be;     // LN bias or null
  float* partials;     // [grid][(O + 3) * H1] : dW1 [O][H1], db1, dgamma, dbeta
  int64_t M;
  int O, H1, N2, act, ln;
  const void* W2x;     // BX kernels: fragment-ordered split image of B(k = n2, j = h1 column) (gemm_bx.h), NTx column tiles
  int NTx;
  float gs, gso;       // BX kernels: power-of-two scale of the dZ2 operand and 1 / (gs * X_WSCALE)
  const void* W1x;     // BX kernels: fragment-ordered split image of B(k = obs index, j = h1 column), K padded to 32 (two 16-k blocks)
  const uint32_t* xmax;  // BX kernels, optional: DEVICE bit pattern of max |X| -> power-of-two scale of the X planes (common.h: x_scale_from_max); NULL: X_ASCALE
};

// BX: LDS image of the dZ2 row tile as two fp16 planes (gemm_bx.h), [32 rows][N2 k] with 2 * N2 bytes per row; the 16-byte k-slots of a
// row are XOR-swizzled with the low four row bits (N2 % 128 == 0), which makes the 16 rows of every ds_read_b128 service
// group hit 16 different slots of the bank row
__device__ __forceinline__ int lf_bx_off(int r, int ks, int N2) { return r * 2 * N2 + ((ks ^ (r & 15)) << 4); }
__device__ __forceinline__ void lf_bx_stage4(char* __restrict__ img, int r, int c4, lf_v4 v, int N2, float sc) {
  uint32_t a0, a1, b0, b1;
  bx_split2(v[0] * sc, v[1] * sc, a0, a1);
  bx_split2(v[2] * sc, v[3] * sc, b0, b1);
  char* d = img + lf_bx_off(r, c4 >> 3, N2) + ((c4 & 4) << 1);
  const int plane = LF_ROWS * 2 * N2;
  *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
  *reinterpret_cast<u32x2*>(d + plane) = u32x2{a1, b1};
}

// BX: the X row tile [32 rows][obs index < 32] as fp16 planes (times X_ASCALE) in TWO orientations, 2 KiB per plane each:
//   XA[row][k = obs]       A operand of the z1 recompute              (fragment lane: row li, k = 8 * (2 s + lh) + e)
//   XT[obs][pos(row)]      A operand of dW1 = X^T dZ1, contraction over the tile's rows.  The B operand of that product is built
//                          from the accumulator registers, whose lane (column li, half lh) holds the rows rho(r, lh) =
//                          (r & 3) + 8 (r >> 2) + 4 lh: k-slot e of 16-k step s is row 16 s + (e & 3) + 8 (e >> 2) + 4 lh, i.e. the
//                          rows sit at pos = row with bits 2 and 3 exchanged.
// Both images use bx_off's swizzle (64-byte rows of four 16-byte k-slots).
constexpr int LF_XPLANE = LF_ROWS * X_ROWB;                  // 2 KiB
constexpr int LF_XIMG = 2 * X_NP * LF_XPLANE;                // XA planes | XT planes: 8 KiB per buffer
__device__ __forceinline__ void lf_x_stage(char* __restrict__ img, int r, int k, float v, float xs) {
  uint32_t p0, p1;
  bx_split2(v * xs, 0.f, p0, p1);
  const int pos = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
  char* da = img + bx_off(r, k >> 3) + (k & 7) * 2;
  char* dt = img + X_NP * LF_XPLANE + bx_off(k, pos >> 3) + (pos & 7) * 2;
  *reinterpret_cast<uint16_t*>(da) = (uint16_t)p0;
  *reinterpret_cast<uint16_t*>(da + LF_XPLANE) = (uint16_t)p1;
  *reinterpret_cast<uint16_t*>(dt) = (uint16_t)p0;
  *reinterpret_cast<uint16_t*>(dt + LF_XPLANE) = (uint16_t)p1;
}

// NW waves per workgroup, NT 32-column MFMA tiles per wave: hidden[0] = 32 * NT * NW.  NW = 8 puts two
// waves on every SIMD so one wave's VALU-heavy LayerNorm epilogue fills the issue slots the other
// leaves idle (a single wave per SIMD ran the epilogue at ~7 cycles per instruction).
// TWIN: grid.y == 2, blockIdx.y == 1 works on the argument set a2 (the second of two equally shaped networks on the same rows)
template <int NT, int NW, int ACT, bool LN, bool BX, bool TWIN = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void k_dx_l1bwd(L1FusedArgs a, L1FusedArgs a2) {
  if (TWIN && blockIdx.y) a = a2;
  constexpr int H1 = 32 * NT * NW;
  constexpr int NTHREADS = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int OP = (a.O + 1) & ~1;              // obs dim padded to the MFMA k-step
  float* W1s = smem;                          // [OP][H1]   (exact-fp32 form only: BX takes W1 from its fragment image)
  // the dZ2 / X tiles are double buffered when N2 == LFP_N2: the next tile's rows are fetched into registers before the
  // main loop and stored into the other buffer after it, so their HBM latency hides under 256 MFMAs and the tile costs
  // one staging barrier instead of two exposed round trips (3.5 us of a 28 us tile)
  const bool dbuf = a.N2 == LFP_N2;
  const int nbuf = dbuf ? 2 : 1;
  float* As0 = W1s + (BX ? 0 : OP * H1);              // nbuf x [32][N2+4] dZ2 row tile (BX: nbuf x 2 fp16 planes [32][N2])
  const int a_img = BX ? X_NP * LF_ROWS * a.N2 / 2 : LF_ROWS * (a.N2 + 4);   // floats per buffer
  float* Xs0 = As0 + nbuf * a_img;                    // nbuf x [32][33]   X tile (cols >= O zero); BX: nbuf x LF_XIMG bytes of planes
  const int x_img = BX ? LF_XIMG / 4 : LF_ROWS * LF_XS;   // floats per buffer
  float* red = Xs0 + nbuf * x_img;                    // [2 phases][2 stats][NW][32]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  const bool lb0 = (lane & 1) != 0, lb1 = (lane & 2) != 0;
  const int O = a.O, N2 = a.N2;
  constexpr bool ln = LN;
  // scale of the X planes: from the device-side max |x| of the pass when the caller supplies it (observations of any magnitude
  // stay inside fp16's window at full precision), else the activations' fixed x16
  const float xs = (BX && a.xmax) ? x_scale_from_max(*a.xmax, X_ASCALE) : X_ASCALE;
  const float xinv = 1.0f / xs;       // exact: a power of two

  if (!BX)
    lds_stage<NTHREADS, float>(W1s, a.W1, OP * H1, O * H1, 0.f);
  float bias[NT], gam[NT], bet[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = w * 32 * NT + 32 * j + li;
    bias[j] = a.b1[col];
    gam[j] = ln ? a.g[col] : 1.f;
    bet[j] = ln ? a.be[col] : 0.f;
  }
  f32x16 dW[NT];
  float dgam[NT], dbet[NT], db1[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    dgam[j] = dbet[j] = db1[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dW[j][r] = 0.f;
  }
  const float invH = 1.0f / (float)H1;
  float dw_scale = (RLX_LF_DWSCALE && LN) ? 0.f : a.gs;   // BX: power-of-two scale of the dW1 operand dZ1 (0: not set yet)
  const int AS = N2 + 4;                      // dZ2 tile row stride: 16-B aligned, conflict-free ds_read_b128
  const int nq = N2 >> 3;                     // K-groups of 8
  constexpr int PF = 4;                       // K-groups of B fragments in flight (registers)
  const lf_v4* __restrict__ Wf = reinterpret_cast<const lf_v4*>(a.W2t);  // [nq][2][H1] float4
  const u32x4* __restrict__ Wx = reinterpret_cast<const u32x4*>(a.W2x) + (int64_t)(w * NT) * X_NP * 64 + lane;   // BX
  const int wx_step = a.NTx * X_NP * 64;                                     // u32x4 entries per 16-k block
  const u32x4* __restrict__ W1x = reinterpret_cast<const u32x4*>(a.W1x) + (int64_t)(w * NT) * X_NP * 64 + lane;   // BX, same tiling

  const int64_t ntiles = (a.M + LF_ROWS - 1) / LF_ROWS;
  constexpr int SA_N = LF_ROWS * (LFP_N2 / 4) / NTHREADS, SX_N = LF_ROWS * 32 / NTHREADS;
  lf_v4 sa[SA_N];
  float sx[SX_N];
  auto stage_load = [&](int64_t tl) {          // dbuf only (N2 == LFP_N2)
    const int64_t rr = tl * LF_ROWS;
#pragma unroll
    for (int c = 0; c < SA_N; ++c) {
      const int i = t + c * NTHREADS, r = i / (LFP_N2 / 4), c4 = (i % (LFP_N2 / 4)) * 4;
      sa[c] = (rr + r < a.M) ? *reinterpret_cast<const lf_v4*>(a.dZ2 + (rr + r) * LFP_N2 + c4) : lf_v4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < SX_N; ++c) {
      const int i = t + c * NTHREADS, r = i >> 5, k = i & 31;
      sx[c] = (k < O && rr + r < a.M) ? a.X[(rr + r) * O + k] : 0.f;
    }
  };
  auto stage_store = [&](int b) {
    float* Ad = As0 + b * a_img;
    float* Xd = Xs0 + b * x_img;
#pragma unroll
    for (int c = 0; c < SA_N; ++c) {
      const int i = t + c * NTHREADS, r = i / (LFP_N2 / 4), c4 = (i % (LFP_N2 / 4)) * 4;
      if (BX) lf_bx_stage4(reinterpret_cast<char*>(Ad), r, c4, sa[c], LFP_N2, a.gs);
      else *reinterpret_cast<lf_v4*>(Ad + r * (LFP_N2 + 4) + c4) = sa[c];
    }
#pragma unroll
    for (int c = 0; c < SX_N; ++c) {
      const int i = t + c * NTHREADS;
      if (BX) lf_x_stage(reinterpret_cast<char*>(Xd), i >> 5, i & 31, sx[c], xs);
      else Xd[(i >> 5) * LF_XS + (i & 31)] = sx[c];
    }
  };
  if (dbuf && (int64_t)blockIdx.x < ntiles) {
    stage_load(blockIdx.x);
    stage_store(0);
  }
  int buf = 0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= (dbuf ? 1 : 0)) {
    const int64_t r0 = tile * LF_ROWS;
    float* As = As0 + buf * a_img;
    float* Xs = Xs0 + buf * x_img;
    const bool has_next = tile + gridDim.x < ntiles;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    lf_v4 bq[BX ? 1 : PF][NT];
    constexpr int PFX = RLX_LF_PFX;           // BX: 16-k blocks of weight fragments in flight
    u32x4 bx[BX ? PFX : 1][NT][X_NP];
    if (BX) {
#pragma unroll
      for (int u = 0; u < PFX; ++u)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int p = 0; p < X_NP; ++p) bx[u][j][p] = Wx[(int64_t)u * wx_step + (j * X_NP + p) * 64];
    } else {
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int j = 0; j < NT; ++j) bq[u][j] = Wf[(int64_t)((u * 2 + lh) * H1) + w * 32 * NT + 32 * j + li];
    }
    __syncthreads();  // previous tile's readers of Xs / As are done; (dbuf) this tile's LDS image, stored a tile ago, is visible
    if (dbuf) {
      // in flight during the main loop.  (MEASURED, mb 32768: requesting the rows BEHIND the main loop and storing them at the
      // end of the tile -- so that the loop's first in-order vmcnt wait does not sit out their HBM latency -- is slower, 96.0
      // vs 87.5 us.  Ablation of this kernel by phase, same shape: main product 41 us, element-wise + dW1 phases 29 us, tile
      // staging / launch / slab store 17 us; the phases do not overlap: all eight waves are in the same one.)
      if (has_next) stage_load(tile + gridDim.x);
    } else {
      for (int i = t; i < LF_ROWS * 32; i += NTHREADS) {
        const int r = i >> 5, k = i & 31;
        const float xv = (k < O && r0 + r < a.M) ? a.X[(r0 + r) * O + k] : 0.f;
        if (BX) lf_x_stage(reinterpret_cast<char*>(Xs), r, k, xv, xs);
        else Xs[r * LF_XS + k] = xv;
      }
      for (int i = t; i < LF_ROWS * (N2 >> 2); i += NTHREADS) {   // whole dZ2 row tile [32][N2]
        const int r = i / (N2 >> 2), c4 = (i - r * (N2 >> 2)) * 4;
        const lf_v4 v = (r0 + r < a.M) ? *reinterpret_cast<const lf_v4*>(a.dZ2 + (r0 + r) * N2 + c4) : lf_v4{0.f, 0.f, 0.f, 0.f};
        if (BX) lf_bx_stage4(reinterpret_cast<char*>(As), r, c4, v, N2, a.gs);
        else *reinterpret_cast<lf_v4*>(As + r * AS + c4) = v;
      }
      __syncthreads();
    }
    // ---- main GEMM: dH1 tile; barrier-free K loop
    const float* a0 = As + li * AS + 4 * lh;
    if (RLX_LF_ABL & 1) {
    } else if (BX) {
      // split-fp32 operands on the half-precision pipe: three MFMAs per 16 k and column tile (gemm_bx.h)
      const char* img = reinterpret_cast<const char*>(As);
      const int plane = LF_ROWS * 2 * N2, nb16 = N2 >> 4;
      for (int q = 0; q < nb16; q += PFX) {
#pragma unroll
        for (int u = 0; u < PFX; ++u) {
          u32x4 av[X_NP];
          const char* ab = img + lf_bx_off(li, 2 * (q + u) + lh, N2);
#pragma unroll
          for (int p = 0; p < X_NP; ++p) av[p] = *reinterpret_cast<const u32x4*>(ab + p * plane);
#define RLX_LF_BX_STEP(P, Q)                                                                                      \
  _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                  \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[P]),                           \
                                                      __builtin_bit_cast(f16x8, bx[u][j][Q]), acc[j], 0, 0, 0);
          RLX_LF_BX_STEP(0, 1)
          RLX_LF_BX_STEP(1, 0)
          RLX_LF_BX_STEP(0, 0)
#undef RLX_LF_BX_STEP
          if (q + u + PFX < nb16) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
              for (int p = 0; p < X_NP; ++p) bx[u][j][p] = Wx[(int64_t)(q + u + PFX) * wx_step + (j * X_NP + p) * 64];
          }
        }
      }
    } else
    for (int q = 0; q < nq; q += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const lf_v4 av = *reinterpret_cast<const lf_v4*>(a0 + 8 * (q + u));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bq[u][j][i], acc[j], 0, 0, 0);
        if (q + u + PF < nq) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
            bq[u][j] = Wf[(int64_t)(((q + u + PF) * 2 + lh) * H1) + w * 32 * NT + 32 * j + li];
        }
      }
    }
    if (BX) {   // back to unscaled dH1
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] *= a.gso;
    }
    if (dbuf && has_next) stage_store(buf ^ 1);        // its last readers finished before this tile's first barrier
    if (RLX_LF_ABL & 2) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[j][r]));
      continue;
    }
    // ---- recompute z1 = X @ W1 + b1 in the same accumulator layout
    f32x16 z[NT];
    if (BX) {
      // z1 on the fp16 pipe: A = XA planes (LDS), B = W1's fragment image (two 16-k blocks, L1 / L2 resident), three plane
      // products per block; the accumulators start at bias * X_ASCALE * X_WSCALE and leave the scale in one multiply
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[j][r] = bias[j] * (xs * X_WSCALE);
      const char* xa = reinterpret_cast<const char*>(Xs);
      const int nks = O > 16 ? 2 : 1;           // (uniform) obs indices >= 16 live in the second 16-k block
      for (int s_ = 0; s_ < ((RLX_LF_ABL & 8) ? 0 : nks); ++s_) {
        u32x4 xf[X_NP];
#pragma unroll
        for (int p = 0; p < X_NP; ++p) xf[p] = *reinterpret_cast<const u32x4*>(xa + p * LF_XPLANE + bx_off(li, 2 * s_ + lh));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const u32x4 w0 = W1x[(int64_t)s_ * wx_step + (j * X_NP + 0) * 64], w1 = W1x[(int64_t)s_ * wx_step + (j * X_NP + 1) * 64];
          z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[0]), __builtin_bit_cast(f16x8, w1), z[j], 0, 0, 0);
          z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[1]), __builtin_bit_cast(f16x8, w0), z[j], 0, 0, 0);
          z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[0]), __builtin_bit_cast(f16x8, w0), z[j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[j][r] *= xinv * X_WINV;
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[j][r] = bias[j];
      const float* x0 = Xs + li * LF_XS + lh;
      const float* w0 = W1s + lh * H1 + w * 32 * NT + li;
      for (int kk = 0; kk < ((RLX_LF_ABL & 8) ? 0 : OP); kk += 2) {
        const float av = x0[kk];
#pragma unroll
        for (int j = 0; j < NT; ++j)
          z[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w0[kk * H1 + 32 * j], z[j], 0, 0, 0);
      }
    }
    // accumulator register r of half lh is row rho = (r&3) + 8*(r>>2) + 4*lh of the tile
    // LayerNorm row statistics.  Per-wave partials (4 rows per 16-B LDS store, lane 0 of each half), then
    // ONE wave folds the NW partials per row in fixed order, then every lane fetches its 16 rows with four
    // 16-B reads per statistic (row rho(r, lh) = 8*(r>>2) + 4*lh + (r&3): registers 4g..4g+3 are contiguous rows).
    constexpr bool red_on = ln;
    // per-wave partial sums [2 stats][NW][32 rows]; after ONE barrier every wave folds the NW partials for itself (fixed
    // order: the same bits in every wave) into a wave-private slot and reads its 16 rows back -- same-wave LDS write ->
    // read needs no barrier, so a reduction costs one workgroup barrier instead of two
    float* redA = red;                       // [2][NW][32]
    float* redB = red + 2 * NW * 32;         // [2][NW][32]
    float* totA = redB + 2 * NW * 32 + w * 64;            // [2][32], this wave's copy
    float* totB = redB + 2 * NW * 32 + NW * 64 + w * 64;  // [2][32], this wave's copy
    if (red_on) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float sv[4], ssv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float s = 0.f, ss = 0.f;
#pragma unroll
          for (int j = 0; j < NT; ++j) { s += z[j][r]; ss += z[j][r] * z[j][r]; }
          sv[e] = s;
          ssv[e] = ss;
        }
        const float st = half_sum4(sv[0], sv[1], sv[2], sv[3], lb0, lb1);       // row 8g + 4lh + (li & 3)
        const float sst = half_sum4(ssv[0], ssv[1], ssv[2], ssv[3], lb0, lb1);
        if (li < 4) {
          redA[(0 * NW + w) * 32 + 8 * g + 4 * lh + li] = st;
          redA[(1 * NW + w) * 32 + 8 * g + 4 * lh + li] = sst;
        }
      }
      __syncthreads();
      {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < NW; ++q) v += redA[((lane >> 5) * NW + q) * 32 + (lane & 31)];
        totA[lane] = v;
      }
      if (BX && RLX_LF_DWSCALE && dw_scale == 0.f) {
        // scale of the dW1 operand (see the dW1 block below): gs / (largest 1 / std(z1) of this tile's 32 rows, as a power of
        // two) -- one value per lane from the wave's copy of the row statistics, a wave maximum, and the result lives in an SGPR
        const float mean = totA[lane & 31] * invH;
        float rs = rsqrtf(fmaxf(0.f, totA[32 + (lane & 31)] * invH - mean * mean) + 1e-6f);
        if (r0 + (lane & 31) >= a.M) rs = 0.f;        // rows past the end of a ragged last tile are all-zero observations
        rs = fmaxf(rs, dpp_f(rs, 0));
        rs = fmaxf(rs, dpp_f(rs, 1));
        rs = fmaxf(rs, dpp_f(rs, 2));
        rs = fmaxf(rs, dpp_f(rs, 3));
        const unsigned u = (unsigned)__float_as_int(rs);
        const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        rs = fmaxf(__int_as_float((int)r16[0]), __int_as_float((int)r16[1]));
        dw_scale = a.gs * x_pow2_inv(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rs))));
      }
    }
    // dy = dH1 * act'(h);  z <- xhat;  acc <- d xhat;  row sums m1, m2
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      lf_v4 sv = {0.f, 0.f, 0.f, 0.f}, ssv = {0.f, 0.f, 0.f, 0.f};
      float a1v[4], a2v[4];
      if (red_on) {
        sv = *reinterpret_cast<const lf_v4*>(totA + 8 * g + 4 * lh);
        ssv = *reinterpret_cast<const lf_v4*>(totA + 32 + 8 * g + 4 * lh);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        float mean = 0.f, rstd_r = 1.f;
        if (red_on) {
          mean = sv[e] * invH;
          rstd_r = rsqrtf(fmaxf(0.f, ssv[e] * invH - mean * mean) + 1e-6f);
        }
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float xh = (z[j][r] - mean) * rstd_r;
          const float y = ln ? xh * gam[j] + bet[j] : z[j][r];
          const float dy = acc[j][r] * act_grad_pre_t<ACT>(y);
          dgam[j] += dy * xh;
          dbet[j] += dy;
          const float dxh = dy * gam[j];
          z[j][r] = xh;
          acc[j][r] = dxh;
          a1 += dxh;
          a2 += dxh * xh;
        }
        a1v[e] = a1;
        a2v[e] = a2;
      }
      if (red_on) {
        const float a1t = half_sum4(a1v[0], a1v[1], a1v[2], a1v[3], lb0, lb1);   // row 8g + 4lh + (li & 3)
        const float a2t = half_sum4(a2v[0], a2v[1], a2v[2], a2v[3], lb0, lb1);
        if (li < 4) {
          redB[(0 * NW + w) * 32 + 8 * g + 4 * lh + li] = a1t;
          redB[(1 * NW + w) * 32 + 8 * g + 4 * lh + li] = a2t;
        }
      }
    }
    if (red_on) {
      __syncthreads();
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) v += redB[((lane >> 5) * NW + q) * 32 + (lane & 31)];
      totB[lane] = v * invH;
    }
    // dZ1 (in acc), bias gradient, and dW1 += X^T dZ1 with the accumulator registers as the B operand:
    // MFMA step r contracts row rho(r,0) (lanes 0-31) and row rho(r,1) (lanes 32-63).
    const float* xt = Xs + li;  // exact form, A operand: A[i = obs index li][k = lh] = X[rho(r, lh)][li]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      lf_v4 m1v = {0.f, 0.f, 0.f, 0.f}, m2v = {0.f, 0.f, 0.f, 0.f}, sv2 = {0.f, 0.f, 0.f, 0.f}, ssv2 = {0.f, 0.f, 0.f, 0.f};
      if (red_on) {
        m1v = *reinterpret_cast<const lf_v4*>(totB + 8 * g + 4 * lh);
        m2v = *reinterpret_cast<const lf_v4*>(totB + 32 + 8 * g + 4 * lh);
        sv2 = *reinterpret_cast<const lf_v4*>(totA + 8 * g + 4 * lh);           // the row statistics again (wave-private copy):
        ssv2 = *reinterpret_cast<const lf_v4*>(totA + 32 + 8 * g + 4 * lh);     // 1 / std is recomputed instead of kept in 16 registers
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const int rho = 8 * g + 4 * lh + e;
        float rstd_r = 1.f;
        if (red_on) {
          const float mean = sv2[e] * invH;
          rstd_r = rsqrtf(fmaxf(0.f, ssv2[e] * invH - mean * mean) + 1e-6f);
        }
        const float av = BX ? 0.f : xt[rho * LF_XS];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float dz = ln ? rstd_r * (acc[j][r] - m1v[e] - z[j][r] * m2v[e]) : acc[j][r];
          db1[j] += dz;
          if (BX) acc[j][r] = dz;      // kept for the fp16-pipe product below
          else if (!(RLX_LF_ABL & 4)) dW[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, dz, dW[j], 0, 0, 0);
        }
      }
    }
    if (BX && !(RLX_LF_ABL & 4)) {
      // dW1 += X^T dZ1 on the fp16 pipe: A = XT planes (LDS), B = the lane's dZ1 values split in registers -- k-slot e of 16-k
      // step s_ is accumulator register 8 s_ + e (see lf_x_stage for the row order).
      // The B operand's scale follows the DATA: dZ1 = LN'(...) carries 1 / std(z1), i.e. it shrinks with the observations'
      // magnitude, and the fixed gradient scale would push it under fp16's full-precision window (measured: observations x 1e4 ->
      // 2e-5 relative error in dW1).  With the tile's largest 1 / std(z1) divided out (as a power of two) the operand sits where
      // it sits for unit-scale observations -- for those the scale IS the fixed one, bit for bit.  (A scale from max |dZ1| itself
      // -- a wave reduction over the finished values -- cost 23 spilled registers and 13 us per launch.)
      // (fixed for the workgroup by its FIRST row tile, above: observation scales do not change from tile to tile, and re-basing
      //  the accumulators per tile, or a scale from max |dZ1| itself, made hipcc spill 23 - 86 registers in this kernel)
      const float sw = dw_scale;
      const char* xtp = reinterpret_cast<const char*>(Xs) + X_NP * LF_XPLANE;
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        u32x4 xf[X_NP];
#pragma unroll
        for (int p = 0; p < X_NP; ++p) xf[p] = *reinterpret_cast<const u32x4*>(xtp + p * LF_XPLANE + bx_off(li, 2 * s_ + lh));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          u32x4 b0, b1;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            uint32_t q0, q1;
            bx_split2(acc[j][8 * s_ + 2 * m] * sw, acc[j][8 * s_ + 2 * m + 1] * sw, q0, q1);
            b0[m] = q0;
            b1[m] = q1;
          }
          dW[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[0]), __builtin_bit_cast(f16x8, b1), dW[j], 0, 0, 0);
          dW[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[1]), __builtin_bit_cast(f16x8, b0), dW[j], 0, 0, 0);
          dW[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[0]), __builtin_bit_cast(f16x8, b0), dW[j], 0, 0, 0);
        }
      }
    }
  }
  // ---- one slab per workgroup
  float* out = a.partials + (int64_t)blockIdx.x * (O + 3) * H1;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = w * 32 * NT + 32 * j + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;  // obs index
      if (row < O) out[(int64_t)row * H1 + col] = BX ? dW[j][r] * (xinv / dw_scale) : dW[j][r];
    }
    // the two halves hold different rows of the same column: fold them
    float v0 = db1[j], v1 = dgam[j], v2 = dbet[j];
    {
      const unsigned u0 = (unsigned)__float_as_int(v0), u1 = (unsigned)__float_as_int(v1), u2 = (unsigned)__float_as_int(v2);
      const auto s0 = __builtin_amdgcn_permlane32_swap(u0, u0, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
      const auto s2 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
      v0 = __int_as_float((int)s0[0]) + __int_as_float((int)s0[1]);
      v1 = __int_as_float((int)s1[0]) + __int_as_float((int)s1[1]);
      v2 = __int_as_float((int)s2[0]) + __int_as_float((int)s2[1]);
    }
    if (lh == 0) {
      out[(int64_t)O * H1 + col] = v0;
      out[(int64_t)(O + 1) * H1 + col] = v1;
      out[(int64_t)(O + 2) * H1 + col] = v2;
    }
  }
}

// ---- first-layer FORWARD on the matrix pipe ---------------------------------------------------------------------
// h1[M, H1] = act(LayerNorm(X[M, O] @ W1 + b1)) for O <= 32.  k_l1<fwd> (mlp.hip) does the K = O product on the VALU: one
// wave per row, 136 FMAs per lane next to the LayerNorm / activation work -- VALU-bound at 28 us for mb = 32768, twice the
// HBM time of its 67 MB store.  Here the product is 9 MFMA steps per 32 x 32 tile (the z1 recompute of k_dx_l1bwd), the
// row statistics use half_sum4 + one workgroup barrier, and the accumulator layout stores 128-B row segments.
// Persistent workgroups (W1 stays in LDS), 8 waves x 64 columns.  Same arithmetic as k_dx_l1bwd's recompute, i.e. the
// forward activations and the backward's recomputed ones agree bit for bit (k_l1<fwd> differs from both in summation
// order only).
template <int NT, int NW, int ACT, bool LN, int KS>      // KS: MFMA k-steps held in registers, (O + 1) / 2 <= KS
__global__ __launch_bounds__(64 * NW, 2) void k_l1fwd_mfma(const float* __restrict__ X, const float* __restrict__ W1,
                                                           const float* __restrict__ b1, const float* __restrict__ g,
                                                           const float* __restrict__ be, float* __restrict__ Hout,
                                                           int64_t M, int O, const int32_t* __restrict__ m_dev,
                                                           int64_t pdelta, float* __restrict__ Hout1) {
  // twin launch (grid.y == 2): blockIdx.y == 1 is a second network of the same first-layer shape on the same rows -- its
  // parameters sit pdelta floats behind the first one's (same layout), its output goes to Hout1
  if (blockIdx.y) { W1 += pdelta; b1 += pdelta; g += pdelta; be += pdelta; Hout = Hout1; }
  if (m_dev && (int64_t)*m_dev < M) M = *m_dev;
  constexpr int H1 = 32 * NT * NW;
  constexpr int NTHREADS = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int OP = (O + 1) & ~1;
  float* Xs = smem;                           // [32][33]
  float* redA = Xs + LF_ROWS * LF_XS;         // [2][NW][32]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  const bool lb0 = (lane & 1) != 0, lb1 = (lane & 2) != 0;
  float* totA = redA + 2 * NW * 32 + w * 64;  // this wave's folded [2][32]
  const int colbase = w * 32 * NT + li;
  // W1 lives in REGISTERS: MFMA step s contracts obs indices 2s (lanes 0-31) and 2s + 1 (lanes 32-63) -- lane (li, lh) needs
  // W1[2s + lh][its column] for every step: (O + 1) / 2 values per column tile, loaded once per workgroup, coalesced over li.
  // (An LDS copy of W1 -- 36 KB filled by every workgroup in front of its two tiles -- cost 5.4 of the kernel's 31 us.)
  float wreg[KS][NT];
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      wreg[s_][j] = (2 * s_ + lh < O) ? W1[(int64_t)(2 * s_ + lh) * H1 + colbase + 32 * j] : 0.f;
  float bias[NT], gam[NT], bet[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    bias[j] = b1[colbase + 32 * j];
    gam[j] = LN ? g[colbase + 32 * j] : 1.f;
    bet[j] = LN ? be[colbase + 32 * j] : 0.f;
  }
  const float invH = 1.0f / (float)H1;
  const int64_t ntiles = (M + LF_ROWS - 1) / LF_ROWS;
  // the X tile of the NEXT row tile travels in registers under this tile's LayerNorm / activation / store passes
  // (MEASURED: taking the A operand straight from global memory instead -- nine strided 4-byte loads per lane and tile, every
  //  wave reading the same 2.2 KB, no LDS tile, one barrier per tile with parity-buffered LayerNorm partials -- is slower,
  //  31.3 vs 26.6 us: the eightfold redundant narrow loads cost more than the two barriers they remove)
  constexpr int XN = LF_ROWS * 32 / NTHREADS;
  float xr[XN];
  auto x_load = [&](int64_t tl) {
#pragma unroll
    for (int c = 0; c < XN; ++c) {
      const int i = t + c * NTHREADS, r = i >> 5, k = i & 31;
      xr[c] = (k < O && tl * LF_ROWS + r < M) ? X[(tl * LF_ROWS + r) * O + k] : 0.f;
    }
  };
  if ((int64_t)blockIdx.x < ntiles) x_load(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * LF_ROWS;
    __syncthreads();  // previous tile's readers of Xs are done
#pragma unroll
    for (int c = 0; c < XN; ++c) {
      const int i = t + c * NTHREADS;
      Xs[(i >> 5) * LF_XS + (i & 31)] = xr[c];
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) x_load(tile + gridDim.x);
    f32x16 z[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) z[j][r] = bias[j];
    {
      const float* x0 = Xs + li * LF_XS + lh;
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) {
        if (2 * s_ < OP) {                    // (uniform)
          const float av = x0[2 * s_];
#pragma unroll
          for (int j = 0; j < NT; ++j) z[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wreg[s_][j], z[j], 0, 0, 0);
        }
      }
    }
    if (LN) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float sv[4], ssv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * gq + e;
          float s_ = 0.f, ss = 0.f;
#pragma unroll
          for (int j = 0; j < NT; ++j) { s_ += z[j][r]; ss += z[j][r] * z[j][r]; }
          sv[e] = s_;
          ssv[e] = ss;
        }
        const float st_ = half_sum4(sv[0], sv[1], sv[2], sv[3], lb0, lb1);
        const float sst = half_sum4(ssv[0], ssv[1], ssv[2], ssv[3], lb0, lb1);
        if (li < 4) {
          redA[(0 * NW + w) * 32 + 8 * gq + 4 * lh + li] = st_;
          redA[(1 * NW + w) * 32 + 8 * gq + 4 * lh + li] = sst;
        }
      }
      __syncthreads();
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) v += redA[((lane >> 5) * NW + q) * 32 + (lane & 31)];
      totA[lane] = v;
    }
    // normalise, activate, store: register r of half lh is row (r&3) + 8*(r>>2) + 4*lh; lanes 0-31 write 128 contiguous bytes
    float* hb = Hout + (r0 + 4 * lh) * H1 + colbase;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      lf_v4 sv = {0.f, 0.f, 0.f, 0.f}, ssv = {0.f, 0.f, 0.f, 0.f};
      if (LN) {
        sv = *reinterpret_cast<const lf_v4*>(totA + 8 * gq + 4 * lh);
        ssv = *reinterpret_cast<const lf_v4*>(totA + 32 + 8 * gq + 4 * lh);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * gq + e;
        const int rho = 8 * gq + e;            // row inside the tile, minus the 4*lh already in hb
        float mean = 0.f, rs = 1.f;
        if (LN) {
          mean = sv[e] * invH;
          rs = rsqrtf(fmaxf(0.f, ssv[e] * invH - mean * mean) + 1e-6f);
        }
        if (r0 + rho + 4 * lh < M) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float xh = (z[j][r] - mean) * rs;
            const float y = LN ? xh * gam[j] + bet[j] : z[j][r];
            hb[(int64_t)rho * H1 + 32 * j] = act_fwd_t<ACT>(y);
          }
        }
      }
    }
  }
}

// ---- first AND second layer forward in one launch -----------------------------------------------------------------
// h1 = act(LayerNorm(X @ W1 + b1)) [M, 512], h2 = act(h1 @ W2 + b2) [M, N2] per 32-row tile without reading h1 back: the
// first layer is the z1 recompute of k_dx_l1bwd<BX> (observation planes x first-layer weight image on the fp16 pipe, row
// statistics with half_sum4 + one barrier), its output goes to HBM once (the layer-2 weight gradient reads it) AND, split into
// fp16 planes, into an LDS image [32 rows][512 k] that is the A operand of the second layer -- whose weight fragments stream
// from the forward split image in L2 exactly as in k_dx_l1bwd's main product.  Replaces k_l1fwd_mfma + k_gemm_bx<0> (layer 2):
// 2 + 67 + 34 MB per launch at 32768 rows instead of 70 + 105 -- the update runs at the ~3 TB/s these kernels reach together
// (profiles/r05_pmc_traffic.md), so bytes are time -- and one dependent launch instead of two.
// 8 waves: wave w owns columns [64 w, 64 w + 64) of layer 1 and columns [NT2 * 32 w, ...) of layer 2.  76 KB of LDS, <= 128
// VGPRs: two workgroups per CU.  TWIN: grid.y == 2, blockIdx.y == 1 takes the second argument set (same shapes, same rows).
#ifndef RLX_L12_STAMPS
#define RLX_L12_STAMPS 0
#endif
struct L12Args {
  const float* X;        // [M, O]
  const void* W1x;       // forward split image of W1 (K = O padded to 32, N = 512)
  const float* b1;
  const float* g;
  const float* be;
  const void* W2x;       // forward split image of W2 (K = 512, N = N2)
  const float* b2;
  float* H1;             // [M, 512] out (read by the layer-2 weight gradient), or NULL: that kernel recomputes it (gemm_bx.hip)
  float* H2;             // [M, N2] out
  float* stats;          // optional [2][M]: LayerNorm mean and 1 / std of every row
  const uint32_t* xmax;  // optional: bit pattern of max |X| (scale of the observation planes; common.h)
  unsigned long long* stamps;   // tuning aid (rlx_dbg_set_stamps): clock64() of thread 0 of workgroup 0 at the phase boundaries of its first two tiles
  int64_t M;
  int O;
};
constexpr int L12_H1 = 512;
constexpr int L12_AROW = 2 * L12_H1 + 16;     // bytes per row of one plane of the h1 image in LDS (padded: see k_l12fwd)
__device__ __forceinline__ void l12_xa_stage(char* __restrict__ img, int r, int k, float v, float xs) {
  uint32_t p0, p1;
  bx_split2(v * xs, 0.f, p0, p1);
  char* da = img + bx_off(r, k >> 3) + (k & 7) * 2;
  *reinterpret_cast<uint16_t*>(da) = (uint16_t)p0;
  *reinterpret_cast<uint16_t*>(da + LF_XPLANE) = (uint16_t)p1;
}

template <int ACT, int NT2, bool TWIN, bool NTS = false>
__global__ __launch_bounds__(512, NT2 == 1 ? 4 : 2) void k_l12fwd(L12Args a, L12Args a2) {
  if (TWIN && blockIdx.y) a = a2;
  constexpr int NW = 8, NT = 2, H1 = L12_H1, NTHREADS = 512, N2 = NW * NT2 * 32;
  // h1 image: rows of 2 * 512 bytes PADDED by 16 (stride 260 dwords: the 16 rows of a ds_read_b128 service group start 4 banks
  // apart -- conflict free without an XOR swizzle, so every store / fragment address is one per-lane base + a compile-time
  // offset; the swizzled form cost ~4 VALU and a live register per element and made hipcc spill 200 registers)
  constexpr int AROW = L12_AROW, APLANE = LF_ROWS * AROW;
  extern __shared__ __attribute__((aligned(16))) char l12_smem[];
  char* Aimg = l12_smem;                                      // 2 planes [32][512 k]
  char* Xs0 = Aimg + X_NP * APLANE;                           // 2 buffers x 2 planes [32][32 k] of observations
  float* redA = reinterpret_cast<float*>(Xs0 + 2 * X_NP * LF_XPLANE);   // [2][NW][32]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  const bool lb0 = (lane & 1) != 0, lb1 = (lane & 2) != 0;
  float* totA = redA + 2 * NW * 32 + w * 64;                  // this wave's folded [2][32]
  const int O = a.O;
  const float xs = a.xmax ? x_scale_from_max(*a.xmax, X_ASCALE) : X_ASCALE;
  const float xinv = 1.0f / xs;
  float bias[NT], gam[NT], bet[NT], b2v[NT2];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = w * 32 * NT + 32 * j + li;
    bias[j] = a.b1[col];
    gam[j] = a.g[col];
    bet[j] = a.be[col];
  }
#pragma unroll
  for (int j = 0; j < NT2; ++j) b2v[j] = a.b2[w * 32 * NT2 + 32 * j + li];
  const u32x4* __restrict__ W1x = reinterpret_cast<const u32x4*>(a.W1x) + (int64_t)(w * NT) * X_NP * 64 + lane;
  constexpr int w1_step = (H1 / 32) * X_NP * 64;              // u32x4 entries per 16-k block of the W1 image
  const u32x4* __restrict__ W2x = reinterpret_cast<const u32x4*>(a.W2x) + (int64_t)(w * NT2) * X_NP * 64 + lane;
  constexpr int w2_step = (N2 / 32) * X_NP * 64;
  const float invH = 1.0f / (float)H1;
  const int64_t ntiles = (a.M + LF_ROWS - 1) / LF_ROWS;
  constexpr int XN = LF_ROWS * 32 / NTHREADS;                 // 2 observation values per thread and tile
  float xr[XN];
  auto x_load = [&](int64_t tl) {
#pragma unroll
    for (int c = 0; c < XN; ++c) {
      const int i = t + c * NTHREADS, r = i >> 5, k = i & 31;
      xr[c] = (k < O && tl * LF_ROWS + r < a.M) ? a.X[(tl * LF_ROWS + r) * O + k] : 0.f;
    }
  };
  auto x_store = [&](int b) {
    char* Xd = Xs0 + b * X_NP * LF_XPLANE;
#pragma unroll
    for (int c = 0; c < XN; ++c) {
      const int i = t + c * NTHREADS;
      l12_xa_stage(Xd, i >> 5, i & 31, xr[c], xs);
    }
  };
  if ((int64_t)blockIdx.x < ntiles) {
    x_load(blockIdx.x);
    x_store(0);
  }
  int buf = 0;
  // phase stamps (tools/l12_phases.py): compiled in only with -DRLX_L12_STAMPS=1 -- the counter costs the kernel three spilled registers
#if RLX_L12_STAMPS
  int sti = 0;
#define L12_STAMP() if (a.stamps && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && sti < 14) a.stamps[sti++] = clock64();
#define L12_WALL(I) if (a.stamps && t == 0 && blockIdx.x == 0 && blockIdx.y == 0) a.stamps[I] = wall_clock64();      // constant 100 MHz: calibrates the clock64 ticks
#else
#define L12_STAMP()
#define L12_WALL(I)
#endif
  L12_STAMP()
  L12_WALL(14)
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const int64_t r0 = tile * LF_ROWS;
    const bool has_next = tile + gridDim.x < ntiles;
    __syncthreads();      // this tile's observation planes are visible; the previous tile's readers of the h1 image are done
    L12_STAMP()
    if (has_next) x_load(tile + gridDim.x);
    // ---- z1 = X @ W1 + b1 on the fp16 pipe (the recompute of k_dx_l1bwd)
    f32x16 z[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) z[j][r] = bias[j] * (xs * X_WSCALE);
    {
      const char* xa = Xs0 + buf * X_NP * LF_XPLANE;
      const int nks = O > 16 ? 2 : 1;
      const u32x4* w1l = W1x;
      asm volatile("" : "+v"(w1l));      // re-load the 8 fragments per tile (L1 / L2 hits): hoisted out of the tile loop they pin 32 registers
      // (behind the asm the pointer is GENERIC to the compiler: it emitted flat_load_dwordx4 + s_waitcnt vmcnt(0) lgkmcnt(0) per
      //  fragment pair -- four serial L2 round trips per tile.  Back to the global address space: global loads, counted waits.)
      typedef const u32x4 __attribute__((address_space(1))) * gfrag_t;
      gfrag_t w1p = (gfrag_t)w1l;
      for (int s_ = 0; s_ < nks; ++s_) {
        u32x4 xf[X_NP];
#pragma unroll
        for (int p = 0; p < X_NP; ++p) xf[p] = *reinterpret_cast<const u32x4*>(xa + p * LF_XPLANE + bx_off(li, 2 * s_ + lh));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const u32x4 w0 = w1p[(int64_t)s_ * w1_step + (j * X_NP + 0) * 64], w1 = w1p[(int64_t)s_ * w1_step + (j * X_NP + 1) * 64];
          z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[0]), __builtin_bit_cast(f16x8, w1), z[j], 0, 0, 0);
          z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[1]), __builtin_bit_cast(f16x8, w0), z[j], 0, 0, 0);
          z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[0]), __builtin_bit_cast(f16x8, w0), z[j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[j][r] *= xinv * X_WINV;
    }
    if (has_next) x_store(buf ^ 1);       // (its last readers finished before this tile's first barrier)
    L12_STAMP()
    // ---- LayerNorm row statistics (as k_l1fwd_mfma / k_dx_l1bwd)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      float sv[4], ssv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * gq + e;
        float s_ = 0.f, ss = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) { s_ += z[j][r]; ss += z[j][r] * z[j][r]; }
        sv[e] = s_;
        ssv[e] = ss;
      }
      const float st_ = half_sum4(sv[0], sv[1], sv[2], sv[3], lb0, lb1);
      const float sst = half_sum4(ssv[0], ssv[1], ssv[2], ssv[3], lb0, lb1);
      if (li < 4) {
        redA[(0 * NW + w) * 32 + 8 * gq + 4 * lh + li] = st_;
        redA[(1 * NW + w) * 32 + 8 * gq + 4 * lh + li] = sst;
      }
    }
    __syncthreads();
    {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) v += redA[((lane >> 5) * NW + q) * 32 + (lane & 31)];
      totA[lane] = v;
    }
    if (a.stats && w == 0 && lane < 32 && r0 + lane < a.M) {      // [2][M]: what the recomputing weight-gradient kernel reads
      const float mean = totA[lane] * invH;
      a.stats[r0 + lane] = mean;
      a.stats[a.M + r0 + lane] = rsqrtf(fmaxf(0.f, totA[32 + lane] * invH - mean * mean) + 1e-6f);
    }
    L12_STAMP()
    // ---- normalise, activate; h1 -> HBM (128-byte row segments) and, as fp16 planes, into the LDS image of the second layer's A operand
    float* hb = a.H1 + (r0 + 4 * lh) * H1 + w * 32 * NT + li;
    char* awr = Aimg + 4 * lh * AROW + (w * 32 * NT + li) * 2;      // element (row rho + 4 lh, k = 64 w + 32 j + li): + rho * AROW + 64 j
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const lf_v4 sv = *reinterpret_cast<const lf_v4*>(totA + 8 * gq + 4 * lh);
      const lf_v4 ssv = *reinterpret_cast<const lf_v4*>(totA + 32 + 8 * gq + 4 * lh);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * gq + e;
        const int rho = 8 * gq + e;              // row inside the tile, minus the 4 * lh
        const float mean = sv[e] * invH;
        const float rs = rsqrtf(fmaxf(0.f, ssv[e] * invH - mean * mean) + 1e-6f);
        const bool inb = r0 + rho + 4 * lh < a.M;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float xh = (z[j][r] - mean) * rs;
          const float h = act_fwd_t<ACT>(xh * gam[j] + bet[j]);
          // (nontemporal: h1 -- 67 MB per launch -- is next read by the weight-gradient launch three kernels later; written through
          //  the L2 it pushes out what the other chain's kernels are using.  Same-box A/B of two builds: 69.64 -> 68.75 ms per
          //  iteration.  The same hint on h2 (read by the next kernel: +0.65 ms), on the tail's H2 loads (+0.5) and dZ3 stores (+0.9)
          //  and on the weight-gradient producers' loads (+2.8: its column tiles share operand rows through the L2) is a loss.)
          // NTS (launches of at least 16384 rows: h1 is 32 MB and more, the size of the L2s): nontemporal stores.  h1 is next read by
          // the weight-gradient launch three kernels later; written through the L2 it pushes out what the other chain's kernels are
          // using.  Same-box A/B of two builds: 69.64 -> 68.75 ms per iteration at 32768 rows -- and 121.9 -> 124.4 at 4096 rows, where
          // h1 (8 MB per network) does stay in the L2 until it is read: plain stores there.  The same hint on h2 (read by the next
          // kernel: +0.65 ms), on the tail's H2 loads (+0.5) and dZ3 stores (+0.9) and on the weight-gradient producers' loads
          // (+2.8: its column tiles share operand rows through the L2) is a loss.
          if (inb && a.H1) {
            if (NTS) __builtin_nontemporal_store(h, &hb[(int64_t)rho * H1 + 32 * j]);
            else hb[(int64_t)rho * H1 + 32 * j] = h;
          }
          uint32_t p0, p1;
          bx_split2((inb ? h : 0.f) * X_ASCALE, 0.f, p0, p1);
          char* d = awr + rho * AROW + j * 64;
          *reinterpret_cast<uint16_t*>(d) = (uint16_t)p0;
          *reinterpret_cast<uint16_t*>(d + APLANE) = (uint16_t)p1;
        }
        __builtin_amdgcn_sched_barrier(0);     // one row at a time: hipcc otherwise runs all 32 ELU polynomials side by side and spills
      }
    }
    __syncthreads();      // the h1 image is complete
    L12_STAMP()
    // ---- second layer: h2 tile = act(h1 @ W2 + b2); barrier-free K loop, weight fragments PFX 16-k blocks ahead
    f32x16 acc[NT2];
#pragma unroll
    for (int j = 0; j < NT2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    constexpr int PFX = 2, NB16 = H1 / 16;
    const char* ard = Aimg + li * AROW + lh * 16;                   // A fragment of 16-k block kb: + 32 kb (+ APLANE for the low plane)
    u32x4 bx[PFX][NT2][X_NP];
#pragma unroll
    for (int u = 0; u < PFX; ++u)
#pragma unroll
      for (int j = 0; j < NT2; ++j)
#pragma unroll
        for (int p = 0; p < X_NP; ++p) bx[u][j][p] = W2x[(int64_t)u * w2_step + (j * X_NP + p) * 64];
#define RLX_L12_STEP(P, Q)                                                                                        \
  _Pragma("unroll") for (int j = 0; j < NT2; ++j)                                                                 \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[P]),                           \
                                                      __builtin_bit_cast(f16x8, bx[u][j][Q]), acc[j], 0, 0, 0);
#define RLX_L12_BLOCK(QU, REFILL)                                                                                 \
  {                                                                                                               \
    u32x4 av[X_NP];                                                                                               \
    const char* ab = ard + (QU) * 32;                                                                             \
    _Pragma("unroll") for (int p = 0; p < X_NP; ++p) av[p] = *reinterpret_cast<const u32x4*>(ab + p * APLANE);    \
    RLX_L12_STEP(0, 1)                                                                                            \
    RLX_L12_STEP(1, 0)                                                                                            \
    RLX_L12_STEP(0, 0)                                                                                            \
    if (REFILL) {                                                                                                 \
      _Pragma("unroll") for (int j = 0; j < NT2; ++j) _Pragma("unroll") for (int p = 0; p < X_NP; ++p)            \
          bx[u][j][p] = W2x[(int64_t)((QU) + PFX) * w2_step + (j * X_NP + p) * 64];                               \
    }                                                                                                             \
  }
    // (the last PFX blocks are peeled: with the refill behind `if (q + u + PFX < NB16)` the loads outstanding at the loop header
    //  depend on the path and hipcc waits with vmcnt(0) every other block -- half the prefetch depth)
#pragma unroll 1
    for (int q = 0; q < NB16 - PFX; q += PFX) {      // (not unrolled: hipcc otherwise hoists all 32 blocks' operand loads and spills)
#pragma unroll
      for (int u = 0; u < PFX; ++u) RLX_L12_BLOCK(q + u, true)
    }
#pragma unroll
    for (int u = 0; u < PFX; ++u) RLX_L12_BLOCK(NB16 - PFX + u, false)
#undef RLX_L12_BLOCK
#undef RLX_L12_STEP
    L12_STAMP()
    {
      const float so = X_AINV * X_WINV;
      float* cb = a.H2 + (r0 + 4 * lh) * N2 + w * 32 * NT2 + li;
#pragma unroll
      for (int j = 0; j < NT2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rho = (r & 3) + 8 * (r >> 2);
          if (r0 + rho + 4 * lh < a.M) cb[(int64_t)rho * N2 + 32 * j] = act_fwd_t<ACT>(fmaf(acc[j][r], so, b2v[j]));
        }
    }
    L12_STAMP()
    L12_WALL(15)
  }
#undef L12_STAMP
#undef L12_WALL
}

bool l12fwd_supported(const rlx_mlp_desc& d) {
  return l1fwd_mfma_supported(d) && d.n_hidden >= 2 && (d.hidden[1] == 256 || d.hidden[1] == 512);
}

// tw (optional): a second network of the same shapes on the same rows (twin launch)
int launch_l12fwd(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x, float* h1, float* h2,
                  const void* w1x, const void* w2x, int64_t M, hipStream_t st, const L12Twin* tw, float* stats) {
  const LayerOff &o0 = L.layer[0], &o1 = L.layer[1];
  L12Args a;
  a.X = x; a.W1x = w1x; a.b1 = params + o0.b; a.g = params + o0.g; a.be = params + o0.be; a.W2x = w2x; a.b2 = params + o1.b;
  a.H1 = stats ? nullptr : h1; a.H2 = h2; a.stats = stats; a.xmax = ctx->l1_xmax; a.M = M; a.O = o0.in;
  a.stamps = (unsigned long long*)ctx->dbg_stamps;
  L12Args a2 = a;
  if (tw) {
    a2.W1x = tw->w1x; a2.b1 = tw->params + o0.b; a2.g = tw->params + o0.g; a2.be = tw->params + o0.be; a2.W2x = tw->w2x;
    a2.b2 = tw->params + o1.b; a2.H1 = stats ? nullptr : tw->h1; a2.H2 = tw->h2; a2.stats = tw->stats;
  }
  const int N2 = o1.out;
  const double nets = tw ? 2.0 : 1.0;
  // algorithmic: the layer-2 product (the K = O first layer rides along); X in, h1 and h2 out, both weight matrices
  ProfScope prof(ctx, PK_L12FWD, nets * 2.0 * (double)M * L12_H1 * (N2 + o0.in), st,
                 nets * 4.0 * ((double)M * (o0.in + (stats ? 2 : L12_H1) + N2) + (double)L12_H1 * (N2 + o0.in)), M, N2, L12_H1, 1);
  const int64_t nt = (M + LF_ROWS - 1) / LF_ROWS;
  const int per = tw ? ctx->num_cus : 2 * ctx->num_cus;
  const int grid = (int)(nt < per ? nt : per);
  const size_t lds = (size_t)X_NP * LF_ROWS * L12_AROW + 2 * X_NP * LF_XPLANE + (2 * 8 * 32 + 8 * 64) * sizeof(float);
#define RLX_L12_LAUNCH(NT2V)                                                                                       \
  {                                                                                                                \
    static AttrOnce attr_set;                                                                                        \
    if (!attr_set.done()) {                                                                                               \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_l12fwd<RLX_ACT_ELU, NT2V, false>),            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                    \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_l12fwd<RLX_ACT_ELU, NT2V, true>),             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                    \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_l12fwd<RLX_ACT_ELU, NT2V, false, true>),      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                    \
      attr_set.mark();                                                                                               \
    }                                                                                                              \
    if (tw) { RLX_PLAUNCH((k_l12fwd<RLX_ACT_ELU, NT2V, true>), dim3(grid, 2), dim3(512), lds, st, a, a2); }         \
    else if (M >= 16384) { RLX_PLAUNCH((k_l12fwd<RLX_ACT_ELU, NT2V, false, true>), dim3(grid), dim3(512), lds, st, a, a2); } \
    else { RLX_PLAUNCH((k_l12fwd<RLX_ACT_ELU, NT2V, false>), dim3(grid), dim3(512), lds, st, a, a2); }              \
  }
  if (N2 == 256) RLX_L12_LAUNCH(1)
  else RLX_L12_LAUNCH(2)
#undef RLX_L12_LAUNCH
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

bool l1fwd_mfma_supported(const rlx_mlp_desc& d) {
  return d.hidden[0] == 512 && d.act == RLX_ACT_ELU && d.ln_first && d.in_dim <= 32;
}

// params1 / h1_1 (optional, twin launch): a second network with the same first layer on the same rows (grid.y == 2)
int launch_l1fwd_mfma(const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x, float* h1, int64_t M,
                      int num_cus, hipStream_t st, const int32_t* m_dev, rlx_ctx* prof_ctx, const float* params1, float* h1_1) {
  const LayerOff& o = L.layer[0];
  const int O = o.in, OP = (O + 1) & ~1;
  const unsigned gy = params1 ? 2 : 1;
  // algorithmic bytes: the rows once in, the activations once out, the layer's parameters
  ProfScope prof(m_dev ? nullptr : prof_ctx, PK_L1FWD, 0.0, st, gy * 4.0 * ((double)M * (O + 512) + (double)(O + 3) * 512), M, 512, 0, PROF_ENGINE_HBM);
  const int64_t nt = (M + LF_ROWS - 1) / LF_ROWS;
  const int per = params1 ? num_cus : 2 * num_cus;    // two resident workgroups per CU over both networks
  const int grid = (int)(nt < per ? nt : per);
  const size_t lds = ((size_t)LF_ROWS * LF_XS + 2 * 8 * 32 + 8 * 64) * sizeof(float);
  const int64_t pdelta = params1 ? (int64_t)(params1 - params) : 0;
  // (9 register-resident k-steps keep the kernel at 4 waves per SIMD; wider observations take the 16-step form)
  if (OP <= 18)
    RLX_PLAUNCH((k_l1fwd_mfma<2, 8, RLX_ACT_ELU, true, 9>), dim3(grid, gy), dim3(512), lds, st, x, params + o.W, params + o.b,
                params + o.g, params + o.be, h1, M, O, m_dev, pdelta, h1_1);
  else
    RLX_PLAUNCH((k_l1fwd_mfma<2, 8, RLX_ACT_ELU, true, 16>), dim3(grid, gy), dim3(512), lds, st, x, params + o.W, params + o.b,
                params + o.g, params + o.be, h1, M, O, m_dev, pdelta, h1_1);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// W2[H1][N2] (flax Dense kernel of layer 1: rows = input = hidden[0] index c, cols = k) ->
// MFMA-B-fragment order Wf[q][h][c] = float4{ W2[c][8q + 4h + 0..3] }
__global__ __launch_bounds__(256) void k_frag_reorder(const float* __restrict__ W2, float* __restrict__ Wf, int H1,
                                                      int N2) {
  const int64_t total = (int64_t)(N2 >> 2) * H1;  // float4 count
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % H1);
    const int qh = (int)(e / H1);  // q*2 + h
    const lf_v4 v = *reinterpret_cast<const lf_v4*>(W2 + (int64_t)c * N2 + qh * 4);
    reinterpret_cast<lf_v4*>(Wf)[e] = v;
  }
}

bool l1fused_supported(const rlx_mlp_desc& d) {
  const bool combo = (d.hidden[0] == 512 && d.act == RLX_ACT_ELU && d.ln_first) ||
                     (d.hidden[0] == 256 && d.act == RLX_ACT_TANH && !d.ln_first) ||
                     (d.hidden[0] == 256 && d.act == RLX_ACT_RELU && !d.ln_first);
  return combo && d.n_hidden >= 2 && d.in_dim <= 32 && d.hidden[1] % 32 == 0 && d.hidden[1] <= 512;
}

size_t l1fused_partial_floats(const rlx_mlp_desc& d, int grid) {
  return (size_t)grid * (d.in_dim + 3) * d.hidden[0] + (size_t)d.hidden[0] * d.hidden[1];  // slabs + W2^T
}

int l1fused_grid(int64_t M, int num_cus) {
  const int64_t nt = (M + LF_ROWS - 1) / LF_ROWS;
  return (int)(nt < num_cus ? nt : num_cus);
}

// arena: [grid][(O+3)*H1] slabs followed by the W2^T copy.  Emits the reduce segments.
bool l1fused_bx_images(const rlx_ctx* ctx, const MlpLayout& L, const float* params, const void** w2x, const void** w1x) {
  const LayerOff& o0 = L.layer[0];
  const LayerOff& o1 = L.layer[1];
  const int H1 = o0.out, N2 = o1.out, O = o0.in;
  // the transposed split image of W2 registered for this pass (bx_prepare_mlp): main GEMM on the half-precision pipe
  *w2x = (N2 % 128 == 0 && N2 % (16 * RLX_LF_PFX) == 0) ? bx_lookup(ctx, params + o1.W, 1, N2, H1) : nullptr;
  *w1x = *w2x ? bx_lookup(ctx, params + o0.W, 0, O, H1) : nullptr;     // first-layer image: z1 recompute on the fp16 pipe
  return *w2x != nullptr && *w1x != nullptr;
}

// tw (optional, twin launch; both networks need their split images -- l1fused_bx_images): the second of two equally shaped
// networks on the same rows x in the same launch (grid.y == 2); its reduce segments go to tw->tab
int launch_l1fused(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x,
                   const float* dZ2, float* arena, int grid, float* grads, int64_t M, ReduceTable* tab, hipStream_t st,
                   const L1FusedTwin* tw) {
  const LayerOff& o0 = L.layer[0];
  const LayerOff& o1 = L.layer[1];
  const int H1 = o0.out, N2 = o1.out, O = o0.in;
  float* slabs = arena;
  float* W2t = arena + (size_t)grid * (O + 3) * H1;
  const void *w2x = nullptr, *w1x = nullptr;
  const bool bxk = l1fused_bx_images(ctx, L, params, &w2x, &w1x);
  RLX_REQUIRE(!tw || (bxk && tw->w2x && tw->w1x), RLX_EUNSUP, "l1fused twin launch: both networks need their split weight images");
  if (!bxk) {
    hipLaunchKernelGGL(k_frag_reorder, dim3(div_up((int64_t)(N2 >> 2) * H1, 256)), dim3(256), 0, st, params + o1.W, W2t,
                       H1, N2);
    RLX_LAUNCH_CHECK();
  }
  L1FusedArgs a;
  a.X = x; a.dZ2 = dZ2; a.W2t = W2t; a.W1 = params + o0.W; a.b1 = params + o0.b;
  a.g = o0.g >= 0 ? params + o0.g : nullptr;
  a.be = o0.be >= 0 ? params + o0.be : nullptr;
  a.partials = slabs; a.M = M; a.O = O; a.H1 = H1; a.N2 = N2; a.act = d.act; a.ln = d.ln_first ? 1 : 0;
  a.W2x = w2x;
  a.W1x = w1x;
  a.NTx = 4 * div_up(H1, G_BN);
  a.gs = ctx->bx_gscale;
  a.gso = X_WINV / a.gs;
  a.xmax = ctx->l1_xmax;
  L1FusedArgs a2 = a;
  if (tw) {
    a2.dZ2 = tw->dZ2; a2.W1 = tw->params + o0.W; a2.b1 = tw->params + o0.b;
    a2.g = o0.g >= 0 ? tw->params + o0.g : nullptr;
    a2.be = o0.be >= 0 ? tw->params + o0.be : nullptr;
    a2.partials = tw->arena; a2.W2x = tw->w2x; a2.W1x = tw->w1x;
  }
  const int OP = (O + 1) & ~1;
  const size_t a_img = bxk ? (size_t)X_NP * LF_ROWS * N2 / 2 : (size_t)LF_ROWS * (N2 + 4);
  const size_t lds = bxk ? ((N2 == LFP_N2 ? 2 : 1) * (a_img + LF_XIMG / 4) + 2048) * sizeof(float)
                         : ((size_t)OP * H1 + (N2 == LFP_N2 ? 2 : 1) * (a_img + LF_ROWS * LF_XS) + 2048) * sizeof(float);
  RLX_REQUIRE(lds <= 160 * 1024, RLX_EUNSUP, "l1fused: tile image exceeds the LDS");
  {
    // main GEMM + z recompute + dW1 on the matrix pipe
    const double ntw = tw ? 2.0 : 1.0;
    ProfScope prof(ctx, PK_DX_L1BWD, ntw * 2.0 * (double)M * H1 * (N2 + O), st,                  // algorithmic: dX + dW1
                   ntw * 4.0 * ((double)M * N2 + (double)H1 * N2 + (double)M * O + 2.0 * O * H1), M, H1, N2, bxk ? 1 : 0);
#define RLX_LF_ATTR(KERNEL)                                                                                    \
  {                                                                                                            \
    static AttrOnce attr_set;                                                                                    \
    if (!attr_set.done()) {                                                                                           \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024));                                                            \
      attr_set.mark();                                                                                           \
    }                                                                                                          \
  }
#define RLX_LF_LAUNCH(NTV, NWV, ACTV, LNV)                                                                      \
  {                                                                                                            \
    RLX_LF_ATTR((k_dx_l1bwd<NTV, NWV, ACTV, LNV, false>))                                                      \
    RLX_LF_ATTR((k_dx_l1bwd<NTV, NWV, ACTV, LNV, true>))                                                       \
    RLX_LF_ATTR((k_dx_l1bwd<NTV, NWV, ACTV, LNV, true, true>))                                                 \
    if (tw) { RLX_PLAUNCH((k_dx_l1bwd<NTV, NWV, ACTV, LNV, true, true>), dim3(grid, 2), dim3(64 * NWV), lds, st, a, a2); } \
    else if (bxk) { RLX_PLAUNCH((k_dx_l1bwd<NTV, NWV, ACTV, LNV, true>), dim3(grid), dim3(64 * NWV), lds, st, a, a2); } \
    else { RLX_PLAUNCH((k_dx_l1bwd<NTV, NWV, ACTV, LNV, false>), dim3(grid), dim3(64 * NWV), lds, st, a, a2); }    \
  }
    if (H1 == 512 && d.act == RLX_ACT_ELU && d.ln_first) RLX_LF_LAUNCH(2, 8, RLX_ACT_ELU, true)
    else if (H1 == 256 && d.act == RLX_ACT_TANH && !d.ln_first) RLX_LF_LAUNCH(2, 4, RLX_ACT_TANH, false)
    else if (H1 == 256 && d.act == RLX_ACT_RELU && !d.ln_first) RLX_LF_LAUNCH(2, 4, RLX_ACT_RELU, false)
    else RLX_REQUIRE(false, RLX_EUNSUP, "l1fused: unsupported (hidden[0], act, ln) combination");
#undef RLX_LF_LAUNCH
#undef RLX_LF_ATTR
  }
  RLX_LAUNCH_CHECK();
  const int64_t PS = (int64_t)(O + 3) * H1;
  tab->seg[tab->n++] = ReduceSeg{slabs, grads + o0.W, (int64_t)O * H1, PS, grid, 0, 1.f, 0.f, 1};
  tab->seg[tab->n++] = ReduceSeg{slabs + (int64_t)O * H1, grads + o0.b, (int64_t)H1, PS, grid, 0, 1.f, 0.f, 1};
  if (d.ln_first) {
    tab->seg[tab->n++] = ReduceSeg{slabs + (int64_t)(O + 1) * H1, grads + o0.g, (int64_t)H1, PS, grid, 0, 1.f, 0.f, 1};
    tab->seg[tab->n++] = ReduceSeg{slabs + (int64_t)(O + 2) * H1, grads + o0.be, (int64_t)H1, PS, grid, 0, 1.f, 0.f, 1};
  }
  if (tw) {
    ReduceTable* t2 = tw->tab;
    float* s2 = tw->arena;
    t2->seg[t2->n++] = ReduceSeg{s2, tw->grads + o0.W, (int64_t)O * H1, PS, grid, 0, 1.f, 0.f, 1};
    t2->seg[t2->n++] = ReduceSeg{s2 + (int64_t)O * H1, tw->grads + o0.b, (int64_t)H1, PS, grid, 0, 1.f, 0.f, 1};
    if (d.ln_first) {
      t2->seg[t2->n++] = ReduceSeg{s2 + (int64_t)(O + 1) * H1, tw->grads + o0.g, (int64_t)H1, PS, grid, 0, 1.f, 0.f, 1};
      t2->seg[t2->n++] = ReduceSeg{s2 + (int64_t)(O + 2) * H1, tw->grads + o0.be, (int64_t)H1, PS, grid, 0, 1.f, 0.f, 1};
    }
  }
  return RLX_OK;
}

}  // namespace rlx
