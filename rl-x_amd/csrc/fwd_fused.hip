// fwd_fused.hip -- the whole trunk forward of the full-jit nets (Dense(512)+LayerNorm+ELU, Dense(256)+ELU, Dense(128)+ELU:
// rl_x/algorithms/ppo/flax_full_jit/policy.py:31-39, critic.py:22-30) in ONE launch per network and minibatch.
//
// The unfused forward is k_l1fwd_mfma -> k_gemm_bx<0> (layer 2) -> k_gemm_bx<0> (layer 3): H1 [mb, 512] is written (67 MB at
// mb = 32768) and read back, H2 likewise, and every launch pays its own prologue / store epilogue with nothing to overlap them
// (28 + 53 + 19 us alone on the chip, against 29 us of matrix-pipe time).  Here a workgroup (8 waves) owns 64 rows and walks
// the three layers with the activations staying on chip:
//   layer 1  z1 = X W1 + b1 on the exact-fp32 MFMA (K = obs <= 32), LayerNorm statistics, ELU -> H1 block in REGISTERS,
//            stored to HBM once (the backward needs it: dW2) -- the stores drain under layer 2;
//   layer 2  the 512-deep contraction in eight 64-k chunks: chunk c is exactly wave c's 64 x 64 block of H1, which that wave
//            splits into the three bf16 planes and puts into an LDS chunk buffer one step ahead of its use (two buffers, one
//            barrier per chunk); weights come from the fragment-ordered split image in L2 (gemm_bx.h);
//   layer 3  the same over eight 32-k chunks of H2.
// All products are computed TRANSPOSED -- D[i = output column][j = batch row] = sum_k W[k][i] H[j][k], i.e. the weight image is
// the MFMA's A operand and the activation planes its B operand (both have the per-lane shape "index l & 31, eight consecutive
// k", so the same registers serve either role).  In the accumulator layout a LANE then owns one batch row and four CONSECUTIVE
// output columns per register quad: the next layer's activation planes are written with 8-byte LDS stores straight from the
// accumulators (no cross-lane transposition), LayerNorm row sums are lane-local adds, and HBM stores are 16-byte pieces.
// Arithmetic: fp32 throughout with the split-bf16 engine's six plane products per fp32 product for layers 2 / 3 (same
// contraction order per output as k_gemm_bx), LayerNorm as var = max(0, E[z^2] - E[z]^2), eps 1e-6 (oracle/nets.py).
#include "mlp.h"
#include "gemm_bx.h"

namespace rlx {

constexpr int FF_ROWS = 64, FF_NW = 8, FF_THREADS = 64 * FF_NW;
constexpr int FF_H1 = 512, FF_H2 = 256, FF_H3 = 128;
constexpr int FF_XS = 33;                         // X tile row stride (floats)
constexpr int FF_PL = FF_ROWS * 512;              // bytes of one bf16 plane of the activation image: [64 rows][256 k]
constexpr int FF_IMG = 3 * FF_PL;                 // 96 KiB: half of H1's columns, later all of H2

typedef float ff_v4 __attribute__((ext_vector_type(4)));

// byte offset of 16-byte k-slot ks (0..31) of row r in a [64][256 k] plane: 512-byte rows, the slot index XOR-ed with the low
// four row bits -- the 16 rows of every ds_read_b128 service group and of every 16-lane ds_write_b64 group hit 16 distinct slots
__device__ __forceinline__ int ff_off(int r, int ks) { return r * 512 + ((ks ^ (r & 15)) << 4); }

struct FwdFusedArgs {
  const float* X;        // [M, O]
  const float* W1;       // [O, 512]
  const float *b1, *g1, *be1;
  const u32x4* Wf2;      // split image of W2 [512, 256] (k_bx_wfrag; 8 column tiles)
  const float* b2;
  const u32x4* Wf3;      // split image of W3 [256, 128] (4 column tiles)
  const float* b3;
  float *H1, *H2, *H3;   // [M, 512], [M, 256], [M, 128]
  int64_t M;
  int O;
  int dbg;   // timing ablations (results invalid): bit 0 no H1 store, 1 no H2 / H3 stores, 2 no layer-2 MFMAs, 3 no layer-3 MFMAs,
             // 4 no layer-1 MFMAs
};

// six plane products of one 16-k block: acc (D[i = out column][j = row]) += W-fragment (A operand) x activation planes (B)
__device__ __forceinline__ void ff_mma(const u32x4 (&wf)[3], const u32x4 (&hf)[3], f32x16& acc) {
#define RLX_FF_STEP(P, Q)                                                                                     \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[P]), __builtin_bit_cast(bf16x8, hf[Q]), acc, 0, 0, 0);
  RLX_FF_STEP(1, 1)
  RLX_FF_STEP(2, 0)
  RLX_FF_STEP(0, 2)
  RLX_FF_STEP(1, 0)
  RLX_FF_STEP(0, 1)
  RLX_FF_STEP(0, 0)
#undef RLX_FF_STEP
}

// four consecutive k (columns) of one row, as three bf16 planes: 8 bytes per plane at `d`
__device__ __forceinline__ void ff_put4(char* d, float v0, float v1, float v2, float v3) {
  uint32_t a0, a1, a2, b0, b1, b2;
  bx_split2(v0, v1, a0, a1, a2);
  bx_split2(v2, v3, b0, b1, b2);
  *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
  *reinterpret_cast<u32x2*>(d + FF_PL) = u32x2{a1, b1};
  *reinterpret_cast<u32x2*>(d + 2 * FF_PL) = u32x2{a2, b2};
}

// DBG: timing ablations, compile-time so that the measured kernel carries no extra control flow (bits as FwdFusedArgs.dbg)
template <int DBG>
__global__ __launch_bounds__(FF_THREADS, 2) void k_fwd_fused(FwdFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int O = a.O, OP = (O + 1) & ~1;
  char* img = lds;                                                  // FF_IMG: activation planes [64 rows][256 k] x 3
  float* W1s = reinterpret_cast<float*>(lds + FF_IMG);              // [OP][512]
  float* Xs = W1s + OP * FF_H1;                                     // [64][33]
  float* red = Xs + FF_ROWS * FF_XS;                                // [2 stats][8 waves][64 rows]
  float* prm = red + 2 * FF_NW * FF_ROWS;                           // b1 | g1 | be1 (512 each) | b2 (256) | b3 (128)
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  const int64_t r0 = (int64_t)blockIdx.x * FF_ROWS;
  // ---- stage W1, the X tile and the small vectors
  for (int i = t; i < OP * FF_H1; i += FF_THREADS) W1s[i] = (i < O * FF_H1) ? a.W1[i] : 0.f;
  for (int i = t; i < FF_ROWS * 32; i += FF_THREADS) {
    const int r = i >> 5, k = i & 31;
    Xs[r * FF_XS + k] = (k < O && r0 + r < a.M) ? a.X[(r0 + r) * O + k] : 0.f;
  }
  for (int i = t; i < FF_H1; i += FF_THREADS) {
    prm[i] = a.b1[i];
    prm[FF_H1 + i] = a.g1[i];
    prm[2 * FF_H1 + i] = a.be1[i];
  }
  if (t < FF_H2) prm[3 * FF_H1 + t] = a.b2[t];
  if (t < FF_H3) prm[3 * FF_H1 + FF_H2 + t] = a.b3[t];
  // first weight fragments of layer 2 (this wave's column tile = w), two 16-k blocks ahead
  u32x4 wf[2][3];
  const u32x4* W2p = a.Wf2 + (int64_t)w * 3 * 64 + lane;             // + kb * 8 * 3 * 64
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p) wf[u][p] = W2p[(int64_t)u * 8 * 3 * 64 + p * 64];
  __syncthreads();

  // ================= layer 1: this wave's 64 hidden columns [64 w, 64 w + 64) x 64 rows, transposed product
  // accumulator (jj, ii), register r, half lh of lane li: hidden column 64 w + 32 jj + (r & 3) + 8 (r >> 2) + 4 lh, row 32 ii + li
  f32x16 z[2][2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int r = 0; r < 16; ++r) z[jj][ii][r] = 0.f;
  {
    const float* wq = W1s + lh * FF_H1 + 64 * w + li;               // A operand: A[i = column li][k = lh]
    const float* xq = Xs + li * FF_XS + lh;                         // B operand: B[k = lh][j = row li]
    for (int kk = (DBG & 16) ? OP : 0; kk < OP; kk += 2) {
      const float w0 = wq[kk * FF_H1], w1 = wq[kk * FF_H1 + 32];
      const float x0 = xq[kk], x1 = xq[32 * FF_XS + kk];
      z[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, x0, z[0][0], 0, 0, 0);
      z[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, x1, z[0][1], 0, 0, 0);
      z[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, x0, z[1][0], 0, 0, 0);
      z[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, x1, z[1][1], 0, 0, 0);
    }
  }
  // + bias; LayerNorm row statistics: lane-local over the lane's 32 columns, the two halves of the wave, then the 8 waves
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const ff_v4 bv = *reinterpret_cast<const ff_v4*>(prm + 64 * w + 32 * jj + 8 * q + 4 * lh);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = z[jj][ii][4 * q + e] + bv[e];
          z[jj][ii][4 * q + e] = v;
          s1[ii] += v;
          s2[ii] = fmaf(v, v, s2[ii]);
        }
    }
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const unsigned u1 = (unsigned)__float_as_int(s1[ii]), u2 = (unsigned)__float_as_int(s2[ii]);
    const auto p1 = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
    const auto p2 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    s1[ii] = __int_as_float((int)p1[0]) + __int_as_float((int)p1[1]);
    s2[ii] = __int_as_float((int)p2[0]) + __int_as_float((int)p2[1]);
    if (lh == 0) {
      red[(0 * FF_NW + w) * FF_ROWS + 32 * ii + li] = s1[ii];
      red[(1 * FF_NW + w) * FF_ROWS + 32 * ii + li] = s2[ii];
    }
  }
  __syncthreads();
  float mean[2], rstd[2];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int q = 0; q < FF_NW; ++q) {                                // fixed order: the same bits in every wave
      a1 += red[(0 * FF_NW + q) * FF_ROWS + 32 * ii + li];
      a2 += red[(1 * FF_NW + q) * FF_ROWS + 32 * ii + li];
    }
    mean[ii] = a1 * (1.0f / FF_H1);
    rstd[ii] = rsqrtf(fmaxf(0.f, a2 * (1.0f / FF_H1) - mean[ii] * mean[ii]) + 1e-6f);
  }
  // normalise + ELU in place: z <- H1 (kept in registers: stored to HBM at the very end, see below)
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = 64 * w + 32 * jj + 8 * q + 4 * lh;
      const ff_v4 gv = *reinterpret_cast<const ff_v4*>(prm + FF_H1 + col);
      const ff_v4 ev = *reinterpret_cast<const ff_v4*>(prm + 2 * FF_H1 + col);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (z[jj][ii][4 * q + e] - mean[ii]) * rstd[ii];
          z[jj][ii][4 * q + e] = act_fwd_t<RLX_ACT_ELU>(fmaf(xh, gv[e], ev[e]));
        }
    }

  // ================= layer 2: out columns [32 w, 32 w + 32) x 64 rows.  The 512-deep contraction runs in two halves of 256 k:
  // waves 0-3 (then 4-7) put their 64 x 64 blocks of H1 into the image as bf16 planes -- all four at once --, one barrier, then
  // every wave walks the 16 k-blocks of the half without further synchronisation.
  auto put_h1 = [&]() {                                             // this wave's block -> image k = 64 (w & 3) + 32 jj + 8 q + 4 lh + e
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
          ff_put4(img + ff_off(32 * ii + li, 8 * (w & 3) + 4 * jj + q) + 8 * lh, z[jj][ii][4 * q], z[jj][ii][4 * q + 1],
                  z[jj][ii][4 * q + 2], z[jj][ii][4 * q + 3]);
  };
  f32x16 acc2[2];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[ii][r] = 0.f;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if ((w >> 2) == half) put_h1();
    __syncthreads();                                                // planes of this half visible
#pragma unroll 1
    for (int s0 = 0; s0 < 16; s0 += 2) {                            // 16-k blocks of the half: global block kb = 16 half + s
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s = s0 + u;
        // one row half at a time: its three planes are live for six products; the weight fragments serve both halves
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          u32x4 hf[3];
          const char* hb = img + ff_off(32 * ii + li, 2 * s + lh);
#pragma unroll
          for (int p = 0; p < 3; ++p) hf[p] = *reinterpret_cast<const u32x4*>(hb + p * FF_PL);
          if (!(DBG & 4)) ff_mma(wf[u], hf, acc2[ii]);
        }
        const int kn = 16 * half + s + 2 < 32 ? 16 * half + s + 2 : 31;   // (clamped: the last two re-fetch the final block)
#pragma unroll
        for (int p = 0; p < 3; ++p) wf[u][p] = W2p[(int64_t)kn * 8 * 3 * 64 + p * 64];
      }
    }
    __syncthreads();                                                // every wave is done reading the image
  }
  // first weight fragments of layer 3 (column tile ct = w >> 1), in flight under the layer-2 epilogue
  const int ct = w >> 1, i3 = w & 1;
  const u32x4* W3p = a.Wf3 + (int64_t)ct * 3 * 64 + lane;            // + kb * 4 * 3 * 64
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p) wf[u][p] = W3p[(int64_t)u * 4 * 3 * 64 + p * 64];
  // H2 = ELU(acc2 + b2), in place, and straight into the image: k = 32 w + 8 q + 4 lh + e -- all eight waves at once
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const ff_v4 bv = *reinterpret_cast<const ff_v4*>(prm + 3 * FF_H1 + 32 * w + 8 * q + 4 * lh);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc2[ii][4 * q + e] = act_fwd_t<RLX_ACT_ELU>(acc2[ii][4 * q + e] + bv[e]);
      ff_put4(img + ff_off(32 * ii + li, 4 * w + q) + 8 * lh, acc2[ii][4 * q], acc2[ii][4 * q + 1], acc2[ii][4 * q + 2],
              acc2[ii][4 * q + 3]);
    }
  }
  __syncthreads();

  // ================= layer 3: wave (ct, i3): out columns [32 ct, 32 ct + 32) x rows [32 i3, 32 i3 + 32); 16 k-blocks, no barrier
  f32x16 acc3;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
  {
    u32x4 hf[2][3];
    {
      const char* hb = img + ff_off(32 * i3 + li, lh);
#pragma unroll
      for (int p = 0; p < 3; ++p) hf[0][p] = *reinterpret_cast<const u32x4*>(hb + p * FF_PL);
    }
#pragma unroll 1
    for (int s0 = 0; s0 < 16; s0 += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s = s0 + u;
        const int sn = s + 1 < 16 ? s + 1 : 15;
        const char* hb = img + ff_off(32 * i3 + li, 2 * sn + lh);
#pragma unroll
        for (int p = 0; p < 3; ++p) hf[u ^ 1][p] = *reinterpret_cast<const u32x4*>(hb + p * FF_PL);
        if (!(DBG & 8)) ff_mma(wf[u], hf[u], acc3);
        const int kn = s + 2 < 16 ? s + 2 : 15;
#pragma unroll
        for (int p = 0; p < 3; ++p) wf[u][p] = W3p[(int64_t)kn * 4 * 3 * 64 + p * 64];
      }
    }
  }
  // ================= the three activation blocks go to HBM at the very end.  The memory counter retires loads and stores in
  // order, so a store issued earlier would make the next wait for a weight fragment sit out the store's HBM round trip
  // (measured: 25 us per launch when H1 was stored right after layer 1); here nothing waits behind them -- the workgroup ends
  // and the next one starts while they drain.  16-byte pieces; the four pieces of a register quad row fill one 128-byte line.
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int64_t row = r0 + 32 * ii + li;
    if (row < a.M) {
      if (!(DBG & 1)) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<ff_v4*>(a.H1 + row * FF_H1 + 64 * w + 32 * jj + 8 * q + 4 * lh) =
                ff_v4{z[jj][ii][4 * q], z[jj][ii][4 * q + 1], z[jj][ii][4 * q + 2], z[jj][ii][4 * q + 3]};
      }
      if (!(DBG & 2)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<ff_v4*>(a.H2 + row * FF_H2 + 32 * w + 8 * q + 4 * lh) =
              ff_v4{acc2[ii][4 * q], acc2[ii][4 * q + 1], acc2[ii][4 * q + 2], acc2[ii][4 * q + 3]};
      }
    }
  }
  {
    const int64_t row = r0 + 32 * i3 + li;
    if (row < a.M && !(DBG & 2)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 32 * ct + 8 * q + 4 * lh;
        const ff_v4 bv = *reinterpret_cast<const ff_v4*>(prm + 3 * FF_H1 + FF_H2 + col);
        ff_v4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = act_fwd_t<RLX_ACT_ELU>(acc3[4 * q + e] + bv[e]);
        *reinterpret_cast<ff_v4*>(a.H3 + row * FF_H3 + col) = h;
      }
    }
  }
}

bool fwd_fused_supported(const rlx_mlp_desc& d) {
  return d.n_hidden == 3 && d.hidden[0] == FF_H1 && d.hidden[1] == FF_H2 && d.hidden[2] == FF_H3 && d.act == RLX_ACT_ELU &&
         d.ln_first && d.in_dim <= 32;
}

// acts[0..2] <- H1, H2, H3 of x [M, in_dim]; needs the forward weight images of layers 2 and 3 (bx_prepare_mlp)
int launch_fwd_fused(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, const float* x,
                     float* const* acts, int64_t M, hipStream_t st) {
  const LayerOff &o0 = L.layer[0], &o1 = L.layer[1], &o2 = L.layer[2];
  const void* i2 = bx_lookup(ctx, params + o1.W, 0, o1.in, o1.out);
  const void* i3 = bx_lookup(ctx, params + o2.W, 0, o2.in, o2.out);
  RLX_REQUIRE(i2 && i3, RLX_EINVAL, "fwd_fused: the split weight images of layers 2 and 3 are not registered");
  FwdFusedArgs a;
  a.X = x; a.W1 = params + o0.W; a.b1 = params + o0.b; a.g1 = params + o0.g; a.be1 = params + o0.be;
  a.Wf2 = (const u32x4*)i2; a.b2 = params + o1.b; a.Wf3 = (const u32x4*)i3; a.b3 = params + o2.b;
  a.H1 = acts[0]; a.H2 = acts[1]; a.H3 = acts[2];
  a.M = M; a.O = d.in_dim;
  a.dbg = (ctx->bx_debug >> 13) & 31;
  const int OP = (d.in_dim + 1) & ~1;
  const size_t lds = (size_t)FF_IMG + ((size_t)OP * FF_H1 + FF_ROWS * FF_XS + 2 * FF_NW * FF_ROWS + 3 * FF_H1 + FF_H2 + FF_H3) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_fused<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_fused<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_fused<15>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_fused<31>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  RLX_REQUIRE(lds <= 160 * 1024, RLX_EUNSUP, "fwd_fused: tile image exceeds the LDS");
  const double flops = 2.0 * (double)M * ((double)d.in_dim * FF_H1 + (double)FF_H1 * FF_H2 + (double)FF_H2 * FF_H3);
  const double bytes = 4.0 * ((double)M * (d.in_dim + FF_H1 + FF_H2 + FF_H3) + (double)d.in_dim * FF_H1 + (double)FF_H1 * FF_H2 +
                              (double)FF_H2 * FF_H3);
  ProfScope prof(ctx, PK_FWD_FUSED, flops, st, bytes, M, FF_H1 + FF_H2 + FF_H3, d.in_dim, 1);
  // (timing ablations: bx_debug bits 13.. select no stores / no stores + no MFMAs of layers 2, 3 / nothing but the skeleton)
  const dim3 grid(div_up(M, FF_ROWS)), block(FF_THREADS);
  if (a.dbg == 0) { RLX_PLAUNCH(k_fwd_fused<0>, grid, block, lds, st, a); }
  else if (a.dbg == 3) { RLX_PLAUNCH(k_fwd_fused<3>, grid, block, lds, st, a); }
  else if (a.dbg == 15) { RLX_PLAUNCH(k_fwd_fused<15>, grid, block, lds, st, a); }
  else { RLX_PLAUNCH(k_fwd_fused<31>, grid, block, lds, st, a); }
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

}  // namespace rlx
