// gemm_bx.hip -- the hidden-layer GEMMs of the minibatch update on the half-precision matrix pipe (split-fp32 operands, gemm_bx.h).
//
//   k_gemm_bx<0>   C[M,N]   = act(A[M,K] @ W[K,N] + bias)              forward hidden layer      (mlp.hip: k_gemm_fwd)
//   k_gemm_bx<1>   HD[M,Kd] = (dZ[M,N] @ W[Kd,N]^T) * act'(HD)         input gradient, in place   (mlp.hip: k_gemm_dx)
//   k_gemm_dw_bx   dW[Kd,N] = Hprev[M,Kd]^T @ dZ[M,N] per M-slab       weight gradient slabs      (mlp.hip: k_gemm_dw)
//
// Same tiles, grids, epilogues and slab reduction as the exact-fp32 kernels they stand in for (128 x 128 block tile -- 64 x 128
// for shapes with one column tile --, wave tiles of 64 x 64, XCD-aware tile order); what changes is the main loop: three
// v_mfma_f32_32x32x16_f16 per 16 k instead of eight v_mfma_f32_32x32x2_f32 per 16 k at twice the cycles.  The weight operand
// of <0>/<1> comes from the fragment-ordered split image that bx_prepare_mlp / _nets / _mats lay out in one launch and
// register per scratch bank (inside the whole-update calls k_clip_adam keeps the images current, optim.hip); the kernels are
// used only while such an image is registered -- every other caller keeps the exact-fp32 engine.  k_gemm_dw_bx needs no
// image (both operands are activations) and is wave specialised (see its comment).
#include "gemm_bx.h"
#ifndef RLX_WS_MIN_WAVES
#define RLX_WS_MIN_WAVES 2   // waves per SIMD the wave-specialised kernels are compiled for.  4 would cap them at 128 VGPRs (two
                             // workgroups per CU also for the input-gradient form, 167 VGPRs): MEASURED 47.3 vs 26.1 us at the layer-3
                             // shape, 102.2 vs 97.1 ms per iteration -- its act'(H) epilogue spills 71 registers (round-4 probe script, git history)
#endif
#include "mlp.h"

namespace rlx {

__global__ __launch_bounds__(256) void k_bx_wfrag(BxJobs jobs) {
  int ji = 0;
#pragma unroll
  for (int q = 1; q < BX_MAX_JOBS; ++q)
    if (q < jobs.n && (int)blockIdx.x >= jobs.job[q].first_block) ji = q;
  const BxJob& jb = jobs.job[ji];
  const int idx = ((int)blockIdx.x - jb.first_block) * 256 + threadIdx.x;
  if (idx >= jb.KB * jb.NT * 64) return;
  const int lane = idx & 63, blk = idx >> 6, nt = blk % jb.NT, kb = blk / jb.NT;
  const int j = nt * 32 + (lane & 31), k0 = kb * 16 + 8 * (lane >> 5);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    v[e] = (k < jb.K && j < jb.N) ? (jb.trans ? jb.W[(int64_t)j * jb.ldw + k] : jb.W[(int64_t)k * jb.ldw + j]) : 0.f;
  }
  u32x4 pl[X_NP];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t p0, p1;
    bx_split2(v[2 * e] * X_WSCALE, v[2 * e + 1] * X_WSCALE, p0, p1);
    pl[0][e] = p0;
    pl[1][e] = p1;
  }
  u32x4* o = jb.out + ((int64_t)blk * X_NP) * 64 + lane;
  o[0] = pl[0];
  o[64] = pl[1];
}

// ---------------------------------------------------------------------------------------
// row-major activation operand [M, K] (contraction index contiguous) x weight image
// ---------------------------------------------------------------------------------------
// WS (wave specialisation, MI == 2 only): 512 threads -- waves 0-3 are the 2 x 2 arrangement of consumer waves (weight-fragment
// loads, LDS fragment reads, MFMAs, epilogue), waves 4-7 only fetch / split / store the activation tile of the next K-step.
// TWIN: grid.y == 2, blockIdx.y == 1 takes {A, Wf, bias, C} from tw.
template <int MODE, int ACT, bool APPLY, int MI, bool WS = false, bool TWIN = false>
__global__ __launch_bounds__(WS ? 2 * G_THREADS : G_THREADS, WS ? RLX_WS_MIN_WAVES : 2) void k_gemm_bx(const float* __restrict__ A, const u32x4* __restrict__ Wf,
                                                          const float* __restrict__ bias, float* __restrict__ C,
                                                          int64_t M, int N, int K, int lda, int ldc, int ntn,
                                                          const int32_t* __restrict__ m_dev, Twin tw, float sa, float so,
                                                          const float* __restrict__ Hsrc) {
  // Hsrc (optional, MODE 1 with APPLY): act' from Hsrc instead of C (out-of-place backward); twin launches: the second problem's in tw.p[2]
  // sa: power-of-two scale of the A operand (X_ASCALE for activations, the pass's gradient scale for dZ); so = 1 / (sa * X_WSCALE)
  constexpr int BM = 64 * MI;        // block tile BM x 128: four waves (2 x 2) of (32 * MI) x 64
  if (TWIN && blockIdx.y) {
    A = static_cast<const float*>(tw.p[0]);
    Wf = static_cast<const u32x4*>(tw.p[1]);
    bias = static_cast<const float*>(tw.p[2]);
    C = const_cast<float*>(static_cast<const float*>(tw.p[3]));
    if (MODE == 1) Hsrc = Hsrc ? static_cast<const float*>(tw.p[2]) : nullptr;   // (MODE 1 has no bias: the slot carries the twin's Hsrc)
  }
  __shared__ __attribute__((aligned(16))) char lds[2 * X_OPER];
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(tile / ntn) * BM;
  if (m_dev) {
    const int64_t mv = *m_dev;
    if (mv < M) M = mv;
    if (m0 >= M) return;
  }
  const int n0 = (tile % ntn) * G_BN;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = (wv >> 1) & 1, wn = wv & 1;
  const int a_r = (t & 255) >> 3, a_c = (t & 7) * 4;   // A tile: 8 threads per 32-float row, 32 rows per pass
  const int NT = ntn * 4, nt0 = (n0 >> 5) + wn * 2;
  if (WS && wv >= 4) {
    // ---- producers: stage tile kt + 1 while the consumers multiply tile kt; one barrier per K-tile
    const int nkp = (K + X_BK - 1) / X_BK;
    const bool plain = m0 + BM <= M && K % X_BK == 0;
    const float* ap = A + (m0 + a_r) * lda + a_c;
    float4 ra[2 * MI];
    auto load = [&](int kt) {
      const int kk = kt * X_BK;
#pragma unroll
      for (int p = 0; p < 2 * MI; ++p)
        ra[p] = plain ? *reinterpret_cast<const float4*>(ap + (int64_t)(32 * p) * lda + kk)
                      : ld4(A, m0 + a_r + 32 * p, kk + a_c, M, K, lda);
    };
    load(0);
#pragma unroll
    for (int p = 0; p < 2 * MI; ++p) bx_stage_k4(lds, a_r + 32 * p, a_c, ra[p], sa);
    if (nkp > 1) load(1);
    __syncthreads();
    for (int kt = 0; kt < nkp; ++kt) {
      if (kt + 1 < nkp) {
        char* nxt = lds + ((kt + 1) & 1) * X_OPER;
#pragma unroll
        for (int p = 0; p < 2 * MI; ++p) bx_stage_k4(nxt, a_r + 32 * p, a_c, ra[p], sa);
        if (kt + 2 < nkp) load(kt + 2);
      }
      __syncthreads();
    }
    return;
  }
  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = (K + X_BK - 1) / X_BK;
  float bv[2] = {0.f, 0.f};
  if (MODE == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + acc_col(wn, j, lane);
      bv[j] = col < N ? bias[col] : 0.f;
    }
  }
  if (WS) {
    // ---- consumers: weight fragments one 16-k step ahead in registers, activation fragments from the stage the producers filled
    u32x4 fb0[2][X_NP], fb1[2][X_NP], fa0[MI][X_NP], fa1[MI][X_NP];
    bx_load_b(Wf, 0, NT, nt0, lane, fb0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const char* cur = lds + (kt & 1) * X_OPER;
      bx_load_frag<MI>(cur, wm * 32 * MI, lane, 0, fa0);
      bx_load_b(Wf, 2 * kt + 1, NT, nt0, lane, fb1);
      bx_load_frag<MI>(cur, wm * 32 * MI, lane, 1, fa1);
      bx_mma<MI>(fa0, fb0, acc);
      bx_load_b(Wf, 2 * kt + 2 < 2 * nk ? 2 * kt + 2 : 2 * nk - 1, NT, nt0, lane, fb0);
      bx_mma<MI>(fa1, fb1, acc);
      __syncthreads();
    }
  } else
  // main loop: bx_kloop (gemm_bx.h) -- two LDS stages, the split + stores of the next K-tile interleaved with the MFMAs
  if (m0 + BM <= M && K % X_BK == 0) {
    const float* ap = A + (m0 + a_r) * lda + a_c;
    auto load = [&](int kt, float4 (&r)[2 * MI]) {
      const int kk = (kt < nk ? kt : nk - 1) * X_BK;
#pragma unroll
      for (int p = 0; p < 2 * MI; ++p) r[p] = *reinterpret_cast<const float4*>(ap + (int64_t)(32 * p) * lda + kk);
    };
    bx_kloop<MI>(lds, Wf, nk, NT, nt0, wm, lane, a_r, a_c, load, acc, sa);
  } else {
    auto load = [&](int kt, float4 (&r)[2 * MI]) {
      const int kk = (kt < nk ? kt : nk - 1) * X_BK;
#pragma unroll
      for (int p = 0; p < 2 * MI; ++p) r[p] = ld4(A, m0 + a_r + 32 * p, kk + a_c, M, K, lda);
    };
    bx_kloop<MI>(lds, Wf, nk, NT, nt0, wm, lane, a_r, a_c, load, acc, sa);
  }
  // accumulator register r of row tile i, lane l: row wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
  if (m0 + BM <= M && n0 + G_BN <= N) {
    // interior tile (uniform branch): independent stores off one per-lane base, no exec masking
    float* cb = C + (m0 + wm * 32 * MI + 4 * (lane >> 5)) * ldc + n0 + wn * 64 + (lane & 31);
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            cb[(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + j * 32] = act_fwd_t<ACT>(fmaf(acc[i][j][r], so, bv[j]));
    } else if (APPLY) {
      // one 32-row band at a time: its 32 activation loads are all issued before the first use
      const float* hsb = (Hsrc ? Hsrc : C) + (m0 + wm * 32 * MI + 4 * (lane >> 5)) * ldc + n0 + wn * 64 + (lane & 31);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        float h[2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) h[j][r] = hsb[(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + j * 32];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            cb[(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + j * 32] = acc[i][j][r] * so * act_grad_t<ACT>(h[j][r]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) cb[(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + j * 32] = acc[i][j][r] * so;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + acc_col(wn, j, lane);
    if (col >= N) continue;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) {
          const int64_t o = row * ldc + col;
          float v = acc[i][j][r] * so;
          if (MODE == 0) v = act_fwd_t<ACT>(v + bv[j]);
          else if (APPLY) v *= act_grad_t<ACT>((Hsrc ? Hsrc : C)[o]);
          C[o] = v;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------
// weight gradient: both operands are activations whose contraction index (the row m) is the slow index in memory.  The
// staging pass transposes in registers: a thread fetches 8 consecutive rows x 4 columns (eight 16-byte loads, 512 B
// contiguous per 32 lanes), splits, and stores for each of its 4 columns the 8 m-values as one 16-byte k-slot per plane.
//
// Wave specialisation.  The split-M grid is one workgroup per CU, so the kernel has to overlap its own phases (ablation of the
// four-wave form at the layer-2 shape: 14 us fixed + 22 us staging + 20 us MFMA + 8 us exposed load latency = the 63 us it
// took -- nothing overlapped).  Eight waves: waves 0-3 only multiply (the 2 x 2 arrangement of 64 x 64 wave tiles, accumulators
// and fragments), waves 4-7 only produce (waves 4-5 fetch / split / store the Hprev^T tile, waves 6-7 the dZ^T tile and the
// bias-gradient column sums).  Two LDS stages of 32 rows; while the consumers multiply stage kt the producers fill stage kt + 1
// and fetch tile kt + 2; ONE barrier per 32 rows.  Each SIMD holds one consumer and one producer wave: the VALU work of the
// split runs in the issue slots the MFMAs leave.
// ---------------------------------------------------------------------------------------
constexpr int XW_THREADS = 512;
// One weight-gradient problem (of one or, twin launches, two equally shaped networks)
struct DwArgs {
  const float* Hp;
  const float* dZ;
  float* partW;
  float* partB;
  const float *Hp1, *dZ1;     // TWIN: the second network's operands / outputs (blockIdx.y == 1)
  float *partW1, *partB1;
  int64_t M, Mc;
  int Kd, ldh, N, ntk, ntn;
  // RECOMPUTED Hprev operand (rc.X != NULL): Hprev = act(LayerNorm(X @ W1 + b1)) is not read from memory but rebuilt per 32-row
  // stage by the two Hprev producer waves -- observation planes x first-layer weight fragments on the fp16 pipe, the rows' LayerNorm
  // statistics as the forward kernel (k_l12fwd) left them -- so the [M, 512] first-layer activations never exist in HBM
  BxDwRecompute rc;
};
// TWIN: grid.y == 2.  Two JOBS per launch: blocks [0, nb0) work on job a, blocks [nb0, gridDim.x) on job b -- the weight
// gradients of two layers whose operands are both ready (the update's layer-3 and layer-2 gradients after the tail kernel): one
// launch, and the two jobs share the CUs, so each needs half the M-slabs (half the slab bytes the reduction reads back).
// REC: job a's Hprev operand is recomputed (DwArgs::rc) -- its own instantiation: the recompute path's registers (195 VGPRs) would
// otherwise cap the plain kernel at one workgroup per CU
template <bool TWIN, bool REC = false>
__global__ __launch_bounds__(XW_THREADS, 2) void k_gemm_dw_bx(DwArgs a, DwArgs b, int nb0, float sg, float so) {
  // sg: power-of-two scale of the dZ operand (the pass's gradient scale; Hprev is split times X_ASCALE); so = 1 / (X_ASCALE * sg)
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 2 stages x { Hprev^T tile (rows = kd), dZ^T tile (rows = n) }
  int bid = blockIdx.x, nblk = nb0;
  if (bid >= nb0) {
    a = b;
    bid -= nb0;
    nblk = gridDim.x - nb0;
  }
  const float* __restrict__ Hp = a.Hp;
  const float* __restrict__ dZ = a.dZ;
  float* __restrict__ partW = a.partW;
  float* __restrict__ partB = a.partB;
  if (TWIN && blockIdx.y) {
    Hp = a.Hp1;
    dZ = a.dZ1;
    partW = a.partW1;
    partB = a.partB1;
  }
  const int64_t M = a.M, Mc = a.Mc;
  const int Kd = a.Kd, ldh = a.ldh, N = a.N, ntk = a.ntk, ntn = a.ntn;
  const int ntiles = ntk * ntn;
  const int lb = xcd_remap(bid, nblk);
  const int s = lb / ntiles, tile = lb % ntiles;
  const int k0d = (tile / ntn) * G_BM;
  const int n0 = (tile % ntn) * G_BN;
  const int64_t mbeg = (int64_t)s * Mc;
  int64_t mend = mbeg + Mc;
  if (mend > M) mend = M;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nk = (int)((mend - mbeg + X_BK - 1) / X_BK);
  const bool interior = k0d + G_BM <= Kd && n0 + G_BN <= N;
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
  const int pt = t & 255, op = pt >> 7, tt = pt & 127, cg = tt & 31, mg = tt >> 5;   // producer coordinates
  const bool rec = REC && a.rc.X != nullptr;
  if (REC && wv >= 4 && rec && op == 0) {
    // ------------------------------------------------------------------ Hprev producers, recomputed operand
    // wave wh of the two owns column tiles 2 wh, 2 wh + 1 of the 128-wide kd tile.  Per 32-row stage: the rows' observations
    // -> fp16 planes in a WAVE-PRIVATE LDS tile (same-wave write -> read: no barrier), z1 by 12 MFMAs against the first-layer
    // weight fragments held in registers, LayerNorm with the forward's (mean, 1 / std), activation, and the accumulator layout
    // IS the transposed operand: lane (column li, half lh) holds rows rho(r, lh) = (r & 3) + 8 (r >> 2) + 4 lh, so registers
    // 8 s .. 8 s + 7 are one 16-byte k-slot (index 2 s + lh) of row `column` -- the rows of a stage sit in the k-slots in THAT
    // order, and the dZ producers below fetch their rows in the same order (only the agreement of the two k maps matters).
    const BxDwRecompute& R = a.rc;
    const int wh = wv - 4, li = lane & 31, lh = lane >> 5;
    const int O = R.O;
    const int64_t pdelta = (TWIN && blockIdx.y) ? R.pdelta1 : 0;
    const float* __restrict__ stats = (TWIN && blockIdx.y) ? R.stats1 : R.stats;
    const u32x4* __restrict__ W1x = reinterpret_cast<const u32x4*>((TWIN && blockIdx.y) ? R.W1x1 : R.W1x);
    const float xs = R.xmax ? x_scale_from_max(*R.xmax, X_ASCALE) : X_ASCALE;
    const float xinv = 1.0f / xs;
    u32x4 w1f[2][2][X_NP];
    float bias[2], gam[2], bet[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ctg = (k0d >> 5) + 2 * wh + j;                 // column tile of the whole first layer
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int p = 0; p < X_NP; ++p) w1f[kb][j][p] = W1x[((int64_t)(kb * R.NT1 + ctg) * X_NP + p) * 64 + lane];
      const int col = k0d + 32 * (2 * wh + j) + li;
      bias[j] = R.b1[pdelta + col];
      gam[j] = R.g[pdelta + col];
      bet[j] = R.be[pdelta + col];
    }
    char* xa = lds + 4 * X_OPER + wh * (X_NP * 2048 + 256);     // wave-private: 2 planes [32 rows][32 k], then mean[32], rstd[32]
    float* st_l = reinterpret_cast<float*>(xa + X_NP * 2048);
    for (int i = lane; i < X_NP * 2048 / 16; i += 64) reinterpret_cast<u32x4*>(xa)[i] = u32x4{0u, 0u, 0u, 0u};   // k >= O stays zero
    constexpr int XC = 9;                                        // 64 x 9 >= 32 rows x 17 observation values
    int xr_[XC], xk_[XC];
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      const int e = lane + 64 * c;
      xr_[c] = e / O;
      xk_[c] = e - xr_[c] * O;
    }
    float xv[XC], smean = 0.f, srstd = 0.f;
    auto xload = [&](int kt) {
      const int64_t m0 = mbeg + (int64_t)kt * X_BK;
      const float* xb = R.X + m0 * O;
      const int nvalid = (int)((mend - m0 < X_BK ? mend - m0 : X_BK)) * O;
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        const int e = lane + 64 * c;
        xv[c] = e < nvalid ? xb[e] : 0.f;
      }
      const bool rv = lane < 32 && m0 + lane < mend;
      smean = rv ? stats[m0 + lane] : 0.f;
      srstd = rv ? stats[M + m0 + lane] : 0.f;
    };
    auto produce = [&](int buf) {
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        if (lane + 64 * c < 32 * O) {
          uint32_t p0, p1;
          bx_split2(xv[c] * xs, 0.f, p0, p1);
          char* da = xa + bx_off(xr_[c], xk_[c] >> 3) + (xk_[c] & 7) * 2;
          *reinterpret_cast<uint16_t*>(da) = (uint16_t)p0;
          *reinterpret_cast<uint16_t*>(da + 2048) = (uint16_t)p1;
        }
      }
      if (lane < 32) { st_l[lane] = smean; st_l[32 + lane] = srstd; }
      f32x16 z[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[j][r] = bias[j] * (xs * X_WSCALE);
      const int nks = O > 16 ? 2 : 1;
      for (int kb = 0; kb < nks; ++kb) {
        u32x4 xf[X_NP];
#pragma unroll
        for (int p = 0; p < X_NP; ++p) xf[p] = *reinterpret_cast<const u32x4*>(xa + p * 2048 + bx_off(li, 2 * kb + lh));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[0]), __builtin_bit_cast(f16x8, w1f[kb][j][1]), z[j], 0, 0, 0);
          z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[1]), __builtin_bit_cast(f16x8, w1f[kb][j][0]), z[j], 0, 0, 0);
          z[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[0]), __builtin_bit_cast(f16x8, w1f[kb][j][0]), z[j], 0, 0, 0);
        }
      }
      const float zs = xinv * X_WINV;
      char* dst = lds + buf * 2 * X_OPER;                        // the Hprev^T operand of this stage
#pragma unroll
      for (int s8 = 0; s8 < 2; ++s8) {
        float4 mv[2], rv[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {                         // rows 16 s8 + 8 h2 + 4 lh + (0..3)
          mv[h2] = *reinterpret_cast<const float4*>(st_l + 16 * s8 + 8 * h2 + 4 * lh);
          rv[h2] = *reinterpret_cast<const float4*>(st_l + 32 + 16 * s8 + 8 * h2 + 4 * lh);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float mean = e < 4 ? (&mv[0].x)[e & 3] : (&mv[1].x)[e & 3];
            const float rs = e < 4 ? (&rv[0].x)[e & 3] : (&rv[1].x)[e & 3];
            // ELU as exp(y) - 1 from v_exp_f32 alone: 6e-8 ABSOLUTE error -- below the operand's fp16-plane resolution (the
            // forward's expm1 polynomial buys relative accuracy near 0 that this operand cannot carry; 10 VALU instructions
            // per element less in producers that are VALU-bound)
            const float y = fmaf((z[j][8 * s8 + e] * zs - mean) * rs, gam[j], bet[j]);
            v[e] = y > 0.f ? y : __expf(fminf(y, 0.f)) - 1.0f;
          }
          bx_stage_k8<true>(dst, 32 * (2 * wh + j) + li, 2 * s8 + lh, v, X_ASCALE);
        }
      }
    };
    xload(0);
    produce(0);
    if (nk > 1) xload(1);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) {
        produce((kt + 1) & 1);
        if (kt + 2 < nk) xload(kt + 2);
      }
      __syncthreads();
    }
    if (k0d == 0 && partB) {                      // the column-sum fold of the dZ producers (below): these 128 threads write it out
      const float* red = reinterpret_cast<const float*>(lds);
      __syncthreads();
      if (pt < 128 && n0 + pt < N) partB[(int64_t)s * N + n0 + pt] = (red[pt] + red[128 + pt]) + (red[256 + pt] + red[384 + pt]);
    }
    return;
  }
  if (wv >= 4) {
    // ------------------------------------------------------------------ producers
    const float* __restrict__ src = op ? dZ : Hp;
    const int ld = op ? N : ldh, c0 = (op ? n0 : k0d) + cg * 4, ncols = op ? N : Kd;
    const bool plain = interior && (mend - mbeg) % X_BK == 0;
    float4 rra[8];
    // k-slot mg of a stage holds rows mg * 8 + (0..7) -- or, next to a RECOMPUTED Hprev operand, the rows in the order that
    // operand's accumulator layout puts them: 16 (mg >> 1) + 4 (mg & 1) + (e & 3) + 8 (e >> 2)
    const int rbase = rec ? 16 * (mg >> 1) + 4 * (mg & 1) : mg * 8;
    const float* sp = src + (mbeg + rbase) * ld + c0;
    // (PLAIN is a compile-time argument of the loop below: with `if (plain)` inside, the two paths issue different numbers of loads,
    //  hipcc cannot count them at the join and every wait of the loop becomes vmcnt(0))
    auto load_plain = [&](int kt, float4 (&rr)[8]) {
      const int64_t m0 = (int64_t)kt * X_BK;
#pragma unroll
      for (int e = 0; e < 8; ++e) rr[e] = *reinterpret_cast<const float4*>(sp + (m0 + (rec ? (e & 3) + 8 * (e >> 2) : e)) * ld);
    };
    auto load_edge = [&](int kt, float4 (&rr)[8]) {
      const int64_t m0 = (int64_t)kt * X_BK;
#pragma unroll
      for (int e = 0; e < 8; ++e) rr[e] = ld4(src, mbeg + m0 + rbase + (rec ? (e & 3) + 8 * (e >> 2) : e), c0, mend, ncols, ld);
    };
    const float sc = op ? sg : X_ASCALE;
    auto stage = [&](int buf, const float4 (&rr)[8]) {
      char* dst = lds + buf * 2 * X_OPER + op * X_OPER;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = rr[e].x;
      bx_stage_k8<true>(dst, cg * 4 + 0, mg, v, sc);
      if (op) {
#pragma unroll
        for (int e = 0; e < 8; ++e) colsum[0] += v[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = rr[e].y;
      bx_stage_k8<true>(dst, cg * 4 + 1, mg, v, sc);
      if (op) {
#pragma unroll
        for (int e = 0; e < 8; ++e) colsum[1] += v[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = rr[e].z;
      bx_stage_k8<true>(dst, cg * 4 + 2, mg, v, sc);
      if (op) {
#pragma unroll
        for (int e = 0; e < 8; ++e) colsum[2] += v[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = rr[e].w;
      bx_stage_k8<true>(dst, cg * 4 + 3, mg, v, sc);
      if (op) {
#pragma unroll
        for (int e = 0; e < 8; ++e) colsum[3] += v[e];
      }
    };
    // TWO stages of rows in flight per producer wave (two register sets, the loop unrolled by two), every load unconditional (a stage
    // index past the end re-reads the last one): the waits are counted -- s_waitcnt vmcnt(8) lets the younger set stay in flight.
    // With ONE set, refilled after its stage was stored, a producer's period was one HBM round trip + the split per 32-row stage
    // (~1 us x 41 stages = the kernel's 47 us), and with the refill behind `if (kt + 2 < nk)` hipcc drained with vmcnt(0) anyway
    // (which is what the round-5 "two K-tiles in flight" experiment measured as neutral).
    float4 rrb[8];
    auto clampk = [&](int kt) { return kt < nk ? kt : nk - 1; };
    auto produce = [&](auto load) {
      load(0, rra);
      load(clampk(1), rrb);
      stage(0, rra);
      load(clampk(2), rra);
      __syncthreads();
      int kt = 0;
      for (; kt + 2 <= nk; kt += 2) {
        if (kt + 1 < nk) stage((kt + 1) & 1, rrb);      // stage kt + 1 (set B), then refill B with stage kt + 3
        load(clampk(kt + 3), rrb);
        __syncthreads();
        if (kt + 2 < nk) stage((kt + 2) & 1, rra);      // stage kt + 2 (set A), refill A with stage kt + 4
        load(clampk(kt + 4), rra);
        __syncthreads();
      }
      if (kt < nk) {                                    // odd count: one more barrier (tile kt + 1 does not exist)
        __syncthreads();
      }
    };
    if (plain) produce(load_plain);
    else produce(load_edge);
    if (k0d == 0 && partB) {
      // column sums: the dZ producers hold 4 columns each over their 8-row groups; fold the 4 row groups in fixed order
      float* red = reinterpret_cast<float*>(lds);   // [4][128]; the loop's last barrier closed every LDS read
      if (op) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[mg * 128 + cg * 4 + q] = colsum[q];
      }
      __syncthreads();
      if (pt < 128 && n0 + pt < N) partB[(int64_t)s * N + n0 + pt] = (red[pt] + red[128 + pt]) + (red[256 + pt] + red[384 + pt]);
    }
    return;
  }
  // -------------------------------------------------------------------- consumers
  const int wm = wv >> 1, wn = wv & 1;
  f32x16 acc[2][2];
  zero_acc(acc);
  u32x4 fa[2][X_NP], fb[2][X_NP];
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = lds + (kt & 1) * 2 * X_OPER;
    bx_load_frag<2, true>(cur, wm * 64, lane, 0, fa);
    bx_load_frag<2, true>(cur + X_OPER, wn * 64, lane, 0, fb);
    bx_mma<2>(fa, fb, acc);
    bx_load_frag<2, true>(cur, wm * 64, lane, 1, fa);
    bx_load_frag<2, true>(cur + X_OPER, wn * 64, lane, 1, fb);
    bx_mma<2>(fa, fb, acc);
    __syncthreads();
  }
  float* outW = partW + (int64_t)s * Kd * N;
  if (interior) {
    float* ob = outW + (int64_t)(k0d + wm * 64 + 4 * (lane >> 5)) * N + n0 + wn * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[(i * 32 + (r & 3) + 8 * (r >> 2)) * N + j * 32] = acc[i][j][r] * so;
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + acc_col(wn, j, lane);
      if (col >= N) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = k0d + acc_row(wm, i, r, lane);
          if (row < Kd) outW[(int64_t)row * N + col] = acc[i][j][r] * so;
        }
    }
  }
  if (k0d == 0 && partB) __syncthreads();   // matches the producers' barrier of the column-sum fold
}

// ---------------------------------------------------------------------------------------
// host side: weight images of one network, registered per scratch bank (policy / critic chains run concurrently)
// ---------------------------------------------------------------------------------------
static void add_job(BxJobs& jobs, int& blocks, int64_t& entries, const float* W, int ldw, int K, int N, int trans) {
  BxJob& j = jobs.job[jobs.n++];
  j.W = W;
  j.ldw = ldw;
  j.K = K;
  j.N = N;
  j.trans = trans;
  j.KB = 2 * div_up(K, X_BK);
  j.NT = 4 * div_up(N, G_BN);
  j.first_block = blocks;
  j.out = reinterpret_cast<u32x4*>(entries);   // offset for now; rebased once the arena is known
  blocks += div_up(j.KB * j.NT * 64, 256);
  entries += (int64_t)j.KB * j.NT * X_NP * 64;
}

int bx_prepare_mlp(rlx_ctx* ctx, const rlx_mlp_desc& d, const MlpLayout& L, const float* params, bool with_bwd,
                   hipStream_t st) {
  const int bank = ctx->bank;
  ctx->bx_n[bank] = 0;
  if (!ctx->gemm_bx) return RLX_OK;
  BxJobs jobs;
  jobs.n = 0;
  int blocks = 0;
  int64_t entries = 0;
  for (int l = 1; l < d.n_hidden; ++l) {
    const LayerOff& o = L.layer[l];
    if (o.in % 4 != 0 || o.out % 4 != 0) continue;
    add_job(jobs, blocks, entries, params + o.W, o.out, o.in, o.out, 0);                 // forward: B = W[in, out]
    if (with_bwd) add_job(jobs, blocks, entries, params + o.W, o.out, o.out, o.in, 1);   // input gradient: B = W^T
  }
  // narrow first layer: its forward image feeds the z1 recompute of the fused first-layer backward (l1fused.hip)
  if (with_bwd && jobs.n > 0 && d.in_dim <= 32 && l1fused_supported(d))
    add_job(jobs, blocks, entries, params + L.layer[0].W, L.layer[0].out, L.layer[0].in, L.layer[0].out, 0);
  if (jobs.n == 0) return RLX_OK;
  u32x4* arena = (u32x4*)scratch(ctx, SL_WFRAG, (size_t)entries * sizeof(u32x4));
  if (!arena) return RLX_ENOMEM;
  for (int i = 0; i < jobs.n; ++i) {
    BxJob& j = jobs.job[i];
    j.out = arena + reinterpret_cast<int64_t>(j.out);
    rlx_ctx::BxImage& im = ctx->bx_img[bank][i];
    im.W = j.W;
    im.trans = j.trans;
    im.K = j.K;
    im.N = j.N;
    im.img = j.out;
  }
  hipLaunchKernelGGL(k_bx_wfrag, dim3(blocks), dim3(256), 0, st, jobs);
  RLX_LAUNCH_CHECK();
  ctx->bx_n[bank] = jobs.n;
  return RLX_OK;
}

void bx_launch_wfrag(const BxJobs& jobs, int blocks, hipStream_t st) {
  hipLaunchKernelGGL(k_bx_wfrag, dim3(blocks), dim3(256), 0, st, jobs);
}

void bx_release(rlx_ctx* ctx) { ctx->bx_n[ctx->bank] = 0; }

int bx_prepare_nets(rlx_ctx* ctx, const BxNetSpec* nets, int n_nets, hipStream_t st, ScratchSlot slot, bool launch) {
  ctx->bx_n[0] = ctx->bx_n[1] = 0;
  if (!ctx->gemm_bx) return RLX_OK;
  BxJobs jobs;
  jobs.n = 0;
  int blocks = 0;
  int64_t entries = 0;
  for (int i = 0; i < n_nets; ++i) {
    const rlx_mlp_desc& d = *nets[i].d;
    const MlpLayout L = make_layout(d);
    for (int l = nets[i].first_layer ? 0 : 1; l < d.n_hidden; ++l) {
      const LayerOff& o = L.layer[l];
      if (o.out % 4 != 0 || (l > 0 && o.in % 4 != 0)) continue;
      if (jobs.n + 2 > BX_MAX_JOBS) break;
      add_job(jobs, blocks, entries, nets[i].params + o.W, o.out, o.in, o.out, 0);
      if (nets[i].with_bwd && l > 0) add_job(jobs, blocks, entries, nets[i].params + o.W, o.out, o.out, o.in, 1);
    }
  }
  if (jobs.n == 0) return RLX_OK;
  // (the arena of the CURRENT scratch bank: a caller working on the side stream under bank 1 must not share it with bank 0's users)
  u32x4* arena = (u32x4*)scratch(ctx, slot, (size_t)entries * sizeof(u32x4));
  if (!arena) return RLX_ENOMEM;
  for (int i = 0; i < jobs.n; ++i) {
    BxJob& j = jobs.job[i];
    j.out = arena + reinterpret_cast<int64_t>(j.out);
    for (int b = 0; b < 2; ++b) {
      rlx_ctx::BxImage& im = ctx->bx_img[b][i];
      im.W = j.W;
      im.trans = j.trans;
      im.K = j.K;
      im.N = j.N;
      im.img = j.out;
    }
  }
  if (launch) {
    hipLaunchKernelGGL(k_bx_wfrag, dim3(blocks), dim3(256), 0, st, jobs);
    RLX_LAUNCH_CHECK();
  }
  ctx->bx_n[0] = ctx->bx_n[1] = jobs.n;
  return RLX_OK;
}

void bx_release_all(rlx_ctx* ctx) { ctx->bx_n[0] = ctx->bx_n[1] = 0; }

BxEmit bx_emit_table(const rlx_ctx* ctx, const rlx_mlp_desc& d, const float* params) {
  BxEmit e;
  e.n = 0;
  if (!ctx->gemm_bx || !ctx->bx_keep[ctx->bank] || ctx->bx_n[ctx->bank] == 0) return e;
  const MlpLayout L = make_layout(d);
  for (int l = 0; l < d.n_hidden && e.n < 3; ++l) {      // (layer 0: only the narrow first layer of the fused backward has an image)
    const LayerOff& o = L.layer[l];
    const void* nn = bx_lookup(ctx, params + o.W, 0, o.in, o.out);
    const void* tt = l == 0 ? nullptr : bx_lookup(ctx, params + o.W, 1, o.out, o.in);
    if (!nn && !tt) continue;
    BxEmitLayer& q = e.l[e.n++];
    q.w_off = o.W;
    q.in = o.in;
    q.out = o.out;
    q.nn = const_cast<void*>(nn);
    q.tt = const_cast<void*>(tt);
    q.nt_nn = 4 * div_up(o.out, G_BN);
    q.nt_tt = 4 * div_up(o.in, G_BN);
  }
  return e;
}

int bx_prepare_mats(rlx_ctx* ctx, const BxMat* mats, int n, hipStream_t st) {
  const int bank = ctx->bank;
  ctx->bx_n[bank] = 0;
  if (!ctx->gemm_bx) return RLX_OK;
  BxJobs jobs;
  jobs.n = 0;
  int blocks = 0;
  int64_t entries = 0;
  for (int i = 0; i < n; ++i) {
    if (mats[i].K % 4 != 0 || mats[i].N % 4 != 0) continue;
    if (jobs.n + 2 > BX_MAX_JOBS) break;
    if (mats[i].fwd) add_job(jobs, blocks, entries, mats[i].W, mats[i].N, mats[i].K, mats[i].N, 0);
    if (mats[i].trans) add_job(jobs, blocks, entries, mats[i].W, mats[i].N, mats[i].N, mats[i].K, 1);
  }
  if (jobs.n == 0) return RLX_OK;
  u32x4* arena = (u32x4*)scratch(ctx, SL_WFRAG, (size_t)entries * sizeof(u32x4));
  if (!arena) return RLX_ENOMEM;
  for (int i = 0; i < jobs.n; ++i) {
    BxJob& j = jobs.job[i];
    j.out = arena + reinterpret_cast<int64_t>(j.out);
    rlx_ctx::BxImage& im = ctx->bx_img[bank][i];
    im.W = j.W;
    im.trans = j.trans;
    im.K = j.K;
    im.N = j.N;
    im.img = j.out;
  }
  hipLaunchKernelGGL(k_bx_wfrag, dim3(blocks), dim3(256), 0, st, jobs);
  RLX_LAUNCH_CHECK();
  ctx->bx_n[bank] = jobs.n;
  return RLX_OK;
}

const void* bx_lookup(const rlx_ctx* ctx, const float* W, int trans, int K, int N) {
  if (!ctx->gemm_bx) return nullptr;
  if ((ctx->bx_debug & 16) && !trans) return nullptr;
  if ((ctx->bx_debug & 32) && trans && K <= 128) return nullptr;
  if ((ctx->bx_debug & 128) && trans && K > 128) return nullptr;
  const int bank = ctx->bank;
  for (int i = 0; i < ctx->bx_n[bank]; ++i) {
    const rlx_ctx::BxImage& im = ctx->bx_img[bank][i];
    if (im.W == W && im.trans == trans && im.K == K && im.N == N) return im.img;
  }
  return nullptr;
}

#define RLX_BX_LAUNCH_WS(MODE, ACTV, APPLYV, GRID, ST, ...)                                                                          \
  if ((MODE) == 0) {                                                                                                              \
    switch (ACTV) {                                                                                                               \
      case RLX_ACT_TANH: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_TANH, false, 2, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break; \
      case RLX_ACT_ELU: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_ELU, false, 2, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;   \
      case RLX_ACT_RELU: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_RELU, false, 2, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break; \
      default: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_NONE, false, 2, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;           \
    }                                                                                                                             \
  } else if (!(APPLYV)) {                                                                                                         \
    RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_NONE, false, 2, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__);                     \
  } else {                                                                                                                        \
    switch (ACTV) {                                                                                                               \
      case RLX_ACT_TANH: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_TANH, true, 2, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;  \
      case RLX_ACT_ELU: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_ELU, true, 2, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;    \
      case RLX_ACT_RELU: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_RELU, true, 2, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;  \
      default: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_NONE, false, 2, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;           \
    }                                                                                                                             \
  }

#define RLX_BX_LAUNCH_WS_TWIN(MODE, ACTV, APPLYV, GRID, ST, ...)                                                                          \
  if ((MODE) == 0) {                                                                                                              \
    switch (ACTV) {                                                                                                               \
      case RLX_ACT_TANH: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_TANH, false, 2, true, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break; \
      case RLX_ACT_ELU: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_ELU, false, 2, true, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;   \
      case RLX_ACT_RELU: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_RELU, false, 2, true, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break; \
      default: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_NONE, false, 2, true, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;           \
    }                                                                                                                             \
  } else if (!(APPLYV)) {                                                                                                         \
    RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_NONE, false, 2, true, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__);                     \
  } else {                                                                                                                        \
    switch (ACTV) {                                                                                                               \
      case RLX_ACT_TANH: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_TANH, true, 2, true, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;  \
      case RLX_ACT_ELU: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_ELU, true, 2, true, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;    \
      case RLX_ACT_RELU: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_RELU, true, 2, true, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;  \
      default: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_NONE, false, 2, true, true>), GRID, dim3(2 * G_THREADS), 0, ST, __VA_ARGS__); break;           \
    }                                                                                                                             \
  }

#define RLX_BX_LAUNCH_MI(MIV, MODE, ACTV, APPLYV, GRID, ST, ...)                                                                   \
  if ((MODE) == 0) {                                                                                                              \
    switch (ACTV) {                                                                                                               \
      case RLX_ACT_TANH: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_TANH, false, MIV>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break; \
      case RLX_ACT_ELU: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_ELU, false, MIV>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;   \
      case RLX_ACT_RELU: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_RELU, false, MIV>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break; \
      default: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_NONE, false, MIV>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;           \
    }                                                                                                                             \
  } else if (!(APPLYV)) {                                                                                                         \
    RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_NONE, false, MIV>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__);                             \
  } else {                                                                                                                        \
    switch (ACTV) {                                                                                                               \
      case RLX_ACT_TANH: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_TANH, true, MIV>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;  \
      case RLX_ACT_ELU: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_ELU, true, MIV>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;    \
      case RLX_ACT_RELU: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_RELU, true, MIV>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;  \
      default: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_NONE, false, MIV>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;           \
    }                                                                                                                             \
  }

#define RLX_BX_LAUNCH_TWIN(MODE, ACTV, APPLYV, GRID, ST, ...)                                                                       \
  if ((MODE) == 0) {                                                                                                              \
    switch (ACTV) {                                                                                                               \
      case RLX_ACT_TANH: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_TANH, false, 1, false, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break; \
      case RLX_ACT_ELU: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_ELU, false, 1, false, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;   \
      case RLX_ACT_RELU: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_RELU, false, 1, false, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break; \
      default: RLX_PLAUNCH((k_gemm_bx<0, RLX_ACT_NONE, false, 1, false, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;           \
    }                                                                                                                             \
  } else if (!(APPLYV)) {                                                                                                         \
    RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_NONE, false, 1, false, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__);                  \
  } else {                                                                                                                        \
    switch (ACTV) {                                                                                                               \
      case RLX_ACT_TANH: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_TANH, true, 1, false, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;  \
      case RLX_ACT_ELU: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_ELU, true, 1, false, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;    \
      case RLX_ACT_RELU: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_RELU, true, 1, false, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;  \
      default: RLX_PLAUNCH((k_gemm_bx<1, RLX_ACT_NONE, false, 1, false, true>), GRID, dim3(G_THREADS), 0, ST, __VA_ARGS__); break;           \
    }                                                                                                                             \
  }

// 64-row block tiles when 128-row tiles would not give every CU its two workgroups
static inline int bx_row_tiles(const rlx_ctx* ctx, int64_t M, int ntn) {
  if (ctx->bx_force_mi == 1 || ctx->bx_force_mi == 2) return ctx->bx_force_mi;
  return (div_up(M, G_BM) * ntn < 2 * ctx->num_cus) ? 1 : 2;
}

// twin launches of the 64-row tile form (batches whose single launch leaves most of the chip idle): what sac.hip asks for; the
// launchers also take a twin at the 128-row wave-specialised shapes (ppo.hip's policy || critic launches)
bool bx_twin_usable(const rlx_ctx* ctx, int64_t M, int N) { return bx_row_tiles(ctx, M, div_up(N, G_BN)) == 1; }

// tw (optional; bx_twin_usable(ctx, M, N)): {A, image, bias, C} of a second problem of the same shape, same launch
int bx_launch_fwd(rlx_ctx* ctx, const float* A, const void* img, const float* bias, float* C, int64_t M, int N, int K,
                  int act, hipStream_t st, int lda, const int32_t* m_dev, const Twin* tw) {
  ProfScope prof(m_dev ? nullptr : ctx, PK_GEMM_FWD, (tw ? 4.0 : 2.0) * (double)M * N * K, st, (tw ? 2.0 : 1.0) * gemm_bytes(M, N, K), M, N, K, 1);
  const int ntn = div_up(N, G_BN);
  if (tw && bx_row_tiles(ctx, M, ntn) != 1) {      // large batches: the wave-specialised 128-row form, both problems in one grid
    RLX_BX_LAUNCH_WS_TWIN(0, act, 0, dim3(div_up(M, G_BM) * ntn, 2), st, A, (const u32x4*)img, bias, C, M, N, K, lda > 0 ? lda : K, N,
                          ntn, m_dev, *tw, X_ASCALE, X_AINV * X_WINV, (const float*)nullptr);
  } else if (tw) {
    RLX_BX_LAUNCH_TWIN(0, act, 0, dim3(div_up(M, 64) * ntn, 2), st, A, (const u32x4*)img, bias, C, M, N, K, lda > 0 ? lda : K, N,
                       ntn, m_dev, *tw, X_ASCALE, X_AINV * X_WINV, (const float*)nullptr);
  } else if (bx_row_tiles(ctx, M, ntn) == 1) {
    RLX_BX_LAUNCH_MI(1, 0, act, 0, dim3(div_up(M, 64) * ntn), st, A, (const u32x4*)img, bias, C, M, N, K, lda > 0 ? lda : K, N,
                     ntn, m_dev, Twin{}, X_ASCALE, X_AINV * X_WINV, (const float*)nullptr);
  } else {
    RLX_BX_LAUNCH_WS(0, act, 0, dim3(div_up(M, G_BM) * ntn), st, A, (const u32x4*)img, bias, C, M, N, K, lda > 0 ? lda : K, N,
                     ntn, m_dev, Twin{}, X_ASCALE, X_AINV * X_WINV, (const float*)nullptr);
  }
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// HD[M, Kd(ldo)] = (dZ[M, N] @ W[Kd, N]^T) (* act'(HD or hsrc));  tw (optional): {dZ, image, hsrc or NULL, HD} of the second problem
int bx_launch_dx(rlx_ctx* ctx, const float* dZ, const void* img, float* HD, int64_t M, int N, int Kd, int ldo, int act,
                 int apply, hipStream_t st, const Twin* tw, const float* hsrc) {
  RLX_REQUIRE(!tw || (hsrc != nullptr) == (tw->p[2] != nullptr), RLX_EINVAL, "bx_launch_dx: a twin launch is out of place for both problems or for neither");
  const float gs = ctx->bx_gscale;   // the pass's gradient scale (gemm_bx.h)
  ProfScope prof(ctx, PK_GEMM_DX, (tw ? 4.0 : 2.0) * (double)M * N * Kd, st, (tw ? 2.0 : 1.0) * gemm_bytes(M, Kd, N, apply), M, Kd, N, 1);
  const int ntn = div_up(Kd, G_BN);
  if (tw && bx_row_tiles(ctx, M, ntn) != 1) {
    RLX_BX_LAUNCH_WS_TWIN(1, act, apply, dim3(div_up(M, G_BM) * ntn, 2), st, dZ, (const u32x4*)img, (const float*)nullptr, HD, M, Kd,
                          N, N, ldo, ntn, (const int32_t*)nullptr, *tw, gs, X_WINV / gs, hsrc);
  } else if (tw) {
    RLX_BX_LAUNCH_TWIN(1, act, apply, dim3(div_up(M, 64) * ntn, 2), st, dZ, (const u32x4*)img, (const float*)nullptr, HD, M, Kd,
                       N, N, ldo, ntn, (const int32_t*)nullptr, *tw, gs, X_WINV / gs, hsrc);
  } else if (bx_row_tiles(ctx, M, ntn) == 1) {
    RLX_BX_LAUNCH_MI(1, 1, act, apply, dim3(div_up(M, 64) * ntn), st, dZ, (const u32x4*)img, (const float*)nullptr, HD, M, Kd, N,
                     N, ldo, ntn, (const int32_t*)nullptr, Twin{}, gs, X_WINV / gs, hsrc);
  } else {
    RLX_BX_LAUNCH_WS(1, act, apply, dim3(div_up(M, G_BM) * ntn), st, dZ, (const u32x4*)img, (const float*)nullptr, HD, M, Kd,
                     N, N, ldo, ntn, (const int32_t*)nullptr, Twin{}, gs, X_WINV / gs, hsrc);
  }
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

bool bx_dw_usable(const rlx_ctx* ctx, int64_t M, int Kd, int ldh, int N) {
  return !(ctx->bx_debug & 64) && ctx->gemm_bx && M >= 4096 && N % 4 == 0 && ldh % 4 == 0 && ldh >= ((Kd + 3) & ~3);   // 16-byte row loads: a ragged Kd needs padded rows
}

static int dw_attr() {
  static AttrOnce attr_set;      
  if (!attr_set.done()) {
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_dw_bx<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    4 * X_OPER + 16384));
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_dw_bx<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    4 * X_OPER + 16384));
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_dw_bx<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    4 * X_OPER + 16384));
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_dw_bx<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    4 * X_OPER + 16384));
    attr_set.mark();  
  }
  return RLX_OK;
}

static DwArgs dw_args(const float* Hp, const float* dZ, float* pW, float* pB, int64_t M, int Kd, int ldh, int N, int64_t Mc, int ntk,
                      int ntn, const Twin* tw) {
  DwArgs a;
  a.Hp = Hp; a.dZ = dZ; a.partW = pW; a.partB = pB;
  a.Hp1 = tw ? static_cast<const float*>(tw->p[0]) : nullptr;
  a.dZ1 = tw ? static_cast<const float*>(tw->p[1]) : nullptr;
  a.partW1 = tw ? const_cast<float*>(static_cast<const float*>(tw->p[2])) : nullptr;
  a.partB1 = tw ? const_cast<float*>(static_cast<const float*>(tw->p[3])) : nullptr;
  a.M = M; a.Mc = Mc; a.Kd = Kd; a.ldh = ldh; a.N = N; a.ntk = ntk; a.ntn = ntn;
  a.rc = BxDwRecompute{};
  return a;
}
constexpr size_t DW_REC_LDS = 2 * (X_NP * 2048 + 256);   // the two Hprev producer waves' private observation tiles + row statistics

// tw (optional): {Hp, dZ, pW, pB} of a second problem of the same shape (grid.y == 2)
int bx_launch_dw(rlx_ctx* ctx, const float* Hp, const float* dZ, float* pW, float* pB, int64_t M, int Kd, int ldh, int N,
                 int64_t Mc, int S, int ntk, int ntn, hipStream_t st, const Twin* tw) {
  int rc = dw_attr();
  if (rc) return rc;
  const float gs = ctx->bx_gscale;   // the pass's gradient scale (gemm_bx.h)
  ProfScope prof(ctx, PK_GEMM_DW, (tw ? 4.0 : 2.0) * (double)M * Kd * N, st, (tw ? 2.0 : 1.0) * gemm_bytes(Kd, N, M), Kd, N, (int)M, 1);
  const DwArgs a = dw_args(Hp, dZ, pW, pB, M, Kd, ldh, N, Mc, ntk, ntn, tw);
  const int nb = S * ntk * ntn;
  if (tw) {
    RLX_PLAUNCH(k_gemm_dw_bx<true>, dim3(nb, 2), dim3(XW_THREADS), 4 * X_OPER, st, a, a, nb, gs, X_AINV / gs);
  } else {
    RLX_PLAUNCH(k_gemm_dw_bx<false>, dim3(nb), dim3(XW_THREADS), 4 * X_OPER, st, a, a, nb, gs, X_AINV / gs);
  }
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// two weight-gradient problems over the same rows in ONE launch (k_gemm_dw_bx's two jobs); tw0 / tw1: their twins (both or neither)
int bx_launch_dw2(rlx_ctx* ctx, const BxDwJob& j0, const BxDwJob& j1, int64_t M, hipStream_t st, const Twin* tw0, const Twin* tw1) {
  int rc = dw_attr();
  if (rc) return rc;
  RLX_REQUIRE((tw0 == nullptr) == (tw1 == nullptr), RLX_EINVAL, "bx_launch_dw2: both jobs or neither must have a twin");
  const float gs = ctx->bx_gscale;
  const double nets = tw0 ? 2.0 : 1.0;
  // (one profiler row per launch, keyed by the larger job's shape; flops and bytes of both jobs)
  ProfScope prof(ctx, PK_GEMM_DW, nets * 2.0 * (double)M * ((double)j0.Kd * j0.N + (double)j1.Kd * j1.N), st,
                 nets * (gemm_bytes(j0.Kd, j0.N, M) + gemm_bytes(j1.Kd, j1.N, M)), j0.Kd + j1.Kd, j0.N + j1.N, (int)M, 1);
  DwArgs a = dw_args(j0.Hp, j0.dZ, j0.pW, j0.pB, M, j0.Kd, j0.ldh, j0.N, j0.Mc, j0.ntk, j0.ntn, tw0);
  const DwArgs b = dw_args(j1.Hp, j1.dZ, j1.pW, j1.pB, M, j1.Kd, j1.ldh, j1.N, j1.Mc, j1.ntk, j1.ntn, tw1);
  if (j0.rc) {
    RLX_REQUIRE(j0.rc->X && j0.rc->W1x && j0.rc->stats && j0.Kd == 512 && j0.rc->O <= 32 && (!tw0 || (j0.rc->W1x1 && j0.rc->stats1)),
                RLX_EINVAL, "bx_launch_dw2: incomplete recompute description");
    a.rc = *j0.rc;
  }
  const int nb0 = j0.S * j0.ntk * j0.ntn, nb1 = j1.S * j1.ntk * j1.ntn;
  const size_t lds = 4 * X_OPER + (j0.rc ? DW_REC_LDS : 0);
  if (j0.rc) {
    if (tw0) { RLX_PLAUNCH((k_gemm_dw_bx<true, true>), dim3(nb0 + nb1, 2), dim3(XW_THREADS), lds, st, a, b, nb0, gs, X_AINV / gs); }
    else { RLX_PLAUNCH((k_gemm_dw_bx<false, true>), dim3(nb0 + nb1), dim3(XW_THREADS), lds, st, a, b, nb0, gs, X_AINV / gs); }
  } else if (tw0) {
    RLX_PLAUNCH(k_gemm_dw_bx<true>, dim3(nb0 + nb1, 2), dim3(XW_THREADS), lds, st, a, b, nb0, gs, X_AINV / gs);
  } else {
    RLX_PLAUNCH(k_gemm_dw_bx<false>, dim3(nb0 + nb1), dim3(XW_THREADS), lds, st, a, b, nb0, gs, X_AINV / gs);
  }
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

}  // namespace rlx
