// gemm_px.hip -- the hidden-layer GEMMs of the large-minibatch update on PLANE tensors (gfx950).
//
// gemm_bx.hip's kernels take fp32 activations and split them into two fp16 planes on the consumer side: per K-tile every
// workgroup loads fp32 through VGPRs, spends ~6 VALU instructions per pair of values and writes the planes to LDS -- work that
// sits next to the MFMAs and is repeated by every consumer of a tensor.  Here the PRODUCER of a tensor emits the planes once
// (row-major fp16 [M][C], high plane and low plane, power-of-two scaled: gemm_bx.h) and every consumer moves them with
// global_load_lds (16 bytes per lane straight into LDS, no VGPR, no VALU):
//
//   k_gemm_px<0>   Cp[M,N]  = planes(act(Ap[M,K] @ W[K,N] + bias))        forward hidden layer
//   k_gemm_px<1>   Dp[M,Kd] = planes((dZp[M,N] @ W[Kd,N]^T) * act'(Hp))   input gradient (in place over Hp when Dp == Hp)
//   k_gemm_dw_px   dW[Kd,N] = Hp[M,Kd]^T @ dZp[M,N] per M-slab            weight gradient slabs (fp32 out, k_reduce_segments)
//
// Row-major operand (k_gemm_px): a stage is [128 rows][64 k] per plane = 128-byte rows of eight 16-byte k-slots; one
// global_load_lds piece is 8 rows x 128 B (full cache lines).  The LDS image is lane-linear, so the bank swizzle lives in the
// SOURCE address: the lane that fills (row r, slot d) fetches k-slot d ^ ((r >> 1) & 7), and the ds_read_b128 of a fragment
// applies the same XOR -- the 16 lanes of every service group then fall on 16 distinct 16-byte slots of the bank row.
// Eight waves; wave w owns 32 output columns of ALL rows of the block tile (WM = 1: 128 x 256 tile, N = 256 covered by one
// workgroup, so a row of the operand is staged once) or 64 rows x 32 columns (WM = 2: 128 x 128 tile).  Weight fragments come
// from the fragment-ordered image in L2 as before, one 16-k step ahead in registers.
// Weight-gradient kernel: the contraction index is the ROW of both operands; the row-major tiles are staged as [16 k][32 col]
// sub-tiles (one 1-KiB piece = one MFMA operand fragment) and read with ds_read_b64_tr_b16, the hardware transpose read:
// inside a 16-lane group, lane i receives elements (i & 3) of the four 8-byte pieces addressed by lanes 4 j + (i >> 2), j = 0..3
// -- with lane p pointing at (k = p >> 2, columns 4 (p & 3) ..) lane i gets column i of four consecutive k (probe:
// tools/probes/tr_probe.hip).  No register transposition, no split: zero VALU in the loop.
#include "gemm_bx.h"
#include <type_traits>
#include "mlp.h"

namespace rlx {

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef const __attribute__((address_space(1))) uint32_t glb_u32;

// 16 bytes per lane: global (per-lane address) -> LDS (wave-uniform base + 16 * lane)
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_u32*)g, (lds_u32*)lds_wave_base, 16, 0, 0);
}

constexpr int P_BM = 128, P_BK = 32;
constexpr int P_ROWB = 2 * P_BK;             // 64 bytes per row of one plane of a stage
constexpr int P_PLANE = P_BM * P_ROWB;       // 8 KiB
constexpr int P_ABYTES = X_NP * P_PLANE;     // 16 KiB: the activation part of a stage
constexpr int P_NSTAGE = 3;
constexpr int P_THREADS = 512;
template <int WN> constexpr int p_stage_bytes() { return P_ABYTES + 2 * WN * X_NP * 1024; }   // + weight fragments of two 16-k steps

// byte offset of k-slot `s` (8 k = 16 B, s < 4) of row r inside one plane of a stage (the swizzle of gemm_bx.h's bx_off)
__device__ __forceinline__ int px_off(int r, int s) { return r * P_ROWB + ((s ^ ((r >> 2) & 3)) << 4); }

// One stage = 32 k of the activation tile (row-major planes, 16 pieces of [16 rows][64 B]) + the weight fragments of its two
// 16-k steps for the workgroup's WN column tiles (4 WN pieces, each already one MFMA operand in the image).  All of it moves
// by global_load_lds; wave w issues pieces w, w + 8, ...  (A: piece id = plane * 8 + row group; lane l fills row 16 g + (l >> 2),
// slot l & 3 from k-slot (l & 3) ^ ((row >> 2) & 3).)
template <int WN>
__device__ __forceinline__ void px_issue_stage(const uint16_t* __restrict__ hi, int64_t plane_stride, int64_t m0, int64_t M,
                                               int ld, int kt, const u32x4* __restrict__ Wf, int NT, int nt0,
                                               char* __restrict__ stage, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = wave + 8 * i, pl = id >> 3, rr = 16 * (id & 7) + (lane >> 2);
    const int s = (lane & 3) ^ ((rr >> 2) & 3);
    int64_t row = m0 + rr;
    row = row < M ? row : M - 1;             // ragged tile: any valid row (its outputs are not stored)
    glds16(hi + pl * plane_stride + row * ld + kt * P_BK + 8 * s, stage + pl * P_PLANE + (id & 7) * 1024);
  }
  constexpr int NB = 2 * WN * X_NP;          // weight pieces per stage
#pragma unroll
  for (int i = 0; i < NB / 8; ++i) {
    const int id = wave + 8 * i, ks = id / (WN * X_NP), rem = id % (WN * X_NP), j = rem / X_NP, pl = rem % X_NP;
    glds16(Wf + ((int64_t)((2 * kt + ks) * NT + nt0 + j) * X_NP + pl) * 64 + lane, stage + P_ABYTES + id * 1024);
  }
}

// one value pair per lane after the exchange: a lane with an even column index holds (its own, the odd neighbour's) values of
// accumulator register 2t, the odd lane (the even neighbour's, its own) of register 2t + 1 -- two ADJACENT columns of ONE row, i.e.
// one 4-byte element of a row-major fp16 plane.  v0 / v1: the lane's registers 2t / 2t + 1.
__device__ __forceinline__ void px_pair(float v0, float v1, bool odd, float& lo_col, float& hi_col) {
  const float give = odd ? v0 : v1;
  const float got = dpp_f(give, 0);        // quad_perm [1,0,3,2]: the neighbour's give-away
  lo_col = odd ? got : v0;
  hi_col = odd ? v1 : got;
}

// MODE 0: Cp = planes(act(Ap @ W + bias)) (scale X_ASCALE in, X_ASCALE out);  MODE 1: Dp = planes((Ap @ W^T image) * act'(Hp)),
// Ap = dZ planes (scale gs in and out), Hp = activation planes (scale X_ASCALE); Dp may alias Hp (each lane reads before it writes).
// WM = 1: block tile 128 x 256 (wave = 128 rows x 32 columns); WM = 2: 128 x 128 (wave = 64 rows x 32 columns).
// Pipeline: three LDS stages, two in flight; per stage ONE raw s_barrier behind a COUNTED s_waitcnt vmcnt (the stage issued last
// stays in flight across the barrier -- __syncthreads() would drain the LDS-DMA queue); nothing in the loop is a register load,
// so hipcc adds no waits of its own.
template <int MODE, int ACT, int WM>
__global__ __launch_bounds__(P_THREADS, 2) void k_gemm_px(const uint16_t* __restrict__ Ap, int64_t a_stride, const u32x4* __restrict__ Wf,
                                                          const float* __restrict__ bias, const uint16_t* __restrict__ Hp,
                                                          uint16_t* __restrict__ Cp, int64_t c_stride, int64_t M, int N, int K,
                                                          int ntn, float so, float sc_out) {
  // so: accumulator -> value (1 / (operand scales)); sc_out: scale of the emitted planes
  constexpr int WN = 8 / WM, MI = 4 / WM, BN = 32 * WN, STAGE = p_stage_bytes<WN>();
  constexpr int OPS = 2 + 2 * WN * X_NP / 8;                        // LDS-DMA instructions per wave and stage
  extern __shared__ __attribute__((aligned(16))) char lds[];       // P_NSTAGE stages (the ONLY LDS object of the kernel)
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(tile / ntn) * P_BM;
  const int n0 = (tile % ntn) * BN;
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, lh = lane >> 5;
  const int wm = wv / WN, wn = wv % WN;
  const int NT = N >> 5, nt0 = n0 >> 5;                             // 32-column tiles of the image per 16-k block; the workgroup's first
  const int nk = K / P_BK;
  f32x16 acc[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  px_issue_stage<WN>(Ap, a_stride, m0, M, K, 0, Wf, NT, nt0, lds, wv, lane);
  if (nk > 1) px_issue_stage<WN>(Ap, a_stride, m0, M, K, 1, Wf, NT, nt0, lds + STAGE, wv, lane);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                   // stage kt has landed for every wave; stage kt - 1 is read out
    if (kt + 2 < nk) px_issue_stage<WN>(Ap, a_stride, m0, M, K, kt + 2, Wf, NT, nt0, lds + ((kt + 2) % P_NSTAGE) * STAGE, wv, lane);
    const char* cur = lds + (kt % P_NSTAGE) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[MI][X_NP], fb[X_NP];
#pragma unroll
      for (int p = 0; p < X_NP; ++p) fb[p] = *reinterpret_cast<const u32x4*>(cur + P_ABYTES + ((ks * WN + wn) * X_NP + p) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * 32 * MI + 32 * i + li;
#pragma unroll
        for (int p = 0; p < X_NP; ++p) fa[i][p] = *reinterpret_cast<const u32x4*>(cur + p * P_PLANE + px_off(row, 2 * ks + lh));
      }
#define RLX_PX_STEP(P, Q)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                  \
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][P]),                        \
                                                      __builtin_bit_cast(f16x8, fb[Q]), acc[i], 0, 0, 0);
      RLX_PX_STEP(0, 1)
      RLX_PX_STEP(1, 0)
      RLX_PX_STEP(0, 0)
#undef RLX_PX_STEP
    }
  }
  // ---- epilogue: lane (column li of the wave's 32, rows rho(r, lh) + 32 i) -> pairs of adjacent columns -> fp16 planes
  const bool odd = (li & 1) != 0;
  const int colp = n0 + 32 * wn + (li & ~1);                        // first column of the lane's pair
  float b0 = 0.f, b1 = 0.f;
  if (MODE == 0) { b0 = bias[colp]; b1 = bias[colp + 1]; }
  uint16_t* chi = Cp;
  uint16_t* clo = Cp + c_stride;
  const uint16_t* hhi = Hp;
  const uint16_t* hlo = Hp + c_stride;
  // a full tile takes the branch-free body: every `if (row < M)` block would otherwise start with hipcc's s_waitcnt vmcnt(0) for
  // the bias / activation loads, and on gfx950 that counter also holds the stores of the block before it
  auto emit = [&](auto guard) {
    constexpr bool GUARD = decltype(guard)::value;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int64_t rb = m0 + wm * 32 * MI + 32 * i + 4 * lh;
  #pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const int r = 2 * tt + (odd ? 1 : 0);                          // the register whose row this lane emits
        const int64_t row = rb + (r & 3) + 8 * (r >> 2);
        float x0, x1;
        px_pair(acc[i][2 * tt], acc[i][2 * tt + 1], odd, x0, x1);
        if (!GUARD || row < M) {
          const int64_t o = row * N + colp;
          float y0, y1;
          if (MODE == 0) {
            y0 = act_fwd_t<ACT>(fmaf(x0, so, b0));
            y1 = act_fwd_t<ACT>(fmaf(x1, so, b1));
          } else {
            const uint32_t h0 = *reinterpret_cast<const uint32_t*>(hhi + o), h1 = *reinterpret_cast<const uint32_t*>(hlo + o);
            const float ha = (bx_lo(h0) + bx_lo(h1)) * X_AINV, hb = (bx_hi(h0) + bx_hi(h1)) * X_AINV;
            y0 = x0 * so * act_grad_t<ACT>(ha);
            y1 = x1 * so * act_grad_t<ACT>(hb);
          }
          uint32_t p0, p1;
          bx_split2(y0 * sc_out, y1 * sc_out, p0, p1);
          *reinterpret_cast<uint32_t*>(chi + o) = p0;
          *reinterpret_cast<uint32_t*>(clo + o) = p1;
        }
      }
    }
  };
  if (m0 + P_BM <= M) emit(std::false_type{});
  else emit(std::true_type{});
}

// ---------------------------------------------------------------------------------------
// weight gradient on plane operands: dW[Kd, N] slab = Hp[rows, Kd]^T @ dZp[rows, N]
// Block tile 128 (kd) x 128 (n), four waves (2 x 2) of 64 x 64, two workgroups per CU.  A stage is ONE 16-row contraction step:
// per operand and plane four 32-column sub-tiles of [16 k][32 col] fp16 = 1 KiB each (row pitch 64 B: the four rows a transpose
// read touches cover all 64 banks once).  Piece order inside a stage: (operand * 2 + plane) * 4 + column tile.  Four stage buffers,
// three stages in flight (48 KiB per workgroup: the latency-bandwidth product of the CU's HBM share), one raw s_barrier per step
// behind a counted s_waitcnt vmcnt.
// ---------------------------------------------------------------------------------------
constexpr int PW_STAGE = 2 * X_NP * 4 * 1024;          // 16 KiB
constexpr int PW_NSTAGE = 4;
constexpr int PW_THREADS = 256;

__device__ __forceinline__ void pw_issue_stage(const uint16_t* __restrict__ Hh, int64_t h_stride, int ldh, int c0h,
                                               const uint16_t* __restrict__ Zh, int64_t z_stride, int ldz, int c0z,
                                               int64_t mrow0, int64_t mend, char* __restrict__ stage, int wave, int lane) {
  // 16 pieces per stage, 4 per wave: piece id = wave * 4 + i
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = wave * 4 + i, op = id >> 3, pl = (id >> 2) & 1, ct = id & 3;
    int64_t row = mrow0 + (lane >> 2);
    row = row < mend ? row : mend - 1;        // ragged last step: a clamped copy (its k-slots are masked in the dZ operand)
    const uint16_t* src = op ? Zh + pl * z_stride + row * ldz + c0z + 32 * ct + 8 * (lane & 3)
                             : Hh + pl * h_stride + row * ldh + c0h + 32 * ct + 8 * (lane & 3);
    glds16(src, stage + id * 1024);
  }
}

// two transpose reads = the 8 k-values (half lh of the 16-k step) of column (lane & 31) of one [16 k][32 col] sub-tile.
// Issued WITHOUT a wait: the caller issues all reads of a step, then one s_waitcnt lgkmcnt(0) + sched_barrier.
__device__ __forceinline__ void pw_frag_issue(unsigned a, uint64_t& v0, uint64_t& v1) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:256" : "=&v"(v0), "=&v"(v1) : "v"(a) : "memory");
}
__device__ __forceinline__ u32x4 pw_join(uint64_t v0, uint64_t v1) {
  u32x4 f;
  f[0] = (uint32_t)v0; f[1] = (uint32_t)(v0 >> 32); f[2] = (uint32_t)v1; f[3] = (uint32_t)(v1 >> 32);
  return f;
}

__global__ __launch_bounds__(PW_THREADS, 2) void k_gemm_dw_px(const uint16_t* __restrict__ Hh, int64_t h_stride, int ldh,
                                                              const uint16_t* __restrict__ Zh, int64_t z_stride, float* __restrict__ partW,
                                                              float* __restrict__ partB, int64_t M, int Kd, int N, int64_t Mc,
                                                              int ntk, int ntn, float so, float zinv) {
  // so = 1 / (X_ASCALE * gs): accumulator -> dW;  zinv = 1 / gs: dZ plane value -> dZ (bias-gradient column sums)
  extern __shared__ __attribute__((aligned(16))) char lds[];       // PW_NSTAGE stages (the only LDS object)
  const int ntiles = ntk * ntn;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int s = lb / ntiles, tile = lb % ntiles;
  const int k0d = (tile / ntn) * 128, n0 = (tile % ntn) * 128;
  const int64_t mbeg = (int64_t)s * Mc;
  int64_t mend = mbeg + Mc;
  if (mend > M) mend = M;
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), wm = wv >> 1, wn = wv & 1;
  const int nk = (int)((mend - mbeg + 15) / 16);
  const int tail = (int)((mend - mbeg) & 15);                       // rows of a ragged last step (0: none)
  f32x16 acc[2][2];
  zero_acc(acc);
  // bias gradient (kd tile 0 only): column sums of dZ = ones^T dZ on the matrix pipe -- an all-ones A fragment against the two
  // planes of the staged B fragments (every row of the result holds the column sum; row 0 of the lanes lh = 0 is stored)
  const bool want_b = k0d == 0 && partB != nullptr && wm == 0;
  f32x16 accb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[j][r] = 0.f;
  const u32x4 ones = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  // lane's byte offset inside a sub-tile: 16-lane group G = lane >> 4 -> columns 16 (G & 1) .., k rows 8 (G >> 1) ..;
  // lane p = lane & 15 points at (k + (p >> 2), 4 (p & 3))
  const int G = lane >> 4, pp = lane & 15;
  const unsigned lds0 = (unsigned)(uintptr_t)lds + (8 * (G >> 1) + (pp >> 2)) * 64 + (16 * (G & 1) + 4 * (pp & 3)) * 2;
#pragma unroll
  for (int i = 0; i < PW_NSTAGE - 1; ++i)
    if (i < nk) pw_issue_stage(Hh, h_stride, ldh, k0d, Zh, z_stride, N, n0, mbeg + 16 * i, mend, lds + i * PW_STAGE, wv, lane);
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;                                  // stages issued after stage kt
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 3 < nk)
      pw_issue_stage(Hh, h_stride, ldh, k0d, Zh, z_stride, N, n0, mbeg + 16 * (kt + 3), mend, lds + ((kt + 3) & 3) * PW_STAGE, wv, lane);
    const unsigned cur = lds0 + (kt & 3) * PW_STAGE;
    uint64_t ra[2][X_NP][2], rb[2][X_NP][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < X_NP; ++p) {
        pw_frag_issue(cur + ((0 * 2 + p) * 4 + (2 * wm + i)) * 1024, ra[i][p][0], ra[i][p][1]);
        pw_frag_issue(cur + ((1 * 2 + p) * 4 + (2 * wn + i)) * 1024, rb[i][p][0], rb[i][p][1]);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    u32x4 fa[2][X_NP], fb[2][X_NP];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < X_NP; ++p) {
        fa[i][p] = pw_join(ra[i][p][0], ra[i][p][1]);
        fb[i][p] = pw_join(rb[i][p][0], rb[i][p][1]);
      }
    if (tail && kt == nk - 1) {
      // ragged last step: rows beyond the slab were fetched from a clamped row -- zero their k-slots in the dZ operand
      const int lh = lane >> 5;                                      // lane half lh holds k = 8 lh + e
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < X_NP; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k_lo = 8 * lh + 2 * e, k_hi = k_lo + 1;
            uint32_t v = fb[i][p][e];
            if (k_lo >= tail) v &= 0xffff0000u;
            if (k_hi >= tail) v &= 0x0000ffffu;
            fb[i][p][e] = v;
          }
    }
#define RLX_PW_STEP(P, Q)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                      \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][P]),                      \
                                                         __builtin_bit_cast(f16x8, fb[j][Q]), acc[i][j], 0, 0, 0);
    RLX_PW_STEP(0, 1)
    RLX_PW_STEP(1, 0)
    RLX_PW_STEP(0, 0)
#undef RLX_PW_STEP
    if (want_b) {                                                  // (uniform)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < X_NP; ++p)
          accb[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ones), __builtin_bit_cast(f16x8, fb[j][p]), accb[j], 0, 0, 0);
    }
  }
  float* outW = partW + (int64_t)s * Kd * N;
  float* ob = outW + (int64_t)(k0d + wm * 64 + 4 * (lane >> 5)) * N + n0 + wn * 64 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(i * 32 + (r & 3) + 8 * (r >> 2)) * N + j * 32] = acc[i][j][r] * so;
  if (want_b && lane < 32) {
#pragma unroll
    for (int j = 0; j < 2; ++j) partB[(int64_t)s * N + n0 + wn * 64 + j * 32 + lane] = accb[j][0] * zinv;
  }
}

// ---------------------------------------------------------------------------------------
// fp32 [M, C] (row stride ld) <-> planes (test hooks, and the glue where a plane tensor meets an fp32 kernel)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_to_planes(const float* __restrict__ X, int ld, uint16_t* __restrict__ P, int64_t stride,
                                                   int64_t M, int C, float sc) {
  const int64_t n2 = M * (C >> 1);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / (C >> 1);
    const int c = (int)(e - r * (C >> 1)) * 2;
    uint32_t p0, p1;
    bx_split2(X[r * ld + c] * sc, X[r * ld + c + 1] * sc, p0, p1);
    *reinterpret_cast<uint32_t*>(P + r * C + c) = p0;
    *reinterpret_cast<uint32_t*>(P + stride + r * C + c) = p1;
  }
}

__global__ __launch_bounds__(256) void k_from_planes(const uint16_t* __restrict__ P, int64_t stride, float* __restrict__ X, int ld,
                                                     int64_t M, int C, float inv) {
  const int64_t n2 = M * (C >> 1);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / (C >> 1);
    const int c = (int)(e - r * (C >> 1)) * 2;
    const uint32_t p0 = *reinterpret_cast<const uint32_t*>(P + r * C + c), p1 = *reinterpret_cast<const uint32_t*>(P + stride + r * C + c);
    X[r * ld + c] = (bx_lo(p0) + bx_lo(p1)) * inv;
    X[r * ld + c + 1] = (bx_hi(p0) + bx_hi(p1)) * inv;
  }
}

int px_to_planes(const float* X, int ld, void* P, int64_t M, int C, float sc, hipStream_t st) {
  int grid = div_up(M * (C >> 1), 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_to_planes, dim3(grid), dim3(256), 0, st, X, ld, (uint16_t*)P, M * C, M, C, sc);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int px_from_planes(const void* P, float* X, int ld, int64_t M, int C, float inv, hipStream_t st) {
  int grid = div_up(M * (C >> 1), 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_from_planes, dim3(grid), dim3(256), 0, st, (const uint16_t*)P, M * C, X, ld, M, C, inv);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// ---------------------------------------------------------------------------------------
// launchers.  Plane tensors: base pointer = high plane, low plane at base + M * C elements.
// ---------------------------------------------------------------------------------------
bool px_shape_ok(int64_t M, int N, int K) { return M >= 4096 && K % P_BK == 0 && K >= P_BK && N % 128 == 0; }

#define RLX_PX_ATTR(KERNEL, BYTES)                                                                                       \
  {                                                                                                                      \
    static bool attr_set = false;                                                                                        \
    if (!attr_set) {                                                                                                     \
      RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      BYTES));                                                                           \
      attr_set = true;                                                                                                   \
    }                                                                                                                    \
  }
#define RLX_PX_LAUNCH(MODEV, ACTV, WMV)                                                                                  \
  {                                                                                                                      \
    constexpr int lds_bytes = P_NSTAGE * p_stage_bytes<8 / WMV>();                                                       \
    RLX_PX_ATTR((k_gemm_px<MODEV, ACTV, WMV>), lds_bytes)                                                                \
    RLX_PLAUNCH((k_gemm_px<MODEV, ACTV, WMV>), dim3(grid), dim3(P_THREADS), lds_bytes, st, (const uint16_t*)Ap, a_stride, \
                (const u32x4*)img, bias, (const uint16_t*)Hp, (uint16_t*)Cp, c_stride, M, N, K, ntn, so, sc_out);        \
  }
#define RLX_PX_LAUNCH_ACT(MODEV, WMV)                                  \
  switch (act) {                                                       \
    case RLX_ACT_TANH: RLX_PX_LAUNCH(MODEV, RLX_ACT_TANH, WMV) break;  \
    case RLX_ACT_ELU: RLX_PX_LAUNCH(MODEV, RLX_ACT_ELU, WMV) break;    \
    case RLX_ACT_RELU: RLX_PX_LAUNCH(MODEV, RLX_ACT_RELU, WMV) break;  \
    default: RLX_PX_LAUNCH(MODEV, RLX_ACT_NONE, WMV) break;            \
  }

// Cp[M, N] = planes(act(Ap[M, K] @ W + bias))
int px_launch_fwd(rlx_ctx* ctx, const void* Ap, const void* img, const float* bias, void* Cp, int64_t M, int N, int K, int act,
                  hipStream_t st) {
  RLX_REQUIRE(px_shape_ok(M, N, K), RLX_EUNSUP, "px_launch_fwd: shape");
  ProfScope prof(ctx, PK_GEMM_FWD, 2.0 * (double)M * N * K, st, gemm_bytes(M, N, K), M, N, K, 1);
  const int64_t a_stride = M * K, c_stride = M * N;
  const void* Hp = nullptr;
  const float so = X_AINV * X_WINV, sc_out = X_ASCALE;
  if (N % 256 == 0) {
    const int ntn = N / 256, grid = div_up(M, P_BM) * ntn;
    RLX_PX_LAUNCH_ACT(0, 1)
  } else {
    const int ntn = N / 128, grid = div_up(M, P_BM) * ntn;
    RLX_PX_LAUNCH_ACT(0, 2)
  }
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

// Dp[M, Kd] = planes((dZp[M, N] @ W[Kd, N]^T) * act'(Hp[M, Kd]))   (Dp may alias Hp)
int px_launch_dx(rlx_ctx* ctx, const void* dZp, const void* img, const void* Hp, void* Dp, int64_t M, int N_, int Kd, int act,
                 hipStream_t st) {
  RLX_REQUIRE(px_shape_ok(M, Kd, N_), RLX_EUNSUP, "px_launch_dx: shape");
  ProfScope prof(ctx, PK_GEMM_DX, 2.0 * (double)M * N_ * Kd, st, gemm_bytes(M, Kd, N_, 1), M, Kd, N_, 1);
  const float gs = ctx->bx_gscale;
  const void* Ap = dZp;
  void* Cp = Dp;
  const float* bias = nullptr;
  const int N = Kd, K = N_;
  const int64_t a_stride = M * K, c_stride = M * N;
  const float so = X_WINV / gs, sc_out = gs;
  if (N % 256 == 0) {
    const int ntn = N / 256, grid = div_up(M, P_BM) * ntn;
    RLX_PX_LAUNCH_ACT(1, 1)
  } else {
    const int ntn = N / 128, grid = div_up(M, P_BM) * ntn;
    RLX_PX_LAUNCH_ACT(1, 2)
  }
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

bool px_dw_ok(int64_t M, int Kd, int N) { return M >= 4096 && Kd % 128 == 0 && N % 128 == 0; }

int px_launch_dw(rlx_ctx* ctx, const void* Hp, const void* dZp, float* pW, float* pB, int64_t M, int Kd, int N, int64_t Mc, int S,
                 hipStream_t st) {
  RLX_REQUIRE(px_dw_ok(M, Kd, N), RLX_EUNSUP, "px_launch_dw: shape");
  static bool attr_set = false;
  if (!attr_set) {
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_dw_px), hipFuncAttributeMaxDynamicSharedMemorySize, PW_NSTAGE * PW_STAGE));
    attr_set = true;
  }
  const float gs = ctx->bx_gscale;
  const int ntk = Kd / 128, ntn = N / 128;
  ProfScope prof(ctx, PK_GEMM_DW, 2.0 * (double)M * Kd * N, st, gemm_bytes(Kd, N, M), Kd, N, (int)M, 1);
  RLX_PLAUNCH(k_gemm_dw_px, dim3(S * ntk * ntn), dim3(PW_THREADS), PW_NSTAGE * PW_STAGE, st, (const uint16_t*)Hp, M * Kd, Kd, (const uint16_t*)dZp,
              M * N, pW, pB, M, Kd, N, Mc, ntk, ntn, X_AINV / gs, 1.f / gs);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

}  // namespace rlx

using namespace rlx;

// Debug / micro-benchmark hook (tests/test_gpu_gemm_px.py, tools/gemm_bench.py): ONE plane-tensor GEMM on caller buffers of fp32
// (converted to planes in front of the kernel and back behind it; the conversions are outside the instrumented launch).
//   mode 0: C[M,N]  = act(A[M,K] @ B[K,N] + aux[N])
//   mode 1: C[M,K] <- (A[M,N] @ B[K,N]^T) * act'(C[M,K])        (A = dZ: the context's gradient scale, option "bx_gscale_log2")
//   mode 2: C[K,N]  = A[M,K]^T @ B[M,N], aux[N] = column sums of B
extern "C" int rlx_dbg_gemm_px_f32(rlx_ctx* ctx, int mode, const float* A, const float* B, float* C, float* aux, int64_t M, int N,
                                   int K, int act, void* stream) {
  RLX_REQUIRE(ctx && A && B && C && M > 0 && N > 0 && K > 0 && mode >= 0 && mode <= 2, RLX_EINVAL, "rlx_dbg_gemm_px_f32: bad args");
  hipStream_t st = (hipStream_t)stream;
  const float gs = ctx->bx_gscale;
  if (mode == 2) {
    RLX_REQUIRE(px_dw_ok(M, K, N), RLX_EUNSUP, "rlx_dbg_gemm_px_f32: weight-gradient shape (M >= 4096, K and N multiples of 128)");
    uint16_t* hp = (uint16_t*)scratch(ctx, SL_FWD_A, (size_t)M * K * 4);
    uint16_t* zp = (uint16_t*)scratch(ctx, SL_FWD_B, (size_t)M * N * 4);
    if (!hp || !zp) return RLX_ENOMEM;
    int rc = px_to_planes(A, K, hp, M, K, X_ASCALE, st);
    if (!rc) rc = px_to_planes(B, N, zp, M, N, gs, st);
    if (rc) return rc;
    const int ntk = K / 128, ntn = N / 128;
    int S = 1;
    const int64_t Mc = choose_mc(M, ntk * ntn, ctx->num_cus, &S);
    float* pW = (float*)scratch(ctx, SL_PARTIAL, ((size_t)S * K * N + (size_t)S * N) * sizeof(float));
    if (!pW) return RLX_ENOMEM;
    float* pB = pW + (size_t)S * K * N;
    rc = px_launch_dw(ctx, hp, zp, pW, pB, M, K, N, Mc, S, st);
    if (rc) return rc;
    ReduceTable tab;
    tab.n = 0;
    tab.seg[tab.n++] = ReduceSeg{pW, C, (int64_t)K * N, (int64_t)K * N, S, 0, 1.f, 0.f, 0};
    if (aux) tab.seg[tab.n++] = ReduceSeg{pB, aux, (int64_t)N, (int64_t)N, S, 0, 1.f, 0.f, 0};
    return launch_reduce_segments(tab, nullptr, nullptr, st);
  }
  // the weight image of B [K rows, N cols] as in rlx_dbg_gemm_f32 modes 3 / 4
  rlx_mlp_desc d{};
  MlpLayout L{};
  d.n_hidden = 2;
  L.n_hidden = 2;
  LayerOff& o = L.layer[1];
  o.W = 0;
  o.in = K;
  o.out = N;
  const bool was = ctx->gemm_bx;
  ctx->gemm_bx = true;
  int rc = bx_prepare_mlp(ctx, d, L, B, mode == 1, st);
  if (!rc) {
    if (mode == 0) {
      RLX_REQUIRE(aux && px_shape_ok(M, N, K), RLX_EUNSUP, "rlx_dbg_gemm_px_f32: forward needs bias, M >= 4096, K % 64 == 0, N % 128 == 0");
      uint16_t* ap = (uint16_t*)scratch(ctx, SL_FWD_A, (size_t)M * K * 4);
      uint16_t* cp = (uint16_t*)scratch(ctx, SL_FWD_B, (size_t)M * N * 4);
      const void* img = bx_lookup(ctx, B, 0, K, N);
      if (!ap || !cp || !img) rc = RLX_ENOMEM;
      if (!rc) rc = px_to_planes(A, K, ap, M, K, X_ASCALE, st);
      if (!rc) rc = px_launch_fwd(ctx, ap, img, aux, cp, M, N, K, act, st);
      if (!rc) rc = px_from_planes(cp, C, N, M, N, X_AINV, st);
    } else {
      RLX_REQUIRE(px_shape_ok(M, K, N), RLX_EUNSUP, "rlx_dbg_gemm_px_f32: input gradient needs M >= 4096, N % 64 == 0, K % 128 == 0");
      uint16_t* zp = (uint16_t*)scratch(ctx, SL_FWD_A, (size_t)M * N * 4);
      uint16_t* hp = (uint16_t*)scratch(ctx, SL_FWD_B, (size_t)M * K * 4);
      const void* img = bx_lookup(ctx, B, 1, N, K);
      if (!zp || !hp || !img) rc = RLX_ENOMEM;
      if (!rc) rc = px_to_planes(A, N, zp, M, N, gs, st);
      if (!rc) rc = px_to_planes(C, K, hp, M, K, X_ASCALE, st);
      if (!rc) rc = px_launch_dx(ctx, zp, img, hp, hp, M, N, K, act, st);
      if (!rc) rc = px_from_planes(hp, C, K, M, K, 1.f / gs, st);
    }
  }
  bx_release(ctx);
  ctx->gemm_bx = was;
  return rc;
}
