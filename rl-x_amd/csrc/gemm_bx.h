// gemm_bx.h -- fp32 GEMM on the half-precision matrix pipe by operand splitting (gfx950).
//
// v_mfma_f32_32x32x16_f16 issues at 32 cycles / instruction / SIMD for 16 k-steps: 16x the rate of v_mfma_f32_32x32x2_f32.
// An fp32 value a (times a power-of-two scale that keeps it inside fp16's exponent range) splits into two fp16 planes
//      a = a0 + a1 + e,   a0 = rne16(a),  a1 = rne16(a - a0)  (the subtraction is exact in fp32),  |e| <= 2^-22 |a|
// -- 11 + 11 significant bits; below fp16's normal range the planes keep an ABSOLUTE resolution of 2^-25 (scaled units).  Then
//      a * b = a0 b0 + a0 b1 + a1 b0  (+ a1 b1 <= 2^-22 |a b|, dropped),
// every plane product exact in the pipe's fp32 accumulation: THREE fp16 MFMAs per 16 k replace eight fp32 MFMAs at half the
// cycles each (96 instead of 512 matrix-pipe cycles), with operand errors of the size of ONE fp32 rounding of the product
// (rounds 2-3 used three bf16 planes and six products: exact operands, twice the matrix work and 1.5x the operand bytes).
// Accuracy is measured, not assumed: tests/test_gpu_gemm.py holds the split GEMM to the same fp64-referenced error budget as
// the exact-fp32 engine, tests/test_gpu_bench_shapes.py the whole minibatch pass at the bench shape to the 1e-5 bar.
//
// Scales (powers of two, exact) put each operand class inside fp16's window -- full 22-bit precision for scaled magnitudes in
// [2^-3, 65504), an absolute floor of 2^-25 below:
//   activations  x X_ASCALE = 16   |h| < 4094, full precision from 0.0078 up, floor 1.9e-9
//   weights      x X_WSCALE = 64   |w| < 1023, full precision from 0.002 up, floor 4.7e-10
//   gradients    x gscale          dZ ~ g / minibatch: rlx_ctx::bx_gscale = bx_grad_scale(minibatch) = 8 * 2^ceil(log2 mb):
//                                  per-sample |g| < 8190, full precision from 0.016 up
// The epilogues multiply by the inverse.  A scaled value beyond fp16's range becomes inf and poisons the result visibly
// (non-finite losses / gradient norms); the plugins check the metrics once per iteration and name this engine in the error.
//
// Operand layout (per lane l of a wave, li = l & 31, lh = l >> 5), 8 fp16 = 4 VGPRs per operand:
//   A: A[i = li][k = 8 * lh + (0..7)]        B: B[k = 8 * lh + (0..7)][j = li]
//   C/D: identical to the f32 form (gemm.h).
// Only the agreement of the two k maps matters for the product, so both sides simply use "element e of the lane's
// vector is k = 8 * lh + e".
//
//   Activation operands (contraction index contiguous per lane): fp32 tile -> split in registers -> two fp16 planes in
//     LDS, [128 rows][32 k] = 64 B per row, the four 16-byte k-slots of a row XOR-swizzled with bits 2-3 of the row so
//     that the 16 lanes of every ds_read_b128 service group fall on 16 distinct slots of the 256-byte bank row (and the
//     8-/16-byte staging stores spread over the banks as well); fragments by one ds_read_b128 per (row tile, plane, 16 k).
//   Weight operands (small, shared by every row tile): split ONCE per weight update into fragment order in global
//     memory (k_bx_wfrag): [K/16][N/32][2 planes][64 lanes] x 16 B, so a wave's fragment is one fully coalesced
//     1-KiB global_load_dwordx4 served by L2 / L1 -- no LDS traffic and no barrier for that operand at all.
#pragma once
#include "gemm.h"

namespace rlx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int X_NP = 2;                  // planes per operand
constexpr float X_ASCALE = 16.f;         // activation operands are split times 16
constexpr float X_AINV = 1.f / 16.f;
constexpr float X_WSCALE = 64.f;         // weight images hold W * 64 (|W| < 1023): typical |w| ~ 0.05 gets a normal low plane
constexpr float X_WINV = 1.f / 64.f;
constexpr float X_WLIMIT = 1023.f;       // |w| at or above this leaves the fp16 window of the weight images (65504 / 64 = 1023.5)
constexpr int X_BK = 32;                 // k per staged tile (two 16-k MFMA steps)
constexpr int X_ROWB = 2 * X_BK;         // bytes per row of one plane
constexpr int X_PLANE = G_BM * X_ROWB;   // 8 KiB
constexpr int X_OPER = X_NP * X_PLANE;   // 16 KiB: one staged operand tile (two planes)

// host: the power-of-two scale of the gradient operands of a pass whose loss is a mean over `rows` samples: 8 * 2^ceil(log2 rows)
// (dZ ~ g / rows with per-sample g of order 1e-3 .. 1e2 lands in fp16's normal range; |g| < 8190 is representable)
inline float bx_grad_scale(int64_t rows) {
  float s = 8.f;
  for (int64_t r = 1; r < rows && s < 1.0e9f; r <<= 1) s *= 2.f;
  return s;
}

// two fp32 -> packed fp16 (v_cvt_pk_f16_f32: round to nearest even; low half = x)
__device__ __forceinline__ uint32_t bx_pack(float x, float y) {
  f32x2 v = {x, y};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float bx_lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
__device__ __forceinline__ float bx_hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }

// (x, y) -> two packed fp16 planes with x = x0 + x1 (to 2^-22 |x|, or 2^-25 absolute), same for y; hipcc folds a power-of-two
// scale applied by the caller into v_fma_mix_f32: six VALU instructions per pair
__device__ __forceinline__ void bx_split2(float x, float y, uint32_t& p0, uint32_t& p1) {
  p0 = bx_pack(x, y);
  p1 = bx_pack(x - bx_lo(p0), y - bx_hi(p0));
}

// byte offset of k-slot `ks` (8 k = 16 B) of row `r` inside one plane
__device__ __forceinline__ int bx_off(int r, int ks) { return r * X_ROWB + ((ks ^ ((r >> 2) & 3)) << 4); }

// Layout of the weight-gradient kernel's operand tiles.  Its staging pass stores one 16-byte k-slot per lane with the lanes of
// a store group 4 rows apart (a thread owns 4 adjacent columns = rows of the transposed tile), which bx_off serves with a
// 2-way bank conflict (PMC: a third of the kernel's LDS cycles).  Found by exhaustive search over bit swaps of the row and
// linear swizzles: rows stored at p = r with bits 0 and 2 exchanged, slot XOR {bit 3, bit 2 ^ bit 4} of p -- conflict free for
// the 8-lane groups of the ds_write_b128 staging stores AND the 16-lane service groups of the ds_read_b128 fragment reads.
__device__ __forceinline__ int bx_off_dw(int r, int ks) {
  const int p = (r & ~5) | ((r & 1) << 2) | ((r >> 2) & 1);
  const int f = ((p >> 3) & 1) | ((((p >> 2) ^ (p >> 4)) & 1) << 1);
  return p * X_ROWB + ((ks ^ f) << 4);
}

// four consecutive k (kc % 4 == 0) of row r -> the three planes of the staged operand at `sb`
__device__ __forceinline__ void bx_stage_k4(char* __restrict__ sb, int r, int kc, float4 v, float sc) {
  uint32_t a0, a1, b0, b1;
  bx_split2(v.x * sc, v.y * sc, a0, a1);
  bx_split2(v.z * sc, v.w * sc, b0, b1);
  char* d = sb + bx_off(r, kc >> 3) + ((kc & 4) << 1);
  *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
  *reinterpret_cast<u32x2*>(d + X_PLANE) = u32x2{a1, b1};
}

// eight consecutive k (k-slot ks) of row r, gathered by the caller from eight memory rows (transposing stage of the
// weight-gradient kernel: the contraction index is the slow index in memory)
template <bool DW = false>
__device__ __forceinline__ void bx_stage_k8(char* __restrict__ sb, int r, int ks, const float (&v)[8], float sc) {
  u32x4 p0, p1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t a, b;
    bx_split2(v[2 * e] * sc, v[2 * e + 1] * sc, a, b);
    p0[e] = a;
    p1[e] = b;
  }
  char* d = sb + (DW ? bx_off_dw(r, ks) : bx_off(r, ks));
  *reinterpret_cast<u32x4*>(d) = p0;
  *reinterpret_cast<u32x4*>(d + X_PLANE) = p1;
}

// fragments of 16-k step s (0 / 1) for MI 32-row tiles starting at row `r0` of a staged operand: f[tile][plane]
template <int MI, bool DW = false>
__device__ __forceinline__ void bx_load_frag(const char* __restrict__ sb, int r0, int lane, int s, u32x4 (&f)[MI][X_NP]) {
  const int r = r0 + (lane & 31), ks = 2 * s + (lane >> 5);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const char* base = sb + (DW ? bx_off_dw(r + 32 * i, ks) : bx_off(r + 32 * i, ks));   // rows r and r + 32: same swizzle
#pragma unroll
    for (int p = 0; p < X_NP; ++p) f[i][p] = *reinterpret_cast<const u32x4*>(base + p * X_PLANE);
  }
}

// B fragments of global 16-k block g for the wave's two 32-column tiles (first one = column tile nt0): fb[j][plane]
__device__ __forceinline__ void bx_load_b(const u32x4* __restrict__ Wf, int g, int NT, int nt0, int lane, u32x4 (&fb)[2][X_NP]) {
  const u32x4* base = Wf + ((int64_t)(g * NT + nt0) * X_NP) * 64 + lane;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int p = 0; p < X_NP; ++p) fb[j][p] = base[(j * X_NP + p) * 64];
}

template <int MI>
__device__ __forceinline__ void bx_mma(const u32x4 (&fa)[MI][X_NP], const u32x4 (&fb)[2][X_NP], f32x16 (&acc)[MI][2]) {
  // smallest products first; the accumulators alternate so no MFMA waits on its predecessor
#define RLX_BX_STEP(P, Q)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                   \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][P]),                    \
                                                         __builtin_bit_cast(f16x8, fb[j][Q]), acc[i][j], 0, 0, 0);
  RLX_BX_STEP(0, 1)
  RLX_BX_STEP(1, 0)
  RLX_BX_STEP(0, 0)
#undef RLX_BX_STEP
}

// ---------------------------------------------------------------------------------------
// Main loop of the row-major-activation x weight-image kernels, software pipelined by hand.
//   iteration kt:  (1) fetch the rows of K-tile kt + 2 into the register set that tile kt + 1 does not use,
//                  (2) first 16-k step of tile kt on the matrix pipe -- its 12 MFMAs carry, interleaved by
//                      sched_group_barrier, the split + LDS stores of tile kt + 1 (about five VALU instructions per MFMA
//                      slot: hipcc on its own emits the whole staging pass in front of the MFMA block, where only the
//                      other wave of the SIMD can hide it),
//                  (3) second 16-k step, (4) one barrier.
// Branch free inside (the pipeline stages past the last tile re-fetch / re-stage the last tile into the stage nobody
// reads), so each iteration is one scheduling region.  load(kt, regs) must tolerate any kt (it clamps).
// MI = 32-row tiles per wave: 2 for the 128-row block tile, 1 for a 64-row block tile (shapes with a single column tile,
// where 128-row tiles would leave half of the CUs without a second workgroup).
// ---------------------------------------------------------------------------------------
template <int MI, class LoadFn>
__device__ __forceinline__ void bx_kloop(char* __restrict__ lds, const u32x4* __restrict__ Wf, int nk, int NT, int nt0, int wm,
                                         int lane, int a_r, int a_c, LoadFn load, f32x16 (&acc)[MI][2], float sa) {
  constexpr int NP = 2 * MI;          // staging passes of 32 rows: the block tile has 64 * MI rows (wave tile 32 * MI x 64)
  float4 ra0[NP], ra1[NP];
  u32x4 fb0[2][X_NP], fb1[2][X_NP], fa0[MI][X_NP], fa1[MI][X_NP];
  load(0, ra0);
  bx_load_b(Wf, 0, NT, nt0, lane, fb0);
#pragma unroll
  for (int p = 0; p < NP; ++p) bx_stage_k4(lds, a_r + 32 * p, a_c, ra0[p], sa);
  load(1, ra1);
  __syncthreads();
#define RLX_BX_ITER(KT, RCUR, RNXT)                                                                       \
  {                                                                                                        \
    const char* cur = lds + ((KT) & 1) * X_OPER;                                                           \
    char* nxt = lds + (((KT) + 1) & 1) * X_OPER;                                                           \
    bx_load_frag<MI>(cur, wm * 32 * MI, lane, 0, fa0);                                                     \
    bx_load_b(Wf, 2 * (KT) + 1, NT, nt0, lane, fb1);                                                       \
    load((KT) + 2, RCUR);                                                                                  \
    _Pragma("unroll") for (int p = 0; p < NP; ++p) bx_stage_k4(nxt, a_r + 32 * p, a_c, RNXT[p], sa);       \
    bx_mma<MI>(fa0, fb0, acc);                                                                             \
    _Pragma("unroll") for (int q = 0; q < 6 * MI; ++q) {                                                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                                   \
      if (q % 3 != 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                   \
    }                                                                                                      \
    bx_load_frag<MI>(cur, wm * 32 * MI, lane, 1, fa1);                                                     \
    bx_load_b(Wf, 2 * (KT) + 2 < 2 * nk ? 2 * (KT) + 2 : 2 * nk - 1, NT, nt0, lane, fb0);                  \
    bx_mma<MI>(fa1, fb1, acc);                                                                             \
    __syncthreads();                                                                                       \
  }
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    RLX_BX_ITER(kt, ra0, ra1)
    RLX_BX_ITER(kt + 1, ra1, ra0)
  }
  if (kt < nk) RLX_BX_ITER(kt, ra0, ra1)
#undef RLX_BX_ITER
}

// One weight operand to lay out: B(k, j) = trans ? W[j * ldw + k] : W[k * ldw + j], zero beyond [K, N]; the image covers
// KB 16-k blocks x NT 32-column tiles (padded to what the consuming kernel's tiles read).
struct BxJob {
  const float* W;
  u32x4* out;
  int ldw, K, N, trans, KB, NT;
  int first_block;   // first 256-thread block of this job inside the launch
};
constexpr int BX_MAX_JOBS = 32;   // SAC with three-hidden-layer nets lays out 21 images in one launch (policy 5, critics 2 x 5, targets 2 x 3)
struct BxJobs {
  int n;
  BxJob job[BX_MAX_JOBS];
};

// gemm_bx.hip: launches the image builder for a prepared job table
void bx_launch_wfrag(const BxJobs& jobs, int blocks, hipStream_t st);

}  // namespace rlx
