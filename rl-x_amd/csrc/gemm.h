// gemm.h -- exact-fp32 MFMA tile engine for the MLP layers (gfx950).
//
// v_mfma_f32_32x32x2_f32 (`__builtin_amdgcn_mfma_f32_32x32x2f32`): exact f32 (bitwise an
// fmaf chain), 64 cycles / instruction / SIMD = the 157.3 TFLOP/s f32 matrix peak.
// The 1e-5 parity bar against the reference's fp32 XLA:CPU path rules out bf16 MFMA.
//   A operand: lane l holds A[i = l&31][k = l>>5]
//   B operand: lane l holds B[k = l>>5][j = l&31]
//   C/D (16 regs): col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
//
// Block tile 128 x 128, K-step 32, 256 threads = 4 waves (2 x 2), each wave a 64 x 64
// sub-tile = 2 x 2 MFMA tiles (64 accumulator VGPRs).  Operands are staged through LDS
// (conflict-free ds_read_b32: lanes 0-31 / 32-63 are separate bank groups); the next
// K-tile's global loads are issued into registers before the MFMA block so HBM/L2 latency
// hides under 4096 MFMA cycles per tile.  One 64-cycle MFMA needs half an A and half a B
// dword per lane, so LDS bandwidth is ~12 % utilised: the matrix pipe is the bound.
#pragma once
#include "common.h"

namespace rlx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int G_BM = 128, G_BN = 128, G_BK = 32;
constexpr int G_THREADS = 256;
constexpr int G_SA_ROW = G_BK + 1;   // As[m][k] (k contiguous), odd stride -> conflict-free column reads
constexpr int G_SB = G_BN + 4;       // Bs[k][n] (n contiguous), 16-B aligned rows
constexpr int G_SBT = G_BN + 2;      // Bs[n][kd] written TRANSPOSED by 4-B stores (input-gradient kernel): stride 130 makes
                                     // the 64 lanes of one store hit 64 different banks (stride 132 gave 2-way conflicts)
constexpr int G_SA_COL = G_BM + 4;   // As[k][m] (m contiguous) for the TN kernel
constexpr int G_LDS_A = (G_BM * G_SA_ROW > G_BK * G_SA_COL) ? G_BM * G_SA_ROW : G_BK * G_SA_COL;
constexpr int G_LDS_B = G_BK * G_SB;

// MFMA over one staged K-tile.  A element (i,k) at As[i*A_I + k*A_K]; B element (k,j) at Bs[k*B_S + j].
// The LDS->register fragment reads are software-pipelined by hand: the fragments of k-group g+1
// (4 k-pairs = 16 MFMAs = 1024 matrix-pipe cycles) are issued BEFORE the MFMAs of group g, and a
// sched_barrier pins that order (hipcc otherwise sinks every ds_read next to its consumer and the
// wave stalls on LDS latency twice per 8 MFMAs -- visible at 1 wave/SIMD).
constexpr int G_KG = 4;  // k-pairs per fragment group

template <int A_I, int A_K, int B_S>
__device__ __forceinline__ void load_frags(const float* __restrict__ a0, const float* __restrict__ b0, int kk0,
                                           float (&fa)[2][G_KG], float (&fb)[2][G_KG]) {
#pragma unroll
  for (int q = 0; q < G_KG; ++q) {
    const int kk = kk0 + 2 * q;
    fa[0][q] = a0[kk * A_K];
    fa[1][q] = a0[32 * A_I + kk * A_K];
    fb[0][q] = b0[kk * B_S];
    fb[1][q] = b0[kk * B_S + 32];
  }
}

__device__ __forceinline__ void mma_frags(const float (&fa)[2][G_KG], const float (&fb)[2][G_KG], f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int q = 0; q < G_KG; ++q) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][q], fb[0][q], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][q], fb[1][q], acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][q], fb[0][q], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][q], fb[1][q], acc[1][1], 0, 0, 0);
  }
}

template <int A_I, int A_K, int B_S = G_SB>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x16 (&acc)[2][2], int wm, int wn, int lane) {
  const int li = lane & 31, lh = lane >> 5;
  const float* a0 = As + (wm * 64 + li) * A_I + lh * A_K;
  const float* b0 = Bs + lh * B_S + wn * 64 + li;
  float fa0[2][G_KG], fb0[2][G_KG], fa1[2][G_KG], fb1[2][G_KG];
  load_frags<A_I, A_K, B_S>(a0, b0, 0, fa0, fb0);
#pragma unroll
  for (int g = 0; g < G_BK / (2 * G_KG); g += 2) {
    load_frags<A_I, A_K, B_S>(a0, b0, (g + 1) * 2 * G_KG, fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    mma_frags(fa0, fb0, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 2 < G_BK / (2 * G_KG)) load_frags<A_I, A_K, B_S>(a0, b0, (g + 2) * 2 * G_KG, fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    mma_frags(fa1, fb1, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// guarded 16-byte global load of row-major src[row][col..col+3]; zero outside [rows, cols).
// cols must be a multiple of 4 for the vector path (checked by the host).
__device__ __forceinline__ float4 ld4(const float* __restrict__ src, int64_t row, int col, int64_t rows, int cols,
                                      int64_t ld) {
  if (row < rows && col < cols) return *reinterpret_cast<const float4*>(src + row * ld + col);
  return make_float4(0.f, 0.f, 0.f, 0.f);
}

// XCD-aware block id remap (8 XCDs, private L2s): consecutive logical tiles -> same XCD so the n-tiles of one row panel
// share that panel in one L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// accumulator element -> (row, col) inside the 128x128 block tile
__device__ __forceinline__ int acc_row(int wm, int i, int r, int lane) {
  return wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int wn, int j, int lane) { return wn * 64 + j * 32 + (lane & 31); }

}  // namespace rlx
