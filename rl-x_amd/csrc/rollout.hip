// rollout.hip -- ONE launch per acting step: policy forward + critic forward + Gaussian sample
// + log-prob (+ the synthetic env transition) for a [N, obs] observation batch.
// Replaces the per-step body of the reference's acting loop:
//   get_action_and_value        rl_x/algorithms/ppo/flax/ppo.py:110-119
//   single_rollout (full-jit)   rl_x/algorithms/ppo/flax_full_jit/ppo.py:130-153
//   env.step + Batch row store  rl_x/algorithms/ppo/flax/ppo.py:277-294
// The unfused path (rlx_actor_critic_fwd_sample_f32 + rlx_env_step_f32) costs 10 launches of
// 4096-row kernels per step; at N = 4096 those GEMMs fill a quarter of the chip.
//
// Workgroup = 32 rows x one network (grid = ceil(N/32) x 2 -> 256 workgroups at N = 4096).
// Every activation of the 32-row tile stays in LDS; weights stream from L2 through a
// register-prefetched LDS stage; hidden layers run on the exact-fp32 MFMA (32x32x2), the K=obs
// first layer and the tiny head on the VALU.  The policy workgroups finish with the sampling
// epilogue and (optionally) the env transition of their 32 envs; observations are double
// buffered by the caller (obs_in = Batch.states[t], obs_out = Batch.states[t+1]).
#include "env_device.h"
#include "gemm.h"
#include "mlp.h"
#include "gemm_bx.h"
#include "ppo_internal.h"

namespace rlx {

constexpr int RO_ROWS = 32;
constexpr int RO_THREADS = 256;
constexpr int RO_MAXH = 512;           // widest activation tile
constexpr int RO_MAXN = 256;           // widest MFMA layer output
constexpr int RO_A0 = RO_ROWS * (RO_MAXH + 1);   // activation buffer 0 (floats)
constexpr int RO_A1 = RO_ROWS * (RO_MAXN + 1);   // activation buffer 1
constexpr int RO_BS = G_BK * (RO_MAXN + 4);      // weight stage
constexpr int RO_MISC = 1024 + 64;
constexpr int RO_LDS_FLOATS = RO_A0 + RO_A1 + RO_BS + RO_MISC;
constexpr float RO_LOG_2PI = 1.8378770664093453f;

// The activation is a run-time field of the net descriptor.  Called per element as act_fwd(v, act), its switch sits between every two
// elements of the unrolled epilogues: hipcc keeps the branches, each element becomes its own dependent chain of ~30 instructions and
// -- one wave per SIMD, nothing to switch to -- the epilogues ran at ~425 clocks per element (phase stamps, round 6: 20 of the
// step's 41 us).  The switch is taken ONCE around the loop instead; inside, `actf` is a compile-time activation and the elements'
// chains interleave.  Same functions, same bits.
#define RO_ACT_SWITCH(ACTV, ...)                                                                                      \
  if ((ACTV) == RLX_ACT_ELU) { auto actf = [](float v_) { return act_fwd_t<RLX_ACT_ELU>(v_); }; __VA_ARGS__ }          \
  else if ((ACTV) == RLX_ACT_TANH) { auto actf = [](float v_) { return act_fwd_t<RLX_ACT_TANH>(v_); }; __VA_ARGS__ }   \
  else { const int act_rt_ = (ACTV); auto actf = [act_rt_](float v_) { return act_fwd(v_, act_rt_); }; __VA_ARGS__ }

struct RolloutNet {
  const float* params;
  int n_hidden;
  int hidden[3];
  int out_dim;
  int act, ln_first;
  int64_t W[3], b[3], g0, be0, headW, headb, logstd;
  int wide_in;           // 0: first layer on the VALU from obs [N, O <= 32]; else its input width K0 (multiple of 64, <= 256):
  const float* x_wide;   //    rows come from x_wide [N, K0 - xb_dim] and the first layer runs on the MFMA (+ LayerNorm iff ln_first)
  const float* x_b;      //    optional second source [N, xb_dim = 64]: the row tile is [x_wide | act(LayerNorm(x_b))]
  int xb_dim;            //    (the recurrent policy's cell output, normalised here instead of in two more launches)
  int64_t xb_g, xb_be;   //    LayerNorm scale / bias of x_b (offsets into params)
  const void* img[3];    // split weight images of the hidden layers l >= 1 (gemm_bx.h; NULL: exact-fp32 layer), laid out by
  int img_nt[3];         // rlx_ppo_rollout_begin; img_nt = 32-column tiles per 16-k block of the image
};

struct RolloutEnv {  // fused synthetic env (enabled iff enabled != 0)
  int enabled;
  uint32_t seed;
  int env_id_offset;
  uint32_t t;
  int horizon;
  float p_term, reward_noise;
  float* final_obs;   // [N,O]  Batch.next_states[t]
  float* reward;      // [N]
  float* terminated;  // [N]
  int32_t* ep_step;
  float* ep_ret;
  float* last_ret;
  float* last_len;
  float* episode_stats;
};

struct RolloutArgs {
  const RolloutNet* nets;  // DEVICE memory [2]: 0 = policy, 1 = critic (dynamic indexing of a by-value kernarg
                           // struct would copy it to scratch)
  const float* obs_in;   // [N,O]
  float* obs_out;        // [N,O] next observation (env mode) or unused
  float* action;         // [N,A]
  float* processed;      // [N,A] or null
  float* value;          // [N]
  float* logp;           // [N]
  int N, O, A;
  uint32_t k0, k1;       // noise subkey
  int scheme;
  int clip_and_rescale;
  const float* lo;
  const float* hi;
  int noise_row_offset, N_global;
  int deterministic;     // action = mean (no noise)
  int dbg_exit;          // tuning aid (option "ro_exit"): 1 / 2 / 3 = return after layer 0 / the hidden layers / the head
  unsigned long long* stamps;   // tuning aid (rlx_dbg_set_stamps): clock64() of thread 0 of workgroup 0 (policy) at the phase boundaries
  RolloutEnv env;
};

// out[32, N] = act(A_s[32, K] @ W[K, N] + bias); A_s in LDS (row stride a_st), out in LDS (row stride o_st).
// NT = 32-column MFMA tiles per wave (N = 128 * NT).
// The layer may be a column slice [col0, col0 + N) of a wider one: ldw = row stride of W, col0 = first column.
template <int NT>
__device__ __forceinline__ void fused_layer(const float* __restrict__ As, int a_st, int K, const float* __restrict__ W,
                                            const float* __restrict__ bias, float* __restrict__ Bs,
                                            float* __restrict__ Out, int o_st, int act, int t, int ldw = 128 * NT,
                                            int col0 = 0) {
  constexpr int N = 128 * NT;
  constexpr int SB = N + 4;
  constexpr int PER = N / 32;  // float4 per thread per k-tile
  const int lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // two K-tiles of weights in flight in registers (L2 -> VGPR latency at one wave per SIMD is longer
  // than one tile's 2048 MFMA cycles); K is a multiple of 64 so the loop is unrolled by two.
  typedef float v4f __attribute__((ext_vector_type(4)));  // native vector: promoted to VGPRs (HIP's float4 union was not)
  v4f rb0[PER], rb1[PER];
  const int nk = K / G_BK;
  const int f_row = t / (N / 4), f_col = (t % (N / 4)) * 4;   // thread's float4 slot inside a [.., N] row block
  constexpr int ROWS_PER_PASS = RO_THREADS / (N / 4);         // k-rows covered by one pass of 256 threads
  // (W comes out of the net descriptor in device memory: generic to the compiler -> flat loads and vmcnt(0) lgkmcnt(0) waits; the
  //  same address in the global address space gives global loads and counted waits)
  typedef const v4f __attribute__((address_space(1))) * gv4f_t;
  const float __attribute__((address_space(1))) * Wg = (const float __attribute__((address_space(1))) *)W;
#define RO_LDW(RB, KT)                                                                                         \
  _Pragma("unroll") for (int p = 0; p < PER; ++p)                                                              \
      RB[p] = *(gv4f_t)(Wg + (int64_t)((KT) * G_BK + f_row + ROWS_PER_PASS * p) * ldw + col0 + f_col);
#define RO_STW(RB)                                                                                             \
  _Pragma("unroll") for (int p = 0; p < PER; ++p)                                                              \
      *reinterpret_cast<v4f*>(Bs + (f_row + ROWS_PER_PASS * p) * SB + f_col) = RB[p];
#define RO_MMA(KT)                                                                                             \
  {                                                                                                            \
    const float* a0 = As + li * a_st + (KT) * G_BK + lh;                                                       \
    const float* b0 = Bs + lh * SB + w * 32 * NT + li;                                                         \
    _Pragma("unroll") for (int kk = 0; kk < G_BK; kk += 2) {                                                   \
      const float av = a0[kk];                                                                                 \
      _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                           \
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[kk * SB + 32 * j], acc[j], 0, 0, 0);            \
    }                                                                                                          \
  }
  RO_LDW(rb0, 0)
  RO_LDW(rb1, 1)
  for (int kt = 0; kt < nk; kt += 2) {
    __syncthreads();  // previous tile's fragment reads (and the producer of As) are complete
    RO_STW(rb0)
    __syncthreads();
    if (kt + 2 < nk) { RO_LDW(rb0, kt + 2) }
    RO_MMA(kt)
    __syncthreads();
    RO_STW(rb1)
    __syncthreads();
    if (kt + 3 < nk) { RO_LDW(rb1, kt + 3) }
    RO_MMA(kt + 1)
  }
#undef RO_LDW
#undef RO_STW
#undef RO_MMA
  RO_ACT_SWITCH(act,
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {
      const int col = col0 + w * 32 * NT + 32 * j + li;
      const float bv = ((const float __attribute__((address_space(1))) *)bias)[col];
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        Out[row * o_st + col] = actf(acc[j][r] + bv);
      }
    })
  __syncthreads();
}

// The same layer on the half-precision matrix pipe (gemm_bx.h): the activation tile stays fp32 in LDS and every wave splits ITS
// copy of the 32 x 16 A fragment into the two fp16 planes in registers (8 conflict-free ds_read_b32 + ~24 VALU per 16 k, issued
// under the previous step's MFMAs); the weight fragments come straight from the fragment-ordered image in L2.  No weight stage
// in LDS and NO barrier inside the K loop (the exact-fp32 form needs four per 64 k).
template <int NT>
__device__ __forceinline__ void fused_layer_bx(const float* __restrict__ As, int a_st, int K, const u32x4* __restrict__ Wf,
                                               int NTimg, const float* __restrict__ bias, float* __restrict__ Out, int o_st,
                                               int act, int t) {
  const int lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int nb = K >> 4;                       // 16-k blocks (K % 64 == 0)
  // The image pointer comes out of the net descriptor in device memory, so to the compiler it is a GENERIC pointer: it emitted
  // flat_load_dwordx4, and flat loads may return out of order with LDS traffic -- every wait became s_waitcnt vmcnt(0), which
  // drains the fragments just requested for the blocks ahead (the prefetch depth never existed).  Cast to the global address
  // space: global_load_dwordx4 and counted waits.
  typedef const u32x4 __attribute__((address_space(1))) * gfrag_t;
  gfrag_t wp = (gfrag_t)(Wf + (int64_t)(w * NT) * X_NP * 64 + lane);
  const int wstep = NTimg * X_NP * 64;
  const float* a0 = As + li * a_st + 8 * lh;
  u32x4 fb[4][NT][X_NP];
  float av[4][8];
#define RO_BX_LOAD(SLOT, G)                                                                       \
  {                                                                                               \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) _Pragma("unroll") for (int p = 0; p < X_NP; ++p) \
        fb[SLOT][j][p] = wp[(int64_t)(G) * wstep + (j * X_NP + p) * 64];                          \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) av[SLOT][e] = a0[(G) * 16 + e];                 \
  }
#define RO_BX_STEP(SLOT, PL, P, Q)                                                                \
  _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                  \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, PL[P]),           \
                                                      __builtin_bit_cast(f16x8, fb[SLOT][j][Q]), acc[j], 0, 0, 0);
  // the A fragment of a block is split into its planes UNDER the previous block's MFMAs (issued first, they run 32 clocks each on
  // the matrix pipe while the VALU does the conversions): with one wave per SIMD nothing else would fill those clocks
#define RO_BX_SPLIT(SLOT, PL)                                                                     \
  _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                 \
    uint32_t p0, p1;                                                                              \
    bx_split2(av[SLOT][2 * e] * X_ASCALE, av[SLOT][2 * e + 1] * X_ASCALE, p0, p1);                \
    PL[0][e] = p0; PL[1][e] = p1;                                                                 \
  }
#define RO_BX_MMA(SLOT, PL) { RO_BX_STEP(SLOT, PL, 0, 1) RO_BX_STEP(SLOT, PL, 1, 0) RO_BX_STEP(SLOT, PL, 0, 0) }
  __syncthreads();                              // the producer of As is complete
  // Four 16-k blocks of weight fragments (NT x 2 KiB each) in flight per wave: 8-16 KiB per wave, 32-64 KiB per CU.  A CU pulls
  // ~50 GB/s from L2 with 8 KiB in flight, ~83 with 16, ~116 with 32, ~140 with 64 (tools/probes/l1fill_probe.hip): the stream is
  // latency-bound, and with one wave per SIMD this kernel has the registers (two blocks ahead: layer 1 took 18.9 k clocks for 512 KB).
  u32x4 plA[X_NP], plB[X_NP];
  RO_BX_LOAD(0, 0)
  RO_BX_LOAD(1, 1)
  RO_BX_LOAD(2, 2)
  RO_BX_SPLIT(0, plA)
  // (the last four blocks are peeled: with the refills behind `if (g + 4 < nb)` the number of loads outstanding at the loop header
  //  depends on the path, and hipcc falls back to vmcnt(0) there)
  int g = 0;
  for (; g + 4 < nb; g += 4) {                  // nb = K / 16, K % 64 == 0
    RO_BX_LOAD(3, g + 3)
    RO_BX_MMA(0, plA)
    RO_BX_SPLIT(1, plB)
    RO_BX_LOAD(0, g + 4)
    RO_BX_MMA(1, plB)
    RO_BX_SPLIT(2, plA)
    RO_BX_LOAD(1, g + 5)
    RO_BX_MMA(2, plA)
    RO_BX_SPLIT(3, plB)
    RO_BX_LOAD(2, g + 6)
    RO_BX_MMA(3, plB)
    RO_BX_SPLIT(0, plA)
  }
  RO_BX_LOAD(3, g + 3)
  RO_BX_MMA(0, plA)
  RO_BX_SPLIT(1, plB)
  RO_BX_MMA(1, plB)
  RO_BX_SPLIT(2, plA)
  RO_BX_MMA(2, plA)
  RO_BX_SPLIT(3, plB)
  RO_BX_MMA(3, plB)
#undef RO_BX_LOAD
#undef RO_BX_MMA
#undef RO_BX_STEP
#undef RO_BX_SPLIT
  RO_ACT_SWITCH(act,
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {
      const int col = w * 32 * NT + 32 * j + li;
      const float bv = ((const float __attribute__((address_space(1))) *)bias)[col];
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        Out[row * o_st + col] = actf(fmaf(acc[j][r], X_WINV * X_AINV, bv));
      }
    })
  __syncthreads();
}

__global__ __launch_bounds__(RO_THREADS, 1) void k_rollout_step(RolloutArgs a) {  // 1 wave/SIMD: LDS (136 KB) admits one WG per CU anyway
  extern __shared__ __attribute__((aligned(16))) float smem[];
#define RO_STAMP(I) if (a.stamps && threadIdx.x == 0 && blockIdx.x == 0) a.stamps[I] = clock64();
  RO_STAMP(0)
  float* A0 = smem;
  float* A1 = A0 + RO_A0;
  float* Bs = A1 + RO_A1;
  float* misc = Bs + RO_BS;  // [32][O<=32] obs tile is not needed (readlane); outs [32][A], flags
  float* outs = misc;                       // [32][A]
  int* s_done = reinterpret_cast<int*>(misc + 32 * 32);  // [32]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int which = blockIdx.x & 1;
  const int64_t r0 = (int64_t)(blockIdx.x >> 1) * RO_ROWS;
  // wave-uniform scalars selected field by field (indexing the kernarg struct dynamically would
  // spill it to scratch and make `act` / `ln_first` look divergent -> exec-masked branches per element)
  const RolloutNet* __restrict__ np = a.nets + which;
#define RO_NETF(f) (np->f)
  const float* P = RO_NETF(params);
  // P comes out of the descriptor table in device memory: a GENERIC pointer to the compiler, i.e. flat loads and
  // s_waitcnt vmcnt(0) lgkmcnt(0) at every use.  Pg is the same address in the global address space (global loads, counted waits).
  typedef const float __attribute__((address_space(1))) * gf32_t;
  gf32_t Pg = (gf32_t)P;
  const int n_hidden = RO_NETF(n_hidden), act = RO_NETF(act), ln_first = RO_NETF(ln_first), out_dim = RO_NETF(out_dim);
  const int H0 = RO_NETF(hidden[0]), H1 = RO_NETF(hidden[1]), H2 = RO_NETF(hidden[2]);
  const int64_t oW0 = RO_NETF(W[0]), oW1 = RO_NETF(W[1]), oW2 = RO_NETF(W[2]);
  const int64_t ob0 = RO_NETF(b[0]), ob1 = RO_NETF(b[1]), ob2 = RO_NETF(b[2]);
  const int64_t og0 = RO_NETF(g0), obe0 = RO_NETF(be0), oHW = RO_NETF(headW), oHb = RO_NETF(headb);
  const int64_t oLS = a.nets[0].logstd;
  const int wide_in = RO_NETF(wide_in), xb_dim = RO_NETF(xb_dim);
  const float* x_wide = RO_NETF(x_wide);
  const float* x_b = RO_NETF(x_b);
  const int64_t oXg = RO_NETF(xb_g), oXbe = RO_NETF(xb_be);
  const u32x4* img1 = reinterpret_cast<const u32x4*>(RO_NETF(img[1]));
  const u32x4* img2 = reinterpret_cast<const u32x4*>(RO_NETF(img[2]));
  const int imgnt1 = RO_NETF(img_nt[1]), imgnt2 = RO_NETF(img_nt[2]);
#undef RO_NETF
  const int O = a.O;

  // The policy workgroup's random draws do not depend on the networks: the sampling noise eps[row, action dim] is a function of the step's
  // key and the element index, the env's pre-reset next observation of (seed, env id, env clock).  They are made at the top of the kernel
  // and kept in registers (same functions, same arguments: same bits).
  constexpr int EPI = (RO_ROWS * 32 + RO_THREADS - 1) / RO_THREADS;     // out_dim <= 32: at most 4 (row, action dim) elements per thread
  constexpr int OPI = (RO_ROWS * 16 + RO_THREADS - 1) / RO_THREADS;     // O <= 32: at most 16 observation pairs per row
  float eps_pre[EPI], ox[OPI], oy[OPI];
  const int pairs = (O + 1) / 2;
  if (which == 0) {
    const int A_ = a.A;
    const uint64_t total = (uint64_t)a.N_global * A_;
#pragma unroll
    for (int q = 0; q < EPI; ++q) {
      const int it = t + q * RO_THREADS;
      eps_pre[q] = 0.f;
      if (it < RO_ROWS * A_ && !a.deterministic) {
        const int e = it / A_, j = it - e * A_;
        const int64_t n = r0 + e;
        if (n < a.N) eps_pre[q] = normal_from_bits(random_bits_at(a.k0, a.k1, (uint64_t)(n + a.noise_row_offset) * A_ + j, total, a.scheme));
      }
    }
  }
  auto draw_early = [&]() {
    if (which != 0) return;
#pragma unroll
    for (int q = 0; q < OPI; ++q) {
      const int it = t + q * RO_THREADS;
      ox[q] = oy[q] = 0.f;
      if (a.env.enabled && it < RO_ROWS * pairs) {
        const int e = it / pairs, p = it - e * pairs;
        const int64_t n = r0 + e;
        if (n < a.N) obs_pair(a.env.seed, (uint32_t)(n + a.env.env_id_offset), a.env.t, (uint32_t)p, ox[q], oy[q]);
      }
    }
  };

  if (wide_in > 0) {
    // ---- wide first layer (recurrent policy torso: [obs latent | cell latent] -> 512, LayerNorm, activation):
    // the row tile comes from x_wide, the layer runs on the MFMA in 256-column slices, then one LayerNorm pass
    const int K0 = wide_in, xs = K0 + 1, st0 = H0 + 1, Ka = K0 - xb_dim;
    for (int i = t; i < RO_ROWS * (Ka >> 2); i += RO_THREADS) {
      const int r = i / (Ka >> 2), c4 = (i - r * (Ka >> 2)) * 4;
      const int64_t row = r0 + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < a.N) v = *reinterpret_cast<const float4*>(x_wide + row * Ka + c4);
      float* d = A1 + r * xs + c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    draw_early();
    if (xb_dim == 64) {   // [.. | act(LayerNorm(x_b))]: wave w normalises rows 8w..8w+7, one lane per column
      const float g = Pg[oXg + lane], be = Pg[oXbe + lane];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int64_t row = r0 + 8 * w + r;
        const float z = row < a.N ? x_b[row * 64 + lane] : 0.f;
        const float mean = wave_sum(z) * (1.0f / 64.0f);
        const float var = fmaxf(0.f, wave_sum(z * z) * (1.0f / 64.0f) - mean * mean);
        A1[(8 * w + r) * xs + Ka + lane] = act_fwd((z - mean) * rsqrtf(var + 1e-6f) * g + be, act);
      }
    }
    for (int c0 = 0; c0 < H0; c0 += 256)   // fused_layer opens with a barrier (publishes A1) and closes with one
      fused_layer<2>(A1, xs, K0, P + oW0, P + ob0, Bs, A0, st0, ln_first ? RLX_ACT_NONE : act, t, H0, c0);
    if (ln_first) {
      const int NJ = H0 >> 6;
      const float invH = 1.0f / (float)H0;
      float g8[8], b8[8];                  // this lane's LayerNorm scale / bias: once, not once per row and element
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        g8[j] = j < NJ ? Pg[og0 + lane + 64 * j] : 0.f;
        b8[j] = j < NJ ? Pg[obe0 + lane + 64 * j] : 0.f;
      }
      RO_ACT_SWITCH(act,
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {
          float z[8];
          float s_ = 0.f;
          float ss = 0.f;
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {
            z[j] = j < NJ ? A0[(8 * w + r) * st0 + lane + 64 * j] : 0.f;
            s_ += z[j];
            ss += z[j] * z[j];
          }
          s_ = wave_sum(s_);
          ss = wave_sum(ss);
          const float mean = s_ * invH;
          const float rstd = rsqrtf(fmaxf(0.f, ss * invH - mean * mean) + 1e-6f);
          _Pragma("unroll") for (int j = 0; j < 8; ++j)
            if (j < NJ) A0[(8 * w + r) * st0 + lane + 64 * j] = actf((z[j] - mean) * rstd * g8[j] + b8[j]);
        })
    }
  } else if (H0 == 512 && O <= 32) {
    // ---- layer 0 on the matrix pipe (hidden[0] = 512): z = X @ W0 as 9 MFMA steps per 32 x 32 tile, wave w owns
    // columns [128 w, 128 w + 128), LayerNorm row sums via half_sum4 + one barrier.  (The VALU form below spends 136
    // FMAs per lane and 16 full-wave reductions per wave on the same 32 x 512 tile.)
    constexpr int NT0 = 4, NW0 = RO_THREADS / 64;
    const int st = H0 + 1, OP = (O + 1) & ~1;
    const int li = lane & 31, lh = lane >> 5;
    const bool lb0 = (lane & 1) != 0, lb1 = (lane & 2) != 0;
    float* Xs = misc;                              // [32][33] = 1056 floats <= RO_MISC (outs / s_done are written after layer 0)
    float* red0 = Bs + RO_BS - 2 * NW0 * 32 - NW0 * 64;   // tail of the weight stage: [2][NW0][32] partials + per-wave totals
    const int colbase = w * 32 * NT0 + li;
    // W0 in registers (as in k_l1fwd_mfma): MFMA step s contracts obs indices 2s (lanes 0-31) and 2s + 1 (lanes 32-63), lane
    // (li, lh) holds W0[2s + lh][its four columns] -- loaded straight from global memory, coalesced over li, in flight while
    // the observation tile goes to LDS.  (The former LDS copy of W0 was a 36 KB fill + its share of the barrier per step.)
    constexpr int KS0 = 16;                        // O <= 32
    float w0r[KS0][NT0];
    // (H0 == 512 here: one per-lane base, every other term of the 64 addresses a compile-time offset)
    gf32_t w0g = Pg + oW0 + (int64_t)lh * 512 + colbase;
#pragma unroll
    for (int s_ = 0; s_ < KS0; ++s_)
#pragma unroll
      for (int j = 0; j < NT0; ++j) w0r[s_][j] = (2 * s_ + lh < O) ? w0g[(2 * s_) * 512 + 32 * j] : 0.f;
    // bias, LayerNorm scale / bias of this lane's columns: requested here, with W0, so that their L2 round trip is over when the
    // element-wise pass wants them (loaded where they were used, the pass opened with ~1 us of exposed latency)
    float b0v[NT0], gam[NT0], bet[NT0];
#pragma unroll
    for (int j = 0; j < NT0; ++j) {
      b0v[j] = Pg[ob0 + colbase + 32 * j];
      gam[j] = ln_first ? Pg[og0 + colbase + 32 * j] : 1.f;
      bet[j] = ln_first ? Pg[obe0 + colbase + 32 * j] : 0.f;
    }
    constexpr int XPI = RO_ROWS * 32 / RO_THREADS;
    float xo[XPI];
#pragma unroll
    for (int c = 0; c < XPI; ++c) {
      const int i = t + c * RO_THREADS, r = i >> 5, k = i & 31;
      const bool in = k < O && r0 + r < a.N;
      xo[c] = a.obs_in[in ? (r0 + r) * O + k : 0];      // (unconditional load; masked at the store, behind the draws: the first use is the wait)
    }
    draw_early();      // under the loads above
#pragma unroll
    for (int c = 0; c < XPI; ++c) {
      const int i = t + c * RO_THREADS, r = i >> 5, k = i & 31;
      Xs[r * 33 + k] = (k < O && r0 + r < a.N) ? xo[c] : 0.f;
    }
    __syncthreads();
    RO_STAMP(1)
    f32x16 z[NT0];
#pragma unroll
    for (int j = 0; j < NT0; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) z[j][r] = b0v[j];
    }
    {
      const float* x0 = Xs + li * 33 + lh;
#pragma unroll
      for (int s_ = 0; s_ < KS0; ++s_) {
        if (2 * s_ < OP) {                         // (uniform)
          const float av = x0[2 * s_];
#pragma unroll
          for (int j = 0; j < NT0; ++j) z[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w0r[s_][j], z[j], 0, 0, 0);
        }
      }
    }
    float* tot0 = red0 + 2 * NW0 * 32 + w * 64;
    RO_STAMP(2)
    if (ln_first) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float sv[4], ssv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s_ = 0.f, ss = 0.f;
#pragma unroll
          for (int j = 0; j < NT0; ++j) { s_ += z[j][4 * gq + e]; ss += z[j][4 * gq + e] * z[j][4 * gq + e]; }
          sv[e] = s_;
          ssv[e] = ss;
        }
        const float st_ = half_sum4(sv[0], sv[1], sv[2], sv[3], lb0, lb1);
        const float sst = half_sum4(ssv[0], ssv[1], ssv[2], ssv[3], lb0, lb1);
        if (li < 4) {
          red0[(0 * NW0 + w) * 32 + 8 * gq + 4 * lh + li] = st_;
          red0[(1 * NW0 + w) * 32 + 8 * gq + 4 * lh + li] = sst;
        }
      }
      __syncthreads();
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < NW0; ++q) v += red0[((lane >> 5) * NW0 + q) * 32 + (lane & 31)];
      tot0[lane] = v;
    }
    RO_STAMP(3)
    const float invH = 1.0f / (float)H0;
    if (ln_first) {      // (the run-time flag outside the loops, like the activation)
      RO_ACT_SWITCH(act,
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {
          _Pragma("unroll") for (int e = 0; e < 4; ++e) {
            const int r = 4 * gq + e, row = 8 * gq + 4 * lh + e;
            const float mean = tot0[row] * invH;
            const float rs = rsqrtf(fmaxf(0.f, tot0[32 + row] * invH - mean * mean) + 1e-6f);
            _Pragma("unroll") for (int j = 0; j < NT0; ++j)
              A0[row * st + colbase + 32 * j] = actf((z[j][r] - mean) * rs * gam[j] + bet[j]);
          }
        })
    } else {
      RO_ACT_SWITCH(act,
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {
          _Pragma("unroll") for (int e = 0; e < 4; ++e) {
            const int r = 4 * gq + e, row = 8 * gq + 4 * lh + e;
            _Pragma("unroll") for (int j = 0; j < NT0; ++j) A0[row * st + colbase + 32 * j] = actf(z[j][r]);
          }
        })
    }
    __syncthreads();   // the next layer's weight stage reuses A1 / Bs (W0s, red0)
    RO_STAMP(4)
  } else
  // ---- layer 0 on the VALU: wave w owns rows 8w..8w+7, lane l owns columns l + 64 j
  {
    const int H = H0, NJ = H >> 6, st = H + 1;
    float xv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int64_t row = r0 + 8 * w + r;
      xv[r] = (lane < O && row < a.N) ? a.obs_in[row * O + lane] : 0.f;
    }
    draw_early();
    float z[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) z[r][j] = 0.f;
    // stage W0 [O][H] into the (still unused) A1 + Bs region with coalesced 16-B loads
    float* W0s = A1;
    {
      const float4* src = reinterpret_cast<const float4*>(P + oW0);
      const int n4 = (O * H) >> 2;  // H % 64 == 0
      lds_stage<RO_THREADS, float4>(reinterpret_cast<float4*>(W0s), src, n4, n4, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    __syncthreads();
    for (int k = 0; k < O; ++k) {
      float xs[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) xs[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[r]), k));
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < NJ) {
          const float wv = W0s[k * H + lane + 64 * j];
#pragma unroll
          for (int r = 0; r < 8; ++r) z[r][j] = fmaf(xs[r], wv, z[r][j]);
        }
    }
    const float invH = 1.0f / (float)H;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < NJ) {
          z[r][j] += Pg[ob0 + lane + 64 * j];
          s += z[r][j];
          ss += z[r][j] * z[r][j];
        }
      float mean = 0.f, rstd = 1.f;
      if (ln_first) {
        s = wave_sum(s);
        ss = wave_sum(ss);
        mean = s * invH;
        rstd = rsqrtf(fmaxf(0.f, ss * invH - mean * mean) + 1e-6f);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < NJ) {
          const int c = lane + 64 * j;
          float y = z[r][j];
          if (ln_first) y = (y - mean) * rstd * Pg[og0 + c] + Pg[obe0 + c];
          A0[(8 * w + r) * st + c] = act_fwd(y, act);
        }
    }
  }
  if (a.dbg_exit == 1) return;
  // (fused_layer begins with a barrier, which also publishes A0)
  const float* hin = A0;
  int hst = H0 + 1, hk = H0;
  if (n_hidden > 1) {
    if (img1) {
      if (H1 == 256) fused_layer_bx<2>(A0, H0 + 1, H0, img1, imgnt1, P + ob1, A1, H1 + 1, act, t);
      else fused_layer_bx<1>(A0, H0 + 1, H0, img1, imgnt1, P + ob1, A1, H1 + 1, act, t);
    } else if (H1 == 256) fused_layer<2>(A0, H0 + 1, H0, P + oW1, P + ob1, Bs, A1, H1 + 1, act, t);
    else fused_layer<1>(A0, H0 + 1, H0, P + oW1, P + ob1, Bs, A1, H1 + 1, act, t);
    hin = A1; hst = H1 + 1; hk = H1;
  }
  RO_STAMP(5)
  if (n_hidden > 2) {
    if (img2) {
      if (H2 == 256) fused_layer_bx<2>(A1, H1 + 1, H1, img2, imgnt2, P + ob2, A0, H2 + 1, act, t);
      else fused_layer_bx<1>(A1, H1 + 1, H1, img2, imgnt2, P + ob2, A0, H2 + 1, act, t);
    } else if (H2 == 256) fused_layer<2>(A1, H1 + 1, H1, P + oW2, P + ob2, Bs, A0, H2 + 1, act, t);
    else fused_layer<1>(A1, H1 + 1, H1, P + oW2, P + ob2, Bs, A0, H2 + 1, act, t);
    hin = A0; hst = H2 + 1; hk = H2;
  }
  if (n_hidden == 1) __syncthreads();
  RO_STAMP(6)
  if (a.dbg_exit == 2) return;
  // ---- head on the VALU: outs[r][o] = h[r] . Wh[:, o] + bh[o]
  {
    const int r = t & 31;
    const int OD = out_dim;
    // head weights [hk, OD] staged in LDS (the weight stage is free now): the dot products then run on LDS reads only
    // (reading W from global inside the k loop cost ~8 us per step)
    float* Whs = Bs;
    for (int i = t; i < hk * OD; i += RO_THREADS) Whs[i] = Pg[oHW + i];      // (<= 1536 floats: six loads in flight per thread)
    __syncthreads();
    for (int o = t >> 5; o < OD; o += 8) {
      float acc0 = 0.f, acc1 = 0.f;
      const float* hr = hin + r * hst;
#pragma unroll 4
      for (int k = 0; k < hk; k += 2) {
        acc0 = fmaf(hr[k], Whs[k * OD + o], acc0);
        acc1 = fmaf(hr[k + 1], Whs[(k + 1) * OD + o], acc1);
      }
      outs[r * OD + o] = (acc0 + acc1) + Pg[oHb + o];
    }
  }
  __syncthreads();
  RO_STAMP(7)
  if (a.dbg_exit == 3) return;
  if (which == 1) {  // critic: value
    if (t < RO_ROWS && r0 + t < a.N) a.value[r0 + t] = outs[t];
    return;
  }
  // ---- policy epilogue: sample, log-prob, processed action, env transition
  // One thread per (row, action dim) for everything that is per element (threefry + erfinv + exp, the action cost term of the
  // env); the per-row sums are then added by one lane per row IN INDEX ORDER -- the same order as a serial loop over the action
  // dims, so log-probs and rewards keep their bits.  (32 lanes looping over the action dims cost 13 us of a 48 us step.)
  const int A = a.A;
  float* s_lp = Bs;                         // [32][A] log-prob terms   (the weight stage is free: the head has finished with it)
  float* s_cost = Bs + RO_ROWS * A;         // [32][A] action-cost terms of the env
  // the per-row lanes of the next phase ask for their episode counters now (one L2 round trip less behind the barrier)
  int es_in = 0;
  float er_in = 0.f;
  if (a.env.enabled && t < RO_ROWS && r0 + t < a.N) { es_in = a.env.ep_step[r0 + t]; er_in = a.env.ep_ret[r0 + t]; }
#pragma unroll
  for (int q = 0; q < EPI; ++q) {
    const int it = t + q * RO_THREADS;
    if (it >= RO_ROWS * A) continue;
    const int e = it / A, j = it - e * A;
    const int64_t n = r0 + e;
    if (n >= a.N) continue;
    const float eps = eps_pre[q];
    const float ls = Pg[oLS + j];
    const float sd = expf(ls);
    const float mu = outs[it];
    const float act = mu + sd * eps;
    const float zs = (act - mu) / sd;
    s_lp[it] = -0.5f * zs * zs - 0.5f * RO_LOG_2PI - ls;
    a.action[n * A + j] = act;
    float p = act;
    if (a.clip_and_rescale) {
      const float c = fminf(fmaxf(act, -1.f), 1.f);
      p = a.lo[j] + 0.5f * (c + 1.0f) * (a.hi[j] - a.lo[j]);
    }
    if (a.processed) a.processed[n * A + j] = p;
    if (a.env.enabled) {
      const float d = env_cost_diff(p, a.obs_in[n * O + j % O]);
      s_cost[it] = d * d;
    }
  }
  // The env's next observation does not wait for the reward: the pre-reset draw (Batch.next_states row; made at the top of the
  // kernel, draw_early) is stored here; only the envs that finish draw again, behind the per-row phase that decides it.
  if (a.env.enabled) {
#pragma unroll
    for (int q = 0; q < OPI; ++q) {
      const int it = t + q * RO_THREADS;
      if (it < RO_ROWS * pairs) {
        const int e = it / pairs, p = it - e * pairs;
        const int64_t n = r0 + e;
        if (n < a.N) {
          const int64_t o = n * O + 2 * p;
          a.env.final_obs[o] = ox[q];
          if (2 * p + 1 < O) a.env.final_obs[o + 1] = oy[q];
        }
      }
    }
  }
  __syncthreads();
  RO_STAMP(8)
  if (t < RO_ROWS) {
    const int64_t n = r0 + t;
    int done = 0;
    if (n < a.N) {
      float lp = 0.f;
      for (int j = 0; j < A; ++j) lp += s_lp[t * A + j];
      a.logp[n] = lp;
      if (a.env.enabled) {
        float acc = 0.f;
        for (int j = 0; j < A; ++j) acc += s_cost[t * A + j];
        const EnvLaneOut e = env_lane_finish_v(a.env.seed, (uint32_t)(n + a.env.env_id_offset), a.env.t, A, a.env.horizon,
                                               a.env.p_term, a.env.reward_noise, acc, es_in, er_in, a.env.ep_step, a.env.ep_ret,
                                               a.env.last_ret, a.env.last_len, (int)n);
        done = e.done;
        a.env.reward[n] = e.reward;
        a.env.terminated[n] = e.term ? 1.f : 0.f;
        if (a.env.episode_stats && e.done) {
          atomicAdd(&a.env.episode_stats[0], 1.f);
          atomicAdd(&a.env.episode_stats[1], e.fin_ret);
          atomicAdd(&a.env.episode_stats[2], e.fin_len);
        }
      }
    }
    s_done[t] = done;
  }
  if (!a.env.enabled) return;
  __syncthreads();
  RO_STAMP(9)
#pragma unroll
  for (int q = 0; q < OPI; ++q) {
    const int it = t + q * RO_THREADS;
    if (it >= RO_ROWS * pairs) continue;
    const int e = it / pairs, p = it - e * pairs;
    const int64_t n = r0 + e;
    if (n >= a.N) continue;
    float x = ox[q], y = oy[q];
    if (s_done[e]) obs_pair(a.env.seed, (uint32_t)(n + a.env.env_id_offset), a.env.t, ENV_STREAM_RESET + p, x, y);
    const int64_t o = n * O + 2 * p;
    a.obs_out[o] = x;
    if (2 * p + 1 < O) a.obs_out[o + 1] = y;
  }
  RO_STAMP(10)
#undef RO_STAMP
}

static bool fill_net(const rlx_mlp_desc& d, const float* params, RolloutNet* n) {
  const MlpLayout L = make_layout(d);
  n->params = params;
  n->n_hidden = d.n_hidden;
  n->out_dim = d.out_dim;
  n->act = d.act;
  n->ln_first = d.ln_first;
  if (d.n_hidden < 1 || d.n_hidden > 3 || d.in_dim > 32) return false;
  if (d.hidden[0] % 64 != 0 || d.hidden[0] > RO_MAXH) return false;
  for (int l = 0; l < d.n_hidden; ++l) {
    n->hidden[l] = d.hidden[l];
    n->W[l] = L.layer[l].W;
    n->b[l] = L.layer[l].b;
    if (l >= 1 && d.hidden[l] != 128 && d.hidden[l] != 256) return false;
    if (l >= 1 && d.hidden[l - 1] % (2 * G_BK) != 0) return false;
  }
  n->g0 = L.layer[0].g;
  n->be0 = L.layer[0].be;
  n->headW = L.head.W;
  n->headb = L.head.b;
  n->logstd = L.logstd;
  return d.out_dim * RO_ROWS <= 1024;
}

static int upload_nets_and_launch(rlx_ctx* ctx, const RolloutNet (&hn)[2], RolloutArgs& a, hipStream_t st) {
  // descriptor table lives in device memory; re-uploaded only when it changes
  RolloutNet* dn = (RolloutNet*)scratch(ctx, SL_RO_NETS, sizeof(hn));
  if (!dn) return RLX_ENOMEM;
  if (ctx->ro_nets_shadow.size() != sizeof(hn) || memcmp(ctx->ro_nets_shadow.data(), hn, sizeof(hn)) != 0) {
    RLX_HIP_TRY(hipStreamSynchronize(st));  // earlier launches may still read the old table
    RLX_HIP_TRY(hipMemcpy(dn, hn, sizeof(hn), hipMemcpyHostToDevice));
    ctx->ro_nets_shadow.assign(reinterpret_cast<const char*>(hn), reinterpret_cast<const char*>(hn) + sizeof(hn));
  }
  a.nets = dn;
  static AttrOnce attr_set;      
  const size_t lds = (size_t)RO_LDS_FLOATS * sizeof(float);
  if (!attr_set.done()) {
    RLX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout_step),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark();  
  }
  const int grid = div_up(a.N, RO_ROWS) * 2;
  hipLaunchKernelGGL(k_rollout_step, dim3(grid), dim3(RO_THREADS), lds, st, a);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

bool rollout_decoder_supported(const RolloutDecoder& p, const rlx_mlp_desc& cd) {
  RolloutNet c;
  return p.K0 % 64 == 0 && p.K0 >= 64 && p.K0 <= RO_MAXN && p.hidden[0] % 256 == 0 && p.hidden[0] <= RO_MAXH &&
         (p.hidden[1] == 128 || p.hidden[1] == 256) && (p.hidden[2] == 128 || p.hidden[2] == 256) &&
         p.out_dim * RO_ROWS <= 1024 && fill_net(cd, nullptr, &c) && cd.out_dim == 1;
}

// One launch for the tail of the recurrent acting step: policy torso on x [N, K0] (first layer on the MFMA, LayerNorm),
// head, sampling, log-prob, processed action -- and the feed-forward critic on obs [N, O] in the other half of the grid.
int launch_rollout_decoder(rlx_ctx* ctx, const RolloutDecoder& p, const rlx_mlp_desc& cd, const float* cparams,
                           const float* obs, int O, uint32_t k0, uint32_t k1, int scheme, float* action, float* processed,
                           float* value, float* logp, int N, int clip_and_rescale, const float* lo, const float* hi,
                           int noise_row_offset, int N_global, int deterministic, hipStream_t st) {
  RLX_REQUIRE(rollout_decoder_supported(p, cd), RLX_EUNSUP, "rollout decoder: shape outside the fused kernel's envelope");
  RolloutNet hn[2];
  memset(hn, 0, sizeof(hn));
  RolloutNet& n = hn[0];
  n.params = p.params; n.n_hidden = 3; n.out_dim = p.out_dim; n.act = p.act; n.ln_first = 1;
  for (int l = 0; l < 3; ++l) { n.hidden[l] = p.hidden[l]; n.W[l] = p.W[l]; n.b[l] = p.b[l]; }
  n.g0 = p.g0; n.be0 = p.be0; n.headW = p.headW; n.headb = p.headb; n.logstd = p.logstd;
  n.wide_in = p.K0; n.x_wide = p.x; n.x_b = p.xb; n.xb_dim = p.xb ? 64 : 0; n.xb_g = p.xb_g; n.xb_be = p.xb_be;
  fill_net(cd, cparams, &hn[1]);
  if (ctx->ro_img.valid && ctx->gemm_bx && ctx->ro_img.params[0] == p.params && ctx->ro_img.params[1] == cparams) {
    for (int q = 0; q < 2; ++q)
      for (int l = 1; l < 3; ++l) { hn[q].img[l] = ctx->ro_img.img[q][l]; hn[q].img_nt[l] = ctx->ro_img.nt[q][l]; }
  }
  RolloutArgs a{};
  a.obs_in = obs; a.obs_out = nullptr; a.action = action; a.processed = processed; a.value = value; a.logp = logp;
  a.N = N; a.O = O; a.A = p.out_dim;
  a.k0 = k0; a.k1 = k1; a.scheme = scheme;
  a.clip_and_rescale = clip_and_rescale; a.lo = lo; a.hi = hi;
  a.noise_row_offset = noise_row_offset; a.N_global = N_global; a.deterministic = deterministic;
  a.env.enabled = 0;
  return upload_nets_and_launch(ctx, hn, a, st);
}

}  // namespace rlx

using namespace rlx;

extern "C" {

int rlx_ppo_rollout_step_supported(const rlx_mlp_desc* pdesc, const rlx_mlp_desc* cdesc) {
  if (!pdesc || !cdesc) return 0;
  RolloutNet a, b;
  return (fill_net(*pdesc, nullptr, &a) && fill_net(*cdesc, nullptr, &b) && pdesc->in_dim == cdesc->in_dim &&
          pdesc->has_logstd && cdesc->out_dim == 1)
             ? 1 : 0;
}

// see include/rlx_hip.h
int rlx_ppo_rollout_begin(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams, const rlx_mlp_desc* cdesc,
                          const float* cparams, void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && cdesc && cparams, RLX_EINVAL, "rlx_ppo_rollout_begin: NULL pointer");
  ctx->ro_img = rlx_ctx::RoImages();
  if (!ctx->gemm_bx || !rlx_ppo_rollout_step_supported(pdesc, cdesc)) return RLX_OK;
  const rlx_mlp_desc* ds[2] = {pdesc, cdesc};
  const float* ps[2] = {pparams, cparams};
  // Engine window (gemm_bx.h): a weight at or above X_WLIMIT would become inf in its image and NaN in every action, value and
  // log-prob of the T steps -- data no later step can repair.  So the acting nets are checked BEFORE their images are laid out:
  // max |parameter| of both vectors (two small launches) and ONE blocking 8-byte read per rollout (the stream is idle here: the
  // previous iteration ended with the metrics' device->host copy).  Outside the window: no images -> every layer of the T steps
  // runs on the exact-fp32 MFMA engine, the counter "bx_window_fallbacks" tells the plugin, which logs it.
  {
    const uint32_t* slot[2];
    for (int n = 0; n < 2; ++n) {
      const int rc = x_max_update(ctx, ps[n], make_layout(*ds[n]).n_params, 2 + n, (hipStream_t)stream, &slot[n]);
      if (rc) return rc;
    }
    uint32_t host[2] = {0, 0};
    RLX_HIP_TRY(hipMemcpyAsync(&host[0], slot[0], sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    RLX_HIP_TRY(hipMemcpyAsync(&host[1], slot[1], sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    RLX_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    float m0, m1;
    memcpy(&m0, &host[0], 4);
    memcpy(&m1, &host[1], 4);
    if (!(m0 < X_WLIMIT && m1 < X_WLIMIT)) {     // (also true for inf)
      ++ctx->bx_window_fallbacks;
      return RLX_OK;
    }
  }
  BxJobs jobs;
  jobs.n = 0;
  int blocks = 0;
  int64_t entries = 0, off[2][3] = {};
  for (int n = 0; n < 2; ++n) {
    const MlpLayout L = make_layout(*ds[n]);
    for (int l = 1; l < ds[n]->n_hidden; ++l) {
      const LayerOff& o = L.layer[l];
      BxJob& j = jobs.job[jobs.n++];
      j.W = ps[n] + o.W; j.ldw = o.out; j.K = o.in; j.N = o.out; j.trans = 0;
      j.KB = 2 * div_up(o.in, X_BK); j.NT = 4 * div_up(o.out, G_BN);
      j.first_block = blocks;
      off[n][l] = entries;
      ctx->ro_img.nt[n][l] = j.NT;
      blocks += div_up(j.KB * j.NT * 64, 256);
      entries += (int64_t)j.KB * j.NT * X_NP * 64;
    }
  }
  if (jobs.n == 0) return RLX_OK;
  u32x4* arena = (u32x4*)scratch(ctx, SL_WFRAG_RO, (size_t)entries * sizeof(u32x4));
  if (!arena) return RLX_ENOMEM;
  int q = 0;
  for (int n = 0; n < 2; ++n)
    for (int l = 1; l < ds[n]->n_hidden; ++l) {
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

int rlx_ppo_rollout_end(rlx_ctx* ctx) {
  RLX_REQUIRE(ctx, RLX_EINVAL, "rlx_ppo_rollout_end: ctx is NULL");
  ctx->ro_img.valid = false;
  return RLX_OK;
}

int rlx_ppo_rollout_step_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams, const rlx_mlp_desc* cdesc,
                             const float* cparams, const float* obs_in, float* obs_out, uint32_t key_io[2], int scheme,
                             float* action, float* processed, float* value, float* logp, int N, int clip_and_rescale,
                             const float* act_low, const float* act_high, int noise_row_offset, int N_global,
                             int fuse_env, uint32_t env_seed, int env_id_offset, uint32_t env_t, int horizon,
                             float p_term, float reward_noise, float* final_obs, float* reward, float* terminated,
                             int32_t* ep_step, float* ep_ret, float* last_ret, float* last_len, float* episode_stats,
                             void* stream) {
  RLX_REQUIRE(ctx && pdesc && pparams && cdesc && cparams && obs_in && key_io && action && value && logp, RLX_EINVAL,
              "rlx_ppo_rollout_step_f32: NULL pointer");
  RLX_REQUIRE(N > 0 && N_global >= N, RLX_EINVAL, "rlx_ppo_rollout_step_f32: bad sizes");
  RLX_REQUIRE(rlx_ppo_rollout_step_supported(pdesc, cdesc), RLX_EUNSUP,
              "rlx_ppo_rollout_step_f32: network shape outside the fused kernel's envelope "
              "(use rlx_actor_critic_fwd_sample_f32 + rlx_env_step_f32)");
  RLX_REQUIRE(!clip_and_rescale || (act_low && act_high), RLX_EINVAL, "rlx_ppo_rollout_step_f32: clip needs bounds");
  if (fuse_env)
    RLX_REQUIRE(obs_out && final_obs && reward && terminated && ep_step && ep_ret && last_ret && last_len && horizon > 0,
                RLX_EINVAL, "rlx_ppo_rollout_step_f32: fuse_env needs the env state pointers");
  RolloutArgs a{};
  RolloutNet hn[2];
  memset(hn, 0, sizeof(hn));
  fill_net(*pdesc, pparams, &hn[0]);
  fill_net(*cdesc, cparams, &hn[1]);
  if (ctx->ro_img.valid && ctx->gemm_bx && ctx->ro_img.params[0] == pparams && ctx->ro_img.params[1] == cparams) {
    for (int n = 0; n < 2; ++n)
      for (int l = 1; l < 3; ++l) { hn[n].img[l] = ctx->ro_img.img[n][l]; hn[n].img_nt[l] = ctx->ro_img.nt[n][l]; }
  }
  a.obs_in = obs_in; a.obs_out = obs_out; a.action = action; a.processed = processed; a.value = value; a.logp = logp;
  a.N = N; a.O = pdesc->in_dim; a.A = pdesc->out_dim;
  uint32_t ks[4];
  split_host(key_io, ks, 2, scheme);  // key, subkey = split(key)
  key_io[0] = ks[0]; key_io[1] = ks[1];
  a.k0 = ks[2]; a.k1 = ks[3]; a.scheme = scheme;
  a.clip_and_rescale = clip_and_rescale; a.lo = act_low; a.hi = act_high;
  a.noise_row_offset = noise_row_offset; a.N_global = N_global;
  a.dbg_exit = ctx->ro_exit;
  a.stamps = (unsigned long long*)ctx->dbg_stamps;
  a.env.enabled = fuse_env ? 1 : 0;
  a.env.seed = env_seed; a.env.env_id_offset = env_id_offset; a.env.t = env_t; a.env.horizon = horizon;
  a.env.p_term = p_term; a.env.reward_noise = reward_noise; a.env.final_obs = final_obs; a.env.reward = reward;
  a.env.terminated = terminated; a.env.ep_step = ep_step; a.env.ep_ret = ep_ret; a.env.last_ret = last_ret;
  a.env.last_len = last_len; a.env.episode_stats = episode_stats;
  return upload_nets_and_launch(ctx, hn, a, (hipStream_t)stream);
}

// see include/rlx_hip.h: the T acting steps of one rollout, queued by one call
int rlx_ppo_rollout_f32(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams, const rlx_mlp_desc* cdesc,
                        const float* cparams, float* states, float* obs_last, uint32_t key_io[2], int scheme, float* actions,
                        float* values, float* logps, int T, int N, int clip_and_rescale, const float* act_low,
                        const float* act_high, int noise_row_offset, int N_global, uint32_t env_seed, int env_id_offset,
                        uint32_t env_t0, int horizon, float p_term, float reward_noise, float* final_obs, float* rewards,
                        float* terminated, int32_t* ep_step, float* ep_ret, float* last_ret, float* last_len,
                        float* episode_stats, void* stream) {
  RLX_REQUIRE(ctx && pdesc && states && obs_last && actions && values && logps && final_obs && rewards && terminated, RLX_EINVAL,
              "rlx_ppo_rollout_f32: NULL pointer");
  RLX_REQUIRE(T > 0 && N > 0, RLX_EINVAL, "rlx_ppo_rollout_f32: bad sizes");
  const int64_t O = pdesc->in_dim, A = pdesc->out_dim;
  for (int t = 0; t < T; ++t) {
    const int rc = rlx_ppo_rollout_step_f32(
        ctx, pdesc, pparams, cdesc, cparams, states + (int64_t)t * N * O, t + 1 < T ? states + (int64_t)(t + 1) * N * O : obs_last,
        key_io, scheme, actions + (int64_t)t * N * A, nullptr, values + (int64_t)t * N, logps + (int64_t)t * N, N, clip_and_rescale,
        act_low, act_high, noise_row_offset, N_global, 1, env_seed, env_id_offset, env_t0 + (uint32_t)t, horizon, p_term,
        reward_noise, final_obs + (int64_t)t * N * O, rewards + (int64_t)t * N, terminated + (int64_t)t * N, ep_step, ep_ret,
        last_ret, last_len, episode_stats, stream);
    if (rc) return rc;
  }
  return RLX_OK;
}

}  // extern "C"
