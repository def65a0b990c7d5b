// prng.hip -- jax.random restated for gfx950: threefry bits, normal, and the
// bit-exact `permutation(key, x, axis=1, independent=True)` used for PPO
// minibatch indices (reference call site rl_x/algorithms/ppo/flax/ppo.py:191-194), on a hand-written
// segmented stable radix sort (no vendor sort library on the path).
#include "common.h"

namespace rlx {

__global__ void k_random_bits(uint32_t k0, uint32_t k1, uint32_t* __restrict__ out, uint64_t n, int scheme) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = random_bits_at(k0, k1, i, n, scheme);
}

__global__ void k_normal(uint32_t k0, uint32_t k1, float* __restrict__ out, uint64_t n, int scheme) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = normal_from_bits(random_bits_at(k0, k1, i, n, scheme));
}

// vals[e*B + i] = i  (tile(arange(B), (E,1)))
__global__ void k_tile_iota(int32_t* __restrict__ out, int64_t B, int64_t total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) out[i] = (int32_t)(i % B);
}

// ---------------------------------------------------------------------------------------
// K4: segmented, stable LSD radix sort of (uint32 key, int32 value) pairs -- E independent rows of B pairs, ascending by
// key, equal keys keep their order: exactly what jax's `_shuffle` asks of `lax.sort_key_val`, and
// what makes the minibatch permutation bit-exact.  Four 8-bit digit passes per sort, three launches per pass for ALL rows:
//   k_rs_hist    per (row, tile of 4096 pairs): 256-bin digit histogram (LDS integer atomics)
//   k_rs_scan    per row: exclusive scan of the [digit][tile] count table -> where each tile's run of a digit starts
//   k_rs_scatter per (row, tile): stable rank of every pair among its tile's pairs with the same digit, then the move
// A tile is split into four contiguous 1024-pair chunks, one per wave; inside a wave the pairs of equal digit are found
// with eight ballots (one per digit bit), their order is the lane order, and a wave-private LDS counter per digit carries
// the count over the wave's 16 steps -- so a tile needs two workgroup barriers, not one per step.
// ---------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256, RS_ITEMS = 16, RS_TILE = RS_THREADS * RS_ITEMS, RS_RADIX = 256;

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint32_t* __restrict__ keys, int64_t B, int tiles, int shift,
                                                        uint32_t* __restrict__ hist) {
  __shared__ uint32_t s_h[RS_RADIX];
  const int tile = blockIdx.x, e = blockIdx.y, t = threadIdx.x;
  s_h[t] = 0;
  __syncthreads();
  const uint32_t* row = keys + (int64_t)e * B;
  const int64_t base = (int64_t)tile * RS_TILE;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t idx = base + i * RS_THREADS + t;
    if (idx < B) atomicAdd(&s_h[(row[idx] >> shift) & 0xFFu], 1u);
  }
  __syncthreads();
  hist[((int64_t)e * RS_RADIX + t) * tiles + tile] = s_h[t];
}

// one workgroup per row: exclusive scan of hist[e][digit][tile] in (digit, tile) order, in place
__global__ __launch_bounds__(RS_THREADS) void k_rs_scan(uint32_t* __restrict__ hist, int tiles) {
  __shared__ uint32_t s_part[RS_THREADS];
  const int e = blockIdx.x, t = threadIdx.x;
  uint32_t* h = hist + (int64_t)e * RS_RADIX * tiles;     // thread t owns digit t: `tiles` consecutive entries
  uint32_t sum = 0;
  for (int i = 0; i < tiles; ++i) sum += h[(int64_t)t * tiles + i];
  s_part[t] = sum;
  __syncthreads();
  // exclusive scan of the 256 digit totals (Hillis-Steele on LDS)
  uint32_t incl = sum;
  for (int d = 1; d < RS_THREADS; d <<= 1) {
    const uint32_t v = t >= d ? s_part[t - d] : 0u;
    __syncthreads();
    incl += v;
    s_part[t] = incl;
    __syncthreads();
  }
  uint32_t run = incl - sum;
  for (int i = 0; i < tiles; ++i) {
    const uint32_t c = h[(int64_t)t * tiles + i];
    h[(int64_t)t * tiles + i] = run;
    run += c;
  }
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint32_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                                                           int64_t B, int tiles, int shift,
                                                           const uint32_t* __restrict__ offsets) {
  __shared__ uint32_t s_cnt[4][RS_RADIX];    // per wave: running count of each digit, then the wave's start inside the tile's run
  const int tile = blockIdx.x, e = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int i = t; i < 4 * RS_RADIX; i += RS_THREADS) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t rbase = (int64_t)e * B;
  const int64_t base = (int64_t)tile * RS_TILE + (int64_t)w * (RS_TILE / 4);   // this wave's contiguous chunk
  uint32_t key[RS_ITEMS];
  int32_t val[RS_ITEMS];
  uint32_t rank[RS_ITEMS];
  const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t idx = base + i * 64 + lane;
    const bool valid = idx < B;
    key[i] = valid ? keys_in[rbase + idx] : 0u;
    val[i] = valid ? vals_in[rbase + idx] : 0;
    const uint32_t dg = (key[i] >> shift) & 0xFFu;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (dg >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t before = (uint32_t)__popcll(peers & below);
    const uint32_t prev = s_cnt[w][dg];                  // pairs of this digit in the wave's earlier steps
    rank[i] = prev + before;
    // (single wave, program order: every lane has read `prev` before the digit's first lane bumps the counter)
    if (valid && before == 0) s_cnt[w][dg] = prev + (uint32_t)__popcll(peers);
  }
  __syncthreads();
  {  // digit t: the four waves' counts -> their starts inside the tile's run of that digit, plus where the run starts
    const uint32_t g = offsets[((int64_t)e * RS_RADIX + t) * tiles + tile];
    const uint32_t c0 = s_cnt[0][t], c1 = s_cnt[1][t], c2 = s_cnt[2][t];
    s_cnt[0][t] = g;
    s_cnt[1][t] = g + c0;
    s_cnt[2][t] = g + c0 + c1;
    s_cnt[3][t] = g + c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t idx = base + i * 64 + lane;
    if (idx < B) {
      const uint32_t dg = (key[i] >> shift) & 0xFFu;
      const int64_t o = rbase + s_cnt[w][dg] + rank[i];
      keys_out[o] = key[i];
      vals_out[o] = val[i];
    }
  }
}

// sorts every row of (keys, vals) [E, B] in place (kb / vb: equally sized alternates); hist: [E, 256, tiles] uint32
static int radix_sort_rows(uint32_t* keys, int32_t* vals, uint32_t* kb, int32_t* vb, uint32_t* hist, int E, int64_t B,
                           hipStream_t st) {
  const int tiles = (int)((B + RS_TILE - 1) / RS_TILE);
  uint32_t* ki = keys; uint32_t* ko = kb;
  int32_t* vi = vals; int32_t* vo = vb;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 8 * pass;
    hipLaunchKernelGGL(k_rs_hist, dim3(tiles, E), dim3(RS_THREADS), 0, st, ki, B, tiles, shift, hist);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rs_scan, dim3(E), dim3(RS_THREADS), 0, st, hist, tiles);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rs_scatter, dim3(tiles, E), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, B, tiles, shift, hist);
    RLX_LAUNCH_CHECK();
    uint32_t* tk = ki; ki = ko; ko = tk;
    int32_t* tv = vi; vi = vo; vo = tv;
  }
  return RLX_OK;   // four passes: the sorted rows are back in (keys, vals)
}

static inline int grid_for(uint64_t n, int block = 256, int cap = 2048) {
  uint64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > (uint64_t)cap ? cap : g));
}

}  // namespace rlx

using namespace rlx;

extern "C" {

int rlx_threefry_split_host(const uint32_t key_in[2], uint32_t* keys_out, int num, int scheme) {
  RLX_REQUIRE(key_in && keys_out && num > 0, RLX_EINVAL, "rlx_threefry_split_host: bad args");
  split_host(key_in, keys_out, num, scheme);
  return RLX_OK;
}

int rlx_random_bits_u32(rlx_ctx* ctx, const uint32_t key[2], uint32_t* out, int64_t n, int scheme, void* stream) {
  RLX_REQUIRE(ctx && key && out && n >= 0, RLX_EINVAL, "rlx_random_bits_u32: bad args");
  if (n == 0) return RLX_OK;
  hipLaunchKernelGGL(k_random_bits, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, key[0], key[1], out,
                     (uint64_t)n, scheme);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_normal_f32(rlx_ctx* ctx, const uint32_t key[2], float* out, int64_t n, int scheme, void* stream) {
  RLX_REQUIRE(ctx && key && out && n >= 0, RLX_EINVAL, "rlx_normal_f32: bad args");
  if (n == 0) return RLX_OK;
  hipLaunchKernelGGL(k_normal, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, key[0], key[1], out,
                     (uint64_t)n, scheme);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int rlx_permutation_i32(rlx_ctx* ctx, uint32_t key_io[2], int32_t* out, int E, int64_t B, int scheme, void* stream) {
  RLX_REQUIRE(ctx && key_io && out && E > 0 && B > 0, RLX_EINVAL, "rlx_permutation_i32: bad args");
  RLX_REQUIRE(B < (1ll << 31), RLX_EUNSUP, "rlx_permutation_i32: B must fit int32");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)E * B;
  // key, subkey = split(key)
  uint32_t ks[4];
  split_host(key_io, ks, 2, scheme);
  key_io[0] = ks[0]; key_io[1] = ks[1];
  uint32_t key[2] = {ks[2], ks[3]};
  // _shuffle: num_rounds = ceil(3 ln(size) / ln(2^32-1))
  const double sz = (double)(total > 1 ? total : 1);
  const int rounds = (int)ceil(3.0 * log(sz) / log(4294967295.0));

  const int tiles = (int)((B + RS_TILE - 1) / RS_TILE);
  uint32_t* keysA = (uint32_t*)scratch(ctx, SL_SORT_KEYS_A, total * 4);
  uint32_t* keysB = (uint32_t*)scratch(ctx, SL_SORT_KEYS_B, total * 4);
  int32_t* valsB = (int32_t*)scratch(ctx, SL_SORT_VALS_B, total * 4);
  uint32_t* hist = (uint32_t*)scratch(ctx, SL_SORT_TMP, (size_t)E * RS_RADIX * tiles * sizeof(uint32_t));
  if (!keysA || !keysB || !valsB || !hist) return RLX_ENOMEM;
  // every round sorts the rows in place (four digit passes: the pairs end where they started), so `out` holds the
  // permutation from the first launch to the last
  hipLaunchKernelGGL(k_tile_iota, dim3(grid_for(total)), dim3(256), 0, st, out, B, total);
  RLX_LAUNCH_CHECK();
  for (int r = 0; r < rounds; ++r) {
    split_host(key, ks, 2, scheme);  // key, subkey = split(key)
    key[0] = ks[0]; key[1] = ks[1];
    hipLaunchKernelGGL(k_random_bits, dim3(grid_for(total)), dim3(256), 0, st, ks[2], ks[3], keysA, (uint64_t)total,
                       scheme);
    RLX_LAUNCH_CHECK();
    const int rc = radix_sort_rows(keysA, out, keysB, valsB, hist, E, B, st);
    if (rc) return rc;
  }
  return RLX_OK;
}

}  // extern "C"
