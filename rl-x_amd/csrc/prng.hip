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
//   k_rs_hist    per (row, tile of 4096 pairs): 256-bin digit histogram (LDS integer atomics) + the row's digit totals
//   k_rs_scan    per (row, 16 digits): exclusive scan of the [digit][tile] count table -> where each tile's run of a digit starts
//   k_rs_scatter per (row, tile): stable rank of every pair among its tile's pairs with the same digit, the tile put in digit
//                order in LDS, then the move with consecutive threads on consecutive pairs of a run
// A tile is split into four contiguous 1024-pair chunks, one per wave; inside a wave the pairs of equal digit are found
// with eight ballots (one per digit bit), their order is the lane order, and a wave-private LDS counter per digit carries
// the count over the wave's 16 steps -- so ranking a tile needs two workgroup barriers, not one per step.
// ---------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256, RS_ITEMS = 16, RS_TILE = RS_THREADS * RS_ITEMS, RS_RADIX = 256;

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint32_t* __restrict__ keys, int64_t B, int tiles, int shift,
                                                        uint32_t* __restrict__ hist, uint32_t* __restrict__ tot) {
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
  if (s_h[t]) atomicAdd(&tot[e * RS_RADIX + t], s_h[t]);      // the row's digit totals (integer: order-independent), zeroed per sort
}

// exclusive scan of hist[e][digit][tile] in (digit, tile) order: hist -> offs (out of place).  RS_SCAN_SPLIT workgroups per row,
// workgroup q finishes digits [16 q, 16 q + 16), wave w four of them: the tile counts of its digits (one coalesced read per
// digit and 64 tiles) and the row's 256 digit totals (k_rs_hist's atomics) are requested together -- ONE round trip to memory --,
// the totals are scanned in LDS, the tile counts with a wave prefix sum per 64 tiles.  (Rounds 1-5: one workgroup per row, one
// thread per digit walking its `tiles` entries twice: 57-75 us per pass at T*N = 524288, twelve passes per permutation.)
constexpr int RS_SCAN_SPLIT = 16, RS_SCAN_CH = 2;
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
__global__ __launch_bounds__(RS_THREADS) void k_rs_scan(const uint32_t* __restrict__ hist, uint32_t* __restrict__ offs,
                                                        const uint32_t* __restrict__ tot, int tiles) {
  __shared__ uint32_t s_tot[RS_RADIX];
  __shared__ uint32_t s_part[RS_THREADS];
  const int e = blockIdx.x, q = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint32_t* h = hist + (int64_t)e * RS_RADIX * tiles;
  uint32_t* o = offs + (int64_t)e * RS_RADIX * tiles;
  const int nch = (tiles + 63) >> 6, dg0 = 16 * q + 4 * w;
  const uint32_t sum = tot[e * RS_RADIX + t];
  // (fixed trip counts: clamped addresses, masked values -- a run-time `i < tiles` loop per digit is one round trip after the other)
  uint32_t v[4][RS_SCAN_CH];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int c = 0; c < RS_SCAN_CH; ++c) {
      const int i = 64 * c + lane;
      v[u][c] = h[(int64_t)(dg0 + u) * tiles + (i < tiles ? i : 0)];
      if (i >= tiles) v[u][c] = 0;
    }
  // exclusive scan of the 256 digit totals (Hillis-Steele on LDS)
  s_part[t] = sum;
  __syncthreads();
  uint32_t incl = sum;
  for (int d = 1; d < RS_THREADS; d <<= 1) {
    const uint32_t x = t >= d ? s_part[t - d] : 0u;
    __syncthreads();
    incl += x;
    s_part[t] = incl;
    __syncthreads();
  }
  s_tot[t] = incl - sum;      // where digit t's first tile starts
  __syncthreads();
  uint32_t run[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) run[u] = s_tot[dg0 + u];
  for (int c0 = 0; c0 < nch; c0 += RS_SCAN_CH) {
    if (c0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < RS_SCAN_CH; ++c) {
          const int i = 64 * (c0 + c) + lane;
          v[u][c] = h[(int64_t)(dg0 + u) * tiles + (i < tiles ? i : 0)];
          if (i >= tiles) v[u][c] = 0;
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < RS_SCAN_CH; ++c) {
        const int i = 64 * (c0 + c) + lane;
        const uint32_t inc = wave_incl_scan_u32(v[u][c], lane);
        if (i < tiles) o[(int64_t)(dg0 + u) * tiles + i] = run[u] + inc - v[u][c];
        run[u] += (uint32_t)__shfl((int)inc, 63, 64);
      }
  }
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint32_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                                                           int64_t B, int tiles, int shift,
                                                           const uint32_t* __restrict__ offsets) {
  __shared__ uint32_t s_cnt[4][RS_RADIX];    // per wave: running count of each digit, then where the wave's pairs of the digit start in the tile
  __shared__ uint32_t s_gofs[RS_RADIX];
  __shared__ uint32_t s_wsum[4];
  __shared__ uint32_t s_key[RS_TILE];
  __shared__ int32_t s_val[RS_TILE];
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
  {  // digit t: the four waves' counts -> where each wave's pairs of that digit start INSIDE THE TILE (digits ascending: an exclusive
     // scan of the tile's digit counts), and the distance from there to the run's place in the row
    const uint32_t g = offsets[((int64_t)e * RS_RADIX + t) * tiles + tile];
    const uint32_t c0 = s_cnt[0][t], c1 = s_cnt[1][t], c2 = s_cnt[2][t], c3 = s_cnt[3][t];
    const uint32_t tot = c0 + c1 + c2 + c3;
    const uint32_t incl = wave_incl_scan_u32(tot, lane);
    if (lane == 63) s_wsum[w] = incl;
    __syncthreads();
    uint32_t excl = incl - tot;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (q < w) excl += s_wsum[q];
    s_cnt[0][t] = excl;
    s_cnt[1][t] = excl + c0;
    s_cnt[2][t] = excl + c0 + c1;
    s_cnt[3][t] = excl + c0 + c1 + c2;
    s_gofs[t] = g - excl;      // (mod 2^32) position in the row = s_gofs[digit] + position in the tile
  }
  __syncthreads();
  // the tile in digit order in LDS ...
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t idx = base + i * 64 + lane;
    if (idx < B) {
      const uint32_t dg = (key[i] >> shift) & 0xFFu;
      const uint32_t pos = s_cnt[w][dg] + rank[i];
      s_key[pos] = key[i];
      s_val[pos] = val[i];
    }
  }
  __syncthreads();
  // ... and out: consecutive threads move consecutive pairs of a run (a run of a digit averages 16 pairs = 64 B per array; the
  // per-pair stores of rounds 1-5 were one 32-B sector write each: 10.5 M of them per pass at T*N = 524288, 60-73 us)
  const int64_t left = B - (int64_t)tile * RS_TILE;
  const int n_valid = left < RS_TILE ? (int)left : RS_TILE;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int p = i * RS_THREADS + t;
    if (p < n_valid) {
      const uint32_t k = s_key[p];
      const int64_t o = rbase + (uint32_t)(s_gofs[(k >> shift) & 0xFFu] + (uint32_t)p);
      keys_out[o] = k;
      vals_out[o] = s_val[p];
    }
  }
}

// sorts every row of (keys, vals) [E, B] in place (kb / vb: equally sized alternates); hist: 2 x [E, 256, tiles] (counts, offsets) + [4, E, 256] (digit totals per pass) uint32
static int radix_sort_rows(uint32_t* keys, int32_t* vals, uint32_t* kb, int32_t* vb, uint32_t* hist, int E, int64_t B,
                           hipStream_t st) {
  const int tiles = (int)((B + RS_TILE - 1) / RS_TILE);
  uint32_t* offs = hist + (int64_t)E * RS_RADIX * tiles;
  uint32_t* tot = offs + (int64_t)E * RS_RADIX * tiles;        // [4 passes][E][256] digit totals
  RLX_HIP_TRY(hipMemsetAsync(tot, 0, (size_t)4 * E * RS_RADIX * sizeof(uint32_t), st));
  uint32_t* ki = keys; uint32_t* ko = kb;
  int32_t* vi = vals; int32_t* vo = vb;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 8 * pass;
    hipLaunchKernelGGL(k_rs_hist, dim3(tiles, E), dim3(RS_THREADS), 0, st, ki, B, tiles, shift, hist, tot + (int64_t)pass * E * RS_RADIX);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rs_scan, dim3(E, RS_SCAN_SPLIT), dim3(RS_THREADS), 0, st, hist, offs, tot + (int64_t)pass * E * RS_RADIX, tiles);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rs_scatter, dim3(tiles, E), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, B, tiles, shift, offs);
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
  uint32_t* hist = (uint32_t*)scratch(ctx, SL_SORT_TMP, (size_t)(2 * tiles + 4) * E * RS_RADIX * sizeof(uint32_t));
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
