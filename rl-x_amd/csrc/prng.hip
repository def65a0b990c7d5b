// prng.hip -- jax.random restated for gfx950: threefry bits, normal, and the
// bit-exact `permutation(key, x, axis=1, independent=True)` used for PPO
// minibatch indices (reference call site rl_x/algorithms/ppo/flax/ppo.py:191-194).
#include "common.h"
#include <rocprim/device/device_radix_sort.hpp>

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

  uint32_t* keysA = (uint32_t*)scratch(ctx, SL_SORT_KEYS_A, total * 4);
  uint32_t* keysB = (uint32_t*)scratch(ctx, SL_SORT_KEYS_B, (size_t)B * 4);
  int32_t* valsB = (int32_t*)scratch(ctx, SL_SORT_VALS_B, total * 4);
  if (!keysA || !keysB || !valsB) return RLX_ENOMEM;
  size_t tmp_bytes = 0;
  RLX_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keysA, keysB, valsB, out, (size_t)B, 0, 32, st));
  void* tmp = scratch(ctx, SL_SORT_TMP, tmp_bytes);
  if (!tmp) return RLX_ENOMEM;

  // ping-pong so that the last round lands in `out`
  int32_t* src = (rounds % 2 == 0) ? out : valsB;
  int32_t* dst = (rounds % 2 == 0) ? valsB : out;
  hipLaunchKernelGGL(k_tile_iota, dim3(grid_for(total)), dim3(256), 0, st, src, B, total);
  RLX_LAUNCH_CHECK();
  for (int r = 0; r < rounds; ++r) {
    split_host(key, ks, 2, scheme);  // key, subkey = split(key)
    key[0] = ks[0]; key[1] = ks[1];
    hipLaunchKernelGGL(k_random_bits, dim3(grid_for(total)), dim3(256), 0, st, ks[2], ks[3], keysA, (uint64_t)total,
                       scheme);
    RLX_LAUNCH_CHECK();
    for (int e = 0; e < E; ++e) {
      RLX_HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keysA + e * B, keysB, src + e * B, dst + e * B, (size_t)B,
                                            0, 32, st));
    }
    int32_t* t = src; src = dst; dst = t;
  }
  if (rounds == 0) { /* src already holds iota; make sure it is `out` */
    if (src != out) RLX_HIP_TRY(hipMemcpyAsync(out, src, total * 4, hipMemcpyDeviceToDevice, st));
  }
  return RLX_OK;
}

}  // extern "C"
