// How fast can ONE CU stream a weight image that every CU reads (L2 hits) -- the pattern of every row-tile kernel of this library
// (k_l12fwd / k_dx_l1bwd / k_rollout_step / k_fwd2h read a 128..512 KB fragment-ordered image per 32-row tile and measure
// ~27-37 B/clk/CU whatever their prefetch depth)?  Each wave reads 1-KiB fragments (64 lanes x 16 B, one global_load_dwordx4) of a
// shared image over and over; variants: waves per workgroup, workgroups per CU, loads in flight per wave, image size, and whether
// all CUs walk the image in the same order or each workgroup starts at its own offset.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/l1fill_probe.hip -o /tmp/l1fill && /tmp/l1fill
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(512) void k_stream(const u32x4* __restrict__ img, int nfrag, int iters, int stagger, unsigned* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
  u32x4 acc = {0u, 0u, 0u, 0u};
  const int start = stagger ? (int)((blockIdx.x * 37u) % (unsigned)nfrag) : 0;
  for (int it = 0; it < iters; ++it) {
    for (int f0 = w * U; f0 < nfrag; f0 += W * U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int f = f0 + u + start;
        if (f >= nfrag) f -= nfrag;
        const u32x4* p = img + (size_t)f * 64 + lane;
        v[u] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = acc.x;
}

template <int U, int NT>
static double run(const u32x4* img, int nfrag, int waves, int wg_per_cu, int stagger, unsigned* sink) {
  const int iters = (int)(64ull * 1024 * 1024 / ((size_t)nfrag * 1024)) + 1;      // ~64 MB per workgroup
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_stream<U, NT>), dim3(256 * wg_per_cu), dim3(64 * waves), 0, 0, img, nfrag, 2, stagger, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_stream<U, NT>), dim3(256 * wg_per_cu), dim3(64 * waves), 0, 0, img, nfrag, iters, stagger, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = (double)iters * nfrag * 1024.0 * wg_per_cu;
  return bytes_per_cu / (ms * 1e-3) / 1e9;      // GB/s per CU
}

int main() {
  const size_t max_bytes = 8u << 20;
  u32x4* img;
  unsigned* sink;
  hipMalloc(&img, max_bytes); hipMalloc(&sink, 4);
  hipMemset(img, 1, max_bytes);
  printf("# l1fill_probe: GB/s PER CU reading a shared image from L2 (x 256 CUs = chip); at 2.4 GHz 64 B/clk = 153.6 GB/s per CU\n");
  printf("| image | waves/WG | WG/CU | in flight per wave (1 KiB each) | same order | staggered start | staggered, nontemporal |\n|---|---|---|---|---|---|---|\n");
  for (int kb : {128, 512, 2048}) {
    const int nfrag = kb;      // 1 KiB fragments
    for (int waves : {4, 8}) {
      for (int wg : {1, 2}) {
#define ROW(U)                                                                                                          \
        printf("| %d KB | %d | %d | %d | %.1f | %.1f | %.1f |\n", kb, waves, wg, U, run<U, 0>(img, nfrag, waves, wg, 0, sink),  \
               run<U, 0>(img, nfrag, waves, wg, 1, sink), run<U, 1>(img, nfrag, waves, wg, 1, sink));
        ROW(2) ROW(4) ROW(8)
#undef ROW
      }
    }
  }
  return 0;
}
