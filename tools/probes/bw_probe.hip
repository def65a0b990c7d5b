// Why do the update's large kernels move their bytes at ~3 TB/s and not at the 6.3 TB/s a float4 copy reaches?
// A plain fp32 streaming kernel launched with THEIR residency (256 threads, 1 or 2 workgroups per CU through a dynamic-LDS
// footprint of 129 / 76 KB) and a variable number of 16-byte loads in flight per lane, against the same kernel at full occupancy.
//   mode 0: copy (read + write)   mode 1: write only   mode 2: read only (sum)
//   stores: plain / nontemporal;  loads: plain / nontemporal
// Persistent grid-stride form: grid = workgroups-per-CU x 256 CUs, each workgroup walks the buffer (the shape of k_l12fwd /
// k_tail_bx / k_dx_l1bwd: one resident wave of workgroups, each handling several row tiles).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, int MODE, bool NT_ST, bool NT_LD>
__global__ __launch_bounds__(256) void k_stream(const f4* __restrict__ src, f4* __restrict__ dst, size_t n4, float* sink) {
  extern __shared__ float lds[];
  const size_t stride = (size_t)gridDim.x * 256 * U;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256;
      if (MODE != 1) v[u] = (i < n4) ? (NT_LD ? __builtin_nontemporal_load(src + i) : src[i]) : acc;
      else v[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256;
      if (MODE == 2) acc += v[u];
      else if (i < n4) {
        if (NT_ST) __builtin_nontemporal_store(v[u], dst + i);
        else dst[i] = v[u];
      }
    }
  }
  if (MODE == 2 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = lds[threadIdx.x];   // keeps the loads (and the LDS allocation) alive
}

struct Case { const char* name; void (*fn)(const f4*, f4*, size_t, float*); int mode; };

template <int U, int MODE, bool NS, bool NL>
static double run(const f4* src, f4* dst, size_t n4, float* sink, int wg_per_cu, size_t lds, int reps) {
  auto fn = k_stream<U, MODE, NS, NL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = wg_per_cu * 256;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, src, dst, n4, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, src, dst, n4, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  const double bytes = (double)n4 * 16 * (MODE == 0 ? 2 : 1);
  return bytes * reps / (ms * 1e-3) / 1e12;
}

int main() {
  const size_t bytes = 512ull << 20;          // 512 MB per buffer: past the 256 MB Infinity Cache
  const size_t n4 = bytes / 16;
  f4 *src, *dst;
  float* sink;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&sink, 4);
  hipMemset(src, 0, bytes); hipMemset(dst, 0, bytes);
  struct Occ { const char* name; int wg; size_t lds; } occ[3] = {{"8 WG/CU (no LDS)", 8, 0}, {"2 WG/CU (76 KB LDS each)", 2, 76 * 1024},
                                                                 {"1 WG/CU (129 KB LDS)", 1, 129 * 1024}};
  printf("# bw_probe: 512 MB buffers, 256-thread workgroups, persistent grid = WG/CU x 256; TB/s (copy counts read + write)\n");
  printf("| residency | loads in flight per lane (16 B each) | copy | copy nt-store | copy nt-load+store | write | write nt | read | read nt |\n|---|---|---|---|---|---|---|---|---|\n");
  for (auto& o : occ) {
#define ROW(U)                                                                                                              \
    printf("| %s | %d | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |\n", o.name, U,                                       \
           run<U, 0, false, false>(src, dst, n4, sink, o.wg, o.lds, 10), run<U, 0, true, false>(src, dst, n4, sink, o.wg, o.lds, 10), \
           run<U, 0, true, true>(src, dst, n4, sink, o.wg, o.lds, 10), run<U, 1, false, false>(src, dst, n4, sink, o.wg, o.lds, 10), \
           run<U, 1, true, false>(src, dst, n4, sink, o.wg, o.lds, 10), run<U, 2, false, false>(src, dst, n4, sink, o.wg, o.lds, 10), \
           run<U, 2, false, true>(src, dst, n4, sink, o.wg, o.lds, 10));
    ROW(1) ROW(2) ROW(4) ROW(8) ROW(16)
#undef ROW
  }
  // the same with the buffer sized like ONE activation tensor of the update (h2 of a 32768-row minibatch: 32 MB; h1: 64 MB): what
  // the Infinity Cache does for a tensor written by one kernel and read by the next
  for (size_t mb : {32, 64, 128}) {
    const size_t m4 = (mb << 20) / 16;
    printf("| %zu MB buffer, 2 WG/CU, 8 in flight | 8 | %.2f | %.2f | - | %.2f | %.2f | %.2f | - |\n", mb,
           run<8, 0, false, false>(src, dst, m4, sink, 2, 76 * 1024, 40), run<8, 0, true, false>(src, dst, m4, sink, 2, 76 * 1024, 40),
           run<8, 1, false, false>(src, dst, m4, sink, 2, 76 * 1024, 40), run<8, 1, true, false>(src, dst, m4, sink, 2, 76 * 1024, 40),
           run<8, 2, false, false>(src, dst, m4, sink, 2, 76 * 1024, 40));
  }
  return 0;
}
