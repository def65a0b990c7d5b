// Probe of two gfx950 instructions the plane-consuming kernels build on (run on the GPU box: hipcc tr_probe.hip -o tr_probe && ./tr_probe)
//   ds_read_b64_tr_b16   which 16-bit LDS elements land in which lane / slot
//   global_load_lds_dwordx4 (__builtin_amdgcn_global_load_lds, 16 B per lane)   where a lane's 16 bytes land
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_tr(const int* lane_addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;   // element value = its index
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)lds;                       // LDS byte address (low 32 bits of the generic pointer)
  unsigned addr = (unsigned)(__builtin_amdgcn_readfirstlane(0)) + lane_addr[threadIdx.x];
  unsigned a = (unsigned)((uintptr_t)(&lds[0]) & 0xffffffffu) + lane_addr[threadIdx.x];
  (void)base; (void)addr;
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)(v >> (16 * j));
}

__global__ void k_glds(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  // lane l reads 16 bytes at src + 4 * perm(l) dwords where perm reverses the lanes: shows that the LDS destination is
  // base + 16 * lane whatever the source address
  const uint32_t* g = src + 4 * (63 - threadIdx.x);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t*)g, (__attribute__((address_space(3))) uint32_t*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

int main() {
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 256 * 2);
  const char* names[3] = {"addr = 8 * lane (natural)", "addr = row-major 4x16 block: (lane&15)*2 + ... uniform base", "addr = 32 * lane (stride 16 elements)"};
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) a[l] = mode == 0 ? 8 * l : (mode == 1 ? 0 : 32 * l);
    hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    std::vector<uint16_t> o(256);
    hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16, %s\n", names[mode]);
    for (int l = 0; l < 64; ++l) { printf(" l%02d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", o[l * 4 + j]); if (l % 4 == 3) printf("\n"); }
  }
  uint32_t *d_src, *d_o2;
  hipMalloc(&d_src, 1024); hipMalloc(&d_o2, 1024);
  std::vector<uint32_t> s(256); for (int i = 0; i < 256; ++i) s[i] = i;
  hipMemcpy(d_src, s.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_glds, dim3(1), dim3(64), 0, 0, d_src, d_o2);
  std::vector<uint32_t> o2(256); hipMemcpy(o2.data(), d_o2, 1024, hipMemcpyDeviceToHost);
  printf("global_load_lds x4: LDS dword i holds source dword:\n");
  for (int i = 0; i < 256; ++i) { printf(" %3u", o2[i]); if (i % 16 == 15) printf("\n"); }
  return 0;
}
