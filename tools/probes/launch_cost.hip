// Host cost of a kernel launch on this box: empty kernel, small and large by-value argument blocks, with and without an
// event record / stream wait in between.   hipcc --offload-arch=gfx950 -O2 tools/probes/launch_cost.hip -o /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float v[240]; };
__global__ void k_empty(int* p) { if (p && threadIdx.x == 1000) *p = 1; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 1000) *p = (int)b.v[3]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  Big b{}; const int N = 20000;
  for (int mode = 0; mode < 5; ++mode) {
    hipDeviceSynchronize();
    const double t0 = now();
    for (int i = 0; i < N; ++i) {
      if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s, nullptr);
      if (mode == 1) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s, b, nullptr);
      if (mode == 2) { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s, nullptr); hipEventRecord(ev, s); }
      if (mode == 3) { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s, nullptr); hipEventRecord(ev, s); hipStreamWaitEvent(s2, ev, 0); hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s2, nullptr); }
      if (mode == 4) { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, (i & 1) ? s : s2, nullptr); }
    }
    const double t1 = now();
    hipDeviceSynchronize();
    const double t2 = now();
    const char* names[5] = {"empty kernel", "960-byte argument block", "launch + event record", "launch + record + wait + launch on 2nd stream", "alternating two streams"};
    printf("%-48s host %.2f us per iteration, until idle %.2f us\n", names[mode], 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
  }
  return 0;
}
