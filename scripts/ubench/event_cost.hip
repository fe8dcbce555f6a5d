// Host cost of the cross-stream primitives: event record, stream-wait-event, and the stream memory operations.
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/event_cost scripts/ubench/event_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(float *p) { if (p && threadIdx.x == 1024) p[0] = 1.f; }
__global__ void spin(float *p, long long clocks) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < clocks) {} if (threadIdx.x == 1024) p[0] = 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  hipStream_t s[2];
  hipEvent_t ev[64];
  for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  float *d; CK(hipMalloc(&d, 4096));
  uint32_t *flag; CK(hipMalloc(&flag, 4096)); CK(hipMemset(flag, 0, 4096));
  const int N = 600;
  for (int busy = 0; busy < 2; ++busy) {
    printf("--- GPU %s\n", busy ? "backlogged (a 20 ms kernel at the head of stream 0)" : "idle");
    auto head = [&] { if (busy) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[0], d, 2000000ll); };  // 100 MHz wall clock: 20 ms
    double t0, t1;
    head(); t0 = now(); for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], d); t1 = now(); CK(hipDeviceSynchronize());
    printf("launch: %.2f us\n", (t1 - t0) / N * 1e6);
    head(); t0 = now(); for (int i = 0; i < N; ++i) CK(hipEventRecord(ev[i & 63], s[0])); t1 = now(); CK(hipDeviceSynchronize());
    printf("event record: %.2f us\n", (t1 - t0) / N * 1e6);
    head(); for (int i = 0; i < 64; ++i) CK(hipEventRecord(ev[i], s[0]));
    t0 = now(); for (int i = 0; i < N; ++i) CK(hipStreamWaitEvent(s[1], ev[i & 63], 0)); t1 = now(); CK(hipDeviceSynchronize());
    printf("stream wait event (recorded%s): %.2f us\n", busy ? ", pending" : ", complete", (t1 - t0) / N * 1e6);
    head(); t0 = now(); for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], d); CK(hipEventRecord(ev[i & 63], s[0])); CK(hipStreamWaitEvent(s[1], ev[i & 63], 0)); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[1], d); }
    t1 = now(); CK(hipDeviceSynchronize());
    printf("launch + record + wait + launch: %.2f us\n", (t1 - t0) / N * 1e6);
    head(); t0 = now(); for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], d); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[1], d); }
    t1 = now(); CK(hipDeviceSynchronize());
    printf("launch + launch (two streams, no dependency): %.2f us\n", (t1 - t0) / N * 1e6);
    head(); t0 = now(); hipError_t e = hipSuccess;
    for (int i = 0; i < N && e == hipSuccess; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], d); e = hipStreamWriteValue32(s[0], flag, (uint32_t)(i + 1), 0); if (e == hipSuccess) e = hipStreamWaitValue32(s[1], flag, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[1], d); }
    t1 = now(); printf("launch + write value + wait value + launch: %.2f us (%s)\n", (t1 - t0) / N * 1e6, hipGetErrorString(e)); (void)hipDeviceSynchronize();
  }
  return 0;
}
