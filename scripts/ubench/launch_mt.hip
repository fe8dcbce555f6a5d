// Host cost of kernel launches: one thread on one stream vs two threads on a stream each (is the runtime's launch path serialised
// across threads?), with and without event record / wait pairs between the streams.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_mt scripts/ubench/launch_mt.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void tiny(float *p) { if (p && threadIdx.x == 1024) p[0] = 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s[2];
  hipEvent_t ev[64];
  for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  float *d; hipMalloc(&d, 256);
  const int N = 2000;
  auto burst = [&](hipStream_t st, int n) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d); };
  for (int rep = 0; rep < 3; ++rep) {
    burst(s[0], 200); burst(s[1], 200); hipDeviceSynchronize();
    double t0 = now(); burst(s[0], N); double t1 = now(); hipDeviceSynchronize();
    printf("one thread, one stream: %.2f us / launch (host)\n", (t1 - t0) / N * 1e6);
    t0 = now(); for (int i = 0; i < N / 2; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], d); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[1], d); } t1 = now(); hipDeviceSynchronize();
    printf("one thread, two streams alternating: %.2f us / launch\n", (t1 - t0) / N * 1e6);
    t0 = now();
    { std::thread th([&] { burst(s[1], N / 2); }); burst(s[0], N / 2); th.join(); }
    t1 = now(); hipDeviceSynchronize();
    printf("two threads, a stream each: %.2f us / launch (wall over both)\n", (t1 - t0) / N * 1e6);
    // the pattern of a backward pass: main = 2 launches + record, helper = wait + launch
    t0 = now();
    for (int i = 0; i < N / 3; ++i) { burst(s[0], 2); hipEventRecord(ev[i & 63], s[0]); hipStreamWaitEvent(s[1], ev[i & 63], 0); burst(s[1], 1); }
    t1 = now(); hipDeviceSynchronize();
    printf("one thread: (2 launches + record | wait + launch) x %d: %.2f us per group\n", N / 3, (t1 - t0) / (N / 3) * 1e6);
    std::atomic<int> posted{0};
    t0 = now();
    { std::thread th([&] { for (int i = 0; i < N / 3; ++i) { while (posted.load(std::memory_order_acquire) <= i) __builtin_ia32_pause(); hipStreamWaitEvent(s[1], ev[i & 63], 0); burst(s[1], 1); } });
      for (int i = 0; i < N / 3; ++i) { burst(s[0], 2); hipEventRecord(ev[i & 63], s[0]); posted.store(i + 1, std::memory_order_release); }
      double tm = now(); th.join(); t1 = now();
      printf("two threads, same pattern: main %.2f us per group, with the helper's tail %.2f\n", (tm - t0) / (N / 3) * 1e6, (t1 - t0) / (N / 3) * 1e6); }
    hipDeviceSynchronize();
  }
  return 0;
}
