// Diagnostic: the shader clock a running kernel actually sees.  One wave reads the shader-clock counter (s_memtime) and the
// constant reference counter (s_memrealtime) around a ~10 us spin; ratio x reference rate = MHz.
// build: hipcc -O2 --offload-arch=gfx950 -shared -fPIC -o scripts/ubench/libclock_probe.so scripts/ubench/clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void clock_probe_kernel(long long* out, int spin_ref_ticks) {
    if (threadIdx.x != 0) return;
    long long r0 = wall_clock64(), t0 = clock64(), r1 = r0;
    while (r1 - r0 < spin_ref_ticks) r1 = wall_clock64();
    long long t1 = clock64();
    out[0] = t1 - t0;
    out[1] = r1 - r0;
}

extern "C" int clock_probe(void* out, int spin_ref_ticks, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)out, spin_ref_ticks);
    return (int)hipGetLastError();
}

extern "C" int clock_probe_ref_khz() {
    int v = 0, d = 0;
    hipGetDevice(&d);
    hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, d);
    return v;
}

// A single wave that stays resident for n x period reference ticks and writes, per period, the shader-clock ticks it counted:
// the clock the card runs at WHILE other streams' kernels load it.  Ends by itself after n periods.
__global__ void clock_trace_kernel(int* out, int n, int period_ref_ticks) {
    if (threadIdx.x != 0) return;
    long long r = wall_clock64(), t = clock64();
    for (int i = 0; i < n; ++i) {
        long long r1 = r;
        while (r1 - r < period_ref_ticks) { __builtin_amdgcn_s_sleep(32); r1 = wall_clock64(); }
        long long t1 = clock64();
        out[2 * i] = (int)(t1 - t);
        out[2 * i + 1] = (int)(r1 - r);
        r = r1;
        t = t1;
    }
}

extern "C" int clock_trace(void* out, int n, int period_ref_ticks, void* stream) {
    hipLaunchKernelGGL(clock_trace_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)out, n, period_ref_ticks);
    return (int)hipGetLastError();
}
