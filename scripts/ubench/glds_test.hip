// buffer_load ... lds on gfx950: where do a wave's 16-byte lanes land, and what does an out-of-range lane write?
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/glds_test.hip -o scripts/ubench/glds_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float *x, float *y, unsigned bytes) {
  __shared__ __attribute__((aligned(16))) float s[2 * 64 * 4];
  for (int e = threadIdx.x; e < 2 * 64 * 4; e += blockDim.x) s[e] = 777.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)bytes, 0x00020000);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane l reads the 16 bytes at element 4 * (63 - l) (reversed), odd lanes out of range
  const unsigned off = (lane & 1) ? 0x80000000u : 16u * (63u - lane);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3))) *)(s + wave * 256), 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * 64 * 4; e += blockDim.x) y[e] = s[e];
}
int main() {
  const int n = 64 * 4;
  std::vector<float> hx(n), hy(2 * n);
  for (int i = 0; i < n; ++i) hx[i] = (float)i;
  float *x, *y;
  hipMalloc(&x, n * 4), hipMalloc(&y, 2 * n * 4);
  hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  k<<<1, 128>>>(x, y, n * 4);
  hipMemcpy(hy.data(), y, 2 * n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; ++w)
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 4; ++e) {
        const float got = hy[w * 256 + l * 4 + e], want = (l & 1) ? 0.f : (float)(4 * (63 - l) + e);
        if (got != want && bad++ < 8) printf("wave %d lane %d elem %d: got %g want %g\n", w, l, e, got, want);
      }
  printf("glds test: %d mismatches (lane-linear placement, out-of-range lanes write zeros)\n", bad);
  return bad != 0;
}
