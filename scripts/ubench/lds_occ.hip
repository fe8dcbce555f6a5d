// How many workgroups of 256 threads fit on a CU for a given dynamic LDS size?  (hipOccupancyMaxActiveBlocksPerMultiprocessor)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(float *o) {
  extern __shared__ float s[];
  s[threadIdx.x] = o[threadIdx.x];
  __syncthreads();
  o[threadIdx.x] = s[255 - threadIdx.x];
}
int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor);
  for (int kb : {32, 64, 70, 72, 74, 76, 77, 78, 79, 80, 81, 82, 96, 128, 160}) {
    int n = -1;
    hipError_t e1 = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, kb * 1024);
    printf("%d KB: attr %d occ %d blocks/CU (err %d)\n", kb, (int)e1, n, (int)e);
  }
  return 0;
}
