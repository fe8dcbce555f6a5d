// Small dense pieces beside the sparse convolutions (gfx950): the data gradient of the 1x1x1 strided shortcut
// convolution as a plain row-major GEMM + a row scatter-add, and the classifier head.
//
// Why they exist: in a training step these operators sit on the CRITICAL chain of ~100 dependent launches between the
// two stem kernels, each of them a few microseconds of work behind a 5-10 us launch floor (profiles/r03_*): the
// shortcut's data gradient ran through the generic gather-GEMM with a one-column table and a read-modify-write epilogue
// (24-64 us per stage, after the main branch's data gradient), and global pooling + the linear layer + cross-entropy and
// their backward were ~17 library launches.  Here the shortcut product is computed BESIDE the main branch (its own
// stream) and only a 5 us row scatter-add stays on the chain; the head is two launches forward and two backward.
#include <algorithm>

#include "common.h"

namespace mink {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float4 ld4g(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// y[n][N] = x[n][Kd] @ W^T with W stored [N][Kd] (row-major): the data gradient of a 1x1x1 convolution whose forward
// kernel is W[cin = N][cout = Kd] (reference modules/common.py:116-125 with kernel_size 1: `downsample`,
// models/mink/resnet.py:120-128).  64 x 64 output tile per workgroup, four waves of 32 x 32 on v_mfma_f32_32x32x2_f32:
// one accumulator chain per output element, k ascending -- bit for bit the sum the gather-GEMM forms for a one-column table.
constexpr int DT = 64, DK = 32, DLD = DK + 1;  // odd LDS row stride: the 32 lanes of a half-wave read 32 rows of one column

__global__ __launch_bounds__(256) void dense_xwt_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        float *__restrict__ y, int64_t n, int Kd, int N) {
  __shared__ float sA[DT * DLD], sB[DT * DLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * DT;
  const int col0 = blockIdx.y * DT;
  const int wr = 32 * (wave >> 1), wc = 32 * (wave & 1);
  f32x16 acc = (f32x16){0};
  for (int k0 = 0; k0 < Kd; k0 += DK) {
    // stage A = x[row0.., k0..k0+32) and B = w[col0.., k0..k0+32): 64 rows x 8 float4 each, two per thread
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + 256 * it, r = e >> 3, c = (e & 7) * 4;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (row0 + r < n && k0 + c < Kd) a = ld4g(x + (row0 + r) * Kd + k0 + c);
      if (col0 + r < N && k0 + c < Kd) b = ld4g(w + (int64_t)(col0 + r) * Kd + k0 + c);
      float *da = sA + r * DLD + c, *db = sB + r * DLD + c;
      da[0] = a.x, da[1] = a.y, da[2] = a.z, da[3] = a.w;
      db[0] = b.x, db[1] = b.y, db[2] = b.z, db[3] = b.w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < DK; kk += 2) {
      const float a = sA[(wr + (lane & 31)) * DLD + kk + (lane >> 5)];
      const float b = sB[(wc + (lane & 31)) * DLD + kk + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = col0 + wc + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = row0 + wr + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < n && col < N) y[row * N + col] = acc[r];
  }
}

// dst[idx[r]][:] += src[r][:] for every r with idx[r] >= 0; the idx values are distinct (each destination row has one owner)
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const float *__restrict__ src, const int *__restrict__ idx,
                                                               float *__restrict__ dst, int64_t n_src, int C4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_src * C4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    const int d = idx[r];
    if (d < 0) continue;
    float *p = dst + ((int64_t)d * C4) * 4 + c;
    const float4 a = ld4g(src + r * C4 * 4 + c), b = ld4g(p);
    *reinterpret_cast<float4 *>(p) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

}  // namespace mink

using namespace mink;

extern "C" {

int mink_dense_xwt(const float *x, const float *w, int64_t n, int32_t Kd, int32_t N, float *y, void *stream) {
  MINK_REQUIRE(n >= 0 && Kd >= 4 && N >= 1 && (Kd & 3) == 0, "dense_xwt: bad shape (%lld x %d) @ (%d x %d)^T", (long long)n, Kd, N, Kd);
  if (n == 0) return MINK_OK;
  MINK_REQUIRE(x && w && y && (((uintptr_t)x | (uintptr_t)w) & 15) == 0, "dense_xwt: NULL or misaligned pointer");
  dense_xwt_kernel<<<dim3((unsigned)cdiv(n, DT), (unsigned)cdiv(N, DT)), 256, 0, (hipStream_t)stream>>>(x, w, y, n, Kd, N);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_rows_scatter_add(const float *src, const int32_t *idx, int64_t n_src, int32_t C, float *dst, void *stream) {
  MINK_REQUIRE(n_src >= 0 && C >= 4 && (C & 3) == 0, "rows_scatter_add: bad shape");
  if (n_src == 0) return MINK_OK;
  MINK_REQUIRE(src && idx && dst && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "rows_scatter_add: NULL or misaligned pointer");
  const int64_t work = n_src * (C >> 2);
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(work, 256), 8192);
  rows_scatter_add_kernel<<<dim3(grid), 256, 0, (hipStream_t)stream>>>(src, idx, dst, n_src, C >> 2);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

}  // extern "C"
