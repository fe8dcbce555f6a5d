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
constexpr int DT = 64, DK = 32, DLT = DT + 1;  // LDS tiles are k-major [k][row]: the 32 lanes of a half-wave read 32 consecutive rows

__global__ __launch_bounds__(256) void dense_xwt_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        float *__restrict__ y, int64_t n, int Kd, int N) {
  __shared__ float sA[2][DK * DLT], sB[2][DK * DLT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * DT;
  const int col0 = blockIdx.y * DT;
  const int wr = 32 * (wave >> 1), wc = 32 * (wave & 1);
  f32x16 acc = (f32x16){0};
  // A = x[row0.., k0..k0+32) and B = w[col0.., k0..k0+32): 64 rows x 8 float4 each, two per thread; the next chunk's
  // loads are issued before this chunk's MFMAs (double-buffered LDS, one barrier per chunk)
  float4 ra[2], rb[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + 256 * it, r = e >> 3, c = (e & 7) * 4;
      ra[it] = rb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < n && k0 + c < Kd) ra[it] = ld4g(x + (row0 + r) * Kd + k0 + c);
      if (col0 + r < N && k0 + c < Kd) rb[it] = ld4g(w + (int64_t)(col0 + r) * Kd + k0 + c);
    }
  };
  auto sts = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + 256 * it, r = e >> 3, c = (e & 7) * 4;
      float *da = sA[buf] + c * DLT + r, *db = sB[buf] + c * DLT + r;
      da[0] = ra[it].x, da[DLT] = ra[it].y, da[2 * DLT] = ra[it].z, da[3 * DLT] = ra[it].w;
      db[0] = rb[it].x, db[DLT] = rb[it].y, db[2 * DLT] = rb[it].z, db[3 * DLT] = rb[it].w;
    }
  };
  gload(0);
  sts(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < Kd; k0 += DK, buf ^= 1) {
    const bool more = k0 + DK < Kd;
    if (more) gload(k0 + DK);
    const float *a = sA[buf] + (lane >> 5) * DLT + wr + (lane & 31), *b = sB[buf] + (lane >> 5) * DLT + wc + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < DK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk * DLT], b[kk * DLT], acc, 0, 0, 0);
    if (more) sts(buf ^ 1);
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


// ---------------------------------------------------------------------------------- classifier head
// logits[b] = mean_{rows of batch b} x[row] @ W + bias  -- MinkowskiGlobalAvgPooling followed by the kernel-volume-1
// `final` convolution (reference models/mink/resnet.py:15-22,93-99,175-177) in ONE launch; one workgroup per batch
// element.  pooled[B][C] is kept for the backward pass.  Fixed summation orders (rows ascending per row lane, lanes
// ascending; channels ascending per quarter, quarters ascending): bitwise reproducible.
constexpr int HB = 256;

__global__ __launch_bounds__(HB) void head_fwd_kernel(const float *__restrict__ x, const int *__restrict__ boff,
                                                      const float *__restrict__ w, const float *__restrict__ bias,
                                                      float *__restrict__ pooled, float *__restrict__ logits, int C,
                                                      int ncls) {
  extern __shared__ float sh[];  // [rlanes][C] partial column sums, then [C] pooled + [4][ncls] partial logits
  const int b = blockIdx.x, t = threadIdx.x;
  const int r0 = boff[b], r1 = boff[b + 1];
  const float inv = r1 > r0 ? 1.f / (float)(r1 - r0) : 0.f;
  const int C4 = C >> 2;
  const int rlanes = C4 >= HB ? 1 : HB / C4;
  for (int c4 = t % C4; c4 < C4; c4 += HB) {  // (one trip unless C > 1024)
    const int rl = C4 >= HB ? 0 : t / C4;
    if (rl < rlanes) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      const float *px = x + 4 * c4;
      int r = r0 + rl;
      for (; r + 3 * rlanes < r1; r += 4 * rlanes) {  // four independent loads in flight; summed in row order
        const float4 v0 = ld4g(px + (int64_t)r * C), v1 = ld4g(px + (int64_t)(r + rlanes) * C),
                     v2 = ld4g(px + (int64_t)(r + 2 * rlanes) * C), v3 = ld4g(px + (int64_t)(r + 3 * rlanes) * C);
        s.x = (((s.x + v0.x) + v1.x) + v2.x) + v3.x, s.y = (((s.y + v0.y) + v1.y) + v2.y) + v3.y;
        s.z = (((s.z + v0.z) + v1.z) + v2.z) + v3.z, s.w = (((s.w + v0.w) + v1.w) + v2.w) + v3.w;
      }
      for (; r < r1; r += rlanes) {
        const float4 v = ld4g(px + (int64_t)r * C);
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
      }
      float *d = sh + (int64_t)rl * C + 4 * c4;
      d[0] = s.x, d[1] = s.y, d[2] = s.z, d[3] = s.w;
    }
  }
  __syncthreads();
  float *sp = sh + (int64_t)rlanes * C;  // pooled row
  for (int c = t; c < C; c += HB) {
    float s = 0.f;
    for (int rl = 0; rl < rlanes; ++rl) s += sh[(int64_t)rl * C + c];
    s *= inv;
    sp[c] = s;
    pooled[(int64_t)b * C + c] = s;
  }
  __syncthreads();
  // logits: thread (part, lane) sums channels part, part + 4, ... of class j = lane (+64 ...): the loads of eight
  // consecutive trips are independent and issued together; one accumulator, channel order ascending within a part
  float *sl = sp + C;  // [4][ncls]
  const int part = t >> 6, lane = t & 63;
  for (int j = lane; j < ncls; j += 64) {
    float s = 0.f;
    int c = part;
    for (; c + 28 < C; c += 32) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = w[(int64_t)(c + 4 * u) * ncls + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) s = fmaf(sp[c + 4 * u], wv[u], s);
    }
    for (; c < C; c += 4) s = fmaf(sp[c], w[(int64_t)c * ncls + j], s);
    sl[part * ncls + j] = s;
  }
  __syncthreads();
  for (int j = t; j < ncls; j += HB)
    logits[(int64_t)b * ncls + j] = ((sl[j] + sl[ncls + j]) + sl[2 * ncls + j]) + sl[3 * ncls + j] + (bias ? bias[j] : 0.f);
}

// Backward of the above in one launch.  Workgroups [0, B): dx[row] = (1 / N_b) * dl[b] @ W^T for every row of batch b.
// Workgroups [B, B + ceil(C / 8)): dW[c][j] = sum_b pooled[b][c] dl[b][j] for eight channels each; the first of them
// also db[j] = sum_b dl[b][j].  b ascending everywhere.
__global__ __launch_bounds__(HB) void head_bwd_kernel(const float *__restrict__ dl, const float *__restrict__ pooled,
                                                      const float *__restrict__ w, const int *__restrict__ boff,
                                                      float *__restrict__ dw, float *__restrict__ db,
                                                      float *__restrict__ dx, int B, int C, int ncls) {
  extern __shared__ float sh[];
  const int t = threadIdx.x;
  if ((int)blockIdx.x < B) {
    if (!dx) return;
    const int b = blockIdx.x;
    const int r0 = boff[b], r1 = boff[b + 1];
    const float inv = r1 > r0 ? 1.f / (float)(r1 - r0) : 0.f;
    float *sd = sh;       // dl[b][:]
    float *sg = sh + ncls;  // dpooled[b][:] * inv
    for (int j = t; j < ncls; j += HB) sd[j] = dl[(int64_t)b * ncls + j];
    __syncthreads();
    for (int c = t; c < C; c += HB) {
      float s = 0.f;
      const float *wr = w + (int64_t)c * ncls;
      int j = 0;
      for (; j + 7 < ncls; j += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = wr[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fmaf(sd[j + u], wv[u], s);
      }
      for (; j < ncls; ++j) s = fmaf(sd[j], wr[j], s);
      sg[c] = s * inv;
    }
    __syncthreads();
    const int C4 = C >> 2;
    for (int64_t i = t; i < (int64_t)(r1 - r0) * C4; i += HB) {
      const int64_t r = r0 + i / C4;
      const int c = (int)(i % C4) * 4;
      *reinterpret_cast<float4 *>(dx + r * C + c) = make_float4(sg[c], sg[c + 1], sg[c + 2], sg[c + 3]);
    }
    return;
  }
  const int c0 = ((int)blockIdx.x - B) * 8;
  for (int e = t; e < 8 * ncls; e += HB) {
    const int c = c0 + e / ncls, j = e % ncls;
    if (c >= C) break;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s = fmaf(pooled[(int64_t)b * C + c], dl[(int64_t)b * ncls + j], s);
    dw[(int64_t)c * ncls + j] = s;
  }
  if (c0 == 0 && db) {
    for (int j = t; j < ncls; j += HB) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += dl[(int64_t)b * ncls + j];
      db[j] = s;
    }
  }
}

// loss = mean_b ( logsumexp(logits[b]) - logits[b][label_b] )  (torch.nn.functional.cross_entropy with its defaults,
// reference modules/classification_training.py:33); prob[B][ncls] = softmax, kept for the backward pass.  One workgroup,
// a wave per row (rows wave, wave + 4, ...); a label outside [0, ncls) makes the loss NaN (torch raises a device assert).
__global__ __launch_bounds__(HB) void ce_fwd_kernel(const float *__restrict__ logits, const long long *__restrict__ labels,
                                                    int B, int ncls, float *__restrict__ prob, float *__restrict__ loss) {
  extern __shared__ float sh[];  // [B] row losses
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int b = wave; b < B; b += HB / 64) {
    const float *row = logits + (int64_t)b * ncls;
    float mx = -INFINITY;
    for (int j = lane; j < ncls; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.f;
    for (int j = lane; j < ncls; j += 64) se += expf(row[j] - mx);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o, 64);
    const float inv = 1.f / se;
    for (int j = lane; j < ncls; j += 64) prob[(int64_t)b * ncls + j] = expf(row[j] - mx) * inv;
    if (lane == 0) {
      const long long y = labels[b];
      sh[b] = (y >= 0 && y < ncls) ? (logf(se) + mx) - row[y] : NAN;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += sh[b];
    *loss = s / (float)B;
  }
}

// dlogits[b][j] = g * (prob[b][j] - [j == label_b]) / B,  g = the incoming gradient of the loss (a device scalar)
__global__ __launch_bounds__(HB) void ce_bwd_kernel(const float *__restrict__ prob, const long long *__restrict__ labels,
                                                    const float *__restrict__ g, int B, int ncls, float *__restrict__ dl) {
  const float s = *g / (float)B;
  for (int i = blockIdx.x * HB + threadIdx.x; i < B * ncls; i += gridDim.x * HB) {
    const int b = i / ncls, j = i - b * ncls;
    dl[i] = s * (prob[i] - (labels[b] == (long long)j ? 1.f : 0.f));
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


int mink_head_forward(const float *x, const int32_t *batch_offsets, int32_t B, int32_t C, const float *w, const float *bias,
                      int32_t ncls, float *pooled, float *logits, void *stream) {
  MINK_REQUIRE(B >= 1 && C >= 4 && (C & 3) == 0 && ncls >= 1, "head_forward: bad shape (B=%d, C=%d, classes=%d)", B, C, ncls);
  MINK_REQUIRE(x && batch_offsets && w && pooled && logits && ((uintptr_t)x & 15) == 0, "head_forward: NULL or misaligned pointer");
  const int C4 = C >> 2, rlanes = C4 >= HB ? 1 : HB / C4;
  const size_t shm = sizeof(float) * ((size_t)rlanes * C + C + 4 * (size_t)ncls);
  MINK_REQUIRE(shm <= 64 * 1024, "head_forward: %d channels x %d classes do not fit the workgroup's LDS", C, ncls);
  head_fwd_kernel<<<dim3((unsigned)B), HB, shm, (hipStream_t)stream>>>(x, batch_offsets, w, bias, pooled, logits, C, ncls);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_head_backward(const float *dlogits, const float *pooled, const float *w, const int32_t *batch_offsets, int32_t B,
                       int32_t C, int32_t ncls, float *dw, float *dbias, float *dx, void *stream) {
  MINK_REQUIRE(B >= 1 && C >= 4 && (C & 3) == 0 && ncls >= 1, "head_backward: bad shape");
  MINK_REQUIRE(dlogits && pooled && w && batch_offsets && dw && (((uintptr_t)dx) & 15) == 0, "head_backward: NULL or misaligned pointer");
  const size_t shm = sizeof(float) * ((size_t)ncls + C);
  MINK_REQUIRE(shm <= 64 * 1024, "head_backward: %d channels x %d classes do not fit the workgroup's LDS", C, ncls);
  head_bwd_kernel<<<dim3((unsigned)(B + cdiv(C, 8))), HB, shm, (hipStream_t)stream>>>(dlogits, pooled, w, batch_offsets, dw, dbias,
                                                                                      dx, B, C, ncls);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_softmax_ce_forward(const float *logits, const int64_t *labels, int32_t B, int32_t ncls, float *prob, float *loss,
                            void *stream) {
  MINK_REQUIRE(B >= 1 && B <= 8192 && ncls >= 1 && logits && labels && prob && loss, "softmax_ce_forward: bad arguments");
  ce_fwd_kernel<<<dim3(1), HB, sizeof(float) * B, (hipStream_t)stream>>>(logits, (const long long *)labels, B, ncls, prob, loss);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_softmax_ce_backward(const float *prob, const int64_t *labels, const float *grad_loss, int32_t B, int32_t ncls,
                             float *dlogits, void *stream) {
  MINK_REQUIRE(B >= 1 && ncls >= 1 && prob && labels && grad_loss && dlogits, "softmax_ce_backward: bad arguments");
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv((int64_t)B * ncls, HB), 1024);
  ce_bwd_kernel<<<dim3(grid), HB, 0, (hipStream_t)stream>>>(prob, (const long long *)labels, grad_loss, B, ncls, dlogits);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

}  // extern "C"
