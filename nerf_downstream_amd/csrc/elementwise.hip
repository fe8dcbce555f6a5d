// Bandwidth-bound feature-matrix kernels on gfx950: batch norm (stats / apply / backward,
// optional fused ReLU + residual), ReLU, add, sum pooling, global average pooling and the
// TensorField->SparseTensor segment mean.  All are HBM-roofline work: 16-byte accesses per
// lane, channels fastest so a wave reads whole feature rows, no atomics (two-stage
// deterministic reductions through a small workspace).
#include <algorithm>

#include "common.h"

namespace mink {

constexpr int EB = 256;
static bool g_bn_small = true;  // few-row layers: the one-launch batch norm (mink_bn_set_small)
static bool g_bn_fold_force = false;
static int g_bn_fold = 0;       // finalize inside the apply pass when there are at most this many partial rows AND re-reading them is cheap (fold_ok; mink_bn_set_fold).  0 = never, the default: measured slower in every configuration
constexpr int kRedBlocks = 2048;  // workgroups of the column reductions (8 per CU: the passes are latency-bound)

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// four consecutive elements of a row-major matrix stored as fp32 or (B16) as bf16; `i` counts ELEMENTS
template <bool B16>
__device__ __forceinline__ float4 ldx4(const float *p, int64_t i) {
  if constexpr (!B16) return *reinterpret_cast<const float4 *>(p + i);
  const uint2 u = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(p) + i);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}

// (contraction off: the same source line became a multiply + fma in one kernel and two multiplies + add in another -- the
//  running statistics of the folded and the separate finalize must be equal bit for bit; __fmul_rn is a plain `*` in HIP)
__device__ __forceinline__ void bn_running_update(float *running_mean, float *running_var, int c, float momentum, float m, double var, int64_t n) {
#pragma clang fp contract(off)
  const double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
  const float keep = 1.f - momentum, ub = (float)unbiased;
  const float a0 = keep * running_mean[c], a1 = momentum * m;
  running_mean[c] = a0 + a1;
  const float b0 = keep * running_var[c], b1 = momentum * ub;
  running_var[c] = b0 + b1;
}
__device__ __forceinline__ double bn_variance(double s, double ss, int64_t n, double &m) {
#pragma clang fp contract(off)
  m = s / (double)n;
  const double mm = m * m;
  const double var = ss / (double)n - mm;
  return var < 0.0 ? 0.0 : var;
}

__device__ __forceinline__ float4 bn_affine(float4 v, float4 mu, float4 is, float4 g, float4 b) {
#pragma clang fp contract(off)  // (two kernels share this line and must round it the same way: see bn_running_update)
  float4 o;
  o.x = (v.x - mu.x) * is.x * g.x + b.x, o.y = (v.y - mu.y) * is.y * g.y + b.y;
  o.z = (v.z - mu.z) * is.z * g.z + b.z, o.w = (v.w - mu.w) * is.w * g.w + b.w;
  return o;
}

__device__ __forceinline__ float4 bn_dx(float4 g, float4 v, float4 mu, float4 is, float4 ga, float4 dg, float4 db, float inv_n) {
#pragma clang fp contract(off)
  float4 o;
  o.x = ga.x * is.x * (g.x - db.x * inv_n - (v.x - mu.x) * is.x * dg.x * inv_n);
  o.y = ga.y * is.y * (g.y - db.y * inv_n - (v.y - mu.y) * is.y * dg.y * inv_n);
  o.z = ga.z * is.z * (g.z - db.z * inv_n - (v.z - mu.z) * is.z * dg.z * inv_n);
  o.w = ga.w * is.w * (g.w - db.w * inv_n - (v.w - mu.w) * is.w * dg.w * inv_n);
  return o;
}

// one row's contribution to (sum g, sum g xhat) of the batch-norm backward; shared (and contraction-free) because the slab-summing
// form below must accumulate exactly what colreduce_kernel<1> accumulates
__device__ __forceinline__ void bn_bwd_accumulate(float4 g, float4 x, float4 mu, float4 is, float4 &s0, float4 &s1) {
#pragma clang fp contract(off)
  s0.x += g.x, s0.y += g.y, s0.z += g.z, s0.w += g.w;
  s1.x += g.x * (x.x - mu.x) * is.x, s1.y += g.y * (x.y - mu.y) * is.y, s1.z += g.z * (x.z - mu.z) * is.z,
      s1.w += g.w * (x.w - mu.w) * is.w;
}

// ---------------------------------------------------------------------- column sums
// Generic two-quantity column reduction over rows: each thread owns 4 channels (one float4
// column) and strides over rows; partial[blk][2][C] in double.
// MODE 0: (sum x, sum x^2)      MODE 1: (sum g, sum g*xhat) with g = dy * (relu ? y>0 : 1)
// MODE 2: as MODE 1 for the fused bn+relu+sum-pool: g = dy_pool[in2out[row]] * (gamma*xhat+beta > 0)
template <int MODE, bool B16 = false>  // B16 (MODE 2 only): `b`, the convolution output, is stored as bf16
__global__ __launch_bounds__(EB) void colreduce_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                       const float *__restrict__ yrelu, int64_t n, int C,
                                                       const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, double *__restrict__ partial,
                                                       const int *__restrict__ in2out = nullptr,
                                                       const float *__restrict__ gamma = nullptr,
                                                       const float *__restrict__ beta = nullptr, int ld = 0) {
  // more than 4 * EB channels (the Bottleneck nets reach 2048): blockIdx.y walks slabs of C channels
  // of rows that are LD floats long; `partial` keeps the [blk][2][LD] layout
  const int LD = ld ? ld : C;
  {
    const int c0 = blockIdx.y * C;
    a += c0, b = b ? b + c0 : b, yrelu = yrelu ? yrelu + c0 : yrelu, partial += c0;
    mean = mean ? mean + c0 : mean, invstd = invstd ? invstd + c0 : invstd;
    gamma = gamma ? gamma + c0 : gamma, beta = beta ? beta + c0 : beta;
  }
  extern __shared__ double s_red[];  // [rows_in_block][2][C] reduced over the row lanes
  const int tpr = C >> 2;            // threads per row
  const int rlanes = EB / tpr;       // rows handled concurrently
  const int c4 = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  float4 s0 = make_float4(0, 0, 0, 0), s1 = make_float4(0, 0, 0, 0);
  float4 mu = make_float4(0, 0, 0, 0), is = make_float4(1, 1, 1, 1);
  const bool active = rl < rlanes;
  float4 ga = make_float4(1, 1, 1, 1), be = make_float4(0, 0, 0, 0);
  if (MODE >= 1 && active) mu = ld4(mean + 4 * c4), is = ld4(invstd + 4 * c4);
  if (MODE == 2 && active) ga = ld4(gamma + 4 * c4), be = ld4(beta + 4 * c4);
  if (active) {
    for (int64_t row = (int64_t)blockIdx.x * rlanes + rl; row < n; row += (int64_t)gridDim.x * rlanes) {
      const int64_t off = row * LD + 4 * c4;
      if (MODE == 0) {
        const float4 v = ld4(a + off);
        s0.x += v.x, s0.y += v.y, s0.z += v.z, s0.w += v.w;
        s1.x += v.x * v.x, s1.y += v.y * v.y, s1.z += v.z * v.z, s1.w += v.w * v.w;
      } else if (MODE == 2) {
        // four rows per trip: the dependent (index -> pooled gradient) loads of all four are in
        // flight together, otherwise this pass is latency-bound
        const int64_t stride = (int64_t)gridDim.x * rlanes;
        int64_t rows[4];
        int par[4];
        float4 xs[4], gs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          rows[u] = row + u * stride;
          par[u] = in2out[rows[u] < n ? rows[u] : row];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t r = rows[u] < n ? rows[u] : row;
          xs[u] = ldx4<B16>(b, r * LD + 4 * c4);
          gs[u] = ld4(a + (int64_t)par[u] * LD + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (rows[u] < n) {
            const float4 x = xs[u];
            float4 g = gs[u];
            const float4 xh = make_float4((x.x - mu.x) * is.x, (x.y - mu.y) * is.y, (x.z - mu.z) * is.z, (x.w - mu.w) * is.w);
            g.x = xh.x * ga.x + be.x > 0.f ? g.x : 0.f, g.y = xh.y * ga.y + be.y > 0.f ? g.y : 0.f;
            g.z = xh.z * ga.z + be.z > 0.f ? g.z : 0.f, g.w = xh.w * ga.w + be.w > 0.f ? g.w : 0.f;
            s0.x += g.x, s0.y += g.y, s0.z += g.z, s0.w += g.w;
            s1.x += g.x * xh.x, s1.y += g.y * xh.y, s1.z += g.z * xh.z, s1.w += g.w * xh.w;
          }
        }
        row += 3 * stride;  // the loop header adds the fourth stride
      } else {
        float4 g = ld4(a + off);
        const float4 x = ld4(b + off);
        if (yrelu) {
          const float4 y = ld4(yrelu + off);
          g.x = y.x > 0.f ? g.x : 0.f, g.y = y.y > 0.f ? g.y : 0.f, g.z = y.z > 0.f ? g.z : 0.f,
          g.w = y.w > 0.f ? g.w : 0.f;
        }
        bn_bwd_accumulate(g, x, mu, is, s0, s1);
      }
    }
    double *d = s_red + ((int64_t)rl * 2) * C + 4 * c4;
    d[0] = s0.x, d[1] = s0.y, d[2] = s0.z, d[3] = s0.w;
    d[C + 0] = s1.x, d[C + 1] = s1.y, d[C + 2] = s1.z, d[C + 3] = s1.w;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * C; e += EB) {
    double s = 0.0;
    for (int r = 0; r < rlanes; ++r) s += s_red[(int64_t)r * 2 * C + e];
    partial[(int64_t)blockIdx.x * 2 * LD + (e < C ? e : LD + e - C)] = s;
  }
}

// colreduce_kernel<1> whose gradient operand is still in the split-K slabs of the data-gradient convolution that produced it:
// g = slab 0 + slab 1 + ... (slab order: the sum splitk_reduce_kernel would have written), stored to `g_sum` on the way -- the
// reduce launch and one pass over the gradient disappear from the chain (8 per ResNet14 step, 28 per ResNet34 step); `addend`
// (optional): the gradient of the residual branch, added behind the slabs -- the identity blocks' add launch goes too.  Same thread
// -> row mapping and the same accumulation as colreduce_kernel<1>: bit-identical statistics.  C <= 1024.
__global__ __launch_bounds__(EB) void colreduce_slabs_kernel(const float *__restrict__ slabs, int nslab, const float *__restrict__ addend,
                                                             float *__restrict__ g_sum, const float *__restrict__ b, const float *__restrict__ yrelu, int64_t n, int C,
                                                             const float *__restrict__ mean, const float *__restrict__ invstd,
                                                             double *__restrict__ partial) {
  extern __shared__ double s_red[];
  const int tpr = C >> 2, rlanes = EB / tpr;
  const int c4 = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const int64_t total = n * C;
  float4 s0 = make_float4(0, 0, 0, 0), s1 = make_float4(0, 0, 0, 0);
  if (rl < rlanes) {
    const float4 mu = ld4(mean + 4 * c4), is = ld4(invstd + 4 * c4);
    for (int64_t row = (int64_t)blockIdx.x * rlanes + rl; row < n; row += (int64_t)gridDim.x * rlanes) {
      const int64_t off = row * C + 4 * c4;
      const float4 x = ld4(b + off);
      float4 y = make_float4(1, 1, 1, 1);
      if (yrelu) y = ld4(yrelu + off);
      float4 g = ld4(slabs + off);
      int z = 1;
      for (; z + 3 < nslab; z += 4) {  // four slabs in flight, added in slab order
        const float4 t0 = ld4(slabs + (int64_t)z * total + off), t1 = ld4(slabs + (int64_t)(z + 1) * total + off),
                     t2 = ld4(slabs + (int64_t)(z + 2) * total + off), t3 = ld4(slabs + (int64_t)(z + 3) * total + off);
        g.x += t0.x, g.y += t0.y, g.z += t0.z, g.w += t0.w;
        g.x += t1.x, g.y += t1.y, g.z += t1.z, g.w += t1.w;
        g.x += t2.x, g.y += t2.y, g.z += t2.z, g.w += t2.w;
        g.x += t3.x, g.y += t3.y, g.z += t3.z, g.w += t3.w;
      }
      for (; z < nslab; ++z) {
        const float4 t = ld4(slabs + (int64_t)z * total + off);
        g.x += t.x, g.y += t.y, g.z += t.z, g.w += t.w;
      }
      if (addend) {  // (the residual branch's gradient: what the add launch behind the reduce would have added)
        const float4 t = ld4(addend + off);
        g.x += t.x, g.y += t.y, g.z += t.z, g.w += t.w;
      }
      st4(g_sum + off, g);
      g.x = y.x > 0.f ? g.x : 0.f, g.y = y.y > 0.f ? g.y : 0.f, g.z = y.z > 0.f ? g.z : 0.f, g.w = y.w > 0.f ? g.w : 0.f;
      bn_bwd_accumulate(g, x, mu, is, s0, s1);
    }
    double *d = s_red + ((int64_t)rl * 2) * C + 4 * c4;
    d[0] = s0.x, d[1] = s0.y, d[2] = s0.z, d[3] = s0.w;
    d[C + 0] = s1.x, d[C + 1] = s1.y, d[C + 2] = s1.z, d[C + 3] = s1.w;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * C; e += EB) {
    double t = 0.0;
    for (int r = 0; r < rlanes; ++r) t += s_red[(int64_t)r * 2 * C + e];
    partial[(int64_t)blockIdx.x * 2 * C + e] = t;
  }
}

// Sum the per-workgroup partials: one wave per channel (4 channels per workgroup), one lane per
// slice of workgroups, then a fixed-order butterfly over the wave; returns (sum0, sum1) of
// channel c to lane 0 of its wave.  All loads of a lane are independent, so the pass costs one
// memory round trip instead of nblk / 16 dependent ones.
constexpr int kFinCh = 4;  // channels per finalize workgroup
// WIDE: the whole workgroup sums ONE channel (thread t takes partial rows t, t + 256, ...; waves combined through LDS in
// wave order) -- for the 2048-row partials of the finest level, where a single wave per channel needs 32 dependent-issue
// loads per lane (26 us for the stem's batch-norm backward; 6 us this way).
template <bool WIDE = false>
__device__ __forceinline__ bool finalize_sums(const double *__restrict__ partial, int nblk, int C, int &c, double &s,
                                              double &ss) {
  const int lane = threadIdx.x & 63;
  if constexpr (WIDE) {
    __shared__ double s_fin[2][4];
    c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int blk = threadIdx.x; blk < nblk; blk += 256) a += partial[(int64_t)blk * 2 * C + c], b += partial[(int64_t)blk * 2 * C + C + c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64), b += __shfl_xor(b, off, 64);
    if (lane == 0) s_fin[0][threadIdx.x >> 6] = a, s_fin[1][threadIdx.x >> 6] = b;
    __syncthreads();
    s = ((s_fin[0][0] + s_fin[0][1]) + s_fin[0][2]) + s_fin[0][3];
    ss = ((s_fin[1][0] + s_fin[1][1]) + s_fin[1][2]) + s_fin[1][3];
    return threadIdx.x == 0;
  }
  c = blockIdx.x * kFinCh + (threadIdx.x >> 6);
  if (c >= C) return false;  // whole wave
  double a = 0.0, b = 0.0;
  for (int blk = lane; blk < nblk; blk += 64) a += partial[(int64_t)blk * 2 * C + c], b += partial[(int64_t)blk * 2 * C + C + c];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64), b += __shfl_xor(b, off, 64);
  s = a, ss = b;
  return lane == 0;
}
constexpr int kWideFinalizeRows = 512;  // partial rows from which the wide form is used

__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const double *__restrict__ partial, int nblk, int64_t n,
                                                                int C, float eps, float momentum,
                                                                float *__restrict__ mean, float *__restrict__ invstd,
                                                                float *running_mean, float *running_var) {
  int c;
  double s, ss;
  if (!finalize_sums(partial, nblk, C, c, s, ss)) return;
  double m;
  const double var = bn_variance(s, ss, n, m);  // (no contraction: bn_apply_fold_kernel must reproduce this bit for bit)
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) bn_running_update(running_mean, running_var, c, momentum, (float)m, var, n);
}

template <bool WIDE>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double *__restrict__ partial, int nblk, int C,
                                                              const float *__restrict__ gamma,
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta) {
  int c;
  double s, ss;
  if (!finalize_sums<WIDE>(partial, nblk, C, c, s, ss)) return;
  dbeta[c] = (float)s;
  dgamma[c] = (float)ss;
}

// ---- split form used by SyncBatchNorm: partials -> sums[2][C] (double), all-reduced by the host
__global__ __launch_bounds__(256) void bn_sum_partials_kernel(const double *__restrict__ partial, int nblk, int C,
                                                              double *__restrict__ sums) {
  int c;
  double s, ss;
  if (!finalize_sums(partial, nblk, C, c, s, ss)) return;
  sums[c] = s;
  sums[C + c] = ss;
}

__global__ void bn_stats_from_sums_kernel(const double *__restrict__ sums, const double *__restrict__ n_total, int C,
                                          float eps, float momentum, float *__restrict__ mean,
                                          float *__restrict__ invstd, float *running_mean, float *running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double n = *n_total;
  const double m = sums[c] / n;
  double var = sums[C + c] / n - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// sums (double, possibly all-reduced) -> float dbeta / dgamma scaled for the apply pass
__global__ void bn_bwd_sums_to_float_kernel(const double *__restrict__ sums, const double *__restrict__ n_total, int C,
                                            float *__restrict__ dbeta_mean, float *__restrict__ dgamma_mean) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta_mean[c] = (float)(sums[c] / *n_total);
  dgamma_mean[c] = (float)(sums[C + c] / *n_total);
}

// ------------------------------------------------------------------ few-row layers: a whole batch norm in ONE launch
// Below ~1 k rows (the deepest stages; every stage of Mink-ResNet34 at four scenes per GPU) a batch norm is three dependent
// launches of 5-9 us each (column partials, finalize, apply) for a few hundred KB of data: launch latency, not bytes.  Here a
// workgroup owns a slab of 16 CHANNELS and ALL rows, so the column statistics never leave the workgroup:
//   forward : y = sum of the convolution's split-K slabs (in slab order) -> statistics -> out = act(bn(y) [+ residual])
//   backward: (sum g, sum g xhat) -> dgamma, dbeta -> dx, dresidual
// 1024 threads = 256 row lanes x 4 lanes of 16 bytes: a thread owns rows rl, rl + 256, rl + 512, rl + 768 -- ONE trip, every
// load of a pass independent (two slabs x four rows in flight, double-buffered: the pass costs ~nslab / 2 memory round trips,
// not rows x nslab / 64 as a 256-thread form did: 40-80 us).  The second pass re-reads what the same thread wrote or read in
// the first (L1 / L2 hits).  Sums: fp32 per thread over its <= 4 rows, double across the 256 row lanes in lane order
// (bitwise reproducible; NOT the summation order of the three-launch form).
constexpr int kSmallT = 1024;     // threads per workgroup
constexpr int kSmallRows = 1024;  // largest row count taken (mink_bn_small_rows)

// CH channels per workgroup (16: 4 lanes x 16 bytes per row, 256 row lanes x 4 rows; 8: 2 lanes, 512 row lanes x 2 rows --
// twice the workgroups: a single CU takes in ~25 GB/s of slabs that other XCDs wrote, 16 workgroups were 20 us for 0.5 MB)
template <int CH>
struct SmallCfg {
  static constexpr int LANES = CH / 4, RL = kSmallT / LANES, NU = kSmallRows / RL, NQ = 8 / NU, QN = 2 * CH, NP = 256 / QN;
};

template <int CH>
__device__ __forceinline__ void small_reduce(float4 s0, float4 s1, int c4, int rl, double *s_red, double *s_part, double *tot) {
  using K = SmallCfg<CH>;
  // s_red [RL row lanes][2][CH] -> s_part [NP][QN] -> tot [2][CH]
  double *d = s_red + rl * K::QN + 4 * c4;
  d[0] = s0.x, d[1] = s0.y, d[2] = s0.z, d[3] = s0.w;
  d[CH] = s1.x, d[CH + 1] = s1.y, d[CH + 2] = s1.z, d[CH + 3] = s1.w;
  __syncthreads();
  if (threadIdx.x < 256) {
    const int q = threadIdx.x % K::QN, p = threadIdx.x / K::QN;
    constexpr int per = K::RL / K::NP;
    double t = 0.0;
    for (int r = 0; r < per; ++r) t += s_red[(p * per + r) * K::QN + q];
    s_part[p * K::QN + q] = t;
  }
  __syncthreads();
  if (threadIdx.x < K::QN) {
    double t = 0.0;
    for (int p = 0; p < K::NP; ++p) t += s_part[p * K::QN + threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
}

template <int CH>
__global__ __launch_bounds__(kSmallT) void bn_small_fwd_kernel(const float *__restrict__ ws, int nslab, int64_t n, int C, float *y,
                                                               float eps, float momentum, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta, const float *__restrict__ residual,
                                                               int relu, float *__restrict__ out, float *__restrict__ mean,
                                                               float *__restrict__ invstd, float *running_mean, float *running_var) {
  using K = SmallCfg<CH>;
  constexpr int NU = K::NU, NQ = K::NQ;
  extern __shared__ double s_dyn[];  // [RL][QN] + [NP][QN] + [QN]
  double *s_red = s_dyn, *s_part = s_dyn + K::RL * K::QN, *s_tot = s_part + 256;
  __shared__ __attribute__((aligned(16))) float s_mu[CH], s_is[CH];
  const int c4 = threadIdx.x % K::LANES, rl = threadIdx.x / K::LANES;
  const int c = blockIdx.x * CH + 4 * c4;
  const int64_t total = n * C;
  int64_t off[NU];
  bool live[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int64_t r = rl + K::RL * u;
    live[u] = r < n;
    off[u] = (live[u] ? r : 0) * C + c;
  }
  float4 v[NU];
  if (nslab == 0) {
#pragma unroll
    for (int u = 0; u < NU; ++u) v[u] = ld4(y + off[u]);
  } else {
    // NQ slabs at a time, the next group requested before the current one is added (slab order kept)
    float4 t[2][NQ][NU];
    auto fetch = [&](int buf, int z) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int u = 0; u < NU; ++u) t[buf][q][u] = ld4(ws + (int64_t)(z + q < nslab ? z + q : z) * total + off[u]);
    };
    auto add = [&](int buf, int z) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (z + q < nslab) {
#pragma unroll
          for (int u = 0; u < NU; ++u) v[u].x += t[buf][q][u].x, v[u].y += t[buf][q][u].y, v[u].z += t[buf][q][u].z, v[u].w += t[buf][q][u].w;
        }
    };
#pragma unroll
    for (int u = 0; u < NU; ++u) v[u] = make_float4(0, 0, 0, 0);
    fetch(0, 0);
    for (int z = 0; z < nslab; z += 2 * NQ) {
      if (z + NQ < nslab) fetch(1, z + NQ);
      add(0, z);
      if (z + 2 * NQ < nslab) fetch(0, z + 2 * NQ);
      add(1, z + NQ);
    }
  }
  float4 s0 = make_float4(0, 0, 0, 0), s1 = s0;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (live[u]) {
      if (nslab) st4(y + off[u], v[u]);
      s0.x += v[u].x, s0.y += v[u].y, s0.z += v[u].z, s0.w += v[u].w;
      s1.x += v[u].x * v[u].x, s1.y += v[u].y * v[u].y, s1.z += v[u].z * v[u].z, s1.w += v[u].w * v[u].w;
    }
  }
  float4 rs[NU];  // requested now: in flight across the reduction
#pragma unroll
  for (int u = 0; u < NU; ++u) rs[u] = residual ? ld4(residual + off[u]) : make_float4(0, 0, 0, 0);
  small_reduce<CH>(s0, s1, c4, rl, s_red, s_part, s_tot);
  if (threadIdx.x < CH) {
    const int cc = blockIdx.x * CH + threadIdx.x;
    const double m = s_tot[threadIdx.x] / (double)n;
    double var = s_tot[CH + threadIdx.x] / (double)n - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)eps));
    s_mu[threadIdx.x] = mf, s_is[threadIdx.x] = isf;
    mean[cc] = mf, invstd[cc] = isf;
    if (running_mean) {
      const double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
      running_mean[cc] = (1.f - momentum) * running_mean[cc] + momentum * mf;
      running_var[cc] = (1.f - momentum) * running_var[cc] + momentum * (float)unbiased;
    }
  }
  __syncthreads();
  const float4 mu = ld4(s_mu + 4 * c4), is = ld4(s_is + 4 * c4), g = ld4(gamma + c), b = ld4(beta + c);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (live[u]) {
      float4 o;
      o.x = (v[u].x - mu.x) * is.x * g.x + b.x, o.y = (v[u].y - mu.y) * is.y * g.y + b.y;
      o.z = (v[u].z - mu.z) * is.z * g.z + b.z, o.w = (v[u].w - mu.w) * is.w * g.w + b.w;
      o.x += rs[u].x, o.y += rs[u].y, o.z += rs[u].z, o.w += rs[u].w;
      if (relu) o.x = fmaxf(o.x, 0.f), o.y = fmaxf(o.y, 0.f), o.z = fmaxf(o.z, 0.f), o.w = fmaxf(o.w, 0.f);
      st4(out + off[u], o);
    }
  }
}

// nslab > 0: the incoming gradient is the sum of `nslab` split-K slabs of the data-gradient convolution that produced it
// ([nslab][n][C] at `dy`); the sum is written to `dy_sum`
template <int CH>
__global__ __launch_bounds__(kSmallT) void bn_small_bwd_kernel(const float *__restrict__ dy, int nslab, const float *__restrict__ addend,
                                                               float *dy_sum, const float *__restrict__ x, const float *__restrict__ yrelu, int64_t n,
                                                               int C, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                               const float *__restrict__ gamma, float *__restrict__ dx,
                                                               float *__restrict__ dres, float *__restrict__ dgamma,
                                                               float *__restrict__ dbeta) {
  using K = SmallCfg<CH>;
  constexpr int NU = K::NU, NQ = K::NQ;
  extern __shared__ double s_dyn[];
  double *s_red = s_dyn, *s_part = s_dyn + K::RL * K::QN, *s_tot = s_part + 256;
  const int c4 = threadIdx.x % K::LANES, rl = threadIdx.x / K::LANES;
  const int c = blockIdx.x * CH + 4 * c4;
  const int64_t total = n * C;
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c);
  int64_t off[NU];
  bool live[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int64_t r = rl + K::RL * u;
    live[u] = r < n;
    off[u] = (live[u] ? r : 0) * C + c;
  }
  float4 g[NU], xv[NU], yv[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    g[u] = ld4(dy + off[u]);
    xv[u] = ld4(x + off[u]);
    yv[u] = yrelu ? ld4(yrelu + off[u]) : make_float4(1, 1, 1, 1);
  }
  for (int z = 1; z < nslab; z += NQ) {  // NQ slabs x NU rows in flight
    float4 t[NQ][NU];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int u = 0; u < NU; ++u) t[q][u] = ld4(dy + (int64_t)(z + q < nslab ? z + q : z) * total + off[u]);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (z + q < nslab) {
#pragma unroll
        for (int u = 0; u < NU; ++u) g[u].x += t[q][u].x, g[u].y += t[q][u].y, g[u].z += t[q][u].z, g[u].w += t[q][u].w;
      }
  }
  if (addend) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const float4 t = ld4(addend + off[u]);
      g[u].x += t.x, g[u].y += t.y, g[u].z += t.z, g[u].w += t.w;
    }
  }
  float4 s0 = make_float4(0, 0, 0, 0), s1 = s0;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (nslab && live[u]) st4(dy_sum + off[u], g[u]);
    float4 &q = g[u];
    q.x = yv[u].x > 0.f ? q.x : 0.f, q.y = yv[u].y > 0.f ? q.y : 0.f, q.z = yv[u].z > 0.f ? q.z : 0.f, q.w = yv[u].w > 0.f ? q.w : 0.f;
    if (live[u]) {
      s0.x += q.x, s0.y += q.y, s0.z += q.z, s0.w += q.w;
      s1.x += q.x * (xv[u].x - mu.x) * is.x, s1.y += q.y * (xv[u].y - mu.y) * is.y, s1.z += q.z * (xv[u].z - mu.z) * is.z,
          s1.w += q.w * (xv[u].w - mu.w) * is.w;
    }
  }
  small_reduce<CH>(s0, s1, c4, rl, s_red, s_part, s_tot);
  if (threadIdx.x < CH) {
    const int cc = blockIdx.x * CH + threadIdx.x;
    dbeta[cc] = (float)s_tot[threadIdx.x];
    dgamma[cc] = (float)s_tot[CH + threadIdx.x];
  }
  const float inv_n = 1.f / (float)n;
  const float4 db = make_float4((float)s_tot[4 * c4], (float)s_tot[4 * c4 + 1], (float)s_tot[4 * c4 + 2], (float)s_tot[4 * c4 + 3]);
  const float4 dg = make_float4((float)s_tot[CH + 4 * c4], (float)s_tot[CH + 1 + 4 * c4], (float)s_tot[CH + 2 + 4 * c4], (float)s_tot[CH + 3 + 4 * c4]);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (live[u]) {
      const float4 q = g[u];
      if (dres) st4(dres + off[u], q);
      float4 o;
      o.x = ga.x * is.x * (q.x - db.x * inv_n - (xv[u].x - mu.x) * is.x * dg.x * inv_n);
      o.y = ga.y * is.y * (q.y - db.y * inv_n - (xv[u].y - mu.y) * is.y * dg.y * inv_n);
      o.z = ga.z * is.z * (q.z - db.z * inv_n - (xv[u].z - mu.z) * is.z * dg.z * inv_n);
      o.w = ga.w * is.w * (q.w - db.w * inv_n - (xv[u].w - mu.w) * is.w * dg.w * inv_n);
      st4(dx + off[u], o);
    }
  }
}

// y = [relu]( (x-mean)*invstd*gamma + beta [+ residual] )
__global__ __launch_bounds__(EB) void bn_apply_kernel(const float *__restrict__ x, int64_t n4, int C4,
                                                      const float *__restrict__ mean, const float *__restrict__ invstd,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                      const float *__restrict__ residual, int relu,
                                                      float *__restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EB) {
    const int c = (int)(i % C4) * 4;
    const float4 v = ld4(x + 4 * i), mu = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
    float4 o = bn_affine(v, mu, is, g, b);
    if (residual) {
      const float4 r = ld4(residual + 4 * i);
      o.x += r.x, o.y += r.y, o.z += r.z, o.w += r.w;
    }
    if (relu) o.x = fmaxf(o.x, 0.f), o.y = fmaxf(o.y, 0.f), o.z = fmaxf(o.z, 0.f), o.w = fmaxf(o.w, 0.f);
    st4(y + 4 * i, o);
  }
}

// dx = gamma*invstd*(g - dbeta/n - xhat*dgamma/n);  dresidual = g
__global__ __launch_bounds__(EB) void bn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                          const float *__restrict__ yrelu, int64_t n4, int C4,
                                                          float inv_n, const float *__restrict__ mean,
                                                          const float *__restrict__ invstd,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ dgamma,
                                                          const float *__restrict__ dbeta, float *__restrict__ dx,
                                                          float *__restrict__ dres) {
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EB) {
    const int c = (int)(i % C4) * 4;
    float4 g = ld4(dy + 4 * i);
    if (yrelu) {
      const float4 y = ld4(yrelu + 4 * i);
      g.x = y.x > 0.f ? g.x : 0.f, g.y = y.y > 0.f ? g.y : 0.f, g.z = y.z > 0.f ? g.z : 0.f, g.w = y.w > 0.f ? g.w : 0.f;
    }
    if (dres) st4(dres + 4 * i, g);
    const float4 v = ld4(x + 4 * i), mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c),
                 dg = ld4(dgamma + c), db = ld4(dbeta + c);
    st4(dx + 4 * i, bn_dx(g, v, mu, is, ga, dg, db, inv_n));
  }
}

// ------------------------------------------------------------------ finalize folded into the consumer
// mean / invstd (or dgamma / dbeta) from <= kFoldRows partial rows, computed by EVERY workgroup of the apply pass for the 64
// channels it covers instead of by a launch of its own (5-7 us + a kernel boundary on a chain of ~10 us kernels).  The sums
// are those of finalize_sums<false>, bit for bit: lane l of a wave there adds rows l, l + 64 in order, then the butterfly
// over offsets 32, 16, ..., 1 leaves lane 0 with the balanced tree  s1[j] = v[j] + v[j+32], s2[j] = s1[j] + s1[j+16], ...,
// s6 = s5[0] + s5[1].  Here a lane holds the 32 row sums v[j] of one parity (j = par, par + 2, ...) of one (quantity,
// channel) column -- 32 independent 8-byte loads -- walks the same tree down to s5[par] and takes s5[par ^ 1] from its
// partner: one shuffle instead of 768.  Wave w covers channels c0 + 16 w .. + 15 (32 columns x 2 parities = 64 lanes).
constexpr int kFoldRows = 128;
__device__ __forceinline__ void fold_sums(const double *__restrict__ partial, int nblk, int C, int c0, double *s_out /* [2][64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, par = lane >> 5, qty = col >> 4, ch = 16 * wave + (col & 15);
  const double *p = partial + (int64_t)qty * C + c0 + ch;
  const int64_t rs = 2 * (int64_t)C;
  double v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int j = par + 2 * i;
    double a = 0.0;
    if (j < nblk) a += p[j * rs];
    v[i] = a;
  }
  if (nblk > 64) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int j = par + 2 * i + 64;
      if (j < nblk) v[i] += p[j * rs];
    }
  }
#pragma unroll
  for (int h = 16; h >= 1; h >>= 1)
#pragma unroll
    for (int i = 0; i < h; ++i) v[i] = v[i] + v[i + h];
  const double t = v[0] + __shfl_xor(v[0], 32, 64);
  if (par == 0) s_out[qty * 64 + ch] = t;
}

// bn_stats_finalize_kernel + bn_apply_kernel in one launch; grid (row chunks, C / 64)
__global__ __launch_bounds__(EB) void bn_apply_fold_kernel(const float *__restrict__ x, int64_t n, int C,
                                                           const double *__restrict__ partial, int nblk, float eps, float momentum,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ residual, int relu, float *__restrict__ y,
                                                           float *__restrict__ mean, float *__restrict__ invstd, float *running_mean,
                                                           float *running_var) {
  __shared__ double s_sum[2 * 64];
  __shared__ __attribute__((aligned(16))) float s_mu[64], s_is[64];
  const int c0 = blockIdx.y * 64;
  fold_sums(partial, nblk, C, c0, s_sum);
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = c0 + threadIdx.x;
    double m;
    const double var = bn_variance(s_sum[threadIdx.x], s_sum[64 + threadIdx.x], n, m);
    const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)eps));
    s_mu[threadIdx.x] = mf, s_is[threadIdx.x] = isf;
    if (blockIdx.x == 0) {
      mean[c] = mf, invstd[c] = isf;
      if (running_mean) bn_running_update(running_mean, running_var, c, momentum, mf, var, n);
    }
  }
  __syncthreads();
  const int c4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = c0 + 4 * c4;
  const float4 mu = ld4(s_mu + 4 * c4), is = ld4(s_is + 4 * c4), g = ld4(gamma + c), b = ld4(beta + c);
  for (int64_t row = (int64_t)blockIdx.x * 16 + rl; row < n; row += (int64_t)gridDim.x * 16) {
    const int64_t o_ = row * C + c;
    float4 o = bn_affine(ld4(x + o_), mu, is, g, b);
    if (residual) {
      const float4 r = ld4(residual + o_);
      o.x += r.x, o.y += r.y, o.z += r.z, o.w += r.w;
    }
    if (relu) o.x = fmaxf(o.x, 0.f), o.y = fmaxf(o.y, 0.f), o.z = fmaxf(o.z, 0.f), o.w = fmaxf(o.w, 0.f);
    st4(y + o_, o);
  }
}

// bn_bwd_finalize_kernel<false> + bn_bwd_apply_kernel in one launch; grid (row chunks, C / 64)
__global__ __launch_bounds__(EB) void bn_bwd_apply_fold_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                               const float *__restrict__ yrelu, int64_t n, int C, float inv_n,
                                                               const double *__restrict__ partial, int nblk,
                                                               const float *__restrict__ mean, const float *__restrict__ invstd,
                                                               const float *__restrict__ gamma, float *__restrict__ dgamma,
                                                               float *__restrict__ dbeta, float *__restrict__ dx,
                                                               float *__restrict__ dres) {
  __shared__ double s_sum[2 * 64];
  __shared__ __attribute__((aligned(16))) float s_db[64], s_dg[64];
  const int c0 = blockIdx.y * 64;
  fold_sums(partial, nblk, C, c0, s_sum);
  __syncthreads();
  if (threadIdx.x < 64) {
    const float db = (float)s_sum[threadIdx.x], dg = (float)s_sum[64 + threadIdx.x];
    s_db[threadIdx.x] = db, s_dg[threadIdx.x] = dg;
    if (blockIdx.x == 0) dbeta[c0 + threadIdx.x] = db, dgamma[c0 + threadIdx.x] = dg;
  }
  __syncthreads();
  const int c4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = c0 + 4 * c4;
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c), dg = ld4(s_dg + 4 * c4), db = ld4(s_db + 4 * c4);
  for (int64_t row = (int64_t)blockIdx.x * 16 + rl; row < n; row += (int64_t)gridDim.x * 16) {
    const int64_t o_ = row * C + c;
    float4 g = ld4(dy + o_);
    if (yrelu) {
      const float4 yv = ld4(yrelu + o_);
      g.x = yv.x > 0.f ? g.x : 0.f, g.y = yv.y > 0.f ? g.y : 0.f, g.z = yv.z > 0.f ? g.z : 0.f, g.w = yv.w > 0.f ? g.w : 0.f;
    }
    if (dres) st4(dres + o_, g);
    const float4 v = ld4(x + o_);
    st4(dx + o_, bn_dx(g, v, mu, is, ga, dg, db, inv_n));
  }
}

// ------------------------------------------------------------------ SGD with momentum over flat buffers
// torch.optim.SGD's update (the reference's optimizer: co3d_3d/src/modules/optim.py:12-14, configs/co3d_cls.gin) for parameters,
// gradients and momentum buffers that each live in ONE flat fp32 buffer of the same layout (parallel.BucketedGradAllReduce):
//   g' = g + wd w;  m = mu m + g'  (a zero buffer makes the first step m = g', as torch's clone does);  w = w - lr m
// one pass instead of torch's multi-tensor chunks (ResNet34: 5 launches, 314 us for 85 MB of parameters), and the gradient
// buffer is cleared on the way out (the memset of the next step's zero_grad).
__global__ __launch_bounds__(EB) void sgd_flat_kernel(float *__restrict__ w, float *__restrict__ g, float *__restrict__ m, int64_t n4,
                                                      float lr, float mu, float wd, int zero_grad) {
#pragma clang fp contract(off)  // (the products and sums of torch's own kernel, one rounding each)
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EB) {
    float4 wv = ld4(w + 4 * i), gv = ld4(g + 4 * i), mv = ld4(m + 4 * i);
    gv.x += wd * wv.x, gv.y += wd * wv.y, gv.z += wd * wv.z, gv.w += wd * wv.w;
    mv.x = mu * mv.x + gv.x, mv.y = mu * mv.y + gv.y, mv.z = mu * mv.z + gv.z, mv.w = mu * mv.w + gv.w;
    wv.x -= lr * mv.x, wv.y -= lr * mv.y, wv.z -= lr * mv.z, wv.w -= lr * mv.w;
    st4(w + 4 * i, wv), st4(m + 4 * i, mv);
    if (zero_grad) st4(g + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

// mode 0: y = max(a,0); mode 1: y = b>0 ? a : 0; mode 2: y = a + b
__global__ __launch_bounds__(EB) void eltwise_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                     int64_t count, int mode, float *__restrict__ y) {
  const int64_t n4 = count >> 2;
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EB) {
    const float4 va = ld4(a + 4 * i);
    float4 o;
    if (mode == 0) {
      o = make_float4(fmaxf(va.x, 0.f), fmaxf(va.y, 0.f), fmaxf(va.z, 0.f), fmaxf(va.w, 0.f));
    } else {
      const float4 vb = ld4(b + 4 * i);
      if (mode == 1)
        o = make_float4(vb.x > 0.f ? va.x : 0.f, vb.y > 0.f ? va.y : 0.f, vb.z > 0.f ? va.z : 0.f,
                        vb.w > 0.f ? va.w : 0.f);
      else
        o = make_float4(va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w);
    }
    st4(y + 4 * i, o);
  }
  if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    y[i] = mode == 0 ? fmaxf(a[i], 0.f) : (mode == 1 ? (b[i] > 0.f ? a[i] : 0.f) : a[i] + b[i]);
  }
}

// Pointwise activations of the ME module surface beyond ReLU (reference modules/common.py:36-43 lists them at import):
// forward (gy == nullptr): out = f(x); backward: out = gy * f'(x).  `slope`: per-channel PReLU weights (C entries,
// or one shared entry when C == 1); channels are the fastest axis of the [rows, C] feature matrix.
enum { ACT_LEAKY = 1, ACT_ELU = 2, ACT_CELU = 3, ACT_SELU = 4, ACT_GELU = 5, ACT_PRELU = 6 };

template <bool BWD>
__device__ __forceinline__ float act_eval(int kind, float x, float alpha) {
  constexpr float kSeluAlpha = 1.6732632423543772f, kSeluScale = 1.0507009873554805f;
  switch (kind) {
    case ACT_LEAKY:
    case ACT_PRELU:
      return BWD ? (x > 0.f ? 1.f : alpha) : (x > 0.f ? x : alpha * x);
    case ACT_ELU:
      return BWD ? (x > 0.f ? 1.f : alpha * expf(x)) : (x > 0.f ? x : alpha * (expf(x) - 1.f));
    case ACT_CELU:
      return BWD ? (x > 0.f ? 1.f : expf(x / alpha)) : (x > 0.f ? x : alpha * (expf(x / alpha) - 1.f));
    case ACT_SELU:
      return BWD ? kSeluScale * (x > 0.f ? 1.f : kSeluAlpha * expf(x))
                 : kSeluScale * (x > 0.f ? x : kSeluAlpha * (expf(x) - 1.f));
    default: {  // ACT_GELU, exact (erf) form
      const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
      return BWD ? cdf + x * 0.3989422804014327f * expf(-0.5f * x * x) : x * cdf;
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(EB) void activation_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                        const float *__restrict__ slope, int C, int64_t count, int kind,
                                                        float alpha, float *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < count; i += (int64_t)gridDim.x * EB) {
    const float a = slope ? slope[C > 1 ? (int)(i % C) : 0] : alpha;
    const float v = act_eval<BWD>(kind, x[i], a);
    out[i] = BWD ? gy[i] * v : v;
  }
}

// ------------------------------------------------ fused bn + relu + sum-pool (stem tail)
// y[o] = sum_{i child of o} relu((x[i]-mean)*invstd*gamma+beta): the normalised [N,C] tensor of
// the finest level (the largest activation of the network) is never written to HBM.
template <bool B16>  // B16: x is stored as bf16
__global__ __launch_bounds__(EB) void bn_relu_pool_fwd_kernel(const float *__restrict__ x, int C4,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ invstd,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              const int *__restrict__ nbr, int64_t n_out, int K,
                                                              float *__restrict__ y) {
  const int64_t idx = (int64_t)blockIdx.x * EB + threadIdx.x;
  if (idx >= n_out * C4) return;
  const int64_t o = idx / C4;
  const int c = (int)(idx - o * C4) * 4;
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
  const float4 sc = make_float4(is.x * g.x, is.y * g.y, is.z * g.z, is.w * g.w);
  const float4 sh = make_float4(b.x - mu.x * sc.x, b.y - mu.y * sc.y, b.z - mu.z * sc.z, b.w - mu.w * sc.w);
  float4 s = make_float4(0, 0, 0, 0);
  auto add = [&](const float4 &v) {
    s.x += fmaxf((v.x - mu.x) * is.x * g.x + b.x, 0.f), s.y += fmaxf((v.y - mu.y) * is.y * g.y + b.y, 0.f);
    s.z += fmaxf((v.z - mu.z) * is.z * g.z + b.z, 0.f), s.w += fmaxf((v.w - mu.w) * is.w * g.w + b.w, 0.f);
  };
  if (K == 8) {  // (uniform) the 2^3 pooling of the stem: the eight children's rows are requested together -- one memory
                 // round trip per thread instead of eight dependent index -> row chains (75 -> 5x us at B=16)
    int ch[8];  // (table slices of a prepared batch start on 4-byte boundaries: no wider loads)
#pragma unroll
    for (int k = 0; k < 8; ++k) ch[k] = nbr[o * 8 + k];
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = ldx4<B16>(x, (int64_t)max(ch[k], 0) * (4 * C4) + c);  // (a missing child re-reads row 0)
#pragma unroll
    for (int k = 0; k < 8; ++k)  // same order of additions as the loop below
      if (ch[k] >= 0) add(v[k]);
  } else {
    for (int k = 0; k < K; ++k) {
      const int i = nbr[o * K + k];
      if (i >= 0) add(ldx4<B16>(x, (int64_t)i * (4 * C4) + c));
    }
  }
  (void)sh;
  st4(y + o * (int64_t)(4 * C4) + c, s);
}

// dx[i] = gamma*invstd*(g - dbeta/n - xhat*dgamma/n), g = dy_pool[in2out[i]] * (gamma*xhat+beta > 0)
__global__ __launch_bounds__(EB) void bn_relu_pool_bwd_kernel(const float *__restrict__ dyp,
                                                              const float *__restrict__ x,
                                                              const int *__restrict__ in2out, int64_t n, int C4,
                                                              float inv_n, const float *__restrict__ mean,
                                                              const float *__restrict__ invstd,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              const float *__restrict__ dgamma,
                                                              const float *__restrict__ dbeta,
                                                              float *__restrict__ dx) {
  for (int64_t i4 = (int64_t)blockIdx.x * EB + threadIdx.x; i4 < n * C4; i4 += (int64_t)gridDim.x * EB) {
    const int64_t row = i4 / C4;
    const int c = (int)(i4 - row * C4) * 4;
    const float4 v = ld4(x + 4 * i4), mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c), be = ld4(beta + c),
                 dg = ld4(dgamma + c), db = ld4(dbeta + c);
    float4 g = ld4(dyp + (int64_t)in2out[row] * (4 * C4) + c);
    const float4 xh = make_float4((v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w);
    g.x = xh.x * ga.x + be.x > 0.f ? g.x : 0.f, g.y = xh.y * ga.y + be.y > 0.f ? g.y : 0.f;
    g.z = xh.z * ga.z + be.z > 0.f ? g.z : 0.f, g.w = xh.w * ga.w + be.w > 0.f ? g.w : 0.f;
    float4 o;
    o.x = ga.x * is.x * (g.x - db.x * inv_n - xh.x * dg.x * inv_n);
    o.y = ga.y * is.y * (g.y - db.y * inv_n - xh.y * dg.y * inv_n);
    o.z = ga.z * is.z * (g.z - db.z * inv_n - xh.z * dg.z * inv_n);
    o.w = ga.w * is.w * (g.w - db.w * inv_n - xh.w * dg.w * inv_n);
    st4(dx + 4 * i4, o);
  }
}

// ------------------------------------------------------------------------- pooling
__global__ __launch_bounds__(EB) void pool_sum_fwd_kernel(const float *__restrict__ x, int ldx, int C4,
                                                          const int *__restrict__ nbr, int64_t n_out, int K,
                                                          float *__restrict__ y) {
  const int64_t idx = (int64_t)blockIdx.x * EB + threadIdx.x;
  if (idx >= n_out * C4) return;
  const int64_t o = idx / C4;
  const int c = (int)(idx - o * C4) * 4;
  float4 s = make_float4(0, 0, 0, 0);
  for (int k = 0; k < K; ++k) {
    const int i = nbr[o * K + k];
    if (i >= 0) {
      const float4 v = ld4(x + (int64_t)i * ldx + c);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  }
  st4(y + o * (int64_t)(4 * C4) + c, s);
}

__global__ __launch_bounds__(EB) void pool_sum_bwd_kernel(const float *__restrict__ dy, int C4,
                                                          const int *__restrict__ in2out, int64_t n_in,
                                                          float *__restrict__ dx) {
  const int64_t idx = (int64_t)blockIdx.x * EB + threadIdx.x;
  if (idx >= n_in * C4) return;
  const int64_t i = idx / C4;
  const int c = (int)(idx - i * C4) * 4;
  st4(dx + i * (int64_t)(4 * C4) + c, ld4(dy + (int64_t)in2out[i] * (4 * C4) + c));
}

// Max pooling over a neighbour table (overlapping windows allowed: the dense 2-D baseline's 3x3 stride-2 pooling).
// arg[o][c] = input row holding the maximum (the first one in offset order, as torch's window scan picks it).
__global__ __launch_bounds__(EB) void pool_max_fwd_kernel(const float *__restrict__ x, int C4, const int *__restrict__ nbr,
                                                          int64_t n_out, int K, float *__restrict__ y, int *__restrict__ arg) {
  const int64_t idx = (int64_t)blockIdx.x * EB + threadIdx.x;
  if (idx >= n_out * C4) return;
  const int64_t o = idx / C4;
  const int c = (int)(idx - o * C4) * 4;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  int4 a = make_int4(-1, -1, -1, -1);
  for (int k = 0; k < K; ++k) {
    const int i = nbr[o * K + k];
    if (i >= 0) {
      const float4 v = ld4(x + (int64_t)i * (4 * C4) + c);
      if (v.x > m.x || a.x < 0) m.x = v.x, a.x = i;  // (a NaN never wins, like torch keeps the first element then)
      if (v.y > m.y || a.y < 0) m.y = v.y, a.y = i;
      if (v.z > m.z || a.z < 0) m.z = v.z, a.z = i;
      if (v.w > m.w || a.w < 0) m.w = v.w, a.w = i;
    }
  }
  st4(y + o * (int64_t)(4 * C4) + c, m);
  *reinterpret_cast<int4 *>(arg + o * (int64_t)(4 * C4) + c) = a;
}

// dx[i][c] = sum of dy[o][c] over the windows o that contain i and chose it (gathered through the transposed table:
// no atomics, one owner per element)
__global__ __launch_bounds__(EB) void pool_max_bwd_kernel(const float *__restrict__ dy, const int *__restrict__ arg, int C4,
                                                          const int *__restrict__ nbr_t, int64_t n_in, int K,
                                                          float *__restrict__ dx) {
  const int64_t idx = (int64_t)blockIdx.x * EB + threadIdx.x;
  if (idx >= n_in * C4) return;
  const int64_t i = idx / C4;
  const int c = (int)(idx - i * C4) * 4;
  float4 s = make_float4(0, 0, 0, 0);
  for (int k = 0; k < K; ++k) {
    const int o = nbr_t[i * K + k];
    if (o >= 0) {
      const int4 a = *reinterpret_cast<const int4 *>(arg + (int64_t)o * (4 * C4) + c);
      const float4 g = ld4(dy + (int64_t)o * (4 * C4) + c);
      s.x += a.x == (int)i ? g.x : 0.f, s.y += a.y == (int)i ? g.y : 0.f;
      s.z += a.z == (int)i ? g.z : 0.f, s.w += a.w == (int)i ? g.w : 0.f;
    }
  }
  st4(dx + i * (int64_t)(4 * C4) + c, s);
}

// one workgroup per (batch, 64-channel slab): 16 float4 columns x 16 row lanes
__global__ __launch_bounds__(EB) void global_avg_fwd_kernel(const float *__restrict__ x, int C,
                                                            const int *__restrict__ boff, float *__restrict__ y) {
  __shared__ float4 s_red[EB];
  const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 15) * 4, rl = threadIdx.x >> 4;
  const int beg = boff[b], end = boff[b + 1];
  float4 s = make_float4(0, 0, 0, 0);
  if (c < C)
    for (int r = beg + rl; r < end; r += 16) {
      const float4 v = ld4(x + (int64_t)r * C + c);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  s_red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    for (int r = 1; r < 16; ++r) {
      const float4 v = s_red[r * 16 + (threadIdx.x & 15)];
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    const float inv = end > beg ? 1.f / (float)(end - beg) : 0.f;
    st4(y + (int64_t)b * C + c, make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv));
  }
}

__global__ __launch_bounds__(EB) void global_avg_bwd_kernel(const float *__restrict__ dy, int C4,
                                                            const int *__restrict__ boff, int B, int64_t n,
                                                            float *__restrict__ dx) {
  const int64_t idx = (int64_t)blockIdx.x * EB + threadIdx.x;
  if (idx >= n * C4) return;
  const int64_t i = idx / C4;
  const int c = (int)(idx - i * C4) * 4;
  int lo = 0, hi = B;  // largest b with boff[b] <= i
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (boff[mid] <= i) lo = mid;
    else hi = mid;
  }
  const float inv = 1.f / (float)(boff[lo + 1] - boff[lo]);
  const float4 g = ld4(dy + (int64_t)lo * (4 * C4) + c);
  st4(dx + i * (int64_t)(4 * C4) + c, make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv));
}

__global__ __launch_bounds__(EB) void segment_mean_kernel(const float *__restrict__ x, int ldx, int C,
                                                          const int *__restrict__ members,
                                                          const int *__restrict__ seg, int64_t n_out,
                                                          float *__restrict__ y) {
  const int64_t idx = (int64_t)blockIdx.x * EB + threadIdx.x;
  if (idx >= n_out * C) return;
  const int64_t u = idx / C;
  const int c = (int)(idx - u * C);
  const int beg = seg[u], end = seg[u + 1];
  float s = 0.f;
  for (int j = beg; j < end; ++j) s += x[(int64_t)members[j] * ldx + c];  // input-row order
  y[idx] = s / (float)(end - beg);
}

// Workgroups of a column reduction over n rows with `rlanes` row lanes per workgroup: eight rows per thread where the rows are
// many, but at least (up to) 256 workgroups -- below ~8 k rows eight rows per thread left a few dozen workgroups walking their
// rows one dependent round trip after the other (ResNet34 at four scenes: 23 us per pass).  One rule for every
// caller: the partial rows, and with them the statistics' summation order, are the same on every path.
static inline int64_t colreduce_blocks(int64_t n, int rlanes) {
  int64_t nblk = cdiv(n, (int64_t)rlanes * 8);
  const int64_t wide = std::min<int64_t>(cdiv(n, (int64_t)rlanes), 256);
  if (nblk < wide) nblk = wide;
  if (nblk > 2048) nblk = 2048;
  return nblk < 1 ? 1 : nblk;
}

static inline unsigned ew_grid(int64_t work) {
  int64_t g = cdiv(work, EB);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace mink

using namespace mink;

#define REQ_C4(C, name) MINK_REQUIRE((C) >= 4 && ((C)&3) == 0, name ": channel count %d must be a multiple of 4", (C))
#define REQ_A16(ptr, name) MINK_REQUIRE(((uintptr_t)(ptr)&15) == 0, name ": pointer must be 16-byte aligned")

extern "C" {

static void launch_bwd_finalize(const double *partial, int nblk, int C, const float *gamma, float *dgamma, float *dbeta,
                                hipStream_t st) {
  if (nblk >= kWideFinalizeRows)
    bn_bwd_finalize_kernel<true><<<dim3((unsigned)C), 256, 0, st>>>(partial, nblk, C, gamma, dgamma, dbeta);
  else
    bn_bwd_finalize_kernel<false><<<dim3((unsigned)cdiv(C, kFinCh)), 256, 0, st>>>(partial, nblk, C, gamma, dgamma, dbeta);
}

int64_t mink_bn_workspace_bytes(int64_t n, int32_t C) { return (int64_t)kRedBlocks * 2 * C * sizeof(double); }

static int launch_colreduce(int mode, const float *a, const float *b, const float *yrelu, int64_t n, int C,
                            const float *mean, const float *invstd, double *partial, hipStream_t st, int *nblk_out) {
  int slabs = 1;  // slabs of at most 4 * EB channels
  while (slabs <= C && (C % slabs != 0 || C / slabs > 4 * EB || ((C / slabs) & 3) != 0)) ++slabs;
  MINK_REQUIRE(slabs <= C, "bn: %d channels cannot be cut into slabs of at most %d", C, 4 * EB);
  const int Cs = C / slabs, tpr = Cs >> 2;
  const int rlanes = EB / tpr;
  int64_t nblk = colreduce_blocks(n, rlanes);
  if (nblk > kRedBlocks) nblk = kRedBlocks;
  if (nblk < 1) nblk = 1;
  const size_t shm = (size_t)rlanes * 2 * Cs * sizeof(double);
  const dim3 grid((unsigned)nblk, (unsigned)slabs);
  if (mode == 0)
    colreduce_kernel<0><<<grid, EB, shm, st>>>(a, b, yrelu, n, Cs, mean, invstd, partial, nullptr, nullptr, nullptr, C);
  else
    colreduce_kernel<1><<<grid, EB, shm, st>>>(a, b, yrelu, n, Cs, mean, invstd, partial, nullptr, nullptr, nullptr, C);
  MINK_CHECK_LAUNCH();
  *nblk_out = (int)nblk;
  return MINK_OK;
}

int mink_bn_stats(const float *x, int64_t n, int32_t C, float eps, float momentum, float *mean, float *invstd,
                  float *running_mean, float *running_var, void *workspace, int64_t workspace_bytes, void *stream) {
  REQ_C4(C, "bn_stats");
  MINK_REQUIRE(workspace_bytes >= mink_bn_workspace_bytes(n, C), "bn_stats: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
               (long long)(mink_bn_workspace_bytes(n, C)));
  MINK_REQUIRE(n >= 1 && x && mean && invstd && workspace, "bn_stats: bad arguments (n=%lld)", (long long)n);
  MINK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats: running stats must come in pairs");
  REQ_A16(x, "bn_stats");
  hipStream_t st = (hipStream_t)stream;
  int nblk = 0;
  int rc = launch_colreduce(0, x, nullptr, nullptr, n, C, nullptr, nullptr, (double *)workspace, st, &nblk);
  if (rc) return rc;
  bn_stats_finalize_kernel<<<dim3((unsigned)cdiv(C, kFinCh)), 256, 0, st>>>((const double *)workspace, nblk, n, C, eps,
                                                                       momentum, mean, invstd, running_mean,
                                                                       running_var);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_bn_stats_from_partials(const double *partial, int32_t rows, int64_t n, int32_t C, float eps, float momentum,
                                float *mean, float *invstd, float *running_mean, float *running_var, void *stream) {
  MINK_REQUIRE(partial && rows >= 1 && n >= 1 && C >= 1 && mean && invstd, "bn_stats_from_partials: bad arguments");
  MINK_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
               "bn_stats_from_partials: running stats must come in pairs");
  bn_stats_finalize_kernel<<<dim3((unsigned)cdiv(C, kFinCh)), 256, 0, (hipStream_t)stream>>>(
      partial, rows, n, C, eps, momentum, mean, invstd, running_mean, running_var);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_bn_apply(const float *x, int64_t n, int32_t C, const float *mean, const float *invstd, const float *gamma,
                  const float *beta, const float *residual, int32_t relu, float *y, void *stream) {
  REQ_C4(C, "bn_apply");
  MINK_REQUIRE(n >= 0, "bn_apply: bad n");
  if (n == 0) return MINK_OK;
  MINK_REQUIRE(x && mean && invstd && gamma && beta && y, "bn_apply: NULL pointer");
  REQ_A16(x, "bn_apply");
  REQ_A16(y, "bn_apply");
  const int64_t n4 = n * (C >> 2);
  bn_apply_kernel<<<dim3(ew_grid(n4)), EB, 0, (hipStream_t)stream>>>(x, n4, C >> 2, mean, invstd, gamma, beta, residual,
                                                                    relu, y);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

// Fold the finalize into the apply pass?  Every workgroup of that pass re-reads rows x 1 KB of partials (64 channels x 2 x 8
// bytes): fine for the short passes of the deep layers, ruinous at layer 1 of a large batch (579 workgroups x 128 KB = 74 MB of
// L2 reads for a 19 MB pass: the step got 5 % slower).  So: few rows AND at most ~8 MB of partial re-reads in total.
static bool fold_ok(int rows, int64_t n, int C, unsigned &gx) {
  if (rows > g_bn_fold || (C & 63) != 0) return false;
  gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 64), 512 / (C / 64)));
  return g_bn_fold_force || (int64_t)rows * gx * (C / 64) <= 8192;
}

int mink_bn_set_fold(int32_t max_rows) {
  const int old = g_bn_fold + (g_bn_fold_force ? 1000 : 0);
  g_bn_fold_force = max_rows >= 1000;  // (1000 + r: up to r rows WITHOUT the total-bytes rule -- the bitwise tests of the folded kernels)
  if (g_bn_fold_force) max_rows -= 1000;
  g_bn_fold = max_rows < 0 ? 0 : (max_rows > kFoldRows ? kFoldRows : max_rows);
  return old;
}

int mink_bn_apply_from_partials(const float *x, int64_t n, int32_t C, const double *partial, int32_t rows, float eps, float momentum,
                                const float *gamma, const float *beta, const float *residual, int32_t relu, float *y, float *mean,
                                float *invstd, float *running_mean, float *running_var, void *stream) {
  REQ_C4(C, "bn_apply_from_partials");
  MINK_REQUIRE(partial && rows >= 1 && n >= 1 && x && gamma && beta && y && mean && invstd, "bn_apply_from_partials: bad arguments");
  MINK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_apply_from_partials: running stats must come in pairs");
  REQ_A16(x, "bn_apply_from_partials");
  REQ_A16(y, "bn_apply_from_partials");
  REQ_A16(residual, "bn_apply_from_partials");
  unsigned gx = 0;
  if (fold_ok(rows, n, C, gx)) {
    bn_apply_fold_kernel<<<dim3(gx, (unsigned)(C / 64)), EB, 0, (hipStream_t)stream>>>(x, n, C, partial, rows, eps, momentum, gamma, beta, residual,
                                                                                  relu, y, mean, invstd, running_mean, running_var);
    MINK_CHECK_LAUNCH();
    return MINK_OK;
  }
  int rc = mink_bn_stats_from_partials(partial, rows, n, C, eps, momentum, mean, invstd, running_mean, running_var, stream);
  if (rc) return rc;
  return mink_bn_apply(x, n, C, mean, invstd, gamma, beta, residual, relu, y, stream);
}

int mink_bn_fwd(const float *x, int64_t n, int32_t C, float eps, float momentum, const float *gamma, const float *beta,
                const float *residual, int32_t relu, float *y, float *mean, float *invstd, float *running_mean,
                float *running_var, void *workspace, int64_t workspace_bytes, void *stream) {
  REQ_C4(C, "bn_fwd");
  MINK_REQUIRE(n >= 1 && x && gamma && beta && y && mean && invstd, "bn_fwd: bad arguments (n=%lld)", (long long)n);
  MINK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_fwd: running stats must come in pairs");
  REQ_A16(x, "bn_fwd");
  REQ_A16(y, "bn_fwd");
  int rc = mink_bn_stats(x, n, C, eps, momentum, mean, invstd, running_mean, running_var, workspace, workspace_bytes, stream);
  if (rc) return rc;
  return mink_bn_apply(x, n, C, mean, invstd, gamma, beta, residual, relu, y, stream);
}

int mink_bn_bwd(const float *dy, const float *x, const float *y, int64_t n, int32_t C, const float *mean,
                const float *invstd, const float *gamma, int32_t relu, float *dx, float *dresidual, float *dgamma,
                float *dbeta, void *workspace, int64_t workspace_bytes, void *stream) {
  REQ_C4(C, "bn_bwd");
  MINK_REQUIRE(workspace_bytes >= mink_bn_workspace_bytes(n, C), "bn_bwd: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
               (long long)(mink_bn_workspace_bytes(n, C)));
  MINK_REQUIRE(n >= 1 && dy && x && mean && invstd && gamma && dx && dgamma && dbeta && workspace,
               "bn_bwd: bad arguments");
  MINK_REQUIRE(!relu || y, "bn_bwd: fused ReLU needs the forward output");
  REQ_A16(dy, "bn_bwd");
  REQ_A16(x, "bn_bwd");
  REQ_A16(dx, "bn_bwd");
  hipStream_t st = (hipStream_t)stream;
  const float *yr = relu ? y : nullptr;
  int nblk = 0;
  int rc = launch_colreduce(1, dy, x, yr, n, C, mean, invstd, (double *)workspace, st, &nblk);
  if (rc) return rc;
  unsigned gx = 0;
  if (fold_ok(nblk, n, C, gx)) {  // the finalize runs inside the apply pass
    bn_bwd_apply_fold_kernel<<<dim3(gx, (unsigned)(C / 64)), EB, 0, st>>>(dy, x, yr, n, C, 1.f / (float)n, (const double *)workspace, nblk,
                                                                         mean, invstd, gamma, dgamma, dbeta, dx, dresidual);
    MINK_CHECK_LAUNCH();
    return MINK_OK;
  }
  launch_bwd_finalize((const double *)workspace, nblk, C, gamma, dgamma, dbeta, st);
  MINK_CHECK_LAUNCH();
  const int64_t n4 = n * (C >> 2);
  bn_bwd_apply_kernel<<<dim3(ew_grid(n4)), EB, 0, st>>>(dy, x, yr, n4, C >> 2, 1.f / (float)n, mean, invstd, gamma,
                                                       dgamma, dbeta, dx, dresidual);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

constexpr int kSmallSmem = (1024 / 4 * 32 + 256 + 32) * (int)sizeof(double);  // RL x QN is 8192 doubles for both channel widths
static int g_small_ch = 0;  // 0: by channel count (8 below 512 channels); 8 / 16: forced (mink_bn_set_small(8 | 16))
static bool small_attrs() {
  bool ok = true;
  ok &= hipFuncSetAttribute(reinterpret_cast<const void *>(&bn_small_fwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSmem) == hipSuccess;
  ok &= hipFuncSetAttribute(reinterpret_cast<const void *>(&bn_small_bwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSmem) == hipSuccess;
  ok &= hipFuncSetAttribute(reinterpret_cast<const void *>(&bn_small_fwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSmem) == hipSuccess;
  ok &= hipFuncSetAttribute(reinterpret_cast<const void *>(&bn_small_bwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSmem) == hipSuccess;
  return ok;
}
static int small_width(int C) { return g_small_ch ? g_small_ch : (C >= 1024 ? 16 : 8); }

// 0 when the device refuses the kernels' 67,840 bytes of LDS per workgroup (a part or context with 64 KB): the callers'
// small_layer() test then never picks the one-launch path and the three-launch norm runs instead of failing
int32_t mink_bn_small_rows(void) {
  if (!g_bn_small) return 0;
  static const bool attr_ok = small_attrs();
  return attr_ok ? kSmallRows : 0;
}

int mink_bn_set_small(int32_t on) {
  const int old = g_bn_small ? (g_small_ch ? g_small_ch : 1) : 0;
  g_bn_small = on != 0;
  g_small_ch = (on == 8 || on == 16) ? on : 0;
  return old;
}

int mink_bn_small_fwd(const float *slabs, int32_t nslab, int64_t n, int32_t C, float *y, float eps, float momentum,
                      const float *gamma, const float *beta, const float *residual, int32_t relu, float *out, float *mean,
                      float *invstd, float *running_mean, float *running_var, void *stream) {
  MINK_REQUIRE(n >= 1 && n <= kSmallRows && C >= 16 && C % 16 == 0, "bn_small_fwd: %lld rows x %d channels not supported",
               (long long)n, C);
  MINK_REQUIRE(y && gamma && beta && out && mean && invstd && nslab >= 0 && (nslab == 0 || slabs), "bn_small_fwd: bad arguments");
  MINK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_small_fwd: running stats must come in pairs");
  REQ_A16(y, "bn_small_fwd");
  REQ_A16(out, "bn_small_fwd");
  REQ_A16(slabs, "bn_small_fwd");
  REQ_A16(residual, "bn_small_fwd");
  static const bool attr_ok = small_attrs();
  MINK_REQUIRE(attr_ok, "bn_small: %d bytes of LDS per workgroup refused", kSmallSmem);
  if (small_width(C) == 16)
    bn_small_fwd_kernel<16><<<dim3((unsigned)(C / 16)), kSmallT, kSmallSmem, (hipStream_t)stream>>>(
        slabs, nslab, n, C, y, eps, momentum, gamma, beta, residual, relu, out, mean, invstd, running_mean, running_var);
  else
    bn_small_fwd_kernel<8><<<dim3((unsigned)(C / 8)), kSmallT, kSmallSmem, (hipStream_t)stream>>>(
        slabs, nslab, n, C, y, eps, momentum, gamma, beta, residual, relu, out, mean, invstd, running_mean, running_var);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_bn_small_bwd(const float *dy, int32_t nslab, const float *addend, float *dy_sum, const float *x, const float *y, int64_t n, int32_t C,
                      const float *mean, const float *invstd, const float *gamma, int32_t relu, float *dx, float *dresidual,
                      float *dgamma, float *dbeta, void *stream) {
  MINK_REQUIRE(n >= 1 && n <= kSmallRows && C >= 16 && C % 16 == 0, "bn_small_bwd: %lld rows x %d channels not supported",
               (long long)n, C);
  MINK_REQUIRE(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && nslab >= 0 && (nslab == 0 || dy_sum) && (!addend || nslab > 0),
               "bn_small_bwd: bad arguments");
  REQ_A16(addend, "bn_small_bwd");
  MINK_REQUIRE(!relu || y, "bn_small_bwd: fused ReLU needs the forward output");
  REQ_A16(dy, "bn_small_bwd");
  REQ_A16(x, "bn_small_bwd");
  REQ_A16(dx, "bn_small_bwd");
  REQ_A16(dy_sum, "bn_small_bwd");
  REQ_A16(dresidual, "bn_small_bwd");
  static const bool attr_ok = small_attrs();
  MINK_REQUIRE(attr_ok, "bn_small: %d bytes of LDS per workgroup refused", kSmallSmem);
  if (small_width(C) == 16)
    bn_small_bwd_kernel<16><<<dim3((unsigned)(C / 16)), kSmallT, kSmallSmem, (hipStream_t)stream>>>(
        dy, nslab, addend, dy_sum, x, relu ? y : nullptr, n, C, mean, invstd, gamma, dx, dresidual, dgamma, dbeta);
  else
    bn_small_bwd_kernel<8><<<dim3((unsigned)(C / 8)), kSmallT, kSmallSmem, (hipStream_t)stream>>>(
        dy, nslab, addend, dy_sum, x, relu ? y : nullptr, n, C, mean, invstd, gamma, dx, dresidual, dgamma, dbeta);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_bn_bwd_slabs(const float *dy_slabs, int32_t nslab, const float *addend, float *dy_sum, const float *x, const float *y, int64_t n, int32_t C,
                      const float *mean, const float *invstd, const float *gamma, int32_t relu, float *dx, float *dresidual,
                      float *dgamma, float *dbeta, void *workspace, int64_t workspace_bytes, void *stream) {
  REQ_C4(C, "bn_bwd_slabs");
  MINK_REQUIRE(workspace_bytes >= mink_bn_workspace_bytes(n, C), "bn_bwd_slabs: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
               (long long)(mink_bn_workspace_bytes(n, C)));
  MINK_REQUIRE(n >= 1 && nslab >= 1 && C <= 4 * EB && dy_slabs && dy_sum && x && mean && invstd && gamma && dx && dgamma && dbeta && workspace,
               "bn_bwd_slabs: bad arguments");
  MINK_REQUIRE(!relu || y, "bn_bwd_slabs: fused ReLU needs the forward output");
  REQ_A16(dy_slabs, "bn_bwd_slabs");
  REQ_A16(dy_sum, "bn_bwd_slabs");
  REQ_A16(x, "bn_bwd_slabs");
  REQ_A16(dx, "bn_bwd_slabs");
  hipStream_t st = (hipStream_t)stream;
  const float *yr = relu ? y : nullptr;
  // the launch shape of launch_colreduce (one slab of channels: C <= 1024), so that the statistics are those of mink_bn_bwd
  const int tpr = C >> 2, rlanes = EB / tpr;
  int64_t nblk = colreduce_blocks(n, rlanes);
  if (nblk > kRedBlocks) nblk = kRedBlocks;
  if (nblk < 1) nblk = 1;
  REQ_A16(addend, "bn_bwd_slabs");
  colreduce_slabs_kernel<<<dim3((unsigned)nblk), EB, (size_t)rlanes * 2 * C * sizeof(double), st>>>(dy_slabs, nslab, addend, dy_sum, x, yr, n, C,
                                                                                                  mean, invstd, (double *)workspace);
  MINK_CHECK_LAUNCH();
  unsigned gx = 0;
  if (fold_ok((int)nblk, n, C, gx)) {
    bn_bwd_apply_fold_kernel<<<dim3(gx, (unsigned)(C / 64)), EB, 0, st>>>(dy_sum, x, yr, n, C, 1.f / (float)n, (const double *)workspace, (int)nblk,
                                                                         mean, invstd, gamma, dgamma, dbeta, dx, dresidual);
    MINK_CHECK_LAUNCH();
    return MINK_OK;
  }
  launch_bwd_finalize((const double *)workspace, (int)nblk, C, gamma, dgamma, dbeta, st);
  MINK_CHECK_LAUNCH();
  const int64_t n4 = n * (C >> 2);
  bn_bwd_apply_kernel<<<dim3(ew_grid(n4)), EB, 0, st>>>(dy_sum, x, yr, n4, C >> 2, 1.f / (float)n, mean, invstd, gamma, dgamma, dbeta, dx, dresidual);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_bn_reduce(int32_t mode, const float *a, const float *b, const float *y, int64_t n, int32_t C, const float *mean,
                   const float *invstd, double *sums, void *workspace, int64_t workspace_bytes, void *stream) {
  REQ_C4(C, "bn_reduce");
  MINK_REQUIRE(workspace_bytes >= mink_bn_workspace_bytes(n, C), "bn_reduce: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
               (long long)(mink_bn_workspace_bytes(n, C)));
  MINK_REQUIRE((mode == 0 || mode == 1) && n >= 1 && a && sums && workspace, "bn_reduce: bad arguments");
  MINK_REQUIRE(mode == 0 || (b && mean && invstd), "bn_reduce: mode 1 needs x, mean and invstd");
  REQ_A16(a, "bn_reduce");
  hipStream_t st = (hipStream_t)stream;
  int nblk = 0;
  int rc = launch_colreduce(mode, a, b, y, n, C, mean, invstd, (double *)workspace, st, &nblk);
  if (rc) return rc;
  bn_sum_partials_kernel<<<dim3((unsigned)cdiv(C, kFinCh)), 256, 0, st>>>((const double *)workspace, nblk, C, sums);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_bn_stats_from_sums(const double *sums, const double *n_total, int32_t C, float eps, float momentum, float *mean,
                            float *invstd, float *running_mean, float *running_var, void *stream) {
  MINK_REQUIRE(C >= 1 && sums && n_total && mean && invstd, "bn_stats_from_sums: bad arguments");
  bn_stats_from_sums_kernel<<<dim3((unsigned)cdiv(C, 64)), 64, 0, (hipStream_t)stream>>>(sums, n_total, C, eps, momentum, mean,
                                                                                       invstd, running_mean, running_var);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_bn_bwd_from_sums(const float *dy, const float *x, const float *y, int64_t n, int32_t C, const double *sums,
                          const double *n_total, const float *mean, const float *invstd, const float *gamma,
                          int32_t relu, float *dx, float *dresidual, float *scratch2c, void *stream) {
  REQ_C4(C, "bn_bwd_from_sums");
  MINK_REQUIRE(n >= 1 && dy && x && sums && n_total && mean && invstd && gamma && dx && scratch2c,
               "bn_bwd_from_sums: bad arguments");
  MINK_REQUIRE(!relu || y, "bn_bwd_from_sums: fused ReLU needs the forward output");
  hipStream_t st = (hipStream_t)stream;
  // dbeta/n and dgamma/n as floats; the apply kernel is then called with inv_n = 1
  bn_bwd_sums_to_float_kernel<<<dim3((unsigned)cdiv(C, 64)), 64, 0, st>>>(sums, n_total, C, scratch2c, scratch2c + C);
  MINK_CHECK_LAUNCH();
  const int64_t n4 = n * (C >> 2);
  bn_bwd_apply_kernel<<<dim3(ew_grid(n4)), EB, 0, st>>>(dy, x, relu ? y : nullptr, n4, C >> 2, 1.f, mean, invstd, gamma,
                                                       scratch2c + C, scratch2c, dx, dresidual);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

static int bn_relu_pool_fwd_impl(const float *x, bool b16, int32_t C, const float *mean, const float *invstd, const float *gamma,
                                 const float *beta, const int32_t *nbr, int64_t n_out, int32_t K, float *y, void *stream) {
  REQ_C4(C, "bn_relu_pool_fwd");
  MINK_REQUIRE(n_out >= 0 && K >= 1, "bn_relu_pool_fwd: bad shape");
  if (n_out == 0) return MINK_OK;
  MINK_REQUIRE(x && mean && invstd && gamma && beta && nbr && y, "bn_relu_pool_fwd: NULL pointer");
  REQ_A16(x, "bn_relu_pool_fwd");
  REQ_A16(y, "bn_relu_pool_fwd");
  const dim3 grid((unsigned)cdiv(n_out * (C >> 2), EB));
  if (b16) bn_relu_pool_fwd_kernel<true><<<grid, EB, 0, (hipStream_t)stream>>>(x, C >> 2, mean, invstd, gamma, beta, nbr, n_out, K, y);
  else bn_relu_pool_fwd_kernel<false><<<grid, EB, 0, (hipStream_t)stream>>>(x, C >> 2, mean, invstd, gamma, beta, nbr, n_out, K, y);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_bn_relu_pool_fwd(const float *x, int32_t C, const float *mean, const float *invstd, const float *gamma,
                          const float *beta, const int32_t *nbr, int64_t n_out, int32_t K, float *y, void *stream) {
  return bn_relu_pool_fwd_impl(x, false, C, mean, invstd, gamma, beta, nbr, n_out, K, y, stream);
}

int mink_bn_relu_pool_fwd_b16(const void *xb, int32_t C, const float *mean, const float *invstd, const float *gamma,
                              const float *beta, const int32_t *nbr, int64_t n_out, int32_t K, float *y, void *stream) {
  return bn_relu_pool_fwd_impl((const float *)xb, true, C, mean, invstd, gamma, beta, nbr, n_out, K, y, stream);
}

static int bn_relu_pool_bwd_impl(const float *dy_pool, const float *x, bool b16, int64_t n, int32_t C, const float *mean,
                                 const float *invstd, const float *gamma, const float *beta, const int32_t *in2out, float *dx,
                                 float *dgamma, float *dbeta, void *workspace, int64_t workspace_bytes, void *stream) {
  REQ_C4(C, "bn_relu_pool_bwd");
  MINK_REQUIRE(!b16 || !dx, "bn_relu_pool_bwd_b16: parameter gradients only (dx must be NULL)");
  MINK_REQUIRE(workspace_bytes >= mink_bn_workspace_bytes(n, C), "bn_relu_pool_bwd: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
               (long long)(mink_bn_workspace_bytes(n, C)));
  MINK_REQUIRE(n >= 1 && dy_pool && x && mean && invstd && gamma && beta && in2out && dgamma && dbeta && workspace,
               "bn_relu_pool_bwd: bad arguments");
  REQ_A16(dy_pool, "bn_relu_pool_bwd");
  REQ_A16(x, "bn_relu_pool_bwd");
  REQ_A16(dx, "bn_relu_pool_bwd");  // dx == NULL: parameter gradients only (the consumer recomputes dx on the fly)
  hipStream_t st = (hipStream_t)stream;
  const int tpr = C >> 2;
  MINK_REQUIRE(tpr <= EB, "bn_relu_pool_bwd: too many channels");
  const int rlanes = EB / tpr;
  int64_t nblk = colreduce_blocks(n, rlanes);
  if (nblk > kRedBlocks) nblk = kRedBlocks;
  const size_t shm = (size_t)rlanes * 2 * C * sizeof(double);
  if (b16)
    colreduce_kernel<2, true><<<dim3((unsigned)nblk), EB, shm, st>>>(dy_pool, x, nullptr, n, C, mean, invstd, (double *)workspace,
                                                                      in2out, gamma, beta);
  else
    colreduce_kernel<2><<<dim3((unsigned)nblk), EB, shm, st>>>(dy_pool, x, nullptr, n, C, mean, invstd, (double *)workspace,
                                                              in2out, gamma, beta);
  MINK_CHECK_LAUNCH();
  launch_bwd_finalize((const double *)workspace, (int)nblk, C, gamma, dgamma, dbeta, st);
  MINK_CHECK_LAUNCH();
  if (!dx) return MINK_OK;
  const int64_t n4 = n * (C >> 2);
  bn_relu_pool_bwd_kernel<<<dim3(ew_grid(n4)), EB, 0, st>>>(dy_pool, x, in2out, n, C >> 2, 1.f / (float)n, mean, invstd,
                                                           gamma, beta, dgamma, dbeta, dx);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_bn_relu_pool_bwd(const float *dy_pool, const float *x, int64_t n, int32_t C, const float *mean,
                          const float *invstd, const float *gamma, const float *beta, const int32_t *in2out, float *dx,
                          float *dgamma, float *dbeta, void *workspace, int64_t workspace_bytes, void *stream) {
  return bn_relu_pool_bwd_impl(dy_pool, x, false, n, C, mean, invstd, gamma, beta, in2out, dx, dgamma, dbeta, workspace, workspace_bytes,
                               stream);
}

int mink_bn_relu_pool_bwd_b16(const float *dy_pool, const void *xb, int64_t n, int32_t C, const float *mean, const float *invstd,
                              const float *gamma, const float *beta, const int32_t *in2out, float *dgamma, float *dbeta,
                              void *workspace, int64_t workspace_bytes, void *stream) {
  return bn_relu_pool_bwd_impl(dy_pool, (const float *)xb, true, n, C, mean, invstd, gamma, beta, in2out, nullptr, dgamma, dbeta, workspace,
                               workspace_bytes, stream);
}

int mink_sgd_step(float *w, float *g, float *m, int64_t n, float lr, float momentum, float weight_decay, int32_t zero_grad, void *stream) {
  MINK_REQUIRE(w && g && m && n >= 0 && (n & 3) == 0, "sgd_step: bad arguments (n = %lld must be a multiple of 4)", (long long)n);
  REQ_A16(w, "sgd_step");
  REQ_A16(g, "sgd_step");
  REQ_A16(m, "sgd_step");
  if (n == 0) return MINK_OK;
  const int64_t n4 = n >> 2;
  sgd_flat_kernel<<<dim3((unsigned)std::min<int64_t>(cdiv(n4, EB), 8192)), EB, 0, (hipStream_t)stream>>>(w, g, m, n4, lr, momentum, weight_decay,
                                                                                                   zero_grad);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_eltwise(const float *a, const float *b, int64_t count, int32_t mode, float *y, void *stream) {
  MINK_REQUIRE(count >= 0 && mode >= 0 && mode <= 2, "eltwise: bad arguments");
  if (count == 0) return MINK_OK;
  MINK_REQUIRE(a && y && (mode == 0 || b), "eltwise: NULL pointer");
  REQ_A16(a, "eltwise");
  REQ_A16(y, "eltwise");
  eltwise_kernel<<<dim3(ew_grid(count >> 2)), EB, 0, (hipStream_t)stream>>>(a, b, count, mode, y);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_activation(const float *x, const float *gy, const float *slope, int32_t C, int64_t count, int32_t kind, float alpha,
                    float *out, void *stream) {
  MINK_REQUIRE(count >= 0 && kind >= ACT_LEAKY && kind <= ACT_PRELU && C >= 1, "activation: bad arguments (kind %d)", kind);
  MINK_REQUIRE(kind != ACT_PRELU || slope, "activation: PReLU needs its slope vector");
  MINK_REQUIRE(kind != ACT_CELU || alpha != 0.f, "activation: CELU alpha must be non-zero");
  if (count == 0) return MINK_OK;
  MINK_REQUIRE(x && out, "activation: NULL pointer");
  const float *sl = kind == ACT_PRELU ? slope : nullptr;
  const dim3 grid(ew_grid(count));
  if (gy) activation_kernel<true><<<grid, EB, 0, (hipStream_t)stream>>>(x, gy, sl, C, count, kind, alpha, out);
  else activation_kernel<false><<<grid, EB, 0, (hipStream_t)stream>>>(x, nullptr, sl, C, count, kind, alpha, out);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_pool_sum_fwd(const float *x, int32_t ldx, int32_t C, const int32_t *nbr, int64_t n_out, int32_t K, float *y,
                      void *stream) {
  REQ_C4(C, "pool_sum_fwd");
  MINK_REQUIRE(n_out >= 0 && K >= 1 && (ldx & 3) == 0 && ldx >= C, "pool_sum_fwd: bad shape");
  if (n_out == 0) return MINK_OK;
  MINK_REQUIRE(x && nbr && y, "pool_sum_fwd: NULL pointer");
  REQ_A16(x, "pool_sum_fwd");
  REQ_A16(y, "pool_sum_fwd");
  pool_sum_fwd_kernel<<<dim3((unsigned)cdiv(n_out * (C >> 2), EB)), EB, 0, (hipStream_t)stream>>>(x, ldx, C >> 2, nbr,
                                                                                                 n_out, K, y);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_pool_sum_bwd(const float *dy, int32_t C, const int32_t *in2out, int64_t n_in, float *dx, void *stream) {
  REQ_C4(C, "pool_sum_bwd");
  MINK_REQUIRE(n_in >= 0, "pool_sum_bwd: bad n");
  if (n_in == 0) return MINK_OK;
  MINK_REQUIRE(dy && in2out && dx, "pool_sum_bwd: NULL pointer");
  REQ_A16(dy, "pool_sum_bwd");
  REQ_A16(dx, "pool_sum_bwd");
  pool_sum_bwd_kernel<<<dim3((unsigned)cdiv(n_in * (C >> 2), EB)), EB, 0, (hipStream_t)stream>>>(dy, C >> 2, in2out,
                                                                                                n_in, dx);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_pool_max_fwd(const float *x, int32_t C, const int32_t *nbr, int64_t n_out, int32_t K, float *y, int32_t *arg,
                      void *stream) {
  REQ_C4(C, "pool_max_fwd");
  MINK_REQUIRE(n_out >= 0 && K >= 1, "pool_max_fwd: bad shape");
  if (n_out == 0) return MINK_OK;
  MINK_REQUIRE(x && nbr && y && arg, "pool_max_fwd: NULL pointer");
  REQ_A16(x, "pool_max_fwd");
  REQ_A16(y, "pool_max_fwd");
  REQ_A16(arg, "pool_max_fwd");
  pool_max_fwd_kernel<<<dim3((unsigned)cdiv(n_out * (C >> 2), EB)), EB, 0, (hipStream_t)stream>>>(x, C >> 2, nbr, n_out, K, y, arg);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_pool_max_bwd(const float *dy, const int32_t *arg, int32_t C, const int32_t *nbr_t, int64_t n_in, int32_t K, float *dx,
                      void *stream) {
  REQ_C4(C, "pool_max_bwd");
  MINK_REQUIRE(n_in >= 0 && K >= 1, "pool_max_bwd: bad shape");
  if (n_in == 0) return MINK_OK;
  MINK_REQUIRE(dy && arg && nbr_t && dx, "pool_max_bwd: NULL pointer");
  REQ_A16(dy, "pool_max_bwd");
  REQ_A16(dx, "pool_max_bwd");
  REQ_A16(arg, "pool_max_bwd");
  pool_max_bwd_kernel<<<dim3((unsigned)cdiv(n_in * (C >> 2), EB)), EB, 0, (hipStream_t)stream>>>(dy, arg, C >> 2, nbr_t, n_in, K, dx);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_global_avg_fwd(const float *x, int32_t C, const int32_t *batch_offsets, int32_t B, float *y, void *stream) {
  REQ_C4(C, "global_avg_fwd");
  MINK_REQUIRE(B >= 1 && x && batch_offsets && y, "global_avg_fwd: bad arguments");
  REQ_A16(x, "global_avg_fwd");
  REQ_A16(y, "global_avg_fwd");
  global_avg_fwd_kernel<<<dim3((unsigned)B, (unsigned)cdiv(C, 64)), EB, 0, (hipStream_t)stream>>>(x, C, batch_offsets, y);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_global_avg_bwd(const float *dy, int32_t C, const int32_t *batch_offsets, int32_t B, int64_t n, float *dx,
                        void *stream) {
  REQ_C4(C, "global_avg_bwd");
  MINK_REQUIRE(B >= 1 && n >= 0, "global_avg_bwd: bad arguments");
  if (n == 0) return MINK_OK;
  MINK_REQUIRE(dy && batch_offsets && dx, "global_avg_bwd: NULL pointer");
  REQ_A16(dy, "global_avg_bwd");
  REQ_A16(dx, "global_avg_bwd");
  global_avg_bwd_kernel<<<dim3((unsigned)cdiv(n * (C >> 2), EB)), EB, 0, (hipStream_t)stream>>>(dy, C >> 2,
                                                                                               batch_offsets, B, n, dx);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_segment_mean(const float *x, int32_t ldx, int32_t C, const int32_t *members, const int32_t *seg,
                      int64_t n_out, float *y, void *stream) {
  MINK_REQUIRE(C >= 1 && ldx >= C && n_out >= 0, "segment_mean: bad shape");
  if (n_out == 0) return MINK_OK;
  MINK_REQUIRE(x && members && seg && y, "segment_mean: NULL pointer");
  segment_mean_kernel<<<dim3((unsigned)cdiv(n_out * C, EB)), EB, 0, (hipStream_t)stream>>>(x, ldx, C, members, seg,
                                                                                          n_out, y);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

}  // extern "C"
