// Sparse convolution on gfx950: output-stationary gather -> LDS -> fp32 MFMA.
//
//   forward / dgrad : y[o] = sum_k x[nbr[o][k]] @ W[k]      (mink_conv_gather_gemm)
//   wgrad           : dW[k] = x[nbr[.][k]]^T @ dy            (mink_conv_wgrad)
//
// Both are implicit GEMMs over the neighbour table, accumulated in registers by
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD), with no atomics: every output
// element has exactly one owner, so results are bitwise reproducible run to run.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"

namespace mink {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128;    // output rows per workgroup (4 waves x 32 rows)
constexpr int BN = 64;     // output columns per workgroup (2 MFMA tiles per wave)
constexpr int BK = 32;     // reduction chunk (input channels) per stage
constexpr int LDA = BK + 4;  // LDS row stride of the A tile: conflict-free ds_read_b128
constexpr int KMAX = 27;

struct GemmParams {
  const float *x;
  const float *w;
  const int *nbr;
  const float *bias;
  float *y;
  float *ws;
  const int *row_perm;  // optional: tile row v computes output row row_perm[v] (-1 = padding)
  int64_t n_out;        // rows of y / of the neighbour table
  int64_t n_virtual;    // rows iterated (== n_out without a permutation)
  int ldx, cin, ldy, cout, K, flip_k, kper, stagger;
  int accumulate;  // y += result instead of y = result (un-split launches whose rows are visited at most once)
  float *stats;  // optional [row tiles][2][cout]: per-tile column (sum, sum of squares) of y (un-split launches only)
  int swz_x, swz_y, swz_z;  // compact_gemm_kernel: > 0 = XCD-aware one-dimensional launch over (row tiles, column tiles, slices)
  unsigned long long *trace;  // measurement only (mink_conv_trace; ABL instantiation): [workgroup][5] = clock at start / loop / epilogue / end, HW id
  unsigned sk_q, sk_r;  // compact_gemm_kernel<.., SK>: the F = (row tiles x column tiles) x K offset-tiles of the launch dealt out in runs of
                        // q = F / G (the first r = F % G workers: q + 1)
  int sk_X, sk_S;       // ... row tiles; slabs per tile in the workspace (>= the workers that can share a tile)
};

__device__ __forceinline__ float4 ld4_guard(const float *p, int valid, bool vec) {
  // valid = number of in-bounds floats at p (may be <= 0 or >= 4)
  if (valid >= 4 && vec) return *reinterpret_cast<const float4 *>(p);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid > 0) v.x = p[0];
  if (valid > 1) v.y = p[1];
  if (valid > 2) v.z = p[2];
  if (valid > 3) v.w = p[3];
  return v;
}

// Branch-free 16-byte load: `ok == false` reads element 0 of `base` (always mapped) and returns
// zeros.  Keeps every load of a staging pass independent so they are all in flight together
// (the guarded form above compiles to serialized load/wait branches).
__device__ __forceinline__ float4 ld4_sel(const float *base, int64_t off, bool ok) {
  const float4 v = *reinterpret_cast<const float4 *>(base + (ok ? off : 0));
  return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Same, but the zeroing is a bit mask on the loaded words: the optimiser cannot fold it back
// into an exec-masked (branchy) load, so the load stays an unconditional, schedulable instruction.
__device__ __forceinline__ float4 ld4_mask(const float *base, int64_t off, bool ok) {
  const uint4 u = *reinterpret_cast<const uint4 *>(base + (ok ? off : 0));
  const unsigned m = ok ? 0xFFFFFFFFu : 0u;
  return make_float4(__uint_as_float(u.x & m), __uint_as_float(u.y & m), __uint_as_float(u.z & m),
                     __uint_as_float(u.w & m));
}

// W_T = false: w[K][cin][cout];  W_T = true: w[K][cout][cin] (dgrad reads the forward kernel)
// VEC: every operand is 16-byte aligned with channel counts that are multiples of 4.
template <bool W_T, bool VEC>
__global__ __launch_bounds__(256) void gather_gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float sA[BM * LDA];
  __shared__ __attribute__((aligned(16))) float sB[BK * BN];
  __shared__ int s_nbr[BM * KMAX];
  __shared__ int s_orow[BM];
  __shared__ unsigned s_kmask;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t o0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int K = p.K;
  const int kbeg = blockIdx.z * p.kper;
  const int kend = min(K, kbeg + p.kper);

  // De-phase the workgroups that share a CU: identical programs started together run their
  // MFMA and their load/barrier phases in lockstep and leave the matrix pipe idle in between.
  if (p.stagger & 63) {
    const int ph = (blockIdx.x / 256) % 3;
    for (int i = 0; i < ph * (p.stagger & 63); ++i) __builtin_amdgcn_s_sleep(16);  // 16 * 64 clocks
  }
  // ---- stage this tile's slice of the neighbour table; find offsets with any neighbour
  if (tid == 0) s_kmask = 0u;
  if (tid < BM) {
    const int64_t v = o0 + tid;
    int orow = -1;
    if (v < p.n_virtual) orow = p.row_perm ? p.row_perm[v] : (int)v;
    s_orow[tid] = orow;
  }
  __syncthreads();
  {
    unsigned m = 0u;
    for (int e = tid; e < BM * K; e += 256) {
      const int lr = e / K, kk = e - lr * K;
      const int orow = s_orow[lr];
      const int v = orow >= 0 ? p.nbr[(int64_t)orow * K + kk] : -1;
      s_nbr[e] = v;
      if (v >= 0) m |= 1u << kk;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d);
    if (lane == 0 && m) atomicOr(&s_kmask, m);
  }
  __syncthreads();
  unsigned kmask = s_kmask;
  if (kend < 32) kmask &= (1u << kend) - 1u;
  kmask &= ~((1u << kbeg) - 1u);

  f32x16 acc0 = {0}, acc1 = {0};

  // per-thread staging coordinates
  const int a_cc = tid & 7, a_r = tid >> 3;       // A: rows a_r + 32 i, float4 column a_cc
  const int b_n4 = tid & 15, b_kk = tid >> 4;     // B (!W_T): rows b_kk + 16 i, float4 column b_n4
  const int bt_n = tid & 63, bt_k4 = tid >> 6;    // B (W_T) : column bt_n, float4 of k at 4*(bt_k4 + 4 i)

  float4 ra[4] = {}, rb[2] = {};

  auto load_chunk = [&](int k, int c0) {
    const int kw = p.flip_k ? (K - 1 - k) : k;
    if (p.stagger & 64) return;  // ablation: no global loads
    if (VEC) {
      const int c = c0 + 4 * a_cc;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int src = s_nbr[(a_r + 32 * i) * K + k];
        ra[i] = ld4_sel(p.x, (int64_t)src * p.ldx + c, src >= 0 && c < p.cin);
      }
      if (!W_T) {
        const int n = n0 + 4 * b_n4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int kk = c0 + b_kk + 16 * i;
          rb[i] = ld4_sel(p.w, ((int64_t)kw * p.cin + kk) * p.cout + n, kk < p.cin && n < p.cout);
        }
      } else {
        const int n = n0 + bt_n;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int kk = c0 + 4 * (bt_k4 + 4 * i);
          rb[i] = ld4_sel(p.w, ((int64_t)kw * p.cout + n) * p.cin + kk, n < p.cout && kk < p.cin);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = a_r + 32 * i;
      const int src = s_nbr[r * K + k];
      const int c = c0 + 4 * a_cc;
      ra[i] = src >= 0 ? ld4_guard(p.x + (int64_t)src * p.ldx + c, p.cin - c, false) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!W_T) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = c0 + b_kk + 16 * i;
        const int n = n0 + 4 * b_n4;
        rb[i] = kk < p.cin ? ld4_guard(p.w + ((int64_t)kw * p.cin + kk) * p.cout + n, p.cout - n, false)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = c0 + 4 * (bt_k4 + 4 * i);
        const int n = n0 + bt_n;
        rb[i] = n < p.cout ? ld4_guard(p.w + ((int64_t)kw * p.cout + n) * p.cin + kk, p.cin - kk, false)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };

  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(&sA[(a_r + 32 * i) * LDA + 4 * a_cc]) = ra[i];
    if (!W_T) {
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<float4 *>(&sB[(b_kk + 16 * i) * BN + 4 * b_n4]) = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = 4 * (bt_k4 + 4 * i);
        sB[(kk + 0) * BN + bt_n] = rb[i].x;
        sB[(kk + 1) * BN + bt_n] = rb[i].y;
        sB[(kk + 2) * BN + bt_n] = rb[i].z;
        sB[(kk + 3) * BN + bt_n] = rb[i].w;
      }
    }
  };

  int k = kmask ? __builtin_ctz(kmask) : -1;
  int c0 = 0;
  if (k >= 0) load_chunk(k, c0);
  const int arow = wave * 32 + (lane & 31), h = lane >> 5, col = lane & 31;
  while (k >= 0) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    // advance to the next (offset, channel chunk) and prefetch it into registers
    int nk = k, nc0 = c0 + BK;
    if (nc0 >= p.cin) {
      nc0 = 0;
      const unsigned rest = (k + 1 < 32) ? (kmask >> (k + 1)) : 0u;
      nk = rest ? (k + 1 + __builtin_ctz(rest)) : -1;
    }
    if (nk >= 0) load_chunk(nk, nc0);
    {
      float4 av[4];
      float b0[16], b1[16];
#pragma unroll
      for (int t = 0; t < 4; ++t) av[t] = *reinterpret_cast<const float4 *>(&sA[arow * LDA + 8 * t + 4 * h]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = 8 * t + 4 * h + j;
          b0[4 * t + j] = sB[kk * BN + col];
          b1[4 * t + j] = sB[kk * BN + 32 + col];
        }
      __builtin_amdgcn_sched_barrier(0);  // keep all LDS reads ahead of the MFMA chain
      if (!(p.stagger & 128))  // ablation: no MFMA
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float a4[4] = {av[t].x, av[t].y, av[t].z, av[t].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b0[4 * t + j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b1[4 * t + j], acc1, 0, 0, 0);
        }
      }
    }
    k = nk;
    c0 = nc0;
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool direct = gridDim.z == 1;
  float *dst = direct ? p.y : p.ws + (int64_t)blockIdx.z * p.n_out * p.cout;
  const int ldd = direct ? p.ldy : p.cout;
  const int c_a = n0 + col, c_b = n0 + 32 + col;
  const float bias_a = (direct && p.bias && c_a < p.cout) ? p.bias[c_a] : 0.f;
  const float bias_b = (direct && p.bias && c_b < p.cout) ? p.bias[c_b] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = s_orow[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
    if (row >= 0) {
      if (c_a < p.cout) dst[row * ldd + c_a] = acc0[r] + bias_a;
      if (c_b < p.cout) dst[row * ldd + c_b] = acc1[r] + bias_b;
    }
  }
}

// ------------------------------------------------------------------------ pipelined
// Software-pipelined gather-GEMM (the production kernel for 16-byte aligned operands).
// Per work item (kernel offset k, 32-channel chunk):
//     global loads   run THREE items ahead   (registers ga/gb)
//     LDS tiles      hold items c+1 and c+2  (double buffer, ONE raw s_barrier per item)
//     MFMA operands  are read from LDS ONE item ahead (register sets R0/R1)
// so the 32 MFMAs of an item never wait for a load issued in the same iteration, and the wave's
// own LDS reads / writes / global loads issue in the shadow of the 64-cycle MFMAs.
// STAGE: the tile's slice of the neighbour table is staged in LDS and offsets without any
// neighbour in the tile are skipped (essential for the class-permuted dgrad of strided convs);
// otherwise indices are read straight from the table one offset ahead.
#define MINK_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;
// raw buffer loads with a per-lane byte offset (VGPR) and a per-item byte offset (SGPR): no address arithmetic per load
__device__ f32x4 raw_load_v4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float raw_load_f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ int raw_load_i32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
// descriptor of a raw buffer of `bytes` bytes: a load at byte offset >= bytes touches no memory and returns 0
__device__ __forceinline__ i32x4 raw_rsrc(const void *ptr, unsigned bytes) {
  const unsigned long long a = (unsigned long long)ptr;
  i32x4 r;
  r.x = (int)(unsigned)a, r.y = (int)((a >> 32) & 0xFFFFu), r.z = (int)bytes, r.w = 0x00020000;
  return r;
}

// FLAT > 0 (== cin, not a multiple of 32; forward weights only): the reduction runs over the
// flattened (offset, channel) axis K*cin in 32-wide items that may straddle two offsets, so no
// MFMA step is spent on channel padding (stem: 756 = 27*28 -> 24 items instead of 27).
// MATH: 0 = exact fp32 (v_mfma_f32_32x32x2_f32); 1 = bf16 operands, fp32 accumulate
// (v_mfma_f32_32x32x16_bf16, 16x the matrix rate; BASELINE config "bf16 mixed precision");
// 3 = split-bf16: x = hi + lo, products hi*hi + hi*lo + lo*hi (~2^-17 relative per product).
// The tiles are converted when they are stored to LDS; HBM tensors stay fp32.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int LD16 = BK + 8;  // bf16 LDS row stride (80 bytes): conflict-free ds_read_b128

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) |
         ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}
__device__ __forceinline__ float bf16_residual(float a) { return a - (float)(__bf16)a; }
using bf16x8v = __attribute__((ext_vector_type(8))) __bf16;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8v pack_bits_bf16x8(const float (&v)[8]) {  // registers that already hold bf16 bits in their low halves
  auto lo = [&](int i) { return __float_as_uint(v[i]); };
  const uint4 u = make_uint4(lo(0) | (lo(1) << 16), lo(2) | (lo(3) << 16), lo(4) | (lo(5) << 16), lo(6) | (lo(7) << 16));
  return __builtin_bit_cast(bf16x8v, u);
}
__device__ __forceinline__ bf16x8v pack_bf16x8(const float (&v)[8]) {
  const uint4 u = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
  return __builtin_bit_cast(bf16x8v, u);
}

template <bool W_T, bool STAGE, int FLAT, int MATH>
__global__ __launch_bounds__(256, 2) void gather_gemm2_kernel(GemmParams p) {
  constexpr int NP = MATH == 3 ? 2 : 1;  // bf16 planes (hi, lo)
  constexpr int A_BYTES = MATH == 0 ? BM * LDA * 4 : NP * BM * LD16 * 2;
  constexpr int B_BYTES = MATH == 0 ? BK * BN * 4 : NP * BN * LD16 * 2;
  __shared__ __attribute__((aligned(16))) unsigned char s_a_raw[2][A_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char s_b_raw[2][B_BYTES];
  auto sA = [&](int buf) { return reinterpret_cast<float *>(s_a_raw[buf]); };
  auto sB = [&](int buf) { return reinterpret_cast<float *>(s_b_raw[buf]); };
  auto sA16 = [&](int buf, int pl) { return reinterpret_cast<unsigned short *>(s_a_raw[buf]) + pl * BM * LD16; };
  auto sB16 = [&](int buf, int pl) { return reinterpret_cast<unsigned short *>(s_b_raw[buf]) + pl * BN * LD16; };
  __shared__ int s_nbr[STAGE ? BM * KMAX : 1];
  __shared__ int s_orow[BM];
  __shared__ unsigned s_kmask;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t o0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int K = p.K;
  const int kbeg = blockIdx.z * p.kper;
  const int kend = min(K, kbeg + p.kper);

  if (tid == 0) s_kmask = 0u;
  if (tid < BM) {
    const int64_t v = o0 + tid;
    int orow = -1;
    if (v < p.n_virtual) orow = p.row_perm ? p.row_perm[v] : (int)v;
    s_orow[tid] = orow;
  }
  __syncthreads();
  unsigned kmask = (kend < 32 ? (1u << kend) - 1u : 0xFFFFFFFFu) & ~((1u << kbeg) - 1u);
  if (STAGE) {
    unsigned m = 0u;
    int lr = tid / K, kk = tid - lr * K;  // element e = tid + 256 i  ->  (row lr, offset kk)
    const int dlr = 256 / K, dkk = 256 - dlr * K;
    for (int e = tid; e < BM * K; e += 256) {
      const int orow = s_orow[lr];
      const int v = orow >= 0 ? p.nbr[(int64_t)orow * K + kk] : -1;
      s_nbr[e] = v;
      if (v >= 0) m |= 1u << kk;
      lr += dlr, kk += dkk;
      if (kk >= K) kk -= K, ++lr;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d);
    if (lane == 0 && m) atomicOr(&s_kmask, m);
    __syncthreads();
    kmask &= s_kmask;
  }
  const int ncc = (p.cin + BK - 1) / BK;
  const int n_items = FLAT ? (K * FLAT + BK - 1) / BK : __popc(kmask) * ncc;

  // per-thread staging coordinates
  const int a_cc = tid & 7, a_r = tid >> 3;     // A: rows a_r + 32 i, float4 column a_cc
  const int b_n4 = tid & 15, b_kk = tid >> 4;   // B (!W_T): rows b_kk + 16 i, float4 column b_n4
  const int bt_n = tid & 63, bt_k4 = tid >> 6;  // B (W_T) : column bt_n, float4 of k at 4*(bt_k4 + 4 i)
  int my_orow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) my_orow[i] = s_orow[a_r + 32 * i];

  int gk = kmask ? __builtin_ctz(kmask) : -1, gc0 = 0;  // iterator of the global-load stage
  int idx_g[4] = {-1, -1, -1, -1};  // neighbour rows of the item the next gload() fetches
  auto load_idx = [&](int k, int (&idx)[4]) {  // branch-free: k < 0 (past the end) yields -1
    const int kq = max(k, 0);
    const unsigned dead = k < 0 ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (STAGE) {
        idx[i] = (int)((unsigned)s_nbr[(a_r + 32 * i) * K + kq] | dead);
      } else {
        const unsigned v = (unsigned)p.nbr[(int64_t)max(my_orow[i], 0) * K + kq];
        idx[i] = (int)(v | dead | (my_orow[i] < 0 ? 0xFFFFFFFFu : 0u));
      }
    }
  };
  // raw 16-byte words of the item in flight + validity bits; the zeroing of invalid words is
  // deferred to sts() one iteration later, so nothing in the issuing iteration waits on vmcnt
  uint4 ga[4] = {}, gb[2] = {};
  unsigned g_ok = 0u;
  auto ldraw = [&](const float *base, int64_t off, bool ok) {
    return *reinterpret_cast<const uint4 *>(base + (ok ? off : 0));
  };
  int gj = 0;  // FLAT: item counter of the global-load stage
  // FLAT (the stem): every operand comes through a raw buffer load whose out-of-range offsets return 0 without touching
  // memory, so nothing is selected or masked per load (the fp32 MFMA rate equals the vector-ALU rate and the two share
  // issue slots: the selects, masks and 64-bit address arithmetic of the guarded form were ~3 VALU instructions per
  // MFMA).  A missing neighbour (-1) is row 0xFFFFFF of a descriptor that ends right there; an item column past the
  // flattened axis, a table row of a padding tile row and a weight column past cout get bit 31 in their offset.  The
  // launcher checks ldx <= 32 (so the sums stay below 2^32) and fewer than 2^24 - 1 input rows.
  const unsigned ldx4 = 4u * (unsigned)p.ldx;
  const i32x4 rfx = raw_rsrc(p.x, FLAT ? 0xFFFFFFu * ldx4 : 0u);
  const i32x4 rfw = raw_rsrc(p.w, FLAT ? 4u * (unsigned)K * (unsigned)FLAT * (unsigned)p.cout : 0u);
  const i32x4 rfn = raw_rsrc(p.nbr, FLAT ? 4u * (unsigned)p.n_out * (unsigned)K : 0u);
  unsigned tab_off[4], fb_off[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) tab_off[i] = my_orow[i] >= 0 ? 4u * (unsigned)my_orow[i] * (unsigned)K : 0x80000000u;
#pragma unroll
  for (int i = 0; i < 2; ++i)  // row b_kk + 16 i of the flattened [K*cin][cout] weight matrix, item 0; one item further per gload
    fb_off[i] = (n0 + 4 * b_n4 < p.cout ? 4u * (unsigned)(n0 + 4 * b_n4) : 0x80000000u) + 4u * (unsigned)(b_kk + 16 * i) * (unsigned)p.cout;
  auto flat_idx = [&](int j, int (&idx)[4]) {  // neighbour rows for flat item j (per-thread offset)
    const int off = (BK * j + 4 * a_cc) / (FLAT ? FLAT : 1);
    const unsigned ko = 4u * (unsigned)min(off, K - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) idx[i] = raw_load_i32(rfn, (int)(tab_off[i] + ko), 0, 0);  // (a padding row reads as row 0: never stored)
  };
  auto gload_flat = [&]() {
    int idx_n[4];
    flat_idx(gj + 1, idx_n);
    const int kf = BK * gj + 4 * a_cc;
    const int off = kf / (FLAT ? FLAT : 1);
    const unsigned chb = off < K ? 4u * (unsigned)(kf - off * FLAT) : 0x80000000u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ga[i] = __builtin_bit_cast(uint4, raw_load_v4(rfx, (int)(__umul24((unsigned)idx_g[i], ldx4) + chb), 0, 0));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      gb[i] = __builtin_bit_cast(uint4, raw_load_v4(rfw, (int)fb_off[i], 0, 0));
      fb_off[i] += 4u * BK * (unsigned)p.cout;
    }
    ++gj;
#pragma unroll
    for (int i = 0; i < 4; ++i) idx_g[i] = idx_n[i];
  };
  auto gload = [&]() {  // fetch item (gk, gc0) into registers and step the iterator
    if (FLAT) {
      gload_flat();
      return;
    }
    // the indices of the FOLLOWING item are requested first: they have a whole iteration to
    // arrive before the next gload() turns them into addresses
    const int nc0 = gc0 + BK;
    const bool wrap = nc0 >= p.cin;
    const int gkq0 = max(gk, 0);
    const unsigned rest = (gkq0 + 1 < 32) ? (kmask >> (gkq0 + 1)) : 0u;
    const int nk_wrap = (gk >= 0 && rest) ? (gkq0 + 1 + __builtin_ctz(rest | 0x80000000u)) : -1;
    const int nk = wrap ? nk_wrap : gk;
    int idx_n[4];
    load_idx(nk, idx_n);
    const int gkq = max(gk, 0);
    const int kw = p.flip_k ? (K - 1 - gkq) : gkq;
    const int c = gc0 + 4 * a_cc;
    unsigned okb = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = idx_g[i] >= 0 && c < p.cin;
      ga[i] = ldraw(p.x, (int64_t)idx_g[i] * p.ldx + c, ok);
      okb |= ok ? (1u << i) : 0u;
    }
    if (!W_T) {
      const int n = n0 + 4 * b_n4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = gc0 + b_kk + 16 * i;
        const bool ok = kk < p.cin && n < p.cout;
        gb[i] = ldraw(p.w, ((int64_t)kw * p.cin + kk) * p.cout + n, ok);
        okb |= ok ? (16u << i) : 0u;
      }
    } else {
      const int n = n0 + bt_n;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = gc0 + 4 * (bt_k4 + 4 * i);
        const bool ok = n < p.cout && kk < p.cin;
        gb[i] = ldraw(p.w, ((int64_t)kw * p.cout + n) * p.cin + kk, ok);
        okb |= ok ? (16u << i) : 0u;
      }
    }
    g_ok = okb;
    gc0 = wrap ? 0 : nc0;
    gk = nk;
#pragma unroll
    for (int i = 0; i < 4; ++i) idx_g[i] = idx_n[i];
  };
  auto sts = [&](int buf) {
    auto masked = [&](uint4 u, unsigned bit) {
      if (FLAT) return u;  // (buffer loads: invalid words arrive as zeros)
      const unsigned m = (g_ok & bit) ? 0xFFFFFFFFu : 0u;
      return make_uint4(u.x & m, u.y & m, u.z & m, u.w & m);
    };
    if (MATH == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<uint4 *>(&sA(buf)[(a_r + 32 * i) * LDA + 4 * a_cc]) = masked(ga[i], 1u << i);
      if (!W_T) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          *reinterpret_cast<uint4 *>(&sB(buf)[(b_kk + 16 * i) * BN + 4 * b_n4]) = masked(gb[i], 16u << i);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int kk = 4 * (bt_k4 + 4 * i);
          const uint4 u = masked(gb[i], 16u << i);
          sB(buf)[(kk + 0) * BN + bt_n] = __uint_as_float(u.x);
          sB(buf)[(kk + 1) * BN + bt_n] = __uint_as_float(u.y);
          sB(buf)[(kk + 2) * BN + bt_n] = __uint_as_float(u.z);
          sB(buf)[(kk + 3) * BN + bt_n] = __uint_as_float(u.w);
        }
      }
    } else {  // convert to bf16 (and the residual plane for split-bf16) on the way into LDS
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 u = masked(ga[i], 1u << i);
        float f[4] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          *reinterpret_cast<uint2 *>(&sA16(buf, pl)[(a_r + 32 * i) * LD16 + 4 * a_cc]) =
              make_uint2(pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]));
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = bf16_residual(f[e]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4 u = masked(gb[i], 16u << i);
        float f[4] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          if (!W_T) {  // 4 consecutive columns of one k row -> transposed [col][k] image
            const int kk = b_kk + 16 * i;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              sB16(buf, pl)[(4 * b_n4 + e) * LD16 + kk] = __builtin_bit_cast(unsigned short, (__bf16)f[e]);
          } else {  // 4 consecutive k of one column
            *reinterpret_cast<uint2 *>(&sB16(buf, pl)[bt_n * LD16 + 4 * (bt_k4 + 4 * i)]) =
                make_uint2(pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]));
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = bf16_residual(f[e]);
        }
      }
    }
  };
  const int arow = wave * 32 + (lane & 31), h = lane >> 5, col = lane & 31;
  struct Ops {  // MATH 0: fp32 fragments; else: bf16x8 fragments (as uint4) per plane and k-step
    float4 a[4];
    float b0[16], b1[16];
  };
  struct Ops16 {
    uint4 a[NP][2], b0[NP][2], b1[NP][2];
  };
  using OpsT = typename std::conditional<MATH == 0, Ops, Ops16>::type;
  auto lds_read = [&](int buf, OpsT &rr) {
    if constexpr (MATH == 0) {
      Ops &r = rr;
#pragma unroll
      for (int t = 0; t < 4; ++t) r.a[t] = *reinterpret_cast<const float4 *>(&sA(buf)[arow * LDA + 8 * t + 4 * h]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = 8 * t + 4 * h + j;
          r.b0[4 * t + j] = sB(buf)[kk * BN + col];
          r.b1[4 * t + j] = sB(buf)[kk * BN + 32 + col];
        }
    } else {
      Ops16 &r = rr;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          r.a[pl][t] = *reinterpret_cast<const uint4 *>(&sA16(buf, pl)[arow * LD16 + 16 * t + 8 * h]);
          r.b0[pl][t] = *reinterpret_cast<const uint4 *>(&sB16(buf, pl)[col * LD16 + 16 * t + 8 * h]);
          r.b1[pl][t] = *reinterpret_cast<const uint4 *>(&sB16(buf, pl)[(col + 32) * LD16 + 16 * t + 8 * h]);
        }
    }
  };
  f32x16 acc0 = {0}, acc1 = {0};
  auto mfma_half = [&](const OpsT &rr, int t0) {
    if constexpr (MATH == 0) {
      const Ops &r = rr;
#pragma unroll
      for (int t = t0; t < t0 + 2; ++t) {
        const float a4[4] = {r.a[t].x, r.a[t].y, r.a[t].z, r.a[t].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], r.b0[4 * t + j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], r.b1[4 * t + j], acc1, 0, 0, 0);
        }
      }
    } else {
      const Ops16 &r = rr;
      const int t = t0 >> 1;  // halves 0 / 2 -> k-steps 0 / 1 (16 channels each)
      auto fr = [](uint4 u) { return __builtin_bit_cast(bf16x8, u); };
      // smallest terms first (split-bf16): lo*hi + hi*lo, then hi*hi
      if (MATH == 3) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr(r.a[NP - 1][t]), fr(r.b0[0][t]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr(r.a[NP - 1][t]), fr(r.b1[0][t]), acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr(r.a[0][t]), fr(r.b0[NP - 1][t]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr(r.a[0][t]), fr(r.b1[NP - 1][t]), acc1, 0, 0, 0);
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr(r.a[0][t]), fr(r.b0[0][t]), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr(r.a[0][t]), fr(r.b1[0][t]), acc1, 0, 0, 0);
    }
  };

  OpsT R0 = {}, R1 = {};
  // ---- prologue: items 0,1 -> LDS, item 2 in flight, operands of item 0 in registers
  if (n_items > 0) {
    if (FLAT) flat_idx(0, idx_g);
    else load_idx(gk, idx_g);
    gload();
    sts(0);
  }
  if (n_items > 1) {
    gload();
    sts(1);
  }
  if (n_items > 2) gload();
  MINK_LDS_BARRIER();
  if (n_items > 0) lds_read(0, R0);
  MINK_LDS_BARRIER();

  // steady state (all three stages active, no branch inside): one scheduling region holding the
  // operand reads of item c+1, the 32 MFMAs of item c, the LDS stores of item c+2 and the global
  // loads of item c+3; the group barriers ask the scheduler to slot the memory / address
  // instructions between the 64-cycle MFMAs instead of clustering them.
  auto body_full = [&](int c, const OpsT &cur, OpsT &nxt) {
    sts(c & 1);  // item c+2 replaces item c (every wave read it during the previous iteration)
    gload();     // item c+3: in flight until the sts of the next iteration
    lds_read((c + 1) & 1, nxt);
    mfma_half(cur, 0);
    mfma_half(cur, 2);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
      __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);  // VALU / SALU
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one VMEM read
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // one DS write
    }
    MINK_LDS_BARRIER();
  };
  auto body_tail = [&](int c, const OpsT &cur, OpsT &nxt) {  // last three items: stages drain
    if (c + 1 < n_items) lds_read((c + 1) & 1, nxt);
    mfma_half(cur, 0);
    if (c + 2 < n_items) sts(c & 1);
    mfma_half(cur, 2);
    MINK_LDS_BARRIER();
  };
  int c = 0;
  for (; c + 3 < n_items; c += 2) {
    body_full(c, R0, R1);
    if (c + 4 < n_items) {
      body_full(c + 1, R1, R0);
    } else {
      body_tail(c + 1, R1, R0);
      c += 2;
      break;
    }
  }
  for (; c < n_items; c += 2) {
    body_tail(c, R0, R1);
    if (c + 1 < n_items) body_tail(c + 1, R1, R0);
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool direct = gridDim.z == 1;
  float *dst = direct ? p.y : p.ws + (int64_t)blockIdx.z * p.n_out * p.cout;
  const int ldd = direct ? p.ldy : p.cout;
  const int c_a = n0 + col, c_b = n0 + 32 + col;
  const float bias_a = (direct && p.bias && c_a < p.cout) ? p.bias[c_a] : 0.f;
  const float bias_b = (direct && p.bias && c_b < p.cout) ? p.bias[c_b] : 0.f;
  float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;  // column statistics of this wave's 32 rows (for the batch norm that follows)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = s_orow[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
    if (row >= 0) {
      float va = acc0[r] + bias_a, vb = acc1[r] + bias_b;
      if (p.accumulate) {  // (uniform) scatter-accumulate: each output row belongs to at most one tile row
        if (c_a < p.cout) va += dst[row * ldd + c_a];
        if (c_b < p.cout) vb += dst[row * ldd + c_b];
      }
      if (c_a < p.cout) dst[row * ldd + c_a] = va;
      if (c_b < p.cout) dst[row * ldd + c_b] = vb;
      sa += va, qa += va * va, sb += vb, qb += vb * vb;
    }
  }
  if (p.stats && direct) {  // uniform
    sa += __shfl_xor(sa, 32), qa += __shfl_xor(qa, 32), sb += __shfl_xor(sb, 32), qb += __shfl_xor(qb, 32);
    float *red = reinterpret_cast<float *>(s_a_raw[0]);  // [wave][sum a, sq a, sum b, sq b][32]; the tiles are dead (last barrier)
    if (h == 0) {
      red[(wave * 4 + 0) * 32 + col] = sa, red[(wave * 4 + 1) * 32 + col] = qa;
      red[(wave * 4 + 2) * 32 + col] = sb, red[(wave * 4 + 3) * 32 + col] = qb;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 5, c = n0 + (which >= 2 ? 32 : 0) + (tid & 31);
      const float v = (red[(0 * 4 + which) * 32 + (tid & 31)] + red[(1 * 4 + which) * 32 + (tid & 31)]) +
                      (red[(2 * 4 + which) * 32 + (tid & 31)] + red[(3 * 4 + which) * 32 + (tid & 31)]);
      if (c < p.cout) p.stats[((int64_t)blockIdx.x * 2 + (which & 1)) * p.cout + c] = v;
    }
  }
}

// stage 2 of the fused statistics: per-tile float partials [rows][cols] -> double partials [gridDim.x][cols]
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float *__restrict__ in, int64_t rows, int cols,
                                                         double *__restrict__ out) {
  for (int c = threadIdx.x; c < cols; c += 256) {
    double s = 0.0;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) s += (double)in[r * cols + c];
    out[(int64_t)blockIdx.x * cols + c] = s;
  }
}

// split-K reduce fused with the column statistics: y = sum_z ws[z] + bias, partial[blk][2][C] = (sum y, sum y^2).
// Thread layout of the batch-norm column reduction: a thread owns one float4 column and strides over rows.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float *__restrict__ ws, int64_t n_out, int cout,
                                                                  int ksplit, const float *__restrict__ bias,
                                                                  float *__restrict__ y, int ldy,
                                                                  double *__restrict__ partial) {
  extern __shared__ double s_red[];  // [row lanes][2][C]
  const int tpr = cout >> 2, rlanes = 256 / tpr;
  const int c4 = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const int64_t total = n_out * cout;
  float4 s0 = make_float4(0, 0, 0, 0), s1 = make_float4(0, 0, 0, 0);
  if (rl < rlanes) {
    const float4 b = bias ? *reinterpret_cast<const float4 *>(bias + 4 * c4) : make_float4(0, 0, 0, 0);
    // four rows per trip: all their slab loads are in flight together (the pass is latency-bound)
    const int64_t stride = (int64_t)gridDim.x * rlanes;
    for (int64_t row = (int64_t)blockIdx.x * rlanes + rl; row < n_out; row += 4 * stride) {
      float4 v[4] = {b, b, b, b};
      int64_t off[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = row + u * stride;
        off[u] = (r < n_out ? r : row) * cout + 4 * c4;
      }
      // four slabs per trip as well: sixteen independent 16-byte loads in flight (the deep layers sum 14 slabs of a few
      // hundred rows: one dependent round trip per slab was 11-14 us for a 1 MB output); summed in slab order
      int z = 0;
      for (; z + 3 < ksplit; z += 4) {
        float4 t[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 4; ++u) t[q][u] = *reinterpret_cast<const float4 *>(ws + (int64_t)(z + q) * total + off[u]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u].x += t[q][u].x, v[u].y += t[q][u].y, v[u].z += t[q][u].z, v[u].w += t[q][u].w;
      }
      for (; z < ksplit; ++z) {
        float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4 *>(ws + (int64_t)z * total + off[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u].x += t[u].x, v[u].y += t[u].y, v[u].z += t[u].z, v[u].w += t[u].w;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = row + u * stride;
        if (r < n_out) {
          *reinterpret_cast<float4 *>(y + r * ldy + 4 * c4) = v[u];
          s0.x += v[u].x, s0.y += v[u].y, s0.z += v[u].z, s0.w += v[u].w;
          s1.x += v[u].x * v[u].x, s1.y += v[u].y * v[u].y, s1.z += v[u].z * v[u].z, s1.w += v[u].w * v[u].w;
        }
      }
    }
    double *d = s_red + ((int64_t)rl * 2) * cout + 4 * c4;
    d[0] = s0.x, d[1] = s0.y, d[2] = s0.z, d[3] = s0.w;
    d[cout + 0] = s1.x, d[cout + 1] = s1.y, d[cout + 2] = s1.z, d[cout + 3] = s1.w;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * cout; e += 256) {
    double s = 0.0;
    for (int r = 0; r < rlanes; ++r) s += s_red[(int64_t)r * 2 * cout + e];
    partial[(int64_t)blockIdx.x * 2 * cout + e] = s;
  }
}

// y[row][c] = sum_z ws[z][row][c] + bias[c]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ ws, int64_t n_out, int cout,
                                                            int ksplit, const float *__restrict__ bias,
                                                            float *__restrict__ y, int ldy) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = n_out * cout;
  if (idx >= total) return;
  const int64_t row = idx / cout;
  const int c = (int)(idx - row * cout);
  float s = 0.f;
  for (int z = 0; z < ksplit; ++z) s += ws[(int64_t)z * total + idx];
  y[row * ldy + c] = s + (bias ? bias[c] : 0.f);
}

// ------------------------------------------------ output-stationary gather-GEMM over ROW-COMPACTED offsets (fp32)
// 37-48 % of a mid-layer table is empty, and the kernel above multiplies those entries as zero rows ("zero
// rows").  Here the workgroup still owns a 128-row x 64-column tile of the output, but
//   * the tile's accumulator lives in LDS, not in registers;
//   * for every offset k of the workgroup's slice the tile rows that HAVE a neighbour are compacted into dense 16-row
//     blocks (rulebook of the tile: wave64 ballot + prefix rank, built once from the tile's table slice) -- ~80 rows
//     instead of 128;
//   * a wave owns a 16-column strip and multiplies EVERY compacted block of the offset for it (v_mfma_f32_16x16x4_f32,
//     operands swapped so that a lane ends up with four consecutive columns of one row): the work is balanced whatever
//     the block count is, and padding is to 16 rows, not 32;
//   * the weight fragment of a wave (32 channels x 16 columns per item) goes from global memory straight into its
//     registers -- only the gathered rows pass through LDS;
//   * after the last channel chunk of an offset the wave adds its blocks into the tile rows they belong to (one
//     16-byte LDS read-modify-write per block and lane, through the rulebook; no other wave touches its strip).
// Offsets and channel chunks are visited in ascending order: bitwise reproducible.  Pipeline: LDS double buffer for the
// gathered rows, CD items (offset, 32-channel chunk) of global loads in flight in registers, one raw barrier per item;
// 78 KB of LDS per workgroup -> two workgroups per CU hide each other's barriers.  Split-K / statistics epilogues are
// those of gather_gemm2_kernel.  Offsets per workgroup <= CKP (split launches; the planner's slices are 2-9 offsets).
constexpr int CLDC = BN + 4;         // row stride of the C tile (floats)
constexpr int CKP = 9;               // offsets per workgroup the rulebook has room for
constexpr int CD = 3;                // ring of global-load register sets (an item is requested CD - 1 items ahead; 4 sets spill at 128 VGPRs)
// Layout of the gathered-row tile.  The operand read is lane (n = lane & 15, kq = lane >> 4) -> the 16-byte piece kq (+ 4
// for the second half) of row n; ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32:
// MI355X_MICROARCH.md, LDS), so one group holds rows {0-3, 12-15} at piece kq and rows {4-11} at piece kq + 1.  With the
// padded stride of the dense kernels (36 floats) rows 13 and 4 of such a group start on the same bank: every operand read
// was a 2-way conflict (8 LDS cycles for 4) -- and the LDS, not the vector memory, is what the non-matrix part of an item
// waits for (16 waves x ~130 LDS cycles per item against 3,072 matrix cycles per SIMD).  Here rows are NOT padded (32
// floats) and piece p of row n sits at piece p ^ ((n >> 1) & 7): two rows fill the 64 banks, the XOR spreads the eight
// rows of equal parity in a group over the eight pieces; the staging stores (8 lanes = the 8 pieces of one row) stay
// conflict-free.  MINK_CSWZ=0: the padded layout with stride MINK_CLDA (A/B builds).
#ifndef MINK_CSWZ
#define MINK_CSWZ 1
#endif
#ifndef MINK_CLDA
#define MINK_CLDA 36
#endif
// MINK_CPF: the gather stage keeps the byte offsets of its thread's rows in registers for all channel chunks of an offset
// (one LDS read per offset instead of one per item, requested a step ahead) and the gathers are global loads with a scalar
// base (x + chunk) and that 32-bit lane offset: two instructions per item where there were two LDS round trips and ten
// 64-bit address instructions at the head of every step.
// MINK_CCIN: the accumulators of an offset START as the C rows they belong to (read behind the previous offset's write,
// under the barrier wait) and are written back when the offset is complete: no read-add-write chain at the end of an
// offset, no v_pk_add / zeroing moves beside the MFMAs, one LDS read for the four row indices of a lane.
#ifndef MINK_CPF
#define MINK_CPF 1
#endif
#ifndef MINK_CCIN
#define MINK_CCIN 1
#endif
constexpr bool CSWZ = MINK_CSWZ, CPF = MINK_CPF, CCIN = MINK_CCIN;
constexpr int CLDA = CSWZ ? BK : MINK_CLDA;
constexpr int CKP_SK = 16;           // ... of the stream-K form (a round per segment where the classic rulebook would take two: the act / nbs nibble tables hold 16)
constexpr int compact_smem(int CM, bool P3 = false, bool SK = false) {
  return ((CM + 1) * CLDC + (P3 ? 3 : 2) * CM * CLDA + (SK ? CKP_SK : CKP) * CM + 32) * 4 + (SK ? CKP_SK : CKP) * CM + 4 * CM;
}

// CM: rows per tile.  64: 39 KB of LDS, four workgroups (16 waves) per CU -- the latency of an item's chain (barrier,
// LDS stores, operand reads, scatter) is hidden by the other workgroups; 128: half the weight traffic.
// Instruction diet (PMC: the matrix pipe and the vector ALU do not co-execute on this part, every VALU instruction is
// a slot the MFMAs lose): item metadata (offset list, block counts) lives in SGPRs; weight fragments come through
// buffer loads whose lane offsets are computed once; padding rows of a block are not zeroed -- row n of the gathered
// operand only reaches column n of the product, which the rulebook sends to the spare C row.
// PERM (the data gradient of a strided convolution, rows grouped by parity class): tile row v computes output row
// row_perm[v]; a class-pure tile has at most eight live offsets anywhere among the K, so the slice of a workgroup is not
// an offset range but ALL offsets -- the rulebook holds the live ones (CKP at a time: a tile that is not class-pure
// just takes more rounds) -- and a split launch cuts the CHANNEL chunks instead (every slice sees every live offset:
// balanced whatever the class is).  p.kper is then the number of 32-channel chunks per slice.
// ABL: the timing-only switches of scripts/kbench.py cab (p.stagger bits 2-7) are compiled in -- forward kernel only; the
// production instantiations carry none of their branches
// (Round 5, measured and removed: 128-row tiles with EIGHT waves, two per column strip taking every second block of an offset --
//  half the LDS operand reads and a third of the padding per row, same 16 waves per CU: 4-8 % SLOWER on the stride-1 layers
//  (l1.conv2 92-97 us against 86-91), 14-40 % slower on the class-permuted data gradients: the two waves a workgroup has on
//  each SIMD leave the barrier together and want the matrix pipe and the LDS at the same moment, where the four waves of four
//  different workgroups are out of step by themselves.)
// NWV: waves per workgroup (4: a wave owns ONE 16-column strip; 2: a wave owns TWO strips -- the per-item instruction stream is
// then paid once per 48 MFMAs instead of once per 24 and a block's LDS operands feed both strips; bit-identical results, and
// measured 8-14 % SLOWER on every layer (two waves per SIMD hide each other's latencies worse than four: DESIGN.md Appendix A),
// so only NWV = 4 is instantiated).
// MATH = 1 (--math bf16; the class-permuted data gradients): the same kernel with ONE v_mfma_f32_16x16x32_bf16 per block and item where
// the fp32 form issues eight 16x16x4 -- the item's weight fragment and a block's gathered operands are rounded to bf16 in registers
// (eight cvt_pk per block and item), LDS and global traffic unchanged: the kernel then runs at its "no matrix work" time, which for the
// strided data gradients is far under what the dense bf16 kernel takes (the table there is two thirds empty per offset).
// P3 (round 6: the structural answer to the convoy of section 4's PMC reading): THREE stages of the gathered-row tile in LDS and the
// MFMA operands of an item read one step AHEAD, into a second register set, under the previous item's MFMAs -- a wave leaves the
// barrier with its operands already in registers and starts multiplying at once; what is left in front of a burst is instruction
// issue (two LDS stores, two gather requests, the eight reads for the next item), not an LDS round trip.  The classic form reads an
// item's operands behind the barrier that published them: every wave of the workgroup -- one per SIMD -- then waits out the same
// LDS round trip at the same moment, and the four workgroups of a CU fall into step (57 % of a wave's life waiting to issue).
// Costs: 8 KB more LDS (45 KB: three workgroups per CU instead of four) and 32 more VGPRs (within the 170 of three waves per SIMD);
// the gather stage runs four items ahead of the multiplying stage, the weight stage three (two iterators).  Same arithmetic in the
// same order: bit-identical to the classic form (tests/test_gpu_ops.py).
// SK (round 6, "stream-K"): what a launch of this kernel loses is not inside the item loop.  A per-workgroup clock trace
// (mink_conv_trace, scripts/kbench.py ctrace; profiles/r06_workgroup_trace.txt) of l1.conv2 -- 575 row tiles x 3 offset slices =
// 1,725 workgroups on 1,024 slots -- shows every CU with four resident workgroups for the first 35 % of the launch, 3.7 until
// 55 %, 2.7 until 80 %, then a tail of 2.5 -> 0: 23 % of the slot time has NO workgroup in it and another 17 % is prologue and
// epilogue; the matrix pipe's 0.48 is that, not the 0.73 of the loop.  The work of a launch is (tiles) x (K offsets), each
// offset of a tile costing the same few items -- so here the launch is EXACTLY one resident round (gridDim.x = 4 or 3 workers per
// CU), and worker w takes the run [F w / G, F (w + 1) / G) of the F = tiles x K offset-tiles in tile-major order: every worker,
// hence every CU, multiplies the same number of offsets (+-1), starts at time zero and ends with the others.  A run crosses tile
// boundaries: a worker processes up to three SEGMENTS (tile, offsets k0 .. k1), each with the prologue / rulebook / epilogue of the
// classic form (offsets of a segment in rounds of CKP, as the class-permuted form does), and writes the segment's partial tile
// into slab (w - first worker of the tile); the worker that finishes a tile also zeroes the slabs nobody wrote, so the consumers
// (split-K reduce with statistics, the batch-norm kernels that read slabs) see the fixed slab count they are used to.  Workers are
// numbered so that the eight XCDs each cover one contiguous eighth of the tiles (column-tile major: a deep layer's XCD reads one
// column tile's weights).  Deterministic: a tile's slabs are summed in slab order, offsets ascending across them.
template <bool W_T, int CM, bool PERM = false, bool ABL = false, int NWV = 4, int MATH = 0, bool P3 = false, bool SK = false>
__global__ __launch_bounds__(64 * NWV, P3 ? 3 : (CM == 64 ? NWV : NWV / 2)) void compact_gemm_kernel(GemmParams p) {
  static_assert(!SK || (!PERM && !P3 && !ABL && NWV == 4), "stream-K: the stride-1 form");
  const int abl = ABL ? p.stagger : 0;
  constexpr bool CCIN = mink::CCIN && (W_T || !PERM);  // (the forward-layout class-permuted form -- tests only -- has no registers to spare)
  constexpr int T = 64 * NWV, SPW = 4 / NWV;  // threads, column strips per wave
  constexpr int NBLK = CM / 16, NH = CM / 64, NA = CM / (T / 8);
  constexpr int CKR = SK ? CKP_SK : CKP;  // offsets per rulebook round
  constexpr int NU = ((PERM ? 27 : CKR) + NWV - 1) / NWV;  // offsets of the slice a wave looks at (cs, cs + NWV, ...); SK: of a ROUND
  extern __shared__ __attribute__((aligned(16))) unsigned char c_smem[];
  float *sC = reinterpret_cast<float *>(c_smem);             // [CM + 1][CLDC]; row CM takes the padding lanes
  constexpr int NST = P3 ? 3 : 2;                            // stages of the gathered-row tile
  float *sA = sC + (CM + 1) * CLDC;                          // [NST][CM][CLDA] compacted gathered rows
  int *s_src = reinterpret_cast<int *>(sA + NST * CM * CLDA);  // [CKP][CM] input row of the p-th compacted row (padding: row 0)
  int *s_cnt = s_src + CKR * CM;                             // [32] compacted rows per offset of the slice
  unsigned char *s_lrow = reinterpret_cast<unsigned char *>(s_cnt + 32);  // [CKP][CM] tile row of the p-th compacted row
  int *s_orow = reinterpret_cast<int *>(s_lrow + CKR * CM);  // PERM: [CM] output row of a tile row (-1: padding)

  const int tid = threadIdx.x, lane = tid & 63;
  const int cs = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = column strip
  const int kq = lane >> 4, n = lane & 15;
  // Workgroup -> (row tile bx, column tile by, slice bz).  Plain: the launch grid.  Swizzled (p.swz_x > 0; deep layers, a
  // one-dimensional launch): workgroups are dealt to the eight XCDs round-robin in launch order, and every row tile of a
  // (column tile, slice) reads the SAME weight slice -- 262 KB at l4.conv2, where the eight row tiles of a slice landed on
  // eight XCDs and each L2 fetched it for itself (PMC: 250 MB of HBM traffic for 30 MB of weights).  Here a run of
  // 8 * swz_x launch slots serves eight slices, one per XCD, whose swz_x row tiles follow each other on that XCD.
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z, gz = gridDim.z;
  if (p.swz_x > 0) {  // uniform
    const unsigned X = (unsigned)p.swz_x, Y = (unsigned)p.swz_y, Z = (unsigned)p.swz_z;
    const unsigned L = blockIdx.x, slice = 8u * (L / (8u * X)) + (L & 7u);
    if (slice >= Y * Z) return;  // (the launch is padded to whole runs)
    bx = (L % (8u * X)) >> 3, by = slice % Y, bz = slice / Y, gz = Z;
  }
#ifndef MINK_CPRIO
#define MINK_CPRIO 0
#endif
  // MINK_CPRIO (A/B builds): a static wave priority that differs between the workgroups sharing a CU.  PMC (round 5): in the
  // item loop a wave waits ~2,300 cycles per item to issue its 24 MFMAs (768 cycles of pipe) while the pipe is 73 % busy --
  // the four waves of a SIMD (four workgroups) run their MFMA bursts at the same time and their LDS / scalar phases at the
  // same time.  With distinct priorities the first workgroup would run its burst unimpeded and the others fill its gaps.
  // Measured (kbench all, two boxes): 1 or 2 gain 2-4 % on layer 1 and the strided data gradients alone, nothing on layers 2-4,
  // and nothing in the step (3.58 ms with and without): off.
  if constexpr (MINK_CPRIO == 1) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned pr = (lin >> 8) & 3u;  // (workgroups 256 apart in launch order tend to share a CU)
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  } else if constexpr (MINK_CPRIO == 2) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, whole register
    const unsigned pr = hw & 3u;  // wave slot on its SIMD (bits 3:0)
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
  unsigned long long t_start = 0, t_loop = 0, t_epi = 0;
  if constexpr (ABL) {
    if (p.trace) t_start = __builtin_readcyclecounter();
  }
  // (every quotient below is uniform but comes out of the vector ALU -- the scalar unit does not divide: readfirstlane puts it
  //  back where the loop's scalar bookkeeping and the scalar-base loads want it)
  unsigned sk_f = 0, sk_end = 0, sk_w = 0;
  if constexpr (SK) {
    const unsigned G = gridDim.x;
    sk_w = (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3);  // (G is a multiple of 8: the workers of an XCD are neighbours)
    sk_f = sk_w * p.sk_q + min(sk_w, p.sk_r), sk_end = sk_f + p.sk_q + (sk_w < p.sk_r ? 1u : 0u);
    if (sk_f >= sk_end) return;
  }
  // SK: one pass per segment of the worker's run (a backward goto that only the SK instantiation contains: a loop statement
  // around the body cost the other instantiations registers -- the forward-layout class-permuted form went from 125 to 128
  // VGPRs plus scratch, and its scalar-base loads lost their scalar registers)
sk_next_segment : {
  int sk_k0 = 0, sk_nk = 0, sk_slab = 0, sk_zero = 0;
  if constexpr (SK) {
    const unsigned Ku = (unsigned)p.K;
    const unsigned tile = (unsigned)__builtin_amdgcn_readfirstlane((int)(sk_f / Ku));
    sk_k0 = (int)(sk_f - tile * Ku);
    sk_nk = (int)min(Ku - (unsigned)sk_k0, sk_end - sk_f);
    by = (unsigned)__builtin_amdgcn_readfirstlane((int)(tile / (unsigned)p.sk_X));
    bx = tile - by * (unsigned)p.sk_X;
    // the worker whose run holds the tile's first offset: runs are q + 1 long for the first r workers, q after them
    const unsigned P = tile * Ku, head = p.sk_r * (p.sk_q + 1u);
    const unsigned w_first = (unsigned)__builtin_amdgcn_readfirstlane((int)(P < head ? P / (p.sk_q + 1u) : p.sk_r + (P - head) / p.sk_q));
    sk_slab = (int)(sk_w - w_first);
    sk_zero = (unsigned)(sk_k0 + sk_nk) == Ku ? p.sk_S - 1 - sk_slab : 0;  // the worker that finishes a tile zeroes the slabs nobody wrote
  }
  const int64_t o0 = (int64_t)bx * CM;
  const int n0 = by * BN;
  const int K = p.K;
  const int kbeg = PERM ? 0 : (SK ? sk_k0 : bz * p.kper), nk = PERM ? K : (SK ? sk_nk : min(K, kbeg + p.kper) - kbeg);
  const int rows_here = (int)min((int64_t)CM, (PERM ? p.n_virtual : p.n_out) - o0);
  const int ncc_all = p.cin / BK;
  const int cbeg = PERM ? bz * p.kper : 0;                               // first channel chunk of this slice
  const int ncc = PERM ? min(ncc_all, cbeg + p.kper) - cbeg : ncc_all;   // channel chunks of this slice

  // ---- prologue: table entries of the slice (a wave takes offsets cs, cs + 4, ...; they are requested together,
  // straight from global memory: one round trip), C = 0
  static_assert(CKP <= 12 && CKR <= 16 && (NWV == 4 || NWV == 2) && CM == 64, "the rulebook's slots and the wave layout");
  int orow[NH];
#pragma unroll
  for (int hh = 0; hh < NH; ++hh) {
    const int r = lane + 64 * hh;
    orow[hh] = r < rows_here ? (PERM ? p.row_perm[o0 + r] : (int)(o0 + r)) : -1;
    if (PERM && cs == 0) s_orow[r] = orow[hh];
  }
  int tv[NU][NH];
  if constexpr (!SK) {  // (SK: a segment has up to K offsets -- its table entries are read round by round, below)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int jj = cs + NWV * u;
#pragma unroll
      for (int hh = 0; hh < NH; ++hh)
        tv[u][hh] = (jj < nk && orow[hh] >= 0) ? p.nbr[(int64_t)orow[hh] * K + kbeg + jj] : -1;
    }
  }
  for (int e = tid; e < (CM + 1) * CLDC / 4; e += T) reinterpret_cast<float4 *>(sC)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (PERM) {  // rows per offset first: the live offsets get the rulebook's slots in ascending order
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int jj = cs + NWV * u;
      int cnt = 0;
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) cnt += __popcll(__ballot(tv[u][hh] >= 0));
      if (lane == 0 && jj < 32) s_cnt[jj] = jj < nk ? cnt : 0;
    }
    __syncthreads();
  }
  // s_lrow is laid out [slot][n][block]: the (up to four) tile rows of lane n's compacted rows are one 4-byte read
  auto lri = [](int pos) { return (pos & 15) * (CM / 16) + (pos >> 4); };
  unsigned amask = 0u;  // PERM: live offsets of the tile
  if (PERM) {
    for (int j = 0; j < nk; ++j)
      if (__builtin_amdgcn_readfirstlane(s_cnt[j]) > 0) amask |= 1u << j;
  }
  const int n_live = PERM ? __popc(amask) : (SK ? nk : 1);  // (SK: the segment's offsets, CKP per round)

  // ---- weight fragment of this lane: channels 16 h + 4 kq + s of the chunk, column 16 cs + n; byte offsets inside the
  // chunk are fixed per lane, the chunk's own offset is a scalar
  i32x4 rw;
  {
    const unsigned long long wa = (unsigned long long)p.w;
    rw.x = (int)(unsigned)wa, rw.y = (int)((wa >> 32) & 0xFFFFu), rw.z = (int)(4u * (unsigned)K * (unsigned)p.cin * (unsigned)p.cout), rw.w = 0x00020000;
  }
  int wv[SPW][W_T ? 2 : 8];  // (strip q of this wave: columns 16 (SPW cs + q) + n)
#pragma unroll
  for (int q = 0; q < SPW; ++q) {
    if (!W_T) {
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[q][e] = 4 * ((16 * (e >> 2) + 4 * kq + (e & 3)) * p.cout + 16 * (SPW * cs + q) + n);
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) wv[q][h] = 4 * ((16 * (SPW * cs + q) + n) * p.cin + 16 * h + 4 * kq);
    }
  }
  // ---- staging of the gathered rows: rows a_r + (T / 8) i, float4 column a_cc
  const int a_cc = tid & 7, a_r = tid >> 3;
  constexpr int ARP = T / 8;  // rows per staging pass
  static_assert(!CSWZ || (ARP % 16 == 0 && BK == 32), "the XOR term must not depend on the staging pass");
  const int a_sw = 4 * (CSWZ ? (a_cc ^ ((a_r >> 1) & 7)) : a_cc);                     // staging store: float offset inside the row
  const int r_sw0 = 4 * (CSWZ ? (kq ^ ((n >> 1) & 7)) : kq), r_sw1 = 4 * (CSWZ ? ((kq + 4) ^ ((n >> 1) & 7)) : kq + 4);  // operand read, halves 0 / 1
  const float *xcol = p.x + 4 * a_cc;
  // The ring's global loads are issued through inline asm and waited for with hand-counted s_waitcnt: every step issues the
  // same NA + NW loads in the same order, so "the rows of item it + 1 have arrived" is vmcnt(NA + 2 NW) and "the weights of
  // item it" is vmcnt(3 NA + 2 NW) -- the loads of the last two steps stay in flight.  Left to the compiler (which merges the
  // pending-load state of the loop's entry, its latch and the conditional steps conservatively) two of every three steps
  // began with s_waitcnt vmcnt(0): a full memory round trip behind the loads issued at the END of the previous step, with
  // the matrix pipe idle -- that, not the MFMAs, was the kernel's item time (pipe busy 0.40).  The compiler does not see
  // these loads: their registers must not be copied between the load and its wait (they are only ever named as asm operands
  // and MFMA / LDS-store sources behind the wait), and everything is drained before the ring's registers die.
  constexpr int NW = SPW * (W_T ? 2 : 8);  // weight loads per item
  constexpr bool DBG_WAIT0 = false;
  f32x4 ga[CD][NA];  // (a native vector: copies of the HIP uint4 struct become memcpy calls that keep the ring in scratch)
  float gw[CD][SPW][W_T ? 1 : 8];
  f32x4 gwt[CD][SPW][W_T ? 2 : 1];  // W_T: the fragment as the two 16-byte loads deliver it (component e & 3 of load e >> 2)
  f32x4 acc[SPW][NBLK];
#pragma unroll
  for (int q = 0; q < SPW; ++q)
#pragma unroll
    for (int i = 0; i < NBLK; ++i) acc[q][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float *myC = sC + 16 * SPW * cs + 4 * kq;  // (strip q: + 16 q)

  // PERM: the live offsets CKP at a time (a class-pure tile has at most eight: one round)
  for (int round0 = 0; round0 < n_live; round0 += CKR) {
    if constexpr (SK) {  // the round's table entries (holding all K of a segment costs seven registers through the item loop: spills)
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int jj = round0 + cs + NWV * u;
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
          tv[u][hh] = (jj < nk && cs + NWV * u < CKR && orow[hh] >= 0) ? p.nbr[(int64_t)orow[hh] * K + kbeg + jj] : -1;
      }
    }
    // ---- rulebook of the round: wave64 ballot + prefix rank per offset
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int jj = (SK ? round0 : 0) + cs + NWV * u;
      const int slot = PERM ? __popc(amask & ((1u << jj) - 1u)) - round0 : (SK ? jj - round0 : jj);
      const bool mine = PERM ? (jj < nk && ((amask >> jj) & 1u) && slot >= 0 && slot < CKP) : (SK ? (jj < nk && slot >= 0 && slot < CKR) : jj < nk);
      if (mine) {  // uniform
        int cnt = 0;
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
          const int v = tv[u][hh];
          const unsigned long long b = __ballot(v >= 0);
          if (v >= 0) {
            const int pos = cnt + wave_rank(b);
            s_src[slot * CM + pos] = v, s_lrow[slot * CM + lri(pos)] = (unsigned char)(lane + 64 * hh);
          }
          cnt += __popcll(b);
        }
        // behind the compacted rows: input row 0 (read, multiplied into columns nobody keeps) and the spare C row
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
          if (lane + 64 * hh >= cnt) s_src[slot * CM + lane + 64 * hh] = 0, s_lrow[slot * CM + lri(lane + 64 * hh)] = (unsigned char)CM;
        if (!PERM && lane == 0) s_cnt[jj] = cnt;
      }
    }
    __syncthreads();
    // the round's offsets that have rows and their block counts, four bits each, in scalar registers (<= 9 entries);
    // PERM: entry a is rulebook slot a, its offset is the (round0 + a)-th live one (a byte each in kl0..kl2)
    unsigned act_lo = 0u, act_hi = 0u, nbs_lo = 0u, nbs_hi = 0u, kl0 = 0u, kl1 = 0u, kl2 = 0u;
    int na = 0;
    if (PERM) {
      unsigned rest = amask;
      for (int a = 0; a < round0; ++a) rest &= rest - 1u;
      for (; rest && na < CKP; rest &= rest - 1u) {
        const int j = __builtin_ctz(rest);
        const unsigned nbj = (unsigned)((__builtin_amdgcn_readfirstlane(s_cnt[j]) + 15) >> 4);
        if (na < 8) act_lo |= (unsigned)na << (4 * na), nbs_lo |= nbj << (4 * na);
        else act_hi |= (unsigned)na << (4 * (na - 8)), nbs_hi |= nbj << (4 * (na - 8));
        const unsigned kb_ = (unsigned)j << (8 * (na & 3));
        if (na < 4) kl0 |= kb_;
        else if (na < 8) kl1 |= kb_;
        else kl2 |= kb_;
        ++na;
      }
    } else {
      const int j0 = SK ? round0 : 0, j1 = SK ? min(nk, round0 + CKR) : nk;  // (SK: entry = rulebook slot j - round0)
      for (int j = j0; j < j1; ++j) {
        const int cnt = __builtin_amdgcn_readfirstlane(s_cnt[j]);
        if (cnt > 0) {
          const unsigned nbj = (unsigned)((cnt + 15) >> 4);
          if (na < 8) act_lo |= (unsigned)(j - j0) << (4 * na), nbs_lo |= nbj << (4 * na);
          else act_hi |= (unsigned)(j - j0) << (4 * (na - 8)), nbs_hi |= nbj << (4 * (na - 8));
          ++na;
        }
      }
    }
    auto nib = [](unsigned lo, unsigned hi, int a) { return (int)(((a < 8 ? lo : hi) >> (4 * (a & 7))) & 15u); };
    // (a const copy: naming the round loop's own variable inside the lambdas below took it -- and with it the whole scalar bookkeeping
    //  of the item loop -- out of registers in the class-permuted instantiations, which do not even use it)
    const int kbeg0 = kbeg + (SK ? round0 : 0);  // kernel offset of rulebook slot 0 of this round (stride-1 forms)
    auto kof = [&](int a, int j) {  // kernel offset of entry a (rulebook slot j)
      if constexpr (!PERM) return kbeg0 + j;
      const unsigned w_ = a < 4 ? kl0 : a < 8 ? kl1 : kl2;
      return (int)((w_ >> (8 * (a & 3))) & 255u);
    };
    const int n_items = (abl & 4) ? 0 : na * ncc;  // (bit 2, timing only: prologue and epilogue alone)

    // iterator of the global-load stage.  Scalar diet (round 5; PMC: 3.2 scalar instructions per MFMA, ~77 per item, in the
    // in-order stream of a wave that has 24 MFMAs to issue): the byte offsets of the item's weight block and channel chunk are
    // carried along and stepped by constants; the nibble / byte tables of the round are only unpacked when the offset changes.
    int g_ka = 0, g_cc = 0;
    auto wbase_of = [&](int a) {  // byte offset of the weight block of round entry a, first channel chunk of the slice
      const int j = nib(act_lo, act_hi, a);
      const int k = kof(a, j);
      const int kw = p.flip_k ? K - 1 - k : k;
      if (abl & 8) return 0;  // (bit 3, timing only: one weight block for every item)
      return W_T ? 4 * ((kw * p.cout + n0) * p.cin + cbeg * BK) : 4 * ((kw * p.cin + cbeg * BK) * p.cout + n0);
    };
    const int g_wstep = (abl & 8) ? 0 : (W_T ? 4 * BK : 4 * BK * p.cout);  // one channel chunk further
    int g_wo = na > 0 ? wbase_of(0) : 0;  // weight block of the stage's item (bytes)
    unsigned g_xo = 4u * (unsigned)(cbeg * BK);  // channel chunk of the stage's item (bytes into a row)
    int g_left = n_items - 1;             // steps the iterator may still take: past the end the last item is re-read
    // every call issues the same NA + (W_T ? 2 : 8) loads, whatever the item: the vmcnt distances of the ring are static
    // (a conditional load would make every wait a wait for ALL loads in flight); past the end the last item is re-read
    // the two halves of an item's request: the gathered rows (gload_a, at the start of a step) and the weight fragment
    // (gload_w, behind the step's MFMAs, which read the previous fragment in place -- requested early, the registers would
    // have to be copied out first: eight v_mov beside 24 MFMAs); gload_w steps the iterator
    unsigned sv[NA];  // CPF: byte offset (row and 16-byte column) of this thread's gathered rows of the gather stage's offset
    auto load_src = [&]() __attribute__((always_inline)) {
      const int j = nib(act_lo, act_hi, g_ka);
#pragma unroll
      for (int i = 0; i < NA; ++i)
        sv[i] = (unsigned)((abl & 16) ? 0 : s_src[j * CM + a_r + ARP * i]) * (4u * (unsigned)p.ldx) + 16u * (unsigned)a_cc;  // (bit 16, timing only: every gather reads row 0)
    };
    auto gload_a = [&](int slot) __attribute__((always_inline)) {
      if constexpr (CPF) {
        unsigned long long xb = (unsigned long long)p.x + g_xo;
        if constexpr (PERM && !W_T) {
          // (the forward-layout class-permuted form -- tests only -- sits at the register limit, and the allocator has been seen to
          //  leave this uniform base in vector registers, which the "s" operand cannot take: put it back explicitly)
          xb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xb >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xb);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ga[slot][i]) : "v"(sv[i]), "s"(xb));
        return;
      }
      const int j = nib(act_lo, act_hi, g_ka);
      const int c0 = (cbeg + g_cc) * BK;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const float *src = xcol + (int64_t)((abl & 16) ? 0 : s_src[j * CM + a_r + ARP * i]) * p.ldx + c0;  // (bit 16, timing only: every gather reads row 0)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ga[slot][i]) : "v"(src));
      }
    };
    auto gload_w = [&](int slot, bool advance = true) __attribute__((always_inline)) {
      const int so = g_wo;
      if constexpr (!W_T) {
        if constexpr (PERM) {  // (the compiler does not hold the item's offset in a scalar register there: it goes into the lane offset)
#pragma unroll
          for (int q = 0; q < SPW; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int vo = wv[q][e] + so;
              asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(gw[slot][q][e]) : "v"(vo), "s"(rw));
            }
        } else {
          const int sso = __builtin_amdgcn_readfirstlane(so);  // (uniform; an SGPR operand)
#pragma unroll
          for (int q = 0; q < SPW; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(gw[slot][q][e]) : "v"(wv[q][e]), "s"(rw), "s"(sso));
        }
      } else {  // (the offset is added to the lane offset: see above)
#pragma unroll
        for (int q = 0; q < SPW; ++q)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int vo = wv[q][h] + so;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(gwt[slot][q][h]) : "v"(vo), "s"(rw));
          }
      }
      if (!advance) return;  // (P3 prologue: the weight loads of the step before the first item exists -- same count, no move)
      if (g_left > 0) {  // uniform
        --g_left, g_wo += g_wstep;
        if constexpr (!P3) g_xo += 4u * BK;
        if (++g_cc == ncc) {
          g_cc = 0, ++g_ka;
          g_wo = wbase_of(g_ka);
          if constexpr (!P3) {
            g_xo = 4u * (unsigned)(cbeg * BK);
            if constexpr (CPF) load_src();  // (used by the NEXT step's gload_a: the read has the barrier wait to arrive)
          }
        }
      }
    };
    // P3: the gather stage has an iterator of its own (it runs one item further ahead than the weight stage)
    static_assert(!P3 || CPF, "the three-stage form gathers through the per-offset lane offsets");
    int r_ka = 0, r_cc = 0, r_left = n_items - 1;
    unsigned r_xo = 4u * (unsigned)(cbeg * BK);
    auto load_src3 = [&]() __attribute__((always_inline)) {
      const int j = nib(act_lo, act_hi, r_ka);
#pragma unroll
      for (int i = 0; i < NA; ++i) sv[i] = (unsigned)s_src[j * CM + a_r + ARP * i] * (4u * (unsigned)p.ldx) + 16u * (unsigned)a_cc;
    };
    auto gload_a3 = [&](int slot) __attribute__((always_inline)) {
      const unsigned long long xb = (unsigned long long)p.x + r_xo;
#pragma unroll
      for (int i = 0; i < NA; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ga[slot][i]) : "v"(sv[i]), "s"(xb));
      if (r_left > 0) {  // uniform; past the end the last item is re-read
        --r_left, r_xo += 4u * BK;
        if (++r_cc == ncc) {
          r_cc = 0, ++r_ka, r_xo = 4u * (unsigned)(cbeg * BK);
          load_src3();  // (used by the NEXT step's gather request)
        }
      }
    };
    auto sts = [&](int slot, int buf) __attribute__((always_inline)) {
      float *a = sA + buf * CM * CLDA + a_sw;
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4 *>(&a[(a_r + ARP * i) * CLDA]) = ga[slot][i];
    };

    // The wait itself names no register; the empty asm statements behind it are the definition points of the ring's registers
    // for everything that follows (volatile asm statements keep their order, so nothing that reads a register can be
    // scheduled above its wait).
    auto tie_rows = [&](int sl) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(ga[sl][i]));
    };
    auto tie_weights = [&](int sl) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < SPW; ++q) {
        if constexpr (W_T) {
#pragma unroll
          for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(gwt[sl][q][h]));
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(gw[sl][q][e]));
        }
      }
    };
    static_assert(3 * NA + 2 * NW <= 63 && 2 * NA + 3 * NW <= 63, "vmcnt is six bits");
    auto wait_rows = [&](int sl) __attribute__((always_inline)) {  // the gathered rows of ring slot sl have arrived
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DBG_WAIT0 ? 0 : NA + 2 * NW));
      tie_rows(sl);
    };
    auto wait_weights = [&](int sl) __attribute__((always_inline)) {  // the weight fragment of ring slot sl has arrived
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DBG_WAIT0 ? 0 : 3 * NA + 2 * NW));
      tie_weights(sl);
    };
    auto drain = [&](int sl) __attribute__((always_inline)) {  // nothing of ring slot sl is in flight any more
      asm volatile("s_waitcnt vmcnt(0)");
      tie_rows(sl), tie_weights(sl);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    static_assert(CD == 3, "the steady-state loop is unrolled by hand");
    // items 0 .. CD - 1 requested (slot = item % CD), item 0 -> LDS
    // every load the COMPILER tracks (table entries, row permutation) is complete before the ring starts: a pending score
    // on a register the ring re-uses would make it insert s_waitcnt vmcnt(0) at the loop header -- a full drain every pass
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    float4 u[P3 ? 2 : 1][NBLK][2];  // MFMA operands of an item (P3: of the item being multiplied and of the next one)
    auto xr_at = [&](const float *a, int blk, int half) __attribute__((always_inline)) {
      return *reinterpret_cast<const float4 *>(a + blk * 16 * CLDA + (half ? r_sw1 : r_sw0));
    };
    if constexpr (P3) {
      if (n_items > 0) {  // uniform
        // the steps "before the first": every step issues [rows of item s + 4][weights of item s + 3] -- the same NA + NW loads in the
        // same order as a real one, so the hand-counted waits hold from the first real step on (weights of item -1: item 0's again)
        load_src3();
        gload_a3(0), gload_w(2, false);                  // s = -4
        gload_a3(1), gload_w(0);                         // s = -3
        wait_rows(0), sts(0, 0), gload_a3(2), gload_w(1);  // s = -2: item 0 -> stage 0
        wait_rows(1), sts(1, 1), gload_a3(0), gload_w(2);  // s = -1: item 1 -> stage 1
      }
      MINK_LDS_BARRIER();
      if (n_items > 0) {
        const float *a0 = sA + n * CLDA;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) u[0][b][0] = xr_at(a0, b, 0), u[0][b][1] = xr_at(a0, b, 1);
      }
    } else {
    if (n_items > 0) {  // uniform
      if constexpr (CPF) load_src();
      gload_a(0), gload_w(0), gload_a(1), gload_w(1), gload_a(2), gload_w(2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NA + 3 * NW));
      tie_rows(0);
      sts(0, 0);
    }
    MINK_LDS_BARRIER();
    }
    // CCIN: float index (row * CLDC) of the tile row behind each block of the offset being multiplied, for this lane's n
    int crow[NBLK];
    auto crows = [&](int j) __attribute__((always_inline)) {
      static_assert(!CCIN || CM / 16 == 4, "four row indices per lane in one word");
      const unsigned pk = *reinterpret_cast<const unsigned *>(s_lrow + j * CM + n * (CM / 16));
#pragma unroll
      for (int b = 0; b < NBLK; ++b) crow[b] = (int)((pk >> (8 * b)) & 255u) * CLDC;
    };
    auto cget = [&](auto lo_c, auto hi_c, int nbx) __attribute__((always_inline)) {
      constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
#pragma unroll
      for (int q = 0; q < SPW; ++q)
#pragma unroll
        for (int b = lo; b < hi; ++b)
          if (b < nbx) acc[q][b] = *reinterpret_cast<const f32x4 *>(myC + 16 * q + crow[b]);
    };
    auto cput = [&](auto lo_c, auto hi_c, int nbx) __attribute__((always_inline)) {
      constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
#pragma unroll
      for (int q = 0; q < SPW; ++q)
#pragma unroll
        for (int b = lo; b < hi; ++b)
          if (b < nbx) *reinterpret_cast<f32x4 *>(myC + 16 * q + crow[b]) = acc[q][b];
    };
    if constexpr (CCIN) {
      if (n_items > 0) {  // uniform: the first offset of the round
        const int nb0 = nib(nbs_lo, nbs_hi, 0);
        crows(nib(act_lo, act_hi, 0));
        cget(S0{}, S2{}, nb0);
        if (nb0 > 2) cget(S2{}, std::integral_constant<int, 4>{}, nb0);
      }
    }
    if constexpr (ABL) {
      if (p.trace && t_loop == 0) t_loop = __builtin_readcyclecounter();
    }
    int ka = 0, cc = 0;
    int c_j = nib(act_lo, act_hi, 0), c_nb = nib(nbs_lo, nbs_hi, 0);  // rulebook slot and block count of the offset being multiplied
    auto step = [&](int it, auto slot_c, auto nslot_c, auto par_c) __attribute__((always_inline)) {
      constexpr int slot = decltype(slot_c)::value;
      const int j = c_j, nb = c_nb;  // nb >= 1
      constexpr int nslot = decltype(nslot_c)::value;
      constexpr int par = P3 ? decltype(par_c)::value : 0;
      if constexpr (P3) {
        // ring slot / LDS stage of item i: i % 3.  Item it + 2 has arrived: into its stage (free since the barrier of step it - 2,
        // behind which nobody reads item it - 1 any more); item it + 4 is requested into the registers of item it + 1 (stored
        // at step it - 1); the operands of item it + 1 (published by the barrier of step it - 1) go into the other register set
        constexpr int s2 = (slot + 2) % 3;
        wait_rows(s2);
        sts(s2, s2);
        gload_a3(nslot);
        const float *an = sA + nslot * CM * CLDA + n * CLDA;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) u[par ^ 1][b][0] = xr_at(an, b, 0), u[par ^ 1][b][1] = xr_at(an, b, 1);
        __builtin_amdgcn_sched_barrier(0);  // (the reads stay in front of the burst: they have its length to arrive)
      } else {
      wait_rows(nslot);
      sts(nslot, (it + 1) & 1);  // item it + 1 (after the last item: a re-read copy nobody multiplies)
      }
      const float *a = sA + (it & 1) * CM * CLDA + n * CLDA;
      // item it + CD takes the registers of item it NOW, not after the MFMAs: the compiler's s_waitcnt before the next
      // step's LDS stores is vmcnt(0..9) where the ring would allow 18 (it merges the loop-carried load scores
      // conservatively), so a load issued at the end of a step was waited for a few hundred cycles later; issued here
      // it has this step's MFMAs to arrive
      if constexpr (!P3) gload_a(slot);
      wait_weights(slot);
      auto mfma8 = [&](auto q_c, f32x4 &c, const float4 &u0, const float4 &u1) __attribute__((always_inline)) {
        constexpr int q = decltype(q_c)::value;  // strip of this wave
        auto wf = [&](auto e_c) __attribute__((always_inline)) -> float {
          constexpr int e = decltype(e_c)::value;
          if constexpr (W_T) return gwt[slot][q][e >> 2][e & 3];
          else return gw[slot][q][e];
        };
        using std::integral_constant;
        if constexpr (MATH == 1) {  // channels 16 (e >> 2) + 4 kq + (e & 3) of the item in slot e of both fragments
          const float wv8[8] = {wf(integral_constant<int, 0>{}), wf(integral_constant<int, 1>{}), wf(integral_constant<int, 2>{}), wf(integral_constant<int, 3>{}),
                                wf(integral_constant<int, 4>{}), wf(integral_constant<int, 5>{}), wf(integral_constant<int, 6>{}), wf(integral_constant<int, 7>{})};
          const float xv8[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack_bf16x8(wv8), pack_bf16x8(xv8), c, 0, 0, 0);
          return;
        }
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf(integral_constant<int, 0>{}), u0.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf(integral_constant<int, 1>{}), u0.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf(integral_constant<int, 2>{}), u0.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf(integral_constant<int, 3>{}), u0.w, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf(integral_constant<int, 4>{}), u1.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf(integral_constant<int, 5>{}), u1.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf(integral_constant<int, 6>{}), u1.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf(integral_constant<int, 7>{}), u1.w, c, 0, 0, 0);
      };
      auto xr = [&](int blk, int half) __attribute__((always_inline)) { return xr_at(a, blk, half); };
      // The operands of every block are read up front (rows past the compacted count hold stale data that is read but
      // never multiplied), then one accumulation chain per block that exists.  Each accumulator is written in ONE
      // place per item: the paired form (two blocks' MFMAs alternating, a single-block tail) made the compiler keep
      // the merged accumulators in other registers and copy them back -- 32 v_mov_b64 per item beside 24 MFMAs, on a
      // part where every vector-ALU instruction is a slot the matrix pipe loses; the 8-cycle gap between dependent
      // 16x16x4 MFMAs of one chain is filled by the other three waves of the SIMD.
      if constexpr (!P3) {
#pragma unroll
        for (int b = 0; b < NBLK; ++b) u[0][b][0] = xr((abl & 128) ? 0 : b, 0), u[0][b][1] = xr((abl & 128) ? 0 : b, 1);  // (bit 7, timing only: one block's operands)
      }
#pragma unroll
      for (int b = 0; b < NBLK; ++b)
        if ((b == 0 || b < nb) && !(abl & 32)) {  // uniform; an item has at least one block (bit 5, timing only: no matrix work)
          mfma8(std::integral_constant<int, 0>{}, acc[0][b], u[par][b][0], u[par][b][1]);
          if constexpr (SPW > 1) mfma8(std::integral_constant<int, 1>{}, acc[SPW - 1][b], u[par][b][0], u[par][b][1]);
        }
      if (++cc == ncc) {  // the offset is complete: add the blocks into the tile rows they belong to (padding -> row CM)
        cc = 0, ++ka;
        if (ka < na) c_j = nib(act_lo, act_hi, ka), c_nb = nib(nbs_lo, nbs_hi, ka);  // uniform (the next offset's slot and blocks)
        if (!(abl & 64)) {  // (bit 6, timing only: no scatter)
        if constexpr (CCIN) {
          // (each accumulator is named in ONE place per role: see the note at `scatter` below)
          cput(S0{}, S2{}, nb);
          if (nb > 2) cput(S2{}, std::integral_constant<int, 4>{}, nb);
          if (ka < na) {  // uniform: the next offset's accumulators start as the rows they belong to
            const int j2 = c_j, nb2 = c_nb;
            crows(j2);
            cget(S0{}, S2{}, nb2);
            if (nb2 > 2) cget(S2{}, std::integral_constant<int, 4>{}, nb2);
          }
        } else {
        auto scatter = [&](auto lo_c, auto hi_c) __attribute__((always_inline)) {
          constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
          int lr[hi - lo];
#pragma unroll
          for (int b = lo; b < hi; ++b) lr[b - lo] = b < nb ? (int)s_lrow[j * CM + n * (CM / 16) + b] : CM;
#pragma unroll
          for (int q = 0; q < SPW; ++q) {
            float4 c[hi - lo];
#pragma unroll
            for (int b = lo; b < hi; ++b) c[b - lo] = *reinterpret_cast<const float4 *>(myC + 16 * q + lr[b - lo] * CLDC);
#pragma unroll
            for (int b = lo; b < hi; ++b) {
              float4 &v = c[b - lo];
              v.x += acc[q][b][0], v.y += acc[q][b][1], v.z += acc[q][b][2], v.w += acc[q][b][3];
              *reinterpret_cast<float4 *>(myC + 16 * q + lr[b - lo] * CLDC) = v;
              acc[q][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
        };
        // (each accumulator is named in ONE place: two call sites that differ only in the block index get tail-merged
        //  behind a pointer phi, and the accumulators then live in scratch memory)
        scatter(S0{}, S2{});
        if (nb > 2) scatter(S2{}, std::integral_constant<int, 4>{});
        if constexpr (NBLK > 4) {
          if (nb > 4) scatter(std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{});
          if (nb > 6) scatter(std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{});
        }
        }
        }
      }
      gload_w(slot);
      MINK_LDS_BARRIER();
    };
    if constexpr (P3) {  // (ring slot = item % 3, operand set = item % 2: six step bodies)
      for (int base = 0; base < n_items; base += 6) {
        step(base, S0{}, S1{}, S0{});
        if (base + 1 < n_items) step(base + 1, S1{}, S2{}, S1{});
        if (base + 2 < n_items) step(base + 2, S2{}, S0{}, S0{});
        if (base + 3 < n_items) step(base + 3, S0{}, S1{}, S1{});
        if (base + 4 < n_items) step(base + 4, S1{}, S2{}, S0{});
        if (base + 5 < n_items) step(base + 5, S2{}, S0{}, S1{});
      }
    } else {
      for (int base = 0; base < n_items; base += CD) {
        step(base, S0{}, S1{}, S0{});
        if (base + 1 < n_items) step(base + 1, S1{}, S2{}, S0{});
        if (base + 2 < n_items) step(base + 2, S2{}, S0{}, S0{});
      }
    }
    if (n_items > 0) drain(0), drain(1), drain(2);  // (uniform) the ring's last re-read loads
    if (!PERM && !SK) break;
  }

  if constexpr (ABL) {
    if (p.trace) t_epi = __builtin_readcyclecounter();
  }
  // ---- epilogue: y / slab = C (+ bias), column statistics of the tile for the batch norm that follows
  const bool direct = !SK && gz == 1;
  float *dst = direct ? p.y : p.ws + (int64_t)(SK ? sk_slab : (int)bz) * p.n_out * p.cout;
  const int ldd = direct ? p.ldy : p.cout;
  const int c4 = tid & 15, rg = tid >> 4;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (direct && p.bias) bias = *reinterpret_cast<const float4 *>(p.bias + n0 + 4 * c4);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  constexpr int RG = T / 16;  // row groups of the epilogue (16 lanes of 16 bytes per row)
#pragma unroll
  for (int jr = 0; jr < CM / RG; ++jr) {
    const int r = rg + RG * jr;
    const int64_t row = PERM ? (int64_t)s_orow[r] : (r < rows_here ? o0 + r : -1);
    if (row >= 0) {
      const float4 u = *reinterpret_cast<const float4 *>(&sC[r * CLDC + 4 * c4]);
      const float4 o = make_float4(u.x + bias.x, u.y + bias.y, u.z + bias.z, u.w + bias.w);
      *reinterpret_cast<float4 *>(&dst[row * ldd + n0 + 4 * c4]) = o;
      s.x += o.x, s.y += o.y, s.z += o.z, s.w += o.w;
      q.x += o.x * o.x, q.y += o.y * o.y, q.z += o.z * o.z, q.w += o.w * o.w;
    }
  }
  if (!PERM && p.stats && direct) {  // uniform
    float *red = sA;  // [RG row groups][2][64]; the tiles are dead (last barrier)
    *reinterpret_cast<float4 *>(&red[(rg * 2 + 0) * BN + 4 * c4]) = s;
    *reinterpret_cast<float4 *>(&red[(rg * 2 + 1) * BN + 4 * c4]) = q;
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid >> 6, c = tid & 63;
      float t = 0.f;
      for (int g = 0; g < RG; ++g) t += red[(g * 2 + which) * BN + c];
      p.stats[((int64_t)bx * 2 + which) * p.cout + n0 + c] = t;
    }
  }
  if constexpr (SK) {
    for (int zsl = 1; zsl <= sk_zero; ++zsl) {  // (uniform) slabs of this tile that no worker wrote: zeros for the consumers
      float *dz = p.ws + (int64_t)(sk_slab + zsl) * p.n_out * p.cout;
#pragma unroll
      for (int jr = 0; jr < CM / RG; ++jr) {
        const int r = rg + RG * jr;
        if (r < rows_here) *reinterpret_cast<float4 *>(&dz[(o0 + r) * p.cout + n0 + 4 * c4]) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    sk_f += sk_nk;
    if (sk_f < sk_end) {  // uniform
      __syncthreads();  // (the next segment zeroes the C tile this epilogue read)
      goto sk_next_segment;
    }
  }
  }  // segment
  if constexpr (ABL) {
    if (p.trace) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the tile's stores have left: what the slot waits for before it is handed on)
      if (tid == 0) {
        const unsigned long long lin = blockIdx.x + (unsigned long long)gridDim.x * (blockIdx.y + (unsigned long long)gridDim.y * blockIdx.z);
        unsigned long long *t = p.trace + 5 * lin;
        t[0] = t_start, t[1] = t_loop, t[2] = t_epi, t[3] = __builtin_readcyclecounter();
        t[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);  // HW_ID, XCC_ID
      }
    }
  }
}

// ------------------------------------------------------------------------------ wgrad
// dW[k][ci][co] = sum over pairs (i,o) of offset k:  x[i][ci] * dy[o][co].
//
// Workgroup = (group of G offsets) x (64x64 ci/co super-tile) x (row range).  Per 128-row tile
// of the range the dy tile is staged ONCE in LDS and shared by the G offsets; for each offset
// the rows that really have a neighbour are compacted with a wave64 ballot + prefix rank (the
// rulebook of that tile, built on the fly), only those x rows are gathered, and the MFMA
// contraction runs over the compacted pairs -- no work is spent on missing neighbours.
// Partial sums stay in registers over the whole row range; row splits are reduced
// deterministically through a slab workspace (no atomics).
constexpr int WT = 64;      // ci / co super-tile
constexpr int WROWS = 128;  // rows per tile
constexpr int WLD = WT + 4; // LDS row stride (floats)

struct WgradParams {
  const float *x;
  const float *dy;
  const int *nbr;
  float *out;  // dw (nsplit == 1) or workspace [nsplit][K][cin][cout]
  int64_t n_out, rows_per_split;
  int ldx, cin, ldy, cout, K, ct_tiles, ngroups, ablate;
  unsigned x_bytes, dy_bytes, nbr_bytes;  // buffer descriptors (streaming kernels; tiled kernel when buf_ok)
  int buf_ok;                             // every byte size < 2^31 and n_in < 2^24: 32-bit offset arithmetic is safe
  // streaming kernel, FUSE: `dy` is the batch-norm INPUT y and the B operand is recomputed on the fly as the
  // input gradient of  pool(relu(bn(y)))  from the pooled gradient -- that gradient is never materialised
  const float *dyp;   // [n_pool][cout] gradient of the pooled output
  const int *in2out;  // [n_out] fine row -> pooled row
  const float *mean, *invstd, *gamma, *beta, *dgamma, *dbeta;
  float inv_n;
  unsigned dyp_bytes, i2o_bytes;
};

// G: offsets per workgroup.  NARROW: cin <= 32 -- the x tile is 32 floats wide and the two
// wave rows split the pair list instead of the (empty) second ci tile.
// Software pipeline per tile: all G pair lists are built up front from one nbr load per row;
// then for each offset the x gather of offset g+1 is in flight (registers) while the MFMAs
// of offset g run from LDS.
// BUF (VEC operands whose byte sizes fit 31 bits, fewer than 2^24 input rows): table, dy and gathered x rows come through
// raw buffer loads -- a missing neighbour / a row past the end / a column past the width is an out-of-range offset that
// returns zeros, so a load costs one 24-bit multiply-add instead of a 64-bit address, a select and (as the guarded form
// compiled) a branch around every load.
// (Round 5, measured and removed: 64-row tiles -- 36 KB of LDS, four workgroups per CU instead of two: l1.conv2 102.5 us against 96.7,
//  l3 93 / 82, only l4 63 / 66.5; the step 3.61-3.62 ms against 3.57-3.58.  Half the rows per tile pad an offset's pairs to 16 twice as
//  often and pay the tile's three barriers twice per 128 rows; occupancy was not what held this kernel back.  The bf16 form, whose
//  matrix work is a sixteenth, does gain from four workgroups per CU: wgrad16_kernel.)
template <int G, bool NARROW, bool VEC, bool BUF = false>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
  static_assert(!BUF || VEC, "buffer loads are 16 bytes wide");
  constexpr int XW = NARROW ? 32 : 64;      // x tile width (floats)
  constexpr int XLD = XW + 4;               // LDS row stride
  constexpr int XC4 = XW / 4;               // float4 columns per x row
  constexpr int XRP = 256 / XC4;            // x rows per staging pass
  constexpr int XNI = WROWS / XRP;          // staging passes (float4 registers per thread)
  constexpr int LL = WROWS;                 // list length (pair count is padded to 16, <= 128)
  __shared__ __attribute__((aligned(16))) float sD[WROWS * WLD];
  __shared__ __attribute__((aligned(16))) float sX[WROWS * XLD];
  __shared__ __attribute__((aligned(16))) int s_row[G * LL];
  __shared__ int s_src[G * LL];
  __shared__ int s_cnt[G * 2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroup -> (offset group / weight tile bx, row split by).  All workgroups of one row split read the same rows of x and
  // dy; in plain launch order (x fastest) they are dealt round-robin to the eight XCDs and every L2 fetches those rows for
  // itself (layer 1: 175 MB of traffic beyond L2 for 24 MB of operands, PMC r04).  When the split count is a multiple of
  // eight the workgroups of a split are given to consecutive slots of ONE XCD (as stream_slot does for the stem).
  unsigned wbx = blockIdx.x, wby = blockIdx.y;
  if (!(p.ablate & 4096) && (gridDim.y & 7u) == 0u) {  // uniform
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7u, slot = lin >> 3;
    wbx = slot % gridDim.x, wby = (slot / gridDim.x) * 8u + xcd;
  }
  const int grp = wbx % p.ngroups;
  const int tile_id = wbx / p.ngroups;
  const int ci0 = (tile_id / p.ct_tiles) * WT, co0 = (tile_id % p.ct_tiles) * WT;
  const int k0 = grp * G;
  const int ng = min(G, p.K - k0);  // offsets handled by this workgroup (uniform)
  const int64_t rbeg = (int64_t)wby * p.rows_per_split;
  const int64_t rend = min(p.n_out, rbeg + p.rows_per_split);

  const int d_c4 = tid & 15, d_rr = tid >> 4;       // dy staging: float4 column, rows d_rr + 16 i
  const int x_c4 = tid % XC4, x_rr = tid / XC4;     // x staging: float4 column, rows x_rr + XRP i

  const int wa = wave >> 1, wn = wave & 1, h = lane >> 5, col = lane & 31;
  const int wm = NARROW ? 0 : wa;

  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = (f32x16){0};

  float4 rx[XNI] = {};
  const i32x4 bx = raw_rsrc(p.x, BUF ? p.x_bytes : 0u), bd = raw_rsrc(p.dy, BUF ? p.dy_bytes : 0u), bn = raw_rsrc(p.nbr, BUF ? p.nbr_bytes : 0u);
  const unsigned ldx4 = 4u * (unsigned)p.ldx, ldy4 = 4u * (unsigned)p.ldy, K4 = 4u * (unsigned)p.K;
  const unsigned x_coff = ci0 + 4 * x_c4 < p.cin ? 4u * (unsigned)(ci0 + 4 * x_c4) : 0x80000000u;
  const unsigned d_coff = co0 + 4 * d_c4 < p.cout ? 4u * (unsigned)(co0 + 4 * d_c4) : 0x80000000u;
  auto gather = [&](int g) {  // x rows of the compacted pairs of offset g -> registers
    if (p.ablate & 64) return;
    if constexpr (BUF) {
      // (list entries behind the padded pair count are stale rows of an earlier offset: loaded, stored, never multiplied;
      //  the tail pairs carry -1 = row 0xFFFFFF, beyond x)
#pragma unroll
      for (int i = 0; i < XNI; ++i)
        rx[i] = __builtin_bit_cast(float4, raw_load_v4(bx, (int)(__umul24((unsigned)s_src[g * LL + x_rr + XRP * i], ldx4) + x_coff), 0, 0));
      return;
    }
    const int m = s_cnt[2 * g] + s_cnt[2 * g + 1];
    const int mpad = (m + 15) & ~15;
#pragma unroll
    for (int i = 0; i < XNI; ++i) {
      const int pr = x_rr + XRP * i;
      int src = -1;
      if (pr < mpad) src = s_src[g * LL + pr];
      const int ci = ci0 + 4 * x_c4;
      if (VEC)
        rx[i] = ld4_sel(p.x, (int64_t)src * p.ldx + ci, src >= 0 && ci < p.cin);
      else
        rx[i] = src >= 0 ? ld4_guard(p.x + (int64_t)src * p.ldx + ci, p.cin - ci, false) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < XNI; ++i) *reinterpret_cast<float4 *>(&sX[(x_rr + XRP * i) * XLD + 4 * x_c4]) = rx[i];
  };

  // BUF (round 5): the table entries and the dy tile of the NEXT tile are requested before the MFMAs of this tile's last
  // offset and wait in registers: the tile prologue no longer starts with a memory round trip (one of its two).
  constexpr bool PF = BUF && G <= 3;  // (nine accumulators leave no registers for it)
  int nb_n[G];
  float4 dy_n[8];
  auto request_tile = [&](int64_t r0) __attribute__((always_inline)) {
    const int64_t row = r0 + tid;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const bool ok = tid < WROWS && row < rend && g < ng;
      const int v = raw_load_i32(bn, (int)(ok ? (unsigned)row * K4 + 4u * (unsigned)(k0 + g) : 0x80000000u), 0, 0);
      nb_n[g] = ok ? v : -1;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t rw_ = r0 + d_rr + 16 * i;
      dy_n[i] = __builtin_bit_cast(float4, raw_load_v4(bd, (int)((rw_ < rend ? (unsigned)rw_ * ldy4 : 0x80000000u) + d_coff), 0, 0));
    }
  };
  if constexpr (PF) {
    if (rbeg < rend) request_tile(rbeg);
  }
  for (int64_t r0 = rbeg; r0 < rend; r0 += WROWS) {
    __syncthreads();  // previous tile fully consumed
    // ---- this tile's neighbour entries (one row per thread of waves 0/1) and dy tile
    int nb[G], rank[G];
    if constexpr (PF) {
#pragma unroll
      for (int g = 0; g < G; ++g) nb[g] = nb_n[g];
    } else if (tid < WROWS) {
      const int64_t row = r0 + tid;
      if constexpr (BUF) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const bool ok = row < rend && g < ng;
          const int v = raw_load_i32(bn, (int)(ok ? (unsigned)row * K4 + 4u * (unsigned)(k0 + g) : 0x80000000u), 0, 0);
          nb[g] = ok ? v : -1;
        }
      } else {
#pragma unroll
        for (int g = 0; g < G; ++g) nb[g] = (row < rend && g < ng) ? p.nbr[row * p.K + k0 + g] : -1;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = d_rr + 16 * i;
      const int64_t row = r0 + r;
      const int co = co0 + 4 * d_c4;
      float4 v;
      if constexpr (PF)
        v = dy_n[i];
      else if constexpr (BUF)
        v = __builtin_bit_cast(float4, raw_load_v4(bd, (int)((row < rend ? (unsigned)row * ldy4 : 0x80000000u) + d_coff), 0, 0));
      else if (VEC)
        v = ld4_sel(p.dy, row * p.ldy + co, row < rend && co < p.cout);
      else
        v = row < rend ? ld4_guard(p.dy + row * p.ldy + co, p.cout - co, false) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4 *>(&sD[r * WLD + 4 * d_c4]) = v;
    }
    // ---- rulebook of the tile: wave64 ballot + prefix rank per offset
    if (tid < WROWS) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const unsigned long long mm = __ballot(nb[g] >= 0);
        rank[g] = wave_rank(mm);
        if (lane == 0) s_cnt[2 * g + wave] = __popcll(mm);
      }
    }
    __syncthreads();
    if (tid < WROWS) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (nb[g] >= 0) {
          const int pos = (wave == 1 ? s_cnt[2 * g] : 0) + rank[g];
          s_row[g * LL + pos] = tid * WLD;  // LDS offset of the dy row
          s_src[g * LL + pos] = nb[g];
        }
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) {  // tail pairs: dy row 0 times a zero x row
        const int m = s_cnt[2 * g] + s_cnt[2 * g + 1];
        const int t = tid - WROWS;
        if (t < ((m + 15) & ~15) - m) {
          s_row[g * LL + m + t] = 0;
          s_src[g * LL + m + t] = -1;
        }
      }
    }
    __syncthreads();
    gather(0);
    stash();
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < ng) {  // uniform
      if (g + 1 < ng) gather(g + 1);  // in flight during the MFMAs below
      else if constexpr (PF) {
        if (r0 + WROWS < rend) request_tile(r0 + WROWS);  // (uniform) the next tile's table entries and dy rows, likewise
      }
      const int m = s_cnt[2 * g] + s_cnt[2 * g + 1];
      const int nsteps = ((m + 15) & ~15) >> 1;  // multiple of 8; lane half h takes pairs [h*nsteps, (h+1)*nsteps)
      const int *lrow = s_row + g * LL + h * nsteps;
      const float *xa = sX + (h * nsteps) * XLD + 32 * wm + col;
      const float *db = sD + 32 * wn + col;
      const int sbeg = NARROW ? wa * (nsteps >> 1) : 0;
      const int send = NARROW ? sbeg + (nsteps >> 1) : nsteps;
      // 4 MFMAs per trip.  Software pipeline over the trips (round 5): the dy-row offsets of trip t + 2 and the eight operands
      // of trip t + 1 are requested before the MFMAs of trip t -- as one trip at a time (round 4) every trip stood behind two
      // dependent LDS round trips (row offsets -> dy values) plus a third for its second operand pair: ~300 exposed cycles
      // per 256 of matrix work at two waves per SIMD (pipe busy 0.33-0.38).  Lists are padded to whole trips; the reads one
      // and two trips past the end are clamped to the last trip and never used.
#ifndef MINK_WPIPE
#define MINK_WPIPE 1
#endif
      if (!MINK_WPIPE) {  // (A/B builds: the round-4 loop)
        if (!(p.ablate & 128))
          for (int s = sbeg; s < send; s += 4) {
            const int4 ro = *reinterpret_cast<const int4 *>(lrow + s);
            const float a0 = xa[(s + 0) * XLD], a1 = xa[(s + 1) * XLD], a2 = xa[(s + 2) * XLD], a3 = xa[(s + 3) * XLD];
            const float b0 = db[ro.x], b1 = db[ro.y], b2 = db[ro.z], b3 = db[ro.w];
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc[g], 0, 0, 0);
          }
      } else if (!(p.ablate & 128) && sbeg < send) {  // uniform
        const int last = send - 4;
        auto rows_of = [&](int s_) __attribute__((always_inline)) { return *reinterpret_cast<const int4 *>(lrow + min(s_, last)); };
        struct Ops { float a0, a1, a2, a3, b0, b1, b2, b3; };
        auto ops_of = [&](int s_, const int4 &ro) __attribute__((always_inline)) {
          const int t = min(s_, last);
          return Ops{xa[(t + 0) * XLD], xa[(t + 1) * XLD], xa[(t + 2) * XLD], xa[(t + 3) * XLD], db[ro.x], db[ro.y], db[ro.z], db[ro.w]};
        };
        auto mfma4 = [&](const Ops &o) __attribute__((always_inline)) {
          __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks every read to its use: the round-4 loop again)
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a0, o.b0, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a1, o.b1, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a2, o.b2, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a3, o.b3, acc[g], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        };
        // two trips per pass, two operand sets in turn: no register rotation beside the MFMAs
        int4 r1 = rows_of(sbeg);
        Ops A = ops_of(sbeg, r1);
        r1 = rows_of(sbeg + 4);
        for (int s = sbeg; s < send; s += 8) {
          const int4 r2 = rows_of(s + 8);
          const Ops B = ops_of(s + 4, r1);
          mfma4(A);
          if (s + 4 < send) {  // uniform
            r1 = rows_of(s + 12);
            A = ops_of(s + 8, r2);
            mfma4(B);
          }
        }
      }
      if (g + 1 < ng) {
        __syncthreads();  // everyone done reading sX
        stash();
        __syncthreads();
      }
      }
    }
  }

  // ---- epilogue: (narrow: add the two pair halves through LDS) then store the partial slab
  float *dst = p.out + (int64_t)wby * p.K * p.cin * p.cout;
  const int co = co0 + 32 * wn + col;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g < ng) {  // uniform
      const int k = k0 + g;
      if (NARROW) {
        __syncthreads();
        if (wa == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sD[(wn * 16 + r) * 64 + lane] = acc[g][r];
        }
        __syncthreads();
        if (wa == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[g][r] += sD[(wn * 16 + r) * 64 + lane];
        }
      }
      if (!NARROW || wa == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = ci0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (ci < p.cin && co < p.cout) dst[((int64_t)k * p.cin + ci) * p.cout + co] = acc[g][r];
        }
      }
    }
  }
}

// ----------------------------------------------------------------- streaming wgrad (cin <= 32)
// The stem's weight gradient (28 -> 64 over ~8e5 rows, the largest kernel of a training step)
// as a plain streamed GEMM  dW[k] (32 x 32 per wave) += X[nbr[.][k]]^T (32 x 2) * dY (2 x 32):
// the MFMA operands are loaded straight from global memory in the 32x32x2 register layout
// (a half-wave reads one 128-byte row segment), so there is no LDS tile, no pair list and no
// barrier in the loop.  Missing neighbours and rows past the end become out-of-range buffer
// offsets, which a buffer load returns as 0:  offset = (nb & 0xFFFFFF) * 4 ldx + column is
// >= the size of x for nb = -1 as long as n_in < 2^24 and ldx < 64 (checked by the launcher).
// D row pairs of operands and D pairs of neighbour rows are in flight per wave.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ptr), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

// Workgroup -> (offset group, cout tile, row split) of the streaming kernels.  Workgroups are dealt to the eight XCDs
// round-robin in launch order, and the three offset groups of one row split read the same conv output, pooled gradient
// and table rows: with the plain (x = group, y = split) order they land on three different XCDs and every L2 fetches
// those rows for itself (r01: 2.0 GB of HBM reads against 0.43 GB algorithmic).  When the split count is a multiple of
// eight the groups of a split are given to consecutive slots of ONE XCD instead.
struct StreamSlot {
  int grp, cot, split;
};
__device__ __forceinline__ StreamSlot stream_slot(const WgradParams &p) {
  StreamSlot s;
  if ((int)gridDim.x == p.ngroups && (gridDim.y & 7) == 0 && !(p.ablate & 4096)) {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, slot = lin >> 3;
    s.grp = (int)(slot % (unsigned)p.ngroups), s.cot = 0, s.split = (int)((slot / (unsigned)p.ngroups) * 8 + xcd);
  } else {
    s.grp = blockIdx.x % p.ngroups, s.cot = blockIdx.x / p.ngroups, s.split = blockIdx.y;
  }
  return s;
}

// FLAT: the (offset, channel) axis of dW is tiled as ONE flattened axis f = k * cin + ci in runs of 32 (the forward
// kernel's FLAT=28 idea): 27 x 28 = 756 rows are 24 tiles instead of 27 offset tiles padded from 28 to 32 channels --
// a ninth fewer MFMAs, gathers and address instructions.  A lane's row of a tile then belongs to one of two offsets,
// so every lane picks ITS neighbour entry from the staged table row (a per-lane LDS read at an address that is a
// constant of (lane, tile)) instead of the half-wave sharing a broadcast; a lane past the end of the axis adds an
// out-of-range column offset.  A group is eight tiles = 256 flat rows = at most 16 offsets from kb = 256 grp / cin.
template <int D, bool FUSE = false, bool FLAT = false>
__global__ __launch_bounds__(256, 2) void wgrad_stream_kernel(WgradParams p) {
  static_assert(D % 2 == 0, "the neighbour staging ring has two slots");
  constexpr int G = FLAT ? 8 : 9;        // K == 27: three groups of nine offsets / of eight 32-row tiles of the flat axis
  constexpr int NK = FLAT ? 16 : 9;      // table entries of a row staged per group
  constexpr unsigned OOB = 0x80000000u;  // beyond any descriptor this kernel is launched with
  __shared__ float sR[2 * 16 * 64];
  // wave-private staging of the neighbour entries of one row pair: one lane per entry loads
  // them, every lane of the half reads them back (LDS broadcast) -- a same-address vector load
  // would cost as much L1 return bandwidth as the x rows themselves
  __shared__ __attribute__((aligned(16))) unsigned sN[4][2][2][32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wa = wave >> 1, h = lane >> 5, col = lane & 31;
  const StreamSlot ss = stream_slot(p);
  const int grp = ss.grp;
  const int co0 = ss.cot * WT;
  const int k0 = grp * G;
  constexpr int ng = G;
  const int64_t rbeg = (int64_t)ss.split * p.rows_per_split;
  const int64_t rend = min(p.n_out, rbeg + p.rows_per_split);
  const unsigned ldx4 = 4u * p.ldx, ldy4 = 4u * p.ldy, K4 = 4u * p.K;
  // The descriptors of everything indexed by the OUTPUT row end at this split's last row: a row past the end reads zeros
  // (table entry 0, dy 0, parent 0) without a test per load -- its dy operand is zero (FUSE: forced below), so whatever x
  // row its entries name contributes nothing.
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rd = make_rsrc(p.dy, (unsigned)rend * ldy4),
                               rn = make_rsrc(p.nbr, (unsigned)rend * K4);
  // lanes beyond cin / cout only feed rows / columns of the product that are never stored
  const unsigned xcol = 4u * min(col, p.cin - 1);
  const unsigned dcol = 4u * min(co0 + 32 * wn + col, p.cout - 1);
  const int kb = FLAT ? (256 * grp) / p.cin : k0;  // first offset this group touches
  const unsigned ncol = col < NK && kb + col < p.K ? 4u * (kb + col) : OOB;
  unsigned kidx[G], xoff[G];  // FLAT: this lane's table entry (relative to kb) and column byte offset in tile g
  if (FLAT) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int f = 32 * (grp * G + g) + col, k = f / p.cin;
      const bool in = k < p.K;
      kidx[g] = in ? (unsigned)(k - kb) : 0u;
      xoff[g] = in ? 4u * (unsigned)(f - k * p.cin) : OOB;
    }
  }
  // FUSE: per-lane constants of this lane's output channel
  const int cco = min(co0 + 32 * wn + col, p.cout - 1);
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(FUSE ? (const void *)p.dyp : (const void *)p.dy, FUSE ? p.dyp_bytes : 0u),
                               ri = make_rsrc(FUSE ? (const void *)p.in2out : (const void *)p.nbr, FUSE ? (unsigned)rend * 4u : 0u);
  const float c_mu = FUSE ? p.mean[cco] : 0.f, c_is = FUSE ? p.invstd[cco] : 0.f, c_ga = FUSE ? p.gamma[cco] : 0.f,
              c_be = FUSE ? p.beta[cco] : 0.f, c_dgn = FUSE ? p.dgamma[cco] * p.inv_n : 0.f,
              c_dbn = FUSE ? p.dbeta[cco] * p.inv_n : 0.f;
  const float c_nmu = -c_mu * c_is, c_a = c_ga * c_is;
  float dp[D];        // FUSE: pooled gradient of the row's parent
  unsigned i2or[D];   // FUSE: parent row of a pair whose operands are still to be requested

  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = (f32x16){0};
  float xa[D][G], db[D];
  unsigned nraw[D];  // lane (h, col < 9): nbr[R0 + h][k0 + col] of a pair still to be staged

  // this wave's q-th pair covers rows R0, R0 + 1 (lane half h takes R0 + h); the two wave
  // rows interleave their pairs
  const int64_t npairs = (rend - rbeg + 1) >> 1;
  const int nq = (int)((npairs + 1 - wa) >> 1);
  // 32-bit row arithmetic relative to the split (every VALU instruction issued here is a slot the matrix pipe
  // does not get: PMC shows MFMA busy + 4 x VALU instructions ~ 87 % of the SIMD cycles, no co-execution)
  const int nrel = (int)(rend - rbeg);
  const unsigned nbase = (unsigned)rbeg * K4 + ncol;     // >= 2^31 for the padding lanes (ncol == OOB): stays out of range
  const unsigned ibase = (unsigned)rbeg * 4u;
  const unsigned dbase = (unsigned)rbeg * ldy4 + dcol;
  auto rel_of = [&](int q) { return 2 * (2 * q + wa) + h; };
  auto load_raw = [&](int s, int q) {  // rows past the end read as entry 0; their x offset is forced out of range below
    const int r = rel_of(q);
    nraw[s] = __builtin_amdgcn_raw_buffer_load_b32(rn, (int)(__umul24(r, K4) + nbase), 0, 0);
  };
  auto stash = [&](int s, int slot, int q) {
    sN[wave][slot][h][col] = nraw[s];  // lanes col >= NK store padding: no exec-mask branch in the loop
  };
  auto load_i2o = [&](int s, int q) {  // (a row past the end reads parent 0: its x operand is zero anyway)
    if (FUSE) {
      const int r = rel_of(q);
      i2or[s] = __builtin_amdgcn_raw_buffer_load_b32(ri, (int)(4u * r + ibase), 0, 0);
    }
  };
  auto load_dy = [&](int s, int q) {
    const int r = rel_of(q);
    db[s] = buf_load(rd, __umul24(r, ldy4) + dbase);
    if (FUSE) dp[s] = buf_load(rp, __umul24(i2or[s], ldy4) + dcol);  // the pooled gradient has the same row pitch
  };
  auto b_operand = [&](int s, int q) {
    if (!FUSE) return db[s];
    const float xh = fmaf(db[s], c_is, c_nmu);
    const float m = fmaf(xh, c_ga, c_be) > 0.f ? 1.f : 0.f;  // (a select of the loaded value itself became an exec branch
                                                              //  and, with it, a copy of all 144 accumulators per trip)
    const float v = fmaf(-c_dgn, xh, fmaf(dp[s], m, -c_dbn));
    return (rel_of(q) < nrel ? c_a : 0.f) * v;  // a row past the end: zero
  };
  // FLAT: the lane's table entries of the pair whose x values are requested NEXT, read from the staging slot one stage
  // ahead (right behind the stash that fills it) -- read where they are used, every gather stood behind an LDS round trip
  unsigned nbn[G];
  auto read_entries = [&](int slot) {
    if constexpr (FLAT) {
#pragma unroll
      for (int g = 0; g < G; ++g) nbn[g] = sN[wave][slot][h][kidx[g]];
    }
  };
  auto load_xs = [&](int s, int slot, int q, auto &&between) {  // the nine x values of pair q
    if constexpr (FLAT) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        between(g);
        xa[s][g] = buf_load(rx, __umul24(nbn[g], ldx4) + xoff[g]);
      }
      return;
    }
    const uint4 n0 = *reinterpret_cast<const uint4 *>(&sN[wave][slot][h][0]);
    const uint4 n1 = *reinterpret_cast<const uint4 *>(&sN[wave][slot][h][4]);
    const unsigned n2 = sN[wave][slot][h][8];
    const unsigned nbv[9] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2};
#pragma unroll
    for (int g = 0; g < G; ++g) {
      between(g);
      xa[s][g] = buf_load(rx, __umul24(nbv[g], ldx4) + xcol);
    }
  };

  if (rbeg < rend) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      load_raw(s, s);
      load_i2o(s, s);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < D; ++s) {  // same load order as the loop body: the vmcnt waits there are FIFO distances
      stash(s, s & 1, s);
      read_entries(s & 1);
      load_xs(s, s & 1, s, [](int) {});
      load_dy(s, s);
      load_raw(s, s + D);
      load_i2o(s, s + D);
      __builtin_amdgcn_sched_barrier(0);
    }
    stash(0, 0, D);  // pair D
    read_entries(0);
    for (int q0 = 0; q0 < nq; q0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {  // pair q0 + s from slot s; pairs past nq were loaded as zeros
        const float b = b_operand(s, q0 + s);
        float a[G];
#pragma unroll
        for (int g = 0; g < G; ++g) a[g] = xa[s][g];
        load_xs(s, s & 1, q0 + s + D, [&](int g) {
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g], b, acc[g], 0, 0, 0);
        });
        load_dy(s, q0 + s + D);
        stash((s + 1) % D, (s + 1) & 1, q0 + s + D + 1);  // pair q0 + s + D + 1, loaded D - 1 pairs ago
        read_entries((s + 1) & 1);
        load_raw(s, q0 + s + 2 * D);
        load_i2o(s, q0 + s + 2 * D);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // VALU (offset)
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the slots (and with them the load FIFO) in program order
      }
    }
  }

  // ---- epilogue: add the two wave rows through LDS, store the partial slab
  float *dst = p.out + (int64_t)ss.split * p.K * p.cin * p.cout;
  const int co = co0 + 32 * wn + col;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g < ng) {
      const int k = k0 + g;
      __syncthreads();
      if (wa == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sR[(wn * 16 + r) * 64 + lane] = acc[g][r];
      }
      __syncthreads();
      if (wa == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = acc[g][r] + sR[(wn * 16 + r) * 64 + lane];
          if (FLAT) {
            const int f = 32 * (grp * G + g) + ci;  // flat row of the [K * cin][cout] matrix
            if (f < p.K * p.cin && co < p.cout) dst[(int64_t)f * p.cout + co] = v;
          } else if (ci < p.cin && co < p.cout) dst[((int64_t)k * p.cin + ci) * p.cout + co] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------- streaming wgrad on the bf16 matrix cores (cin <= 32)
// BASELINE config "bf16 mixed precision": the stem's weight gradient with bf16 MFMA operands and fp32 accumulation
// (v_mfma_f32_32x32x16_bf16: sixteen rows per instruction, 16x the fp32 matrix rate).  Same ownership as the fp32
// streaming kernel -- a wave holds the nine 32 x 32 accumulators of one offset group and one 32-column half, two wave
// rows interleave the 16-row blocks -- and the same "operands straight from global memory in register layout" idea:
// lane (h, m) of the A operand holds x[nbr[R + 8h + j][k0 + g]][m], j = 0..7 (eight gathered rows of channel m), lane
// (h, n) of the B operand dY[R + 8h + j][n]; values are loaded as fp32 and packed to bf16 in registers (HBM tensors
// stay fp32).  With the matrix work down 16x the kernel is bound by its gathers, which are the fp32 kernel's.
// Pipeline per wave: table entries of block b+1 are fetched during block b and broadcast through a wave-private LDS
// slot; the x gathers run one three-offset sub-batch ahead of the MFMAs; FUSE recomputes dY from the conv output and
// the pooled gradient as the fp32 kernel does (parents one block ahead).

// B16 (bf16 STORAGE of the full-resolution stage, stem16.hip): x is the bf16 copy of the input ([n][32], 64-byte rows) and
// `dy` -- FUSE: the convolution output -- is bf16 too; the operands are then 2-byte loads that need no conversion.
template <bool FUSE, bool B16 = false>
__global__ __launch_bounds__(256, 2) void wgrad_stream_bf16_kernel(WgradParams p) {
  constexpr int G = 9, SB = 1;           // offsets per group, offsets per sub-batch (gathers run one sub-batch ahead)
  __shared__ float sR[2 * 16 * 64];
  __shared__ __attribute__((aligned(16))) unsigned sN[4][2][12][16];  // [wave][slot][offset (9 used)][row of the block]: a lane's eight rows of one offset are two 16-byte reads
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wa = wave >> 1, h = lane >> 5, col = lane & 31;
  const StreamSlot ss = stream_slot(p);
  const int grp = ss.grp;
  const int co0 = ss.cot * WT;
  const int k0 = grp * G;
  const int64_t rbeg = (int64_t)ss.split * p.rows_per_split;
  const int64_t rend = min(p.n_out, rbeg + p.rows_per_split);
  const int nrel = (int)(rend - rbeg);
  const int nblocks = (nrel + 15) >> 4;
  const int nq = (nblocks + 1 - wa) >> 1;  // this wave's blocks: b = 2 q + wa
  const unsigned ldx4 = (B16 ? 2u : 4u) * p.ldx, ldy4 = (B16 ? 2u : 4u) * p.ldy, ldp4 = 4u * p.ldy, K4 = 4u * p.K;
  // (as in the fp32 kernel: what is indexed by the output row ends at this split's last row -- rows past it read zeros)
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rd = make_rsrc(p.dy, (unsigned)rend * ldy4), rn = make_rsrc(p.nbr, p.nbr_bytes);
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(FUSE ? (const void *)p.dyp : (const void *)p.dy, FUSE ? p.dyp_bytes : 0u),
                               ri = make_rsrc(FUSE ? (const void *)p.in2out : (const void *)p.nbr, FUSE ? (unsigned)rend * 4u : 0u);
  const unsigned xcol = (B16 ? 2u : 4u) * min(col, p.cin - 1);
  const unsigned dcol = (B16 ? 2u : 4u) * min(co0 + 32 * wn + col, p.cout - 1), pcol = 4u * min(co0 + 32 * wn + col, p.cout - 1);
  const int cco = min(co0 + 32 * wn + col, p.cout - 1);
  const float c_mu = FUSE ? p.mean[cco] : 0.f, c_is = FUSE ? p.invstd[cco] : 0.f, c_ga = FUSE ? p.gamma[cco] : 0.f,
              c_be = FUSE ? p.beta[cco] : 0.f, c_dgn = FUSE ? p.dgamma[cco] * p.inv_n : 0.f,
              c_dbn = FUSE ? p.dbeta[cco] * p.inv_n : 0.f;
  const float c_nmu = -c_mu * c_is, c_a = c_ga * c_is;
  const unsigned nbase = (unsigned)rbeg * K4, ibase = (unsigned)rbeg * 4u, dbase = (unsigned)rbeg * ldy4 + dcol;

  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = (f32x16){0};
  if (nq <= 0) goto epilogue;
  {
    // table loader lanes: row kk = lane >> 2 of the block, entries 3 (lane & 3) .. + 2 (lanes with (lane & 3) == 3 idle)
    const int t_row = lane >> 2, t_part = lane & 3;
    unsigned traw[3];
    auto load_table = [&](int q) __attribute__((always_inline)) {  // block 2 q + wa (rows past the end read as "no neighbour")
      const int r = 16 * (2 * q + wa) + t_row;
      const bool ok = q < nq && r < nrel && t_part < 3;
#pragma unroll
      for (int e = 0; e < 3; ++e)
        traw[e] = ok ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rn, (int)(__umul24(r, K4) + nbase + 4u * (k0 + 3 * t_part + e)), 0, 0)
                     : 0xFFFFFFFFu;
    };
    auto stash_table = [&](int slot) __attribute__((always_inline)) {
      if (t_part < 3) {
#pragma unroll
        for (int e = 0; e < 3; ++e) sN[wave][slot][3 * t_part + e][t_row] = traw[e];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slot: in-order LDS, visible to the reads below
    };
    unsigned par[8];  // FUSE: pooled parent of this lane's eight rows, one block ahead
    auto load_par = [&](int q) __attribute__((always_inline)) {
      if (FUSE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 16 * (2 * q + wa) + 8 * h + j;
          par[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ri, (int)(4u * r + ibase), 0, 0);  // (past the end: parent 0)
        }
      }
    };
    float braw[8], bpool[8];
    auto load_b = [&](int q) __attribute__((always_inline)) {  // dY rows (FUSE: conv output rows + pooled gradient of their parents)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 16 * (2 * q + wa) + 8 * h + j;
        if constexpr (B16)
          braw[j] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rd, (int)(__umul24(r, ldy4) + dbase), 0, 0) << 16);
        else
          braw[j] = buf_load(rd, __umul24(r, ldy4) + dbase);
        if (FUSE) bpool[j] = buf_load(rp, __umul24(par[j], ldp4) + pcol);
      }
    };
    auto b_fragment = [&](int q) __attribute__((always_inline)) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!FUSE) {
          v[j] = braw[j];
        } else {
          const int r = 16 * (2 * q + wa) + 8 * h + j;
          const float xh = fmaf(braw[j], c_is, c_nmu);
          const float m = fmaf(xh, c_ga, c_be) > 0.f ? 1.f : 0.f;
          v[j] = (r < nrel ? c_a : 0.f) * fmaf(-c_dgn, xh, fmaf(bpool[j], m, -c_dbn));  // a row past the end must not contribute
        }
      }
      return pack_bf16x8(v);
    };
    float xraw[3][SB][8];
    auto load_x = [&](int buf, int slot, int sb) __attribute__((always_inline)) {  // the x gathers of one sub-batch (three offsets x eight rows)
#pragma unroll
      for (int g = 0; g < SB; ++g) {
        const uint4 n0 = *reinterpret_cast<const uint4 *>(&sN[wave][slot][SB * sb + g][8 * h]);
        const uint4 n1 = *reinterpret_cast<const uint4 *>(&sN[wave][slot][SB * sb + g][8 * h + 4]);
        const unsigned nbv[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if constexpr (B16)  // (the bf16 bits, kept in the low half of the register)
            xraw[buf][g][j] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rx, (int)(__umul24(nbv[j], ldx4) + xcol), 0, 0));
          else
            xraw[buf][g][j] = buf_load(rx, __umul24(nbv[j], ldx4) + xcol);  // (-1 is row 0xFFFFFF: beyond x, reads as zero)
      }
    };
    // ---- prologue: table of block 0 staged, of block 1 in flight; B operands and first x sub-batch of block 0 in flight
    load_table(0);
    load_par(0);
    stash_table(0);
    load_b(0);
    load_table(1);
    load_par(1);
    load_x(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // one block = nine one-offset sub-batches; the gathers run one sub-batch ahead through a ring of three buffers
    // (nine is a multiple of three: every register index is a compile-time constant without unrolling over blocks)
    for (int q = 0; q < nq; ++q) {
      const int slot = q & 1;
      const bf16x8v bfrag = b_fragment(q);
#pragma unroll
      for (int sb = 0; sb < G; ++sb) {
        const int cur = sb % 3, nxt = (sb + 1) % 3;
        if (sb == 0) load_b(q + 1);  // (parents of block q + 1 arrived one block ago)
        if (sb == 1) load_par(q + 2);
        if (sb < G - 1) {
          load_x(nxt, slot, sb + 1);
        } else {
          stash_table(slot ^ 1);  // entries of block q + 1, in flight since the previous block
          load_x(nxt, slot ^ 1, 0);
          load_table(q + 2);
        }
        acc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B16 ? pack_bits_bf16x8(xraw[cur][0]) : pack_bf16x8(xraw[cur][0]), bfrag, acc[sb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // keep the sub-batches (and the load FIFO) in program order
      }
    }
  }
epilogue:
  // ---- add the two wave rows through LDS, store the partial slab
  float *dst = p.out + (int64_t)ss.split * p.K * p.cin * p.cout;
  const int co = co0 + 32 * wn + col;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int k = k0 + g;
    __syncthreads();
    if (wa == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sR[(wn * 16 + r) * 64 + lane] = acc[g][r];
    }
    __syncthreads();
    if (wa == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[g][r] + sR[(wn * 16 + r) * 64 + lane];
        if (ci < p.cin && co < p.cout) dst[((int64_t)k * p.cin + ci) * p.cout + co] = v;
      }
    }
  }
}

// ---------------------------------------------------- streaming wgrad over bf16 STORAGE, operands transposed through LDS
// With x ([n][32] bf16, 64-byte rows) and the convolution output in bf16 (stem16.hip) the kernel above still issues eight
// 2-byte gathers per MFMA operand: a lane of the A operand holds ONE channel of EIGHT different rows.  Here the sixteen
// gathered rows of an operand are fetched the way they lie in memory -- four lanes x 16 bytes per row, ONE load per lane --
// written to a wave-private 1 KB LDS image [16 rows][32 channels] and read back with ds_read_b64_tr_b16, gfx950's
// transposing LDS read (a lane receives its channel of four rows), two reads per operand.  The B operand (dY recomputed
// from the convolution output, the pooled gradient and the batch-norm constants, FUSE of the kernels above) takes the same
// route: a lane computes eight consecutive columns of one row (constants per column from LDS), packs them and the wave
// reads the block back transposed -- 3 loads per block instead of 16.  ~16 load instructions per 16-row block instead of
// ~100.  Ownership, row splits, slabs and epilogue are those of wgrad_stream_bf16_kernel; LDS traffic is wave-private and
// in issue order (no barrier in the loop).  cout == 64, K == 27, x pitch 32.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const unsigned short *p) {  // ds_read_b64_tr_b16 (EXEC must be all ones)
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p));
}
template <int A>
__device__ __forceinline__ unsigned quad_bcast(unsigned v) {  // lane A of every quad
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, A | (A << 2) | (A << 4) | (A << 6), 0xF, 0xF, false);
}

// (A variant whose two column-half waves shared ONE set of gathered operands through a double-buffered image and a
// barrier per block measured 246 us against 244: the gathers are not what bounds this kernel -- PMC: 184 vector-ALU
// instructions per 16-row block and wave, the recomputation of dY, beside nine MFMAs.)
__global__ __launch_bounds__(256, 2) void wgrad_stream_b16t_kernel(WgradParams p) {
  constexpr int G = 9;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NSL = 4 * 3;  // 1 KB operand images: [wave][ring]
  __shared__ float sR[2 * 16 * 64];
  __shared__ __attribute__((aligned(16))) unsigned short sA[NSL][16 * 32];  // [row][channel]
  __shared__ __attribute__((aligned(16))) unsigned short sB[4][16 * 32];    // [wave][row][column of the wave's half]
  __shared__ __attribute__((aligned(16))) float sC[7][64];                  // per column: invstd, -mean*invstd, gamma, beta, gamma*invstd, dgamma/n, dbeta/n
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wa = wave >> 1, h = lane >> 5, col = lane & 31;
  const StreamSlot ss = stream_slot(p);
  const int grp = ss.grp;
  const int k0 = grp * G;
  const int64_t rbeg = (int64_t)ss.split * p.rows_per_split;
  const int64_t rend = min(p.n_out, rbeg + p.rows_per_split);
  const int nrel = (int)(rend - rbeg);
  const int nblocks = (nrel + 15) >> 4;
  const int nq = (nblocks + 1 - wa) >> 1;  // this wave's blocks: b = 2 q + wa
  if (tid < 64) {
    const float is = p.invstd[tid], mu = p.mean[tid], ga = p.gamma[tid];
    sC[0][tid] = is, sC[1][tid] = -mu * is, sC[2][tid] = ga, sC[3][tid] = p.beta[tid], sC[4][tid] = ga * is;
    sC[5][tid] = p.dgamma[tid] * p.inv_n, sC[6][tid] = p.dbeta[tid] * p.inv_n;
  }
  __syncthreads();
  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = (f32x16){0};
  if (nq > 0) {  // (wave-uniform)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), ry = make_rsrc(p.dy, (unsigned)rend * 128u),
                                 rn = make_rsrc(p.nbr, (unsigned)rend * 4u * 27u), rp = make_rsrc(p.dyp, p.dyp_bytes),
                                 ri = make_rsrc(p.in2out, (unsigned)rend * 4u);
    const int g_row = lane >> 2, g_ch = lane & 3;  // gather / compute role: row of the block, 16-byte chunk
    const unsigned nbase = (unsigned)rbeg * 108u + 4u * (unsigned)(k0 + 3 * g_ch), ibase = (unsigned)rbeg * 4u;
    const unsigned ybase = (unsigned)rbeg * 128u + 64u * wn + 16u * g_ch, pcol = 128u * wn + 32u * g_ch;
    unsigned short *sBw = &sB[wave][0];
    const int st_off = g_row * 32 + 8 * g_ch;                                        // halfword offset of this lane's 16 bytes
    const int tr_off = (8 * h + ((lane & 15) >> 2)) * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);  // transposed read, rows 8h..8h+3 (+128: rows 8h+4..)
    unsigned traw[3], par, ent[G];
    u32x4v ga[G], yraw, dp0, dp1;
    auto rel_row = [&](int q) { return 16 * (2 * q + wa) + g_row; };
    auto load_table = [&](int q) __attribute__((always_inline)) {  // entries 3 g_ch .. + 2 of the lane's row (g_ch == 3: idle)
      const int r = rel_row(q);
      const bool ok = q < nq && r < nrel && g_ch < 3;
#pragma unroll
      for (int e = 0; e < 3; ++e)
        traw[e] = ok ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rn, (int)(__umul24(r, 108u) + nbase + 4u * e), 0, 0) : 0xFFFFFFFFu;
    };
    auto load_par = [&](int q) __attribute__((always_inline)) {
      const int r = rel_row(q);
      par = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ri, (int)(q < nq && r < nrel ? 4u * r + ibase : OOB), 0, 0);  // (past the end: parent 0)
    };
    auto spread = [&]() __attribute__((always_inline)) {  // every lane of a row's quad gets the row's nine entries
      ent[0] = quad_bcast<0>(traw[0]), ent[1] = quad_bcast<0>(traw[1]), ent[2] = quad_bcast<0>(traw[2]);
      ent[3] = quad_bcast<1>(traw[0]), ent[4] = quad_bcast<1>(traw[1]), ent[5] = quad_bcast<1>(traw[2]);
      ent[6] = quad_bcast<2>(traw[0]), ent[7] = quad_bcast<2>(traw[1]), ent[8] = quad_bcast<2>(traw[2]);
    };
    auto gather = [&](int g) __attribute__((always_inline)) {  // (-1 is row 0xFFFFFF: beyond x, reads as zeros)
      ga[g] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(__umul24(ent[g], 64u) + 16u * g_ch), 0, 0);
    };
    auto load_b = [&](int q) __attribute__((always_inline)) {  // conv output row chunk + pooled gradient of its parent
      const int r = rel_row(q);
      yraw = __builtin_amdgcn_raw_buffer_load_b128(ry, (int)(q < nq && r < nrel ? __umul24(r, 128u) + ybase : OOB), 0, 0);
      const unsigned po = __umul24(par, 256u) + pcol;
      dp0 = __builtin_amdgcn_raw_buffer_load_b128(rp, (int)po, 0, 0);
      dp1 = __builtin_amdgcn_raw_buffer_load_b128(rp, (int)(po + 16u), 0, 0);
    };
    auto b_operand = [&](int q) __attribute__((always_inline)) {  // dY of block q: computed row-major, read back transposed
      const int cb = 32 * wn + 8 * g_ch;
      const unsigned yw[4] = {yraw[0], yraw[1], yraw[2], yraw[3]};
      const float dpv[8] = {__uint_as_float(dp0[0]), __uint_as_float(dp0[1]), __uint_as_float(dp0[2]), __uint_as_float(dp0[3]),
                            __uint_as_float(dp1[0]), __uint_as_float(dp1[1]), __uint_as_float(dp1[2]), __uint_as_float(dp1[3])};
      const bool live = rel_row(q) < nrel;
      float v[8];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {  // four columns at a time: 28 constants live, not 56
        float4 cs[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) cs[c] = *reinterpret_cast<const float4 *>(&sC[c][cb + 4 * hf]);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int e = 4 * hf + e4;
          auto at = [&](int c) { return e4 == 0 ? cs[c].x : e4 == 1 ? cs[c].y : e4 == 2 ? cs[c].z : cs[c].w; };
          const float y = __uint_as_float((e & 1) ? (yw[e >> 1] & 0xFFFF0000u) : (yw[e >> 1] << 16));
          const float xh = fmaf(y, at(0), at(1));
          const float m = fmaf(xh, at(2), at(3)) > 0.f ? 1.f : 0.f;
          v[e] = (live ? at(4) : 0.f) * fmaf(-at(5), xh, fmaf(dpv[e], m, -at(6)));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      *reinterpret_cast<uint4 *>(sBw + st_off) = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
      const uint2 b_lo = lds_tr16(sBw + tr_off), b_hi = lds_tr16(sBw + tr_off + 128);
      return __builtin_bit_cast(bf16x8v, make_uint4(b_lo.x, b_lo.y, b_hi.x, b_hi.y));
    };
    // ---- prologue
    load_table(0);
    load_par(0);
    spread();
#pragma unroll
    for (int g = 0; g < G; ++g) gather(g);
    load_b(0);
    load_table(1);
    load_par(1);
    __builtin_amdgcn_sched_barrier(0);
    for (int q = 0; q < nq; ++q) {
      spread();  // entries of block q + 1
      {
        unsigned short *sAw = &sA[wave * 3][0];
        const bf16x8v bfrag = b_operand(q);
        load_b(q + 1);  // (its parents arrived one block ago)
        load_par(q + 2);
        load_table(q + 2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- nine offsets: gathered rows of block q -> LDS -> transposed fragment; the registers take block q + 1's rows
        *reinterpret_cast<u32x4v *>(sAw + st_off) = ga[0];
        gather(0);
        uint2 a_lo = lds_tr16(sAw + tr_off), a_hi = lds_tr16(sAw + tr_off + 128);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const bf16x8v afrag = __builtin_bit_cast(bf16x8v, make_uint4(a_lo.x, a_lo.y, a_hi.x, a_hi.y));
          if (g + 1 < G) {
            unsigned short *slot = sAw + ((g + 1) % 3) * 512;
            *reinterpret_cast<u32x4v *>(slot + st_off) = ga[g + 1];
            gather(g + 1);
            a_lo = lds_tr16(slot + tr_off), a_hi = lds_tr16(slot + tr_off + 128);
          }
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[g], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // ---- epilogue: add the two wave rows through LDS, store the partial slab
  float *dst = p.out + (int64_t)ss.split * p.K * p.cin * p.cout;
  const int co = 32 * wn + col;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int k = k0 + g;
    __syncthreads();
    if (wa == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sR[(wn * 16 + r) * 64 + lane] = acc[g][r];
    }
    __syncthreads();
    if (wa == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[g][r] + sR[(wn * 16 + r) * 64 + lane];
        if (ci < p.cin) dst[((int64_t)k * p.cin + ci) * p.cout + co] = v;
      }
    }
  }
}

// out[i] = sum_z ws[z][i]: 64 outputs x 4 slab lanes per workgroup (fixed order -> deterministic)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ ws, int64_t count, int nslab,
                                                          float *__restrict__ out) {
  __shared__ float s_part[4][64];
  const int lane = threadIdx.x >> 6, o = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 64 + o;
  // four independent chains per thread (slabs z, z + 4, z + 8, z + 12 of its lane): sixteen loads in flight instead of the
  // one-after-the-other adds of a single chain -- the stem's 168-slab reduce is the LAST kernel of a training step
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < count) {
    const float *q = ws + i;
    int z = lane;
    for (; z + 12 < nslab; z += 16) {
      const float a = q[(int64_t)z * count], b = q[(int64_t)(z + 4) * count], c = q[(int64_t)(z + 8) * count], d = q[(int64_t)(z + 12) * count];
      s0 += a, s1 += b, s2 += c, s3 += d;
    }
    for (; z < nslab; z += 4) s0 += q[(int64_t)z * count];
  }
  const float s = (s0 + s1) + (s2 + s3);
  s_part[lane][o] = s;
  __syncthreads();
  if (lane == 0 && i < count) out[i] = (s_part[0][o] + s_part[1][o]) + (s_part[2][o] + s_part[3][o]);
}

static int g_wgrad_stream = 1;
static int g_wgrad_force = 0;  // bits 0-3: G code, bits 4..: row splits (scripts/kbench.py wsweep)

struct WgradPlan {
  int G, ngroups, nsplit;
  int64_t rows_per_split;
};

static WgradPlan wgrad_plan(int64_t n_out, int K, int cin, int cout) {
  WgradPlan pl;
  const int64_t tiles = cdiv(cin, WT) * cdiv(cout, WT);
  const int64_t row_tiles = cdiv(n_out, WROWS);
  // offsets per workgroup: share the dy tile between as many offsets as parallelism allows
  pl.G = 1;
  if (K >= 9 && tiles * cdiv(K, 9) * row_tiles >= 1024) pl.G = 9;
  else if (K >= 3 && tiles * cdiv(K, 3) * row_tiles >= 1024) pl.G = 3;
  const bool tiny = row_tiles <= 4 && tiles * K >= 512;  // few rows, many weight tiles: one workgroup per (tile, offset), no slabs
  if (tiny) pl.G = 1;
  // (layer1, one 64 x 64 weight tile and many rows: alone, one offset per workgroup and three times the workgroups win --
  //  kbench wsweep: l1.conv2 G1 z64 89 us against G3 z48 101 -- but inside a step, beside the data-gradient chain, the 1296
  //  workgroups take 208 us where the 432 take 125: the plan stays)
  if (g_wgrad_force & 0xF) pl.G = (g_wgrad_force & 0xF) == 1 ? 1 : (g_wgrad_force & 0xF) == 2 ? 3 : 9;  // tuning hook
  pl.ngroups = (int)cdiv(K, pl.G);
  const int64_t xy = tiles * pl.ngroups;
  int64_t z = cdiv(512, xy);  // ~2 resident workgroups per CU: fewer partial slabs to write and reduce
  if (tiny) z = 1;
  if (g_wgrad_force >> 4) z = g_wgrad_force >> 4;
  if (z > row_tiles) z = row_tiles;
  if (z < 1) z = 1;
  const bool streamed = pl.G == 9 && K == 27 && cin <= 32 && tiles == 1 && z >= 16;  // see stream_slot: splits in eights
  if (streamed) z = z / 8 * 8;  // rounded DOWN: 3 x 176 workgroups no longer fit the 512 resident slots (measured 1.21 ms against 0.86)
  pl.rows_per_split = align_up(cdiv(n_out, z), WROWS);
  pl.nsplit = (int)cdiv(n_out, pl.rows_per_split);
  if (pl.nsplit < 1) pl.nsplit = 1;
  if (streamed) pl.nsplit = (int)align_up(pl.nsplit, 8);  // trailing splits may be empty: they store zero slabs
  return pl;
}

// ------------------------------------------------ tiled weight gradient on the bf16 matrix cores (mid layers, --math bf16)
// BASELINE config #4 ("MFMA bf16 on the rulebook GEMM") for the twelve mid-layer weight gradients, which until round 4 stayed on
// wgrad_kernel's exact-fp32 MFMAs (0.8 ms of that step's weight-gradient stream).  Same ownership as wgrad_kernel -- workgroup =
// (G offsets) x (64 x 64 ci / co tile) x (row range); per 128-row tile the dy tile is staged once, per offset the rows that have
// a neighbour are compacted (wave64 ballot + prefix rank) and only their x rows gathered -- but both tiles live in LDS as
// bf16, ROW-major as they arrive (an 8-byte store per gathered float4), and the MFMA fragments come out of them through
// gfx950's transposing LDS read (ds_read_b64_tr_b16: sixteen lanes hand in four rows x sixteen channels and each receives its
// channel of the four rows): v_mfma_f32_32x32x16_bf16 contracts SIXTEEN pairs per instruction where the fp32 kernel's
// 32x32x2 contracts two.  The dy rows of an offset's pairs are reached through the pair list (a lane's row address is its
// own: no compacted copy of the tile).  LDS: 2 x 128 rows x 128 bytes + lists = 37 KB -> four workgroups per CU where the
// fp32 kernel's 74 KB allow two.  fp32 accumulation, fp32 slabs, deterministic (no atomics).  cin, cout multiples of 64,
// 16-byte aligned operands, 32-bit buffer offsets (the launcher checks).
// Bank conflicts of the transposing read: a half-wave reads four rows x 64 bytes; with 128-byte rows, rows r and r + 2 would
// share banks, so the two 64-byte halves of a row are swapped on rows with bit 1 set (a row's pieces then cover all 256 bytes
// over any four consecutive rows).
template <int G>
__global__ __launch_bounds__(256, 4) void wgrad16_kernel(WgradParams p) {
  constexpr int LL = WROWS;  // list length (pair count padded to 16, <= 128)
  constexpr int PITCH = 64;  // halfwords per row of both images
  __shared__ __attribute__((aligned(16))) unsigned short sD[WROWS * PITCH];  // dy tile, bf16 [row][co]
  __shared__ __attribute__((aligned(16))) unsigned short sX[WROWS * PITCH];  // gathered x rows of one offset, bf16 [pair][ci]
  __shared__ int s_row[G * LL];  // tile row of the p-th pair (padding: row 0)
  __shared__ int s_src[G * LL];  // its x row (padding: -1 = a row beyond x: zeros)
  __shared__ int s_cnt[G * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned wbx = blockIdx.x, wby = blockIdx.y;
  if (!(p.ablate & 4096) && (gridDim.y & 7u) == 0u) {  // uniform: the workgroups of a row split share an XCD (see wgrad_kernel)
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7u, slot = lin >> 3;
    wbx = slot % gridDim.x, wby = (slot / gridDim.x) * 8u + xcd;
  }
  const int grp = wbx % p.ngroups, tile_id = wbx / p.ngroups;
  const int ci0 = (tile_id / p.ct_tiles) * WT, co0 = (tile_id % p.ct_tiles) * WT;
  const int k0 = grp * G;
  const int ng = min(G, p.K - k0);
  const int64_t rbeg = (int64_t)wby * p.rows_per_split;
  const int64_t rend = min(p.n_out, rbeg + p.rows_per_split);
  const int c4 = tid & 15, rr = tid >> 4;  // staging: float4 column, rows rr + 16 i (both tiles)
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, col = lane & 31;
  // image address of (row, halfword column c): the 64-byte halves of a row are swapped where bit 1 of the row is set
  auto img = [](int row, int c) { return row * PITCH + (c ^ ((row & 2) << 4)); };
  // transposing read: this lane hands in four channels (4 (lane & 3) .. of the sixteen at 16 ((lane >> 4) & 1)) of row (lane & 15) >> 2
  const int tr_row = 8 * h + ((lane & 15) >> 2), tr_c = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = (f32x16){0};
  const i32x4 bx = raw_rsrc(p.x, p.x_bytes), bd = raw_rsrc(p.dy, p.dy_bytes), bn = raw_rsrc(p.nbr, p.nbr_bytes);
  const unsigned ldx4 = 4u * (unsigned)p.ldx, ldy4 = 4u * (unsigned)p.ldy, K4 = 4u * (unsigned)p.K;
  const unsigned x_coff = 4u * (unsigned)(ci0 + 4 * c4), d_coff = 4u * (unsigned)(co0 + 4 * c4);
  float4 rx[8];
  auto gather = [&](int g) __attribute__((always_inline)) {  // x rows of the compacted pairs of offset g -> registers (-1: zeros)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      rx[i] = __builtin_bit_cast(float4, raw_load_v4(bx, (int)(__umul24((unsigned)s_src[g * LL + rr + 16 * i], ldx4) + x_coff), 0, 0));
  };
  auto put = [&](unsigned short *im, int row, const float4 &v) __attribute__((always_inline)) {
    *reinterpret_cast<uint2 *>(im + img(row, 4 * c4)) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) put(sX, rr + 16 * i, rx[i]);
  };
  for (int64_t r0 = rbeg; r0 < rend; r0 += WROWS) {
    __syncthreads();  // previous tile fully consumed
    int nb[G], rank[G];
    if (tid < WROWS) {
      const int64_t row = r0 + tid;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const bool ok = row < rend && g < ng;
        const int v = raw_load_i32(bn, (int)(ok ? (unsigned)row * K4 + 4u * (unsigned)(k0 + g) : 0x80000000u), 0, 0);
        nb[g] = ok ? v : -1;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t row = r0 + rr + 16 * i;
      const float4 v = __builtin_bit_cast(float4, raw_load_v4(bd, (int)((row < rend ? (unsigned)row * ldy4 : 0x80000000u) + d_coff), 0, 0));
      put(sD, rr + 16 * i, v);
    }
    if (tid < WROWS) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const unsigned long long mm = __ballot(nb[g] >= 0);
        rank[g] = wave_rank(mm);
        if (lane == 0) s_cnt[2 * g + wave] = __popcll(mm);
      }
    }
    __syncthreads();
    if (tid < WROWS) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (nb[g] >= 0) {
          const int pos = (wave == 1 ? s_cnt[2 * g] : 0) + rank[g];
          s_row[g * LL + pos] = tid, s_src[g * LL + pos] = nb[g];
        }
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) {  // tail pairs: dy row 0 times a zero x row
        const int m = s_cnt[2 * g] + s_cnt[2 * g + 1];
        const int t = tid - WROWS;
        if (t < ((m + 15) & ~15) - m) s_row[g * LL + m + t] = 0, s_src[g * LL + m + t] = -1;
      }
    }
    __syncthreads();
    gather(0);
    stash();
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < ng) {  // uniform
        if (g + 1 < ng) gather(g + 1);  // in flight during the MFMAs below
        const int m = s_cnt[2 * g] + s_cnt[2 * g + 1];
        const int nk16 = (m + 15) >> 4;  // sixteen pairs per MFMA
        for (int kk = 0; kk < nk16; ++kk) {
          const int pa = 16 * kk + tr_row;  // this lane's pair of the low half (high half: + 4)
          const int r_lo = s_row[g * LL + pa], r_hi = s_row[g * LL + pa + 4];
          const uint2 a_lo = lds_tr16(sX + img(pa, 32 * wm + tr_c)), a_hi = lds_tr16(sX + img(pa + 4, 32 * wm + tr_c));
          const uint2 b_lo = lds_tr16(sD + img(r_lo, 32 * wn + tr_c)), b_hi = lds_tr16(sD + img(r_hi, 32 * wn + tr_c));
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, make_uint4(a_lo.x, a_lo.y, a_hi.x, a_hi.y)),
                                                           __builtin_bit_cast(bf16x8v, make_uint4(b_lo.x, b_lo.y, b_hi.x, b_hi.y)), acc[g], 0, 0, 0);
        }
        if (g + 1 < ng) {
          __syncthreads();  // everyone done reading sX
          stash();
          __syncthreads();
        }
      }
    }
  }
  // ---- epilogue: the partial slab (C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h)
  float *dst = p.out + (int64_t)wby * p.K * p.cin * p.cout;
  const int co = co0 + 32 * wn + col;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g < ng) {  // uniform
      const int k = k0 + g;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
        dst[((int64_t)k * p.cin + ci) * p.cout + co] = acc[g][r];
      }
    }
  }
}

template <int G>
static void launch_wgrad(const WgradParams &p, dim3 grid, hipStream_t st) {
  const bool vec = (((uintptr_t)p.x | (uintptr_t)p.dy) & 15) == 0 && ((p.ldx | p.ldy | p.cin | p.cout) & 3) == 0;
  constexpr int buf_on = 1;
  if (p.cin <= 32) {
    if (vec && p.buf_ok && buf_on) wgrad_kernel<G, true, true, true><<<grid, 256, 0, st>>>(p);
    else if (vec) wgrad_kernel<G, true, true><<<grid, 256, 0, st>>>(p);
    else wgrad_kernel<G, true, false><<<grid, 256, 0, st>>>(p);
  } else {
    if (vec && p.buf_ok && buf_on) wgrad_kernel<G, false, true, true><<<grid, 256, 0, st>>>(p);
    else if (vec) wgrad_kernel<G, false, true><<<grid, 256, 0, st>>>(p);
    else wgrad_kernel<G, false, false><<<grid, 256, 0, st>>>(p);
  }
}

}  // namespace mink

using namespace mink;

// ---- kernel timing registry (measurement only; see mink_conv_timing in the header)
namespace {
struct TimedLaunch {
  MinkTimingEntry e;
  hipEvent_t a, b;
};
std::mutex g_time_mu;
std::vector<TimedLaunch> g_timed;
std::vector<hipEvent_t> g_event_pool;
int g_time_mode = 0, g_time_kind = 0, g_time_K = 0, g_time_cin = 0, g_time_cout = 0;

struct ScopedTimer {  // records an event pair around the launches of one convolution call, on the launch stream
  bool on = false;
  TimedLaunch t;
  hipStream_t st;
  ScopedTimer(int kind, int64_t n_in, int64_t n_out, int K, int cin, int cout, const int32_t *nbr, hipStream_t stream) : st(stream) {
    if (g_time_mode == 0) return;
    if (g_time_mode == 2 && !((g_time_kind < 0 || kind == g_time_kind) && K == g_time_K && cin == g_time_cin && cout == g_time_cout)) return;
    std::lock_guard<std::mutex> lk(g_time_mu);
    for (hipEvent_t *ev : {&t.a, &t.b}) {
      if (!g_event_pool.empty()) {
        *ev = g_event_pool.back();
        g_event_pool.pop_back();
      } else if (hipEventCreate(ev) != hipSuccess) {
        return;
      }
    }
    t.e = MinkTimingEntry{kind, K, cin, cout, n_in, n_out, nbr, 0.f};
    on = hipEventRecord(t.a, st) == hipSuccess;
  }
  ~ScopedTimer() {
    if (!on) return;
    (void)hipEventRecord(t.b, st);
    std::lock_guard<std::mutex> lk(g_time_mu);
    g_timed.push_back(t);
  }
};
}  // namespace

static int g_stagger = 0;
static int g_flat = 1;
// Stream-K plan of a stride-1 mid-layer launch: (workers G, slabs S) or G = 0 when the shape does not take it.  G is one resident
// round (4 or 3 workgroups on each of the 256 CUs; a multiple of 8 for the XCD numbering); a worker's run is at least F / G
// offset-tiles long, so at most ceil(K / floor(F / G)) + 1 workers share a tile -- the slab count.  The largest G whose slab
// count stays within the budget (15 slabs, the 128 MB the slabs may take) wins.
struct SkPlan {
  int G, S;
};
static SkPlan sk_plan(int64_t n_rows, int K, int cout) {
  const int64_t F = cdiv(n_rows, 64) * cdiv(cout, BN) * K;
  const int64_t slab_cap = std::max<int64_t>(1, (128ll << 20) / (4 * n_rows * cout));
  // (measured, scripts/kbench.py sk: runs of 15 and 6 offsets -- layers 1 and 2 of the B=16 batch -- gain; runs of 3 -- layer 3, ten
  //  slabs -- do not, and layer 4, where only three workers per CU keep the slab count at 15, loses 25-29 % to the (tile, slice) grid
  //  of 896 workgroups: a run has to be at least five offsets long, in one resident round of four workers per CU)
  const int G = 1024;
  const int64_t run = F / G;
  if (run < 5) return {0, 0};
  const int64_t S = cdiv(K, run) + 1;
  if (S <= slab_cap) return {G, (int)S};
  return {0, 0};
}

template <bool W_T>
static int launch_compact_sk(const GemmParams &p, int G, hipStream_t st) {
  constexpr int smem = compact_smem(64, false, true);
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<W_T, 64, false, false, 4, 0, false, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
  MINK_REQUIRE(ok, "gather_gemm: %d bytes of LDS per workgroup refused", smem);
  compact_gemm_kernel<W_T, 64, false, false, 4, 0, false, true><<<dim3((unsigned)G), 256, smem, st>>>(p);
  return MINK_OK;
}

template <bool W_T, bool PERM, int MATH>
static int launch_compact_p3(const GemmParams &p, dim3 grid, hipStream_t st) {
  constexpr int smem = compact_smem(64, true);
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<W_T, 64, PERM, false, 4, MATH, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
  MINK_REQUIRE(ok, "gather_gemm: %d bytes of LDS per workgroup refused", smem);
  compact_gemm_kernel<W_T, 64, PERM, false, 4, MATH, true><<<grid, 256, smem, st>>>(p);
  return MINK_OK;
}

static int g_math = 0;  // 0 fp32, 1 bf16 MFMA, 3 split-bf16
static int g_wgrad_bf16 = 1;  // bf16 math: stem weight gradient on the bf16 MFMA too (set_stagger bit 28 switches it off: A/B tests)
static int g_wgrad_bf16_off = 0;
static int g_b16t_off = 0;      // set_stagger bit 11: the bf16-storage stem weight gradient without the LDS transposition (A/B tests)
static int g_compact = 1;  // fp32 mid layers on compact_gemm_kernel (set_stagger bit 30: the dense kernel, for the tests that compare the two)
static int g_compact_perm16 = 1;  // --math bf16: the class-permuted data gradients on compact_gemm_kernel<.., MATH = 1> (set_stagger bit 27 = off: the dense bf16 kernel, A/B)
static int g_compact_cin32 = 0;  // set_stagger bit 8 (measurement only, scripts/kbench.py stemc): the class-permuted form also takes cin = 32
static int g_compact_perm = 1;  // ... and the class-permuted strided data gradients (bit 31)
static unsigned long long *g_trace_buf = nullptr;  // mink_conv_trace
static int64_t g_trace_cap = 0;
static int g_compact_p3 = 0;    // mink_conv_set_pipeline: the three-stage form of compact_gemm_kernel (1: stride-1 layers, 2: class-permuted too, 3: both)
static int g_wgrad_xcd = 1;  // streaming wgrad: groups of a row split share an XCD (stream_slot; bit 29: plain order)

extern "C" {

int mink_conv_set_stagger(int units) {
  const int old = g_stagger;
  g_stagger = units & 255;
  g_compact_cin32 = (units >> 8) & 1;
  g_compact_perm16 = !((units >> 27) & 1);
  g_b16t_off = (units >> 11) & 1;      // bit 11: bf16-storage stem weight gradient with 2-byte gathers (A/B)
  g_flat = !(units & 512);      // bit 9: no flattened-K stem path
  g_wgrad_stream = !(units & 1024);  // bit 10: tiled (LDS) wgrad kernel for the stem
  g_wgrad_bf16_off = (units >> 28) & 1;  // bit 28: bf16 math keeps the exact-fp32 weight-gradient kernel (A/B tests)
  g_wgrad_xcd = !((units >> 29) & 1);    // bit 29: plain workgroup order in the streaming weight-gradient kernels (A/B)
  g_compact = !((units >> 30) & 1);      // bit 30: mid layers back on gather_gemm2_kernel (A/B)
  g_compact_perm = !(((unsigned)units >> 31) & 1u);  // bit 31: class-permuted strided data gradients back on gather_gemm2_kernel (A/B)
  g_wgrad_force = (units >> 12) & 0x7FFF;  // bits 12-15: force G (1, 3, 9), bits 16-26: force the row split count
  return old;
}

int mink_conv_trace(void *buf, int64_t capacity_workgroups) {
  g_trace_buf = (unsigned long long *)buf, g_trace_cap = buf ? capacity_workgroups : 0;
  return MINK_OK;
}

int mink_conv_set_pipeline(int mode) {
  const int old = g_compact_p3;
  if (mode >= 0 && mode <= 15) g_compact_p3 = mode;  // (bit 2, measurement only: the two-stage form with the LDS footprint -- hence the occupancy -- of the three-stage one)
  return old;
}

int mink_conv_timing(int32_t mode, int32_t kind, int32_t K, int32_t cin, int32_t cout) {
  MINK_REQUIRE(mode >= 0 && mode <= 2, "conv_timing: mode %d", mode);
  const int old = g_time_mode;
  g_time_mode = mode, g_time_kind = kind, g_time_K = K, g_time_cin = cin, g_time_cout = cout;
  return old;
}

int64_t mink_conv_timing_fetch(MinkTimingEntry *out, int64_t max) {
  std::lock_guard<std::mutex> lk(g_time_mu);
  if (!out) return (int64_t)g_timed.size();
  int64_t n = 0;
  for (TimedLaunch &t : g_timed) {
    if (n < max) {
      float ms = -1.f;
      if (hipEventSynchronize(t.b) == hipSuccess) (void)hipEventElapsedTime(&ms, t.a, t.b);
      t.e.ms = ms;
      out[n++] = t.e;
    }
    g_event_pool.push_back(t.a);
    g_event_pool.push_back(t.b);
  }
  g_timed.clear();
  return n;
}

int mink_conv_get_math(void) { return g_math; }

int mink_conv_set_math(int mode) {
  const int old = g_math;
  if (mode == 0 || mode == 1 || mode == 3) g_math = mode;
  return old;
}

int mink_conv_plan_ksplit(int64_t n_out, int32_t K, int32_t cout, int32_t row_classes) {
  if (n_out <= 0 || K <= 1) return 1;
  const int64_t tiles = cdiv(n_out, BM) * cdiv(cout, BN);
  if (tiles >= 768) return 1;
  // Small row counts: split the offsets over gridDim.z.  Measured on the ResNet layers (kbench
  // ksweep): the best split puts just under one resident round (2 workgroups x 256 CUs) of
  // workgroups on the chip; with more than 256 tiles it is the split that best fills whole
  // rounds, at ~1% per extra slab for the partial-sum write + reduce (the slabs stay in MALL).
  // A class-permuted dgrad (row_classes) has ~K/8 live offsets per tile and wants no split there.
  const int64_t slab_cap = (128ll << 20) / (4 * n_out * cout);  // keep the slabs cache resident
  int best = 1;
  double best_score = -1.0;
  for (int kper = K; kper >= 1; --kper) {
    const int zs = (int)cdiv(K, kper);
    if (kper > 1 && cdiv(K, kper - 1) == zs) continue;  // a smaller kper gives the same split with less work
    if (zs > 1 && zs > slab_cap) break;
    const int64_t blocks = tiles * zs;
    double score;
    if (tiles <= 256) {  // the largest split within one round
      score = blocks <= 512 ? (double)blocks : 0.0;
      if (row_classes && zs == 3) score = 0.0;  // thirds (one dz plane each) cut the parity classes badly,
      if (row_classes && zs == 4 && blocks <= 640) score = (double)blocks;  // quarters slightly over a round are fine
    }
    else if (row_classes) score = zs == 1;
    // (bf16 math: the matrix work is a sixteenth, so a slab's write + reduce weighs five times as much against it --
    //  layer1 at B=16, scripts/ksplit_sweep.py: 7 slabs 65 us, 3 slabs 56)
    else score = zs > 7 ? 0.0 : (double)blocks / (512.0 * cdiv(blocks, 512)) - (g_math == 1 ? 0.05 : 0.01) * zs;
    if (score > best_score) best_score = score, best = zs;
  }
  return best;
}

// XCD-aware launch of compact_gemm_kernel for the layers whose weights do not stay in one L2 (4 MB): few row tiles, many
// (column tile, slice) pairs.  Turns the three-dimensional grid into the padded one-dimensional one the kernel decodes.
static void compact_swizzle(GemmParams &p, dim3 &grid, int cin, int cout, int K) {
  p.swz_x = p.swz_y = p.swz_z = 0;
  p.trace = nullptr;
  const int64_t wbytes = 4ll * K * cin * cout;
  const unsigned slices = grid.y * grid.z;
  if (wbytes <= (2ll << 20) || grid.x > 64 || slices < 16) return;
  p.swz_x = (int)grid.x, p.swz_y = (int)grid.y, p.swz_z = (int)grid.z;
  grid = dim3((unsigned)(cdiv(slices, 8) * 8 * grid.x), 1, 1);
}

// compact_gemm_kernel (fp32, no row permutation): 64-row tiles, four workgroups per CU, at most CKP offsets per workgroup.
// kbench ksweep on the ResNet layers: the best split is the largest one that keeps the launch within two resident rounds
// (2 x 1024 workgroups), capped at 14 slabs -- l2.conv2 83 us at 7 slabs against 94 at the 3 the older rule picks.
static bool compact_shape(int64_t n_rows, int K, int cin, int cout, int row_classes) {
  if (!(g_compact && g_math == 0 && !row_classes && K >= 8 && cin >= 64 && cin % BK == 0 && cout % BN == 0 && n_rows >= 1))
    return false;
  const int64_t zmin = cdiv(K, CKP);  // the fewest slabs the rulebook allows; beyond the slab budget: the dense kernel, un-split
  return zmin == 1 || zmin * 4 * n_rows * cout <= (128ll << 20);
}
static int compact_plan(int64_t n_rows, int K, int cout) {
  if (g_compact_p3 & 8) {  // stream-K: the slab count of its plan
    const SkPlan sk = sk_plan(n_rows, K, cout);
    if (sk.G) return sk.S;
  }
  constexpr int wg_cap = 2048;
  const int64_t tiles = cdiv(n_rows, 64) * cdiv(cout, BN);
  const int64_t slab_cap = std::max<int64_t>(1, (128ll << 20) / (4 * n_rows * cout));
  int best = (int)cdiv(K, CKP);
  for (int kper = CKP; kper >= 1; --kper) {
    const int zs = (int)cdiv(K, kper);
    if (zs > 14 || tiles * zs > wg_cap || (zs > 1 && zs > slab_cap)) break;
    // (seven slabs that already fill a resident round are not worth doubling: layer 1 at four scenes, 144 tiles -- 7 slabs 33-35 us,
    //  14 slabs 37-39, scripts/ksplit_sweep.py fp32 30 4; the reduce reads twice the slabs for the same round count per CU)
    if (zs > 7 && best == 7 && tiles * 7 >= 1000) break;
    best = zs;
  }
  return best;
}

// compact_gemm_kernel<.., PERM> (fp32, rows grouped by parity class: the data gradient of a strided convolution): the split
// is over 32-channel chunks; kbench ksweep: the largest that keeps the launch within ~800 workgroups (l2.conv1 three slices 51 us
// against 57 at two or four, l3.conv1 five 53 against 57 at seven).
static bool compact_perm_shape(int64_t n_rows, int K, int cin, int cout, int row_classes) {
  return g_compact && g_compact_perm && (g_math == 0 || (g_math == 1 && g_compact_perm16)) && row_classes && K >= 8 &&
         cin >= (g_compact_cin32 ? 32 : 64) && cin % BK == 0 && cout % BN == 0 && n_rows >= 1;
}
static int compact_perm_plan(int64_t n_rows, int cin, int cout) {
  constexpr int cap = 850;
  const int ncc = cin / BK;
  // (n_rows is the padded length of the class permutation: up to 127 padding rows per class, whose tiles exit at once)
  const int64_t tiles = cdiv(std::max<int64_t>(n_rows - 512, 64), 64) * cdiv(cout, BN);
  int best = 1;
  // (up to nine slices where the launch stays under the cap: layer 4 at four scenes per GPU, 5 slices 33 us, 9 slices 27 -- ksplit_sweep, round 5)
  for (int zs = 2; zs <= std::min(ncc, 9); ++zs) {  // (a split that does not divide the chunk count just has a shorter last slice)
    if (tiles * zs > cap || zs * 4 * n_rows * cout > (128ll << 20)) break;
    best = zs;
  }
  return best;
}

int mink_conv_plan(int64_t n_rows, int32_t K, int32_t cin, int32_t cout, int32_t row_classes) {
  if (compact_perm_shape(n_rows, K, cin, cout, row_classes)) return compact_perm_plan(n_rows, cin, cout);
  if (compact_shape(n_rows, K, cin, cout, row_classes)) return compact_plan(n_rows, K, cout);
  return mink_conv_plan_ksplit(n_rows, K, cout, row_classes);
}

// stats_out (optional): double [<= 512][2][cout] column (sum, sum of squares) partials of y for the
// batch norm that follows; *stats_rows receives how many partial rows were written (0: this
// launch configuration cannot produce them and the caller reduces y itself).
static int gather_gemm_impl(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *w, int32_t w_transposed,
                            int32_t flip_k, const int32_t *nbr, int64_t n_out, int32_t K, const int32_t *row_perm,
                            int64_t n_virtual, float *y, int32_t ldy, int32_t cout, const float *bias, int32_t ksplit,
                            float *workspace, int64_t workspace_bytes, double *stats_out, int32_t *stats_rows, void *stats_ws,
                            int64_t stats_ws_bytes, void *stream, int32_t *slabs_out = nullptr) {
  // slabs_out != NULL: a split launch LEAVES its partial slabs in `workspace` ([*slabs_out][n_out][cout], summed by the
  // caller's next kernel -- mink_bn_small_fwd); *slabs_out = 1 means y holds the result as usual
  if (slabs_out) *slabs_out = 1;
  if (stats_rows) *stats_rows = 0;
  MINK_REQUIRE(K >= 1 && K <= KMAX, "gather_gemm: kernel volume %d unsupported", K);
  MINK_REQUIRE(cin >= 1 && cout >= 1 && ldx >= cin && ldy >= cout && n_out >= 0 && n_in >= 0, "gather_gemm: bad shape");
  MINK_REQUIRE(ksplit >= 1 && ksplit <= std::max(K, cin / BK), "gather_gemm: bad ksplit %d", ksplit);  // (offsets, or channel chunks: compact_perm_plan)
  MINK_REQUIRE(n_out * (int64_t)K < (1ll << 31), "gather_gemm: table too large");
  if (n_out == 0) return MINK_OK;
  MINK_REQUIRE(x && w && nbr && y, "gather_gemm: NULL pointer");
  MINK_REQUIRE(ksplit == 1 || workspace, "gather_gemm: split-K needs a workspace");
  MINK_REQUIRE(ksplit == 1 || workspace_bytes >= 4ll * ksplit * n_out * cout,
               "gather_gemm: workspace of %lld bytes, %lld needed for %d slabs of %lld x %d", (long long)workspace_bytes,
               (long long)(4ll * ksplit * n_out * cout), ksplit, (long long)n_out, cout);
  MINK_REQUIRE(!stats_ws || stats_ws_bytes >= mink_conv_stats_workspace_bytes(n_out, cout),
               "gather_gemm: statistics workspace of %lld bytes, %lld needed", (long long)stats_ws_bytes,
               (long long)mink_conv_stats_workspace_bytes(n_out, cout));
  if (!row_perm) n_virtual = n_out;
  MINK_REQUIRE(n_virtual >= n_out || (flip_k & 2), "gather_gemm: the row permutation must cover every output row");
  ScopedTimer timer(w_transposed ? 1 : 0, n_in, n_out, K, cin, cout, nbr, (hipStream_t)stream);
  GemmParams p;
  p.row_perm = row_perm, p.n_virtual = n_virtual;
  p.stagger = g_stagger;
  p.x = x, p.w = w, p.nbr = nbr, p.bias = bias, p.y = y, p.ws = workspace;
  p.n_out = n_out, p.ldx = ldx, p.cin = cin, p.ldy = ldy, p.cout = cout, p.K = K, p.flip_k = flip_k & 1;
  p.accumulate = (flip_k >> 1) & 1;
  p.kper = (int)cdiv(K, ksplit);
  p.stats = nullptr;
  p.swz_x = p.swz_y = p.swz_z = 0;
  p.trace = nullptr;
  p.sk_q = p.sk_r = 0, p.sk_X = 0, p.sk_S = 0;
  // stream-K (compact_gemm_kernel<.., SK>): a split stride-1 launch whose caller planned at least the slabs the plan needs
  SkPlan sk = {0, 0};
  if ((g_compact_p3 & 8) && ksplit > 1 && !row_perm && !(g_stagger & 0xFC) && !g_trace_buf && compact_shape(n_out, K, cin, cout, 0)) {
    sk = sk_plan(n_out, K, cout);
    if (sk.G == 0 || sk.S > ksplit) sk = {0, 0};
  }
  const int zs = sk.G ? ksplit : (int)cdiv(K, p.kper);  // (stream-K: exactly the slabs the caller planned; the spare ones are zeroed)
  const dim3 grid((unsigned)cdiv(n_virtual, BM), (unsigned)cdiv(cout, BN), (unsigned)zs);
  hipStream_t st = (hipStream_t)stream;
  const bool al = (((uintptr_t)x | (uintptr_t)w) & 15) == 0 && (ldx & 3) == 0 && (cin & 3) == 0;
  const bool vec = al && (w_transposed ? true : (cout & 3) == 0);
  MINK_REQUIRE(!p.accumulate || (zs == 1 && vec && !stats_out),
               "gather_gemm: accumulation needs an un-split launch of the pipelined kernel (16-byte aligned operands)");
  flip_k &= 1;
  const bool want_stats = stats_out && stats_rows && stats_ws && !row_perm && !w_transposed;
  const bool stats_direct = want_stats && zs == 1 && vec;  // conv epilogue -> per-tile partials -> stage 2
  const bool stats_split = want_stats && zs > 1 && (cout & 3) == 0 && cout <= 1024 && (ldy & 3) == 0 &&
                           (((uintptr_t)y | (uintptr_t)workspace | (uintptr_t)bias) & 15) == 0;
  if (compact_perm_shape(n_virtual, K, cin, cout, row_perm != nullptr) && row_perm && vec && !p.accumulate && !stats_out &&
      (g_math == 0 || w_transposed) &&  // (bf16 math: the data-gradient form only)
      (ldy & 3) == 0 && 4ll * K * cin * cout < (1ll << 31) && 4ll * n_in * ldx < (1ll << 32) &&  // (32-bit gather offsets)
      (((uintptr_t)y | (uintptr_t)workspace | (uintptr_t)bias) & 15) == 0) {
    // class-permuted rows, every live offset of a tile, split over channel chunks (compact_gemm_kernel<.., PERM>)
    constexpr int CMT = 64;
    constexpr int smem = compact_smem(CMT);
    static const bool attr_ok = [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<false, CMT, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<true, CMT, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
    }();
    MINK_REQUIRE(attr_ok, "gather_gemm: %d bytes of LDS per workgroup refused", smem);
    const int ncc = cin / BK;
    p.kper = (int)cdiv(ncc, ksplit);  // channel chunks per slice
    const int zc = (int)cdiv(ncc, p.kper);
    dim3 cgrid((unsigned)cdiv(n_virtual, CMT), grid.y, (unsigned)zc);
    compact_swizzle(p, cgrid, cin, cout, K);
    if (w_transposed && (g_compact_p3 & 2) && !(g_stagger & 0xFC)) {
      const int rc = g_math == 1 ? launch_compact_p3<true, true, 1>(p, cgrid, st) : launch_compact_p3<true, true, 0>(p, cgrid, st);
      if (rc) return rc;
    } else if (w_transposed && (g_stagger & 0xFC)) {  // timing-only switches (kbench cabp): the instantiation that carries them
      static const bool abl_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<true, CMT, true, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
      MINK_REQUIRE(abl_ok, "gather_gemm: %d bytes of LDS per workgroup refused", smem);
      compact_gemm_kernel<true, CMT, true, true><<<cgrid, 256, smem, st>>>(p);
    } else if (w_transposed && g_math == 1) {
      static const bool a16_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<true, CMT, true, false, 4, 1>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
      MINK_REQUIRE(a16_ok, "gather_gemm: %d bytes of LDS per workgroup refused", smem);
      compact_gemm_kernel<true, CMT, true, false, 4, 1><<<cgrid, 256, smem, st>>>(p);
    } else if (w_transposed) compact_gemm_kernel<true, CMT, true><<<cgrid, 256, smem, st>>>(p);
    else compact_gemm_kernel<false, CMT, true><<<cgrid, 256, smem, st>>>(p);
    MINK_CHECK_LAUNCH();
    if (zc > 1 && slabs_out) {
      *slabs_out = zc;
    } else if (zc > 1) {
      splitk_reduce_kernel<<<dim3((unsigned)cdiv(n_out * cout, 256)), 256, 0, st>>>(workspace, n_out, cout, zc, bias, y, ldy);
      MINK_CHECK_LAUNCH();
    }
    return MINK_OK;
  }
  if (stats_direct) p.stats = (float *)stats_ws;
  unsigned tiles_x = grid.x;  // row tiles that wrote statistics partials
  // (--math bf16 keeps the stride-1 mid layers on the dense bf16 kernel: the row-compacted form with one bf16 MFMA per block and item,
  //  measured round 5, is no faster there -- l1.conv2 57 / 60 us against 56 / 56 forward / data gradient, l3 48 / 58 against 44 / 46)
  const bool compact = g_compact && g_math == 0 && vec && !row_perm && !p.accumulate && K >= 8 && (p.kper <= CKP || sk.G) && cin >= 64 &&
                       cin % BK == 0 && cout % BN == 0 && (ldy & 3) == 0 && 4ll * K * cin * cout < (1ll << 31) &&
                       4ll * n_in * ldx < (1ll << 32) && (((uintptr_t)y | (uintptr_t)workspace | (uintptr_t)bias) & 15) == 0;
  if (compact) {  // row-compacted offsets, C tile in LDS (compact_gemm_kernel)
    constexpr int CMT = 64;
    constexpr int smem = compact_smem(CMT);
    static const bool attr_ok = [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<false, CMT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 smem) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<true, CMT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 smem) == hipSuccess;
    }();
    MINK_REQUIRE(attr_ok, "gather_gemm: %d bytes of LDS per workgroup refused", smem);
    dim3 cgrid((unsigned)cdiv(n_out, CMT), grid.y, grid.z);
    tiles_x = cgrid.x;
    compact_swizzle(p, cgrid, cin, cout, K);
    if (sk.G) {
      p.swz_x = 0;
      const int64_t F = (int64_t)cdiv(n_out, CMT) * grid.y * K;
      p.sk_q = (unsigned)(F / sk.G), p.sk_r = (unsigned)(F % sk.G), p.sk_X = (int)cdiv(n_out, CMT), p.sk_S = zs;
      const int rc = w_transposed ? launch_compact_sk<true>(p, sk.G, st) : launch_compact_sk<false>(p, sk.G, st);
      if (rc) return rc;
    } else if (g_compact_p3 & 4) {  // (measurement only: two stages at three workgroups per CU)
      constexpr int smem3 = compact_smem(CMT, true);
      static const bool ok3 = hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<false, CMT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem3) == hipSuccess &&
                              hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<true, CMT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem3) == hipSuccess;
      MINK_REQUIRE(ok3, "gather_gemm: %d bytes of LDS per workgroup refused", smem3);
      if (w_transposed) compact_gemm_kernel<true, CMT><<<cgrid, 256, smem3, st>>>(p);
      else compact_gemm_kernel<false, CMT><<<cgrid, 256, smem3, st>>>(p);
    } else
    if ((g_compact_p3 & 1) && !(g_stagger & 0xFC)) {
      const int rc = w_transposed ? launch_compact_p3<true, false, 0>(p, cgrid, st) : launch_compact_p3<false, false, 0>(p, cgrid, st);
      if (rc) return rc;
    } else if (w_transposed) compact_gemm_kernel<true, CMT><<<cgrid, 256, smem, st>>>(p);
    else if ((g_stagger & 0xFC) || g_trace_buf) {  // timing-only switches (kbench cab) / phase trace (kbench ctrace): the instantiation that carries them
      p.trace = (int64_t)cgrid.x * cgrid.y * cgrid.z <= g_trace_cap ? g_trace_buf : nullptr;
      static const bool abl_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&compact_gemm_kernel<false, CMT, false, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
      MINK_REQUIRE(abl_ok, "gather_gemm: %d bytes of LDS per workgroup refused", smem);
      compact_gemm_kernel<false, CMT, false, true><<<cgrid, 256, smem, st>>>(p);
    } else compact_gemm_kernel<false, CMT><<<cgrid, 256, smem, st>>>(p);
  } else {
    const bool stage = row_perm != nullptr;
    if (vec) {
      // the flat path addresses x rows with a 24-bit multiply: the table must not name a row >= 2^24 - 1 (n_in bounds it)
      const bool flat = cin == 28 && zs == 1 && !flip_k && !w_transposed && !stage && g_flat && ldx <= 32 &&
                        4ll * K * cin * cout < (1ll << 31) && n_in < (1 << 24) - 1;
#define MINK_LAUNCH_GG2(M)                                                                          \
  do {                                                                                              \
    if (w_transposed && stage) gather_gemm2_kernel<true, true, 0, M><<<grid, 256, 0, st>>>(p);      \
    else if (w_transposed) gather_gemm2_kernel<true, false, 0, M><<<grid, 256, 0, st>>>(p);         \
    else if (stage) gather_gemm2_kernel<false, true, 0, M><<<grid, 256, 0, st>>>(p);                \
    else if (flat) gather_gemm2_kernel<false, false, 28, M><<<grid, 256, 0, st>>>(p);               \
    else gather_gemm2_kernel<false, false, 0, M><<<grid, 256, 0, st>>>(p);                          \
  } while (0)
      if (g_math == 1) MINK_LAUNCH_GG2(1);
      else if (g_math == 3) MINK_LAUNCH_GG2(3);
      else MINK_LAUNCH_GG2(0);
#undef MINK_LAUNCH_GG2
    } else if (w_transposed) gather_gemm_kernel<true, false><<<grid, 256, 0, st>>>(p);  // operands that are not 16-byte rows (e.g. 27 channels)
    else gather_gemm_kernel<false, false><<<grid, 256, 0, st>>>(p);
  }
  MINK_CHECK_LAUNCH();
  if (stats_direct) {
    const int rows = (int)std::min<int64_t>(512, cdiv((int64_t)tiles_x, 8));  // >= 8 tiles per partial row: a latency-bound pass
    colsum_f32_kernel<<<dim3((unsigned)rows), 256, 0, st>>>((const float *)stats_ws, (int64_t)tiles_x, 2 * cout, stats_out);
    MINK_CHECK_LAUNCH();
    *stats_rows = rows;
  }
  if (zs > 1 && slabs_out) {
    MINK_REQUIRE(!bias && !stats_out, "gather_gemm: slabs are left to the caller only without bias and statistics");
    *slabs_out = zs;
    return MINK_OK;
  }
  if (zs > 1 && stats_split) {
    const int tpr = cout >> 2, rlanes = 256 / tpr;
    const int rows = (int)std::max<int64_t>(1, std::min<int64_t>(512, cdiv(n_out, (int64_t)rlanes * 4)));  // one four-row trip per thread
    splitk_reduce_stats_kernel<<<dim3((unsigned)rows), 256, (size_t)rlanes * 2 * cout * sizeof(double), st>>>(
        workspace, n_out, cout, zs, bias, y, ldy, stats_out);
    MINK_CHECK_LAUNCH();
    *stats_rows = rows;
  } else if (zs > 1) {
    splitk_reduce_kernel<<<dim3((unsigned)cdiv(n_out * cout, 256)), 256, 0, st>>>(workspace, n_out, cout, zs, bias, y,
                                                                                  ldy);
    MINK_CHECK_LAUNCH();
  }
  return MINK_OK;
}

int mink_conv_gather_gemm(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *w, int32_t w_transposed,
                          int32_t flip_k, const int32_t *nbr, int64_t n_out, int32_t K, const int32_t *row_perm,
                          int64_t n_virtual, float *y, int32_t ldy, int32_t cout, const float *bias, int32_t ksplit,
                          float *workspace, int64_t workspace_bytes, void *stream) {
  return gather_gemm_impl(x, n_in, ldx, cin, w, w_transposed, flip_k, nbr, n_out, K, row_perm, n_virtual, y, ldy, cout, bias,
                          ksplit, workspace, workspace_bytes, nullptr, nullptr, nullptr, 0, stream);
}

int mink_conv_gather_gemm_slabs(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *w, int32_t w_transposed,
                                int32_t flip_k, const int32_t *nbr, int64_t n_out, int32_t K, const int32_t *row_perm,
                                int64_t n_virtual, float *y, int32_t ldy, int32_t cout, int32_t ksplit, float *workspace,
                                int64_t workspace_bytes, int32_t *slabs_out, void *stream) {
  MINK_REQUIRE(slabs_out, "gather_gemm_slabs: NULL slabs_out");
  return gather_gemm_impl(x, n_in, ldx, cin, w, w_transposed, flip_k, nbr, n_out, K, row_perm, n_virtual, y, ldy, cout, nullptr,
                          ksplit, workspace, workspace_bytes, nullptr, nullptr, nullptr, 0, stream, slabs_out);
}

int64_t mink_conv_stats_workspace_bytes(int64_t n_out, int32_t cout) {
  return (int64_t)cdiv(n_out > 0 ? n_out : 1, 64) * 2 * cout * sizeof(float);  // (row tiles of 64: compact_gemm_kernel)
}

int mink_conv_gather_gemm_stats(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *w, const int32_t *nbr,
                                int64_t n_out, int32_t K, float *y, int32_t ldy, int32_t cout, const float *bias,
                                int32_t ksplit, float *workspace, int64_t workspace_bytes, double *stats_out,
                                int32_t *stats_rows, void *stats_ws, int64_t stats_ws_bytes, void *stream) {
  MINK_REQUIRE(stats_out && stats_rows && stats_ws, "gather_gemm_stats: NULL statistics buffers");
  return gather_gemm_impl(x, n_in, ldx, cin, w, 0, 0, nbr, n_out, K, nullptr, 0, y, ldy, cout, bias, ksplit, workspace,
                          workspace_bytes, stats_out, stats_rows, stats_ws, stats_ws_bytes, stream);
}

int64_t mink_conv_wgrad_workspace_bytes(int64_t n_out, int32_t K, int32_t cin, int32_t cout) {
  const WgradPlan pl = wgrad_plan(n_out, K, cin, cout);
  return pl.nsplit > 1 ? (int64_t)pl.nsplit * K * cin * cout * 4 : 0;
}

struct WgradFuse {  // dy = input gradient of pool(relu(bn(y))): see mink_conv_wgrad_bn_relu_pool
  const float *dyp;
  const int32_t *in2out;
  int64_t n_pool;
  const float *mean, *invstd, *gamma, *beta, *dgamma, *dbeta;
  int b16;  // x and y are bf16 (x: [n_in][ldx] with ldx = 32; y: [n_out][cout]) -- bf16 storage of the full-resolution stage
};

static bool wgrad_stream_ok(int64_t n_in, int32_t ldx, int32_t cin, int32_t ldy, int32_t cout, int64_t n_out, int32_t K) {
  return K == 27 && cin <= 32 && ldx < 64 && n_in < (1 << 24) && 4 * n_in * ldx < (1ll << 31) &&
         4 * n_out * ldy < (1ll << 31) && 4 * n_out * K < (1ll << 31);  // what the buffer-offset arithmetic assumes
}

static int wgrad_impl(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *dy, int32_t ldy, int32_t cout,
                      const int32_t *nbr, int64_t n_out, int32_t K, float *dw, void *workspace, int64_t workspace_bytes,
                      const WgradFuse *fuse, void *stream) {
  MINK_REQUIRE(K >= 1 && K <= KMAX && cin >= 1 && cout >= 1 && ldx >= cin && ldy >= cout && n_out >= 0 && n_in >= 0,
               "wgrad: bad shape");
  MINK_REQUIRE(dw, "wgrad: NULL dw");
  hipStream_t st = (hipStream_t)stream;
  if (n_out == 0) {
    MINK_HIP(hipMemsetAsync(dw, 0, sizeof(float) * K * cin * cout, st));
    return MINK_OK;
  }
  MINK_REQUIRE(x && dy && nbr, "wgrad: NULL pointer");
  ScopedTimer timer(2, n_in, n_out, K, cin, cout, nbr, st);
  const WgradPlan pl = wgrad_plan(n_out, K, cin, cout);
  MINK_REQUIRE(pl.nsplit == 1 || workspace, "wgrad: needs a workspace");
  // (the slab count comes from the plan, which tuning knobs can change between the caller's size query and this launch)
  MINK_REQUIRE(pl.nsplit == 1 || workspace_bytes >= (int64_t)pl.nsplit * K * cin * cout * 4,
               "wgrad: workspace of %lld bytes, %lld needed for the %d row splits of this plan", (long long)workspace_bytes,
               (long long)((int64_t)pl.nsplit * K * cin * cout * 4), pl.nsplit);
  WgradParams p;
  p.x = x, p.dy = dy, p.nbr = nbr, p.out = pl.nsplit > 1 ? (float *)workspace : dw;
  p.n_out = n_out, p.rows_per_split = pl.rows_per_split, p.ldx = ldx, p.cin = cin, p.ldy = ldy, p.cout = cout, p.K = K;
  p.ct_tiles = (int)cdiv(cout, WT);
  p.ngroups = pl.ngroups;
  p.ablate = g_stagger | (g_wgrad_xcd ? 0 : 4096);
  const dim3 grid((unsigned)(pl.ngroups * cdiv(cin, WT) * p.ct_tiles), (unsigned)pl.nsplit);
  const int esz = fuse && fuse->b16 ? 2 : 4;
  const int64_t xb = esz * n_in * ldx, db = esz * n_out * ldy, nb = 4 * n_out * K;
  p.x_bytes = (unsigned)xb, p.dy_bytes = (unsigned)db, p.nbr_bytes = (unsigned)nb;
  p.buf_ok = xb < (1ll << 31) && db < (1ll << 31) && nb < (1ll << 31) && n_in < (1 << 24) && xb <= 0xFFFFFFll * 4 * ldx;
  const bool stream_ok = wgrad_stream_ok(n_in, ldx, cin, ldy, cout, n_out, K);
  const bool bf16_stream = g_math == 1 && g_wgrad_bf16 && !g_wgrad_bf16_off && pl.G == 9 && stream_ok && g_wgrad_stream;
  // flattened (offset, channel) tiling: 24 instead of 27 tiles when the axis fits three groups of 256 rows and the
  // padded channels are worth saving; ldx <= 32 keeps "no neighbour" + "past the axis" inside 32-bit offset arithmetic
  constexpr int flat_on = 1;
  const bool flat = flat_on && K * cin <= 768 && K * cin > 512 && ldx <= 32 && cin >= 16;
  if (fuse) {
    MINK_REQUIRE(pl.G == 9 && stream_ok && g_wgrad_stream && 4 * fuse->n_pool * ldy < (1ll << 31),
                 "wgrad_bn_relu_pool: shape not supported by the streaming kernel (ask mink_conv_wgrad_bn_relu_pool_supported)");
    p.dyp = fuse->dyp, p.in2out = fuse->in2out, p.mean = fuse->mean, p.invstd = fuse->invstd, p.gamma = fuse->gamma;
    p.beta = fuse->beta, p.dgamma = fuse->dgamma, p.dbeta = fuse->dbeta, p.inv_n = 1.f / (float)n_out;
    p.dyp_bytes = (unsigned)(4 * fuse->n_pool * ldy), p.i2o_bytes = (unsigned)(4 * n_out);
    MINK_REQUIRE(!fuse->b16 || bf16_stream, "wgrad_bn_relu_pool_b16: needs bf16 math (mink_conv_set_math(1))");
    if (fuse->b16 && cout == 64 && ldx == 32 && !g_b16t_off) wgrad_stream_b16t_kernel<<<grid, 256, 0, st>>>(p);  // (bit 11: the 2-byte-gather kernel, A/B tests)
    else if (fuse->b16) wgrad_stream_bf16_kernel<true, true><<<grid, 256, 0, st>>>(p);
    else if (bf16_stream) wgrad_stream_bf16_kernel<true><<<grid, 256, 0, st>>>(p);  // (four row pairs in flight: 2 / 6 / 8 measured, DESIGN appendix)
    else if (flat) wgrad_stream_kernel<4, true, true><<<grid, 256, 0, st>>>(p);
    else wgrad_stream_kernel<4, true><<<grid, 256, 0, st>>>(p);
  } else if (bf16_stream) wgrad_stream_bf16_kernel<false><<<grid, 256, 0, st>>>(p);
  else if (pl.G == 9 && stream_ok && g_wgrad_stream && flat) wgrad_stream_kernel<4, false, true><<<grid, 256, 0, st>>>(p);
  else if (pl.G == 9 && stream_ok && g_wgrad_stream) wgrad_stream_kernel<4><<<grid, 256, 0, st>>>(p);
  else if (g_math == 1 && g_wgrad_bf16 && !g_wgrad_bf16_off && pl.G != 9 && cin % WT == 0 && cout % WT == 0 && p.buf_ok &&
           (((uintptr_t)x | (uintptr_t)dy) & 15) == 0 && ((ldx | ldy) & 3) == 0) {
    // --math bf16: the mid-layer weight gradients on the bf16 matrix cores too (wgrad16_kernel)
    if (pl.G == 3) wgrad16_kernel<3><<<grid, 256, 0, st>>>(p);
    else wgrad16_kernel<1><<<grid, 256, 0, st>>>(p);
  } else if (pl.G == 9) launch_wgrad<9>(p, grid, st);
  else if (pl.G == 3) launch_wgrad<3>(p, grid, st);
  else launch_wgrad<1>(p, grid, st);
  MINK_CHECK_LAUNCH();
  if (pl.nsplit > 1) {
    const int64_t count = (int64_t)K * cin * cout;
    slab_reduce_kernel<<<dim3((unsigned)cdiv(count, 64)), 256, 0, st>>>((const float *)workspace, count, pl.nsplit, dw);
    MINK_CHECK_LAUNCH();
  }
  return MINK_OK;
}

int mink_conv_wgrad(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *dy, int32_t ldy, int32_t cout,
                    const int32_t *nbr, int64_t n_out, int32_t K, float *dw, void *workspace, int64_t workspace_bytes,
                    void *stream) {
  return wgrad_impl(x, n_in, ldx, cin, dy, ldy, cout, nbr, n_out, K, dw, workspace, workspace_bytes, nullptr, stream);
}

int mink_conv_wgrad_bn_relu_pool_supported(int64_t n_in, int32_t ldx, int32_t cin, int64_t n_out, int32_t K, int32_t cout) {
  if (n_out <= 0) return 0;
  return wgrad_plan(n_out, K, cin, cout).G == 9 && wgrad_stream_ok(n_in, ldx, cin, cout, cout, n_out, K) && g_wgrad_stream;
}

int mink_conv_wgrad_bn_relu_pool(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *y, int32_t cout,
                                 const float *dy_pool, int64_t n_pool, const int32_t *in2out, const float *mean,
                                 const float *invstd, const float *gamma, const float *beta, const float *dgamma,
                                 const float *dbeta, const int32_t *nbr, int64_t n_out, int32_t K, float *dw,
                                 void *workspace, int64_t workspace_bytes, void *stream) {
  MINK_REQUIRE(y && dy_pool && in2out && mean && invstd && gamma && beta && dgamma && dbeta && n_pool >= 1,
               "wgrad_bn_relu_pool: NULL pointer");
  const WgradFuse f = {dy_pool, in2out, n_pool, mean, invstd, gamma, beta, dgamma, dbeta, 0};
  return wgrad_impl(x, n_in, ldx, cin, y, cout, cout, nbr, n_out, K, dw, workspace, workspace_bytes, &f, stream);
}

int mink_conv_wgrad_bn_relu_pool_b16(const void *xb, int64_t n_in, int32_t cin, const void *yb, int32_t cout, const float *dy_pool,
                                     int64_t n_pool, const int32_t *in2out, const float *mean, const float *invstd,
                                     const float *gamma, const float *beta, const float *dgamma, const float *dbeta,
                                     const int32_t *nbr, int64_t n_out, int32_t K, float *dw, void *workspace,
                                     int64_t workspace_bytes, void *stream) {
  MINK_REQUIRE(xb && yb && dy_pool && in2out && mean && invstd && gamma && beta && dgamma && dbeta && n_pool >= 1 && cin >= 1 && cin <= 32,
               "wgrad_bn_relu_pool_b16: bad arguments");
  const WgradFuse f = {dy_pool, in2out, n_pool, mean, invstd, gamma, beta, dgamma, dbeta, 1};
  return wgrad_impl((const float *)xb, n_in, 32, cin, (const float *)yb, cout, cout, nbr, n_out, K, dw, workspace, workspace_bytes, &f,
                    stream);
}

}  // extern "C"
