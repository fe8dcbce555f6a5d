// bf16 STORAGE for the full-resolution stage of the network (BASELINE config "bf16 mixed precision"): the input features
// and the stem convolution's output -- three quarters of all activation bytes of a Mink-ResNet step -- live in HBM as
// bf16; everything from the pooled level down stays fp32 (include/mink_hip.h, "bf16 storage").
//
//   rows_to_bf16_kernel    x fp32 [n][cin] -> bf16 [n][32], zero-padded: a gathered row is ONE 64-byte half cache line
//   stem_fwd_bf16s_kernel  y = conv(x) over the neighbour table (27 offsets, 64 output channels), y stored as bf16,
//                          column statistics of the STORED values for the batch norm that follows
//
// The forward kernel is a streamed GEMM with no operand tile in LDS: with x in bf16 a lane's A operand of
// v_mfma_f32_32x32x16_bf16 -- eight consecutive channels of its row -- is one 16-byte buffer load straight into the
// register the MFMA reads (a missing neighbour is an out-of-range offset that returns zeros), so the per-item barrier,
// LDS store and conversion of the fp32-storage kernel (gather_gemm2_kernel<.., MATH = 1>: 0.39 ms for 22 us of matrix
// work) disappear.  The weights (27 x 28 x 64, converted once per workgroup) sit in LDS for the whole kernel; after that
// one barrier the eight waves of a workgroup never synchronise: each streams its own 32-row blocks with nine offsets of
// gathers in flight.  One workgroup per CU (135 KB of LDS), blocks dealt to waves statically (bitwise reproducible
// statistics).
#include <algorithm>

#include "common.h"

namespace mink {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int S16_CP = 32;     // channels of a bf16 x row (zero-padded)
constexpr int S16_CO = 64;     // output channels
constexpr int S16_K = 27;      // offsets
constexpr int S16_LDW = 40;    // halfwords per (offset, output channel) row of the LDS weight image: 80 bytes, conflict-free ds_read_b128
constexpr int S16_WAVES = 8;
constexpr int S16_D = 9;       // offsets of gathers in flight per wave (27 % D == 0: ring slots are compile-time constants)
constexpr int S16_SMEM = S16_K * S16_CO * S16_LDW * 2;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ptr), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}

__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float *__restrict__ x, int64_t n, int c, int ldx,
                                                           uint4 *__restrict__ xb) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (row, 8-channel chunk)
  if (i >= n * 4) return;
  const int64_t row = i >> 2;
  const int c0 = (int)(i & 3) * 8;
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = c0 + e < c ? x[row * ldx + c0 + e] : 0.f;
  xb[i] = make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
}

struct Stem16Params {
  const void *xb;   // bf16 [n_in][32]
  const float *w;   // fp32 [27][cin][64]
  const int *nbr;   // [n_out][27]
  void *yb;         // bf16 [n_out][64]
  double *stats;    // [gridDim.x][2][64] column (sum, sum of squares) of the stored values
  int64_t n_out;
  unsigned xb_bytes, nbr_bytes, yb_bytes;
  int cin, nblk;
};

__global__ __launch_bounds__(64 * S16_WAVES) void stem_fwd_bf16s_kernel(Stem16Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s16_smem[];
  unsigned short *sW = reinterpret_cast<unsigned short *>(s16_smem);  // [27][64][LDW]: channel c of (k, co) at halfword c
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // ---- weights -> LDS (bf16, channels 28..31 zero)
  for (int rowi = tid; rowi < S16_K * S16_CO; rowi += 64 * S16_WAVES) {
    const int k = rowi >> 6, co = rowi & 63;
    float f[S16_CP];
#pragma unroll
    for (int ch = 0; ch < S16_CP; ++ch) f[ch] = ch < p.cin ? p.w[((int64_t)k * p.cin + ch) * S16_CO + co] : 0.f;
    uint4 *dst = reinterpret_cast<uint4 *>(sW + rowi * S16_LDW);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      dst[q] = make_uint4(pack2(f[8 * q], f[8 * q + 1]), pack2(f[8 * q + 2], f[8 * q + 3]), pack2(f[8 * q + 4], f[8 * q + 5]),
                          pack2(f[8 * q + 6], f[8 * q + 7]));
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rx = rsrc_of(p.xb, p.xb_bytes), rn = rsrc_of(p.nbr, p.nbr_bytes), ry = rsrc_of(p.yb, p.yb_bytes);
  const int gw = blockIdx.x * S16_WAVES + wave, nw = gridDim.x * S16_WAVES;
  // k-slot (step t, half h, j) of the 32x32x16 contraction <-> channel 16 h + 8 t + j: a lane reads 32 contiguous bytes of
  // its row in two loads, and the B fragments below follow the same map
  const unsigned xcol = 32u * (unsigned)h;
  const unsigned short *bw = sW + r * S16_LDW + 16 * h;

  float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;  // column statistics of this lane's columns r and 32 + r
  if (gw < p.nblk) {                             // (wave-uniform)
    unsigned idx[S16_K], idn[S16_K], dead, deadn;
    auto load_tab = [&](int b, unsigned (&dst)[S16_K], unsigned &dd) __attribute__((always_inline)) {
      const int64_t row = (int64_t)b * 32 + r;
      const bool ok = row < p.n_out;  // (a block past the end has no valid row)
      dd = ok ? 0u : 0xFFFFFFFFu;     // an out-of-range table read returns 0, a VALID row: such lanes are forced to "no neighbour"
      const unsigned off = ok ? (unsigned)row * (4u * S16_K) : OOB;
#pragma unroll
      for (int k = 0; k < S16_K; ++k) dst[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rn, (int)(off + 4u * k), 0, 0);
    };
    u32x4 ra[S16_D][2];
    auto issue = [&](int slot, unsigned id, unsigned dd) __attribute__((always_inline)) {
      const unsigned off = __umul24(id | dd, 2u * S16_CP) + xcol;  // -1 -> row 0xFFFFFF: beyond xb, reads as zeros
      ra[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0);
      ra[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(off + 16u), 0, 0);
    };
    int b = gw;
    load_tab(b, idx, dead);
#pragma unroll
    for (int d = 0; d < S16_D; ++d) issue(d, idx[d], dead);
    for (; b < p.nblk; b += nw) {
      load_tab(b + nw, idn, deadn);  // next block's table rows: a whole block to arrive
      f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
      for (int k = 0; k < S16_K; ++k) {
        const int slot = k % S16_D;
        const unsigned short *wk = bw + k * S16_CO * S16_LDW;
        const uint4 b00 = *reinterpret_cast<const uint4 *>(wk), b01 = *reinterpret_cast<const uint4 *>(wk + 8);
        const uint4 b10 = *reinterpret_cast<const uint4 *>(wk + 32 * S16_LDW), b11 = *reinterpret_cast<const uint4 *>(wk + 32 * S16_LDW + 8);
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, ra[slot][0]), a1 = __builtin_bit_cast(bf16x8, ra[slot][1]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, b00), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, b10), acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, b01), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, b11), acc1, 0, 0, 0);
        // the slot's registers are free: offset k + D of this block, or the first offsets of the wave's next block
        if (k + S16_D < S16_K) issue(slot, idx[k + S16_D], dead);
        else issue(slot, idn[k + S16_D - S16_K], deadn);
        __builtin_amdgcn_sched_barrier(0);  // one offset per scheduling region (unbounded, the scheduler pulls all 108 fragment reads up front)
      }
      // ---- epilogue of the block: C/D layout column = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 h
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = (int64_t)b * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const unsigned off = row < p.n_out ? (unsigned)row * (2u * S16_CO) + 2u * (unsigned)r : OOB;  // (a store past the end is dropped)
        const __bf16 va = (__bf16)acc0[q], vb = (__bf16)acc1[q];
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, va), ry, (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, vb), ry, (int)(off + 64u), 0, 0);
        const float fa = (float)va, fb = (float)vb;  // rows past the end hold exact zeros (no neighbours)
        sa += fa, qa += fa * fa, sb += fb, qb += fb * fb;
      }
#pragma unroll
      for (int k = 0; k < S16_K; ++k) idx[k] = idn[k];
      dead = deadn;
    }
  }

  // ---- statistics: halves of a wave, then the waves of the workgroup in a fixed order
  sa += __shfl_xor(sa, 32), qa += __shfl_xor(qa, 32), sb += __shfl_xor(sb, 32), qb += __shfl_xor(qb, 32);
  __syncthreads();  // every wave is done with the weight image
  float *red = reinterpret_cast<float *>(s16_smem);  // [wave][sum a, sq a, sum b, sq b][32]
  if (h == 0) {
    red[(wave * 4 + 0) * 32 + r] = sa, red[(wave * 4 + 1) * 32 + r] = qa;
    red[(wave * 4 + 2) * 32 + r] = sb, red[(wave * 4 + 3) * 32 + r] = qb;
  }
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 5, c = tid & 31;
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < S16_WAVES; ++w) v += (double)red[(w * 4 + which) * 32 + c];
    p.stats[((int64_t)blockIdx.x * 2 + (which & 1)) * S16_CO + (which >= 2 ? 32 : 0) + c] = v;
  }
}

int cu_count() {
  static const int n = [] {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
    return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }();
  return n;
}

}  // namespace
}  // namespace mink

using namespace mink;

extern "C" {

int mink_rows_to_bf16(const float *x, int64_t n, int32_t c, int32_t ldx, void *xb, void *stream) {
  MINK_REQUIRE(n >= 0 && c >= 1 && c <= S16_CP && ldx >= c, "rows_to_bf16: bad shape (%lld x %d, pitch %d; at most %d channels)",
               (long long)n, c, ldx, S16_CP);
  if (n == 0) return MINK_OK;
  MINK_REQUIRE(x && xb && ((uintptr_t)xb & 15) == 0, "rows_to_bf16: NULL or unaligned pointer");
  rows_to_bf16_kernel<<<dim3((unsigned)cdiv(n * 4, 256)), 256, 0, (hipStream_t)stream>>>(x, n, c, ldx, (uint4 *)xb);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_stem_conv_bf16s_supported(int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout) {
  return K == S16_K && cout == S16_CO && cin >= 1 && cin <= S16_CP && n_in >= 1 && n_out >= 1 && n_in < (1 << 24) - 1 &&
         n_out * (4ll * S16_K) < (1ll << 31);
}

int32_t mink_stem_conv_bf16s_stats_rows(void) { return cu_count(); }

int mink_stem_conv_bf16s(const void *xb, int64_t n_in, const float *w, int32_t cin, const int32_t *nbr, int64_t n_out, int32_t K,
                         void *yb, int32_t cout, double *stats_out, int32_t stats_rows, void *stream) {
  MINK_REQUIRE(mink_stem_conv_bf16s_supported(n_in, n_out, K, cin, cout),
               "stem_conv_bf16s: shape not supported (%lld -> %lld rows, K=%d, %d -> %d channels)", (long long)n_in, (long long)n_out,
               K, cin, cout);
  MINK_REQUIRE(xb && w && nbr && yb && stats_out && (((uintptr_t)xb | (uintptr_t)yb) & 15) == 0, "stem_conv_bf16s: NULL or unaligned pointer");
  const int grid = cu_count();
  MINK_REQUIRE(stats_rows == grid, "stem_conv_bf16s: %d statistics rows, the launch writes %d (mink_stem_conv_bf16s_stats_rows)",
               stats_rows, grid);
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&stem_fwd_bf16s_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, S16_SMEM) == hipSuccess;
  MINK_REQUIRE(attr_ok, "stem_conv_bf16s: %d bytes of LDS per workgroup refused", S16_SMEM);
  Stem16Params p;
  p.xb = xb, p.w = w, p.nbr = nbr, p.yb = yb, p.stats = stats_out, p.n_out = n_out;
  p.xb_bytes = (unsigned)(n_in * 2 * S16_CP), p.nbr_bytes = (unsigned)(n_out * 4 * S16_K), p.yb_bytes = (unsigned)(n_out * 2 * S16_CO);
  p.cin = cin, p.nblk = (int)cdiv(n_out, 32);
  stem_fwd_bf16s_kernel<<<dim3((unsigned)grid), 64 * S16_WAVES, S16_SMEM, (hipStream_t)stream>>>(p);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

}  // extern "C"
