// GPU-side scene augmentation: the per-scene random transforms of the reference's
// co3d_3d/src/data/transforms.py (RandomRotation :339-358, RandomAffine :395-427, CoordinateDropout
// :247-265, RandomHorizontalFlip :430-450, CoordinateUniformTranslation :284-294, CoordinateJitter
// :268-281, RandomScale :361-373, RandomTranslation :376-392, RandomFeatureJitter :22-41) applied to a
// whole batch in three launches, between the loader and TensorField.sparse():
//
//   count  : which voxels survive the dropout, per-block survivor counts, per-scene maxima of the
//            pre-flip coordinates (the flip is `max - c` over the surviving voxels)
//   scan   : exclusive scan of the block counts (one block)
//   apply  : transformed coordinates + (jittered) features of the survivors, compacted in order
//
// The host draws the per-SCENE randomness (gates, matrices, offsets -- a few dozen floats per scene,
// see mink_hip.h MINK_AUG_*); the per-VOXEL randomness (dropout, coordinate jitter, feature noise) is
// a counter-based Philox4x32-10 stream keyed (seed; voxel-in-scene, draw, scene stream), so the result
// does not depend on the launch geometry and the numpy oracle (oracle/augment.py) reproduces it.
// Coordinate arithmetic is fp32 with every product and sum rounded separately (no fma contraction):
// the oracle matches it bit for bit, so the voxels land in the same cells after flooring.
#include "common.h"

namespace mink {
namespace {

constexpr int kBlock = 256;
constexpr int kBoundStride = 32;  // words per scene in the flip-bounds array: one 128-byte line each, so scenes do not share an L2 atomic unit

struct Philox {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0, c1 = lo1, c2 = n2, c3 = lo0;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}

__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }  // [0,1), exact

__device__ __forceinline__ int scene_of(const int *__restrict__ scene_offsets, int n_scenes, int64_t i) {
  int lo = 0, hi = n_scenes;  // scene b with scene_offsets[b] <= i < scene_offsets[b+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)scene_offsets[mid] <= i) lo = mid;
    else hi = mid;
  }
  return lo;
}

// order-preserving float <-> uint (for atomicMax)
__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// out_j = ((v0*M[0][j] + v1*M[1][j]) + v2*M[2][j]), every operation rounded (row-vector times matrix)
__device__ __forceinline__ void vec_mat(const float v[3], const float *__restrict__ M, float out[3]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float t0 = v[0] * M[j], t1 = v[1] * M[3 + j], t2 = v[2] * M[6 + j];
    out[j] = (t0 + t1) + t2;
  }
}

__device__ __forceinline__ void pre_flip(const float *__restrict__ P, const float c[3], float p[3]) {
#pragma clang fp contract(off)
  vec_mat(c, P + MINK_AUG_A, p);
#pragma unroll
  for (int j = 0; j < 3; ++j) p[j] = p[j] + P[MINK_AUG_a + j];
}

// row i of a [n][4] coordinate array that is float32 or (as_int) int32, e.g. straight from mink_decode_plenoxel
__device__ __forceinline__ float4 load_coord(const void *__restrict__ coords, int64_t i, bool as_int) {
  if (as_int) {
    const int4 v = *reinterpret_cast<const int4 *>((const int *)coords + 4 * i);
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
  }
  return *reinterpret_cast<const float4 *>((const float *)coords + 4 * i);
}

__device__ __forceinline__ bool keeps(const float *__restrict__ P, uint32_t word) { return u01(word) >= P[MINK_AUG_DROPOUT]; }

__global__ __launch_bounds__(kBlock) void augment_count_kernel(const void *__restrict__ coords, bool as_int, int64_t n,
                                                               const int *__restrict__ scene_offsets, int n_scenes,
                                                               const float *__restrict__ params,
                                                               const uint32_t *__restrict__ streams, uint32_t k0,
                                                               uint32_t k1, int *__restrict__ block_counts,
                                                               uint32_t *__restrict__ bounds) {
  __shared__ int s_count[kBlock / 64];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool keep = false;
  int scene = 0;
  uint32_t ord[3] = {0u, 0u, 0u};  // 0 = below every encoded float
  if (i < n) {
    const int b = scene = scene_of(scene_offsets, n_scenes, i);
    const float *P = params + (int64_t)b * MINK_AUG_PARAMS;
    const uint32_t vox = (uint32_t)(i - scene_offsets[b]);
    const Philox r = philox4x32_10(vox, 0u, streams[b], 0u, k0, k1);
    keep = keeps(P, r.x);
    const bool fx = P[MINK_AUG_FLIP] != 0.f, fy = P[MINK_AUG_FLIP + 1] != 0.f, fz = P[MINK_AUG_FLIP + 2] != 0.f;
    if ((fx || fy || fz) && (keep || P[MINK_AUG_FLIP_ALL] != 0.f)) {
      const float4 c4 = load_coord(coords, i, as_int);
      const float c[3] = {c4.y, c4.z, c4.w};
      float p[3];
      pre_flip(P, c, p);
      if (fx) ord[0] = f2ord(p[0]);
      if (fy) ord[1] = f2ord(p[1]);
      if (fz) ord[2] = f2ord(p[2]);
    }
  }
  // one atomic per wave and axis when the wave lies inside one scene (nearly always: scenes are tens of
  // thousands of rows); per-lane atomics on 3 words per scene would serialise the whole batch
  const int b0 = __shfl(scene, 0);
  const bool uniform = __ballot(i < n && scene != b0) == 0ull;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    uint32_t v = ord[j];
    if (uniform) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
      // most waves do not raise the running maximum: look first (a stale, smaller value only costs an atomic)
      if ((threadIdx.x & 63) == 0 && v > __atomic_load_n(bounds + kBoundStride * b0 + j, __ATOMIC_RELAXED))
        atomicMax(bounds + kBoundStride * b0 + j, v);
    } else if (v != 0u) {
      atomicMax(bounds + kBoundStride * scene + j, v);
    }
  }
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0) s_count[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) t += s_count[w];
    block_counts[blockIdx.x] = t;
  }
}

// one block: exclusive scan of block_counts[nb] -> block_offsets[nb], total -> *n_kept
__global__ __launch_bounds__(kBlock) void augment_scan_kernel(const int *__restrict__ block_counts, int nb,
                                                              int *__restrict__ block_offsets, int *__restrict__ n_kept) {
  __shared__ int s[kBlock];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += kBlock) {
    const int j = base + threadIdx.x;
    const int v = j < nb ? block_counts[j] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < kBlock; d <<= 1) {  // Hillis-Steele inclusive scan
      const int add = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    if (j < nb) block_offsets[j] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry += s[kBlock - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_kept = carry;
}

constexpr int kDraws = MINK_AUG_MAX_CHANNELS / 4;  // Philox draws that can carry feature noise

struct RawCols {
  int inv[MINK_AUG_MAX_CHANNELS];  // feature column of every raw-layout column, -1 = not selected
  int raw[MINK_AUG_MAX_CHANNELS];  // raw-layout column ([xyzs 0:3 | density 3 | sh 4:31], co3d.py:205-214) of every feature column, -1 = none
};

__global__ __launch_bounds__(kBlock) void augment_apply_kernel(
    const void *__restrict__ coords, bool as_int, const float *__restrict__ feats, int64_t ldf, int C, int64_t n,
    const int *__restrict__ scene_offsets, int n_scenes, const float *__restrict__ params,
    const uint32_t *__restrict__ streams, uint32_t k0, uint32_t k1, const int *__restrict__ block_offsets,
    const uint32_t *__restrict__ bounds, RawCols cols, float *__restrict__ out_coords, float *__restrict__ out_feats,
    int64_t ldo) {
#pragma clang fp contract(off)
  __shared__ int s_wave[kBlock / 64];
  __shared__ int s_dst[kBlock];    // output row of each voxel of this block, -1 = dropped
  __shared__ int s_scene[kBlock];
  __shared__ uint32_t s_vox[kBlock];
  __shared__ int s_lo[kBlock], s_hi[kBlock], s_start[kBlock];
  __shared__ float s_std[kBlock];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool keep = false;
  int b = 0;
  uint32_t vox = 0;
  Philox r = {0, 0, 0, 0};
  const float *P = params;
  if (i < n) {
    b = scene_of(scene_offsets, n_scenes, i);
    P = params + (int64_t)b * MINK_AUG_PARAMS;
    vox = (uint32_t)(i - scene_offsets[b]);
    r = philox4x32_10(vox, 0u, streams[b], 0u, k0, k1);
    keep = keeps(P, r.x);
  }
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  int rank = block_offsets[blockIdx.x] + wave_rank(m);
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) rank += s_wave[w];
  s_dst[threadIdx.x] = keep ? rank : -1;
  s_scene[threadIdx.x] = b;
  s_vox[threadIdx.x] = vox;
  {  // raw columns [lo, hi) of this voxel take noise of scale s_std (an empty range without feature jitter)
    const bool fj = keep && P[MINK_AUG_FEAT_STD] != 0.f;
    const int lo = fj ? max((int)P[MINK_AUG_FEAT_START], 0) : 0;
    s_lo[threadIdx.x] = lo;
    s_hi[threadIdx.x] = fj ? max((int)P[MINK_AUG_FEAT_START] + (int)P[MINK_AUG_FEAT_DIM], lo) : 0;
    s_start[threadIdx.x] = fj ? (int)P[MINK_AUG_FEAT_START] : 0;
    s_std[threadIdx.x] = fj ? P[MINK_AUG_FEAT_STD] : 0.f;
  }
  if (keep) {
    const float4 c4 = load_coord(coords, i, as_int);
    const float c[3] = {c4.y, c4.z, c4.w};
    float p[3], q[3], o[3];
    pre_flip(P, c, p);
#pragma unroll
    for (int j = 0; j < 3; ++j) q[j] = P[MINK_AUG_FLIP + j] != 0.f ? ord2f(bounds[kBoundStride * b + j]) - p[j] : p[j];
    vec_mat(q, P + MINK_AUG_B, o);
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = o[j] + P[MINK_AUG_b + j];
    const float amp = P[MINK_AUG_JITTER];
    if (amp != 0.f) {
      const float jit[3] = {amp * (u01(r.y) - 0.5f), amp * (u01(r.z) - 0.5f), amp * (u01(r.w) - 0.5f)};
      float jo[3];
      vec_mat(jit, P + MINK_AUG_BJ, jo);
#pragma unroll
      for (int j = 0; j < 3; ++j) o[j] = o[j] + jo[j];
    }
    *reinterpret_cast<float4 *>(out_coords + 4 * (int64_t)rank) = make_float4(c4.x, o[0], o[1], o[2]);
  }
  __syncthreads();
  // features: the block copies its [256, C] slab cooperatively (coalesced reads, near-coalesced writes) ...
  const int64_t row0 = (int64_t)blockIdx.x * kBlock;
  const int rows = (int)min((int64_t)kBlock, n - row0);
#pragma unroll 4
  for (int idx = threadIdx.x; idx < rows * C; idx += kBlock) {
    const int v = idx / C, col = idx - v * C;
    const int dst = s_dst[v], raw = cols.raw[col];
    if (dst < 0 || (raw >= s_lo[v] && raw < s_hi[v])) continue;  // dropped | takes noise: below
    out_feats[(int64_t)dst * ldo + col] = feats[(row0 + v) * ldf + col];
  }
  // ... except the columns that take noise: one thread per (voxel, Philox draw) = four noise columns.
  // Box-Muller on draw 1 + j/4: words (0,1) -> normals 4(j/4), 4(j/4)+1; words (2,3) -> the other two
  for (int idx = threadIdx.x; idx < rows * kDraws; idx += kBlock) {
    const int v = idx / kDraws, d = idx - v * kDraws;
    const int dst = s_dst[v];
    if (dst < 0) continue;
    const float std = s_std[v];
    const int start = s_start[v], dim = s_hi[v] - start;
    if (std == 0.f || 4 * d >= dim) continue;
    const Philox g = philox4x32_10(s_vox[v], 1u + (uint32_t)d, streams[s_scene[v]], 0u, k0, k1);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t wa = h ? g.z : g.x, wb = h ? g.w : g.y;
      const float u1 = (float)((wa >> 8) + 1u) * 0x1p-24f;  // (0,1]
      const float rad = sqrtf(-2.f * logf(u1)), ang = 6.283185307179586f * u01(wb);
      float sn, cs;
      sincosf(ang, &sn, &cs);
      z[2 * h] = rad * cs, z[2 * h + 1] = rad * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * d + e, raw = start + j;
      if (j >= dim || raw < 0 || raw >= MINK_AUG_MAX_CHANNELS) continue;
      const int col = cols.inv[raw];
      if (col < 0) continue;
      // "randn - 0.5", as the reference (transforms.py:36)
      out_feats[(int64_t)dst * ldo + col] = feats[(row0 + v) * ldf + col] + (z[e] - 0.5f) * std;
    }
  }
}

}  // namespace
}  // namespace mink

using namespace mink;

extern "C" {

int64_t mink_augment_workspace_bytes(int64_t n, int32_t n_scenes) {
  const int64_t nb = cdiv(n > 0 ? n : 1, kBlock);
  return align_up(2 * nb * 4, 128) + (int64_t)n_scenes * kBoundStride * 4;
}

int mink_augment_scenes(const void *coords, int32_t coords_are_int32, const float *feats, int64_t ldf, int32_t C, int64_t n,
                        const int32_t *scene_offsets, int32_t n_scenes, const float *params, const uint32_t *streams,
                        uint64_t seed, const int32_t *raw_cols, float *out_coords, float *out_feats, int64_t ldo,
                        int32_t *n_kept, void *workspace, int64_t workspace_bytes, void *stream) {
  MINK_REQUIRE(workspace_bytes >= mink_augment_workspace_bytes(n, n_scenes) || n == 0,
               "augment_scenes: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
               (long long)mink_augment_workspace_bytes(n, n_scenes));
  MINK_REQUIRE(n >= 0 && n < (int64_t)1 << 31 && n_scenes >= 1 && C >= 1 && C <= MINK_AUG_MAX_CHANNELS && ldf >= C &&
                   ldo >= C,
               "augment_scenes: bad shape (n %lld, scenes %d, C %d, ldf %lld, ldo %lld; at most %d channels)", (long long)n,
               n_scenes, C, (long long)ldf, (long long)ldo, MINK_AUG_MAX_CHANNELS);
  MINK_REQUIRE(n_kept && raw_cols, "augment_scenes: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    MINK_HIP(hipMemsetAsync(n_kept, 0, 4, s));
    return MINK_OK;
  }
  MINK_REQUIRE(coords && feats && scene_offsets && params && streams && out_coords && out_feats && workspace,
               "augment_scenes: NULL pointer");
  MINK_REQUIRE((((uintptr_t)coords | (uintptr_t)out_coords) & 15) == 0, "augment_scenes: coordinates must be 16-byte aligned");
  MINK_REQUIRE(coords != (const void *)out_coords && feats != out_feats, "augment_scenes: not an in-place operation");
  const int nb = (int)cdiv(n, kBlock);
  int *block_counts = (int *)workspace, *block_offsets = block_counts + nb;
  uint32_t *bounds = (uint32_t *)((char *)workspace + align_up(2 * (int64_t)nb * 4, 128));
  RawCols cols;
  for (int c = 0; c < MINK_AUG_MAX_CHANNELS; ++c) cols.raw[c] = c < C ? raw_cols[c] : -1, cols.inv[c] = -1;
  for (int c = 0; c < C; ++c) {
    MINK_REQUIRE(cols.raw[c] < MINK_AUG_MAX_CHANNELS, "augment_scenes: raw column %d of feature column %d", cols.raw[c], c);
    if (cols.raw[c] >= 0) {
      MINK_REQUIRE(cols.inv[cols.raw[c]] < 0, "augment_scenes: raw column %d is selected twice", cols.raw[c]);
      cols.inv[cols.raw[c]] = c;
    }
  }
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  MINK_HIP(hipMemsetAsync(bounds, 0, (size_t)n_scenes * kBoundStride * 4, s));
  augment_count_kernel<<<nb, kBlock, 0, s>>>(coords, coords_are_int32 != 0, n, scene_offsets, n_scenes, params, streams, k0, k1, block_counts, bounds);
  MINK_CHECK_LAUNCH();
  augment_scan_kernel<<<1, kBlock, 0, s>>>(block_counts, nb, block_offsets, n_kept);
  MINK_CHECK_LAUNCH();
  augment_apply_kernel<<<nb, kBlock, 0, s>>>(coords, coords_are_int32 != 0, feats, ldf, C, n, scene_offsets, n_scenes, params, streams, k0, k1,
                                             block_offsets, bounds, cols, out_coords, out_feats, ldo);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

}  // extern "C"
