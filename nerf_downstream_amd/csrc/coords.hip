// Coordinate maps on gfx950: packed-key hash map, first-occurrence unique, stride map,
// kernel map (neighbour table) and the ME-format rulebook (wave64 ballot + prefix sums).
// HBM-bound integer work: one thread per row / per (row,offset), coalesced 16-B row
// loads, hash probes served from L2 / Infinity Cache (table = 12 B per slot, 2-4x rows).
#include <string.h>

#include <mutex>

#include "common.h"

namespace mink {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

constexpr int kBlock = 256;  // 4 waves

// Timing diagnostic (MINK_DIAG_DUP, bit mask): launch an idempotent map kernel a second time -- what the step pays for one
// more of it is what it pays for the first.  1 = 3x3x3 tables, 2 = block insert, 4 = leader pass 2, 8 = level insert,
// 16 = leader pass 1, 32 = block fill, 64 = 0xFF fills.
static int diag_dup() {
  static const int v = [] {
    const char *e = getenv("MINK_DIAG_DUP");
    return e ? atoi(e) : 0;
  }();
  return v;
}

// ------------------------------------------------------------------------------ keys
__device__ __forceinline__ uint64_t dev_table_capacity(int64_t n) {
  uint64_t cap = 64;
  while (cap < (uint64_t)(2 * n)) cap <<= 1;
  return cap;
}

// clear_keys / clear_vals (optional): the hash map these keys are about to be inserted into is emptied by the same launch
// (clear_cap slots, or the capacity for the device-resident row count) -- one launch per level less in the pyramid
template <int MODE>  // 0 = float field rows, 1 = int32 rows
__device__ __forceinline__ bool row_key(const void *__restrict__ coords, int64_t i, int out_ts, uint64_t &key) {
  int b, x, y, z;
  if (MODE == 0) {
    const float4 c = reinterpret_cast<const float4 *>(coords)[i];
    b = (int)floorf(c.x), x = (int)floorf(c.y), y = (int)floorf(c.z), z = (int)floorf(c.w);
  } else {
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    b = c.x, x = c.y, y = c.z, z = c.w;
  }
  if (out_ts > 1) x = floor_to(x, out_ts), y = floor_to(y, out_ts), z = floor_to(z, out_ts);
  return pack_key(b, x, y, z, key);
}

// Level 0 of a pyramid: rows whose keys are STRICTLY ascending (the order of a voxel grid's `links`, of np.nonzero) are
// their own unique rows, and no hash insert is needed to find that out -- every row compares its key with the row before it
// and the first wave that sees a pair out of order sets MINK_STATUS_NOT_ASCENDING (one coherent look at the word per
// offending wave, one atomic until it is seen set).  The unique kernels of level 0 read the word (asc_fast).
__device__ __forceinline__ bool asc_fast(const uint32_t *asc_status) {
  return asc_status && !(*asc_status & MINK_STATUS_NOT_ASCENDING);
}

template <int MODE>
__device__ __forceinline__ void make_keys_body(const void *__restrict__ coords, int64_t n_host, const int *__restrict__ n_dev,
                                               int out_ts, uint64_t *__restrict__ keys, uint32_t *status,
                                               unsigned long long *__restrict__ clear_keys, int *__restrict__ clear_vals,
                                               uint64_t clear_cap, bool check_order = false) {
  const int64_t n = n_dev ? (int64_t)*n_dev : n_host;  // device-resident row count (level chains)
  if (clear_keys) {
    const uint64_t cap = n_dev ? dev_table_capacity(n) : clear_cap;
    for (uint64_t s = (uint64_t)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (uint64_t)gridDim.x * kBlock) {
      clear_keys[s] = kEmptyKey;
      clear_vals[s] = 0x7F7F7F7F;
    }
  }
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint64_t key;
  if (!row_key<MODE>(coords, i, out_ts, key)) {
    atomicOr(status, MINK_STATUS_RANGE);
    key = 0;  // keep the pipeline well-defined; the host raises on the status word
  }
  keys[i] = key;
  if (check_order) {
    uint64_t before = 0;
    const bool bad = i > 0 && (!row_key<MODE>(coords, i - 1, out_ts, before) || before >= key);
    const unsigned long long m = __ballot(bad);
    if (bad && (threadIdx.x & 63) == __builtin_ctzll(m) &&
        !(__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & MINK_STATUS_NOT_ASCENDING))
      atomicOr(status, MINK_STATUS_NOT_ASCENDING);
  }
}
template <int MODE>
__global__ __launch_bounds__(kBlock) void make_keys_kernel(const void *__restrict__ coords, int64_t n_host,
                                                           const int *__restrict__ n_dev, int out_ts,
                                                           uint64_t *__restrict__ keys, uint32_t *status,
                                                           unsigned long long *__restrict__ clear_keys = nullptr,
                                                           int *__restrict__ clear_vals = nullptr, uint64_t clear_cap = 0,
                                                           bool check_order = false) {
  make_keys_body<MODE>(coords, n_host, n_dev, out_ts, keys, status, clear_keys, clear_vals, clear_cap, check_order);
}

// ---------------------------------------------------------------------------- unique
// Hash-map capacity for n keys (== mink_table_capacity): in a level chain the row count of a level is only known on
// the device, and so is the capacity of the next level's map -- sized for the rows it really receives, not for the
// field's row count (a 2 M-slot table per level for 173 k / 37 k / 8 k ... rows cost 125 MB of clearing per batch and
// scattered the few keys over 25 MB each).

// Runs of equal keys in ADJACENT lanes insert once.  Device-scope atomics are what a map build costs the convolutions it runs
// beside (a second launch of a hash insert adds 0.13-0.22 ms to a step, a second launch of the 3x3x3 tables nothing:
// DESIGN.md Appendix A), and rows arrive in scan order: the cells of a 4^3 block, and the children of a coarser cell, sit
// next to each other.  `head_mask` = ballot of the lanes that start a run; a lane's run starts at the highest head at or below it.
__device__ __forceinline__ uint64_t shfl_up1_u64(uint64_t v) {
  const unsigned lo = __shfl_up((unsigned)v, 1), hi = __shfl_up((unsigned)(v >> 32), 1);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int run_head_lane(unsigned long long head_mask, int lane) {
  return 63 - __builtin_clzll(head_mask & ((2ull << lane) - 1ull));
}

__global__ __launch_bounds__(kBlock) void insert_kernel(const uint64_t *__restrict__ keys, int64_t n_host,
                                                        const int *__restrict__ n_dev, unsigned long long *tkeys,
                                                        int *tvals, uint64_t mask, int *__restrict__ slot_of_row,
                                                        const uint32_t *__restrict__ asc_status) {
  if (asc_fast(asc_status)) return;  // (ascending level 0: nothing to look up, the map stays empty)
  const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
  if (n_dev) mask = dev_table_capacity(n) - 1;  // chained level: the map is sized for the rows that arrive
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if ((int64_t)blockIdx.x * kBlock >= n) return;  // (whole workgroups: every lane of a live wave reaches the shuffles)
  const bool live = i < n;
  const int lane = threadIdx.x & 63;
  const uint64_t key = live ? keys[i] : kEmptyKey;  // (no packed key is all ones: an idle lane never joins a run)
  const uint64_t before = shfl_up1_u64(key);
  const bool head = live && (lane == 0 || before != key);
  const unsigned long long head_mask = __ballot(head);
  int s32 = 0;
  if (head) {
    uint64_t s = mix64(key) & mask;
    for (;;) {
      const unsigned long long prev = atomicCAS(&tkeys[s], (unsigned long long)kEmptyKey, (unsigned long long)key);
      if (prev == kEmptyKey || prev == key) break;
      s = (s + 1) & mask;
    }
    atomicMin(&tvals[s], (int)i);  // first occurrence wins (the head is the lowest row of its run)
    s32 = (int)s;
  }
  s32 = __shfl(s32, live ? run_head_lane(head_mask, lane) : lane);
  if (live) slot_of_row[i] = s32;
}

// flag first occurrences, count them per block
__global__ __launch_bounds__(kBlock) void flag_kernel(const int *__restrict__ tvals, const int *__restrict__ slot_of_row,
                                                      int64_t n_host, const int *__restrict__ n_dev,
                                                      uint8_t *__restrict__ flags, int *__restrict__ block_counts,
                                                      const uint32_t *__restrict__ asc_status) {
  __shared__ int s_cnt[kBlock / 64];
  const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool fast = asc_fast(asc_status);
  bool first = false;
  if (i < n) {
    first = fast || tvals[slot_of_row[i]] == (int)i;
    flags[i] = first;
  }
  const unsigned long long m = __ballot(first);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// single-workgroup exclusive scan of int32 (n up to a few 1e5); total -> *total_out.
// 256 threads x 16 consecutive items per trip: wave-level shuffles scan the per-thread sums.  (One wave per SIMD on
// purpose: a 1024-thread workgroup needs four free wave slots on EVERY SIMD of one CU at once, and beside a kernel that
// fills the register files -- the stem convolution runs while the next batch's maps are built -- it was seen waiting
// 590 us for that to happen.)
constexpr int kScanBlock = 256;
__device__ __forceinline__ void scan_body(const int *__restrict__ in, int *__restrict__ out, int64_t n, int *total_out) {
  constexpr int IT = 16;
  __shared__ int s_wave[kScanBlock / 64];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t base = 0; base < n; base += kScanBlock * IT) {
    const int64_t i0 = base + (int64_t)threadIdx.x * IT;
    int v[IT], tsum = 0;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      v[j] = (i0 + j < n) ? in[i0 + j] : 0;
      tsum += v[j];
    }
    int incl = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kScanBlock / 64; ++w) {
      const int t = s_wave[w];
      if (w < wave) woff += t;
      total += t;
    }
    int run = s_carry + woff + incl - tsum;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      if (i0 + j < n) out[i0 + j] = run;
      run += v[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) s_carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = s_carry;
}
__global__ __launch_bounds__(kScanBlock) void scan_kernel(const int *__restrict__ in, int *__restrict__ out, int64_t n,
                                                          int *total_out) {
  scan_body(in, out, n, total_out);
}
struct ScanBatch {  // one workgroup per array
  const int *in[8];
  int *out[8], *total[8];
  int64_t n[8];
};
__global__ __launch_bounds__(kScanBlock) void scan_batch_kernel(ScanBatch b) {
  scan_body(b.in[blockIdx.x], b.out[blockIdx.x], b.n[blockIdx.x], b.total[blockIdx.x]);
}

__global__ __launch_bounds__(kBlock) void assign_kernel(const uint64_t *__restrict__ keys,
                                                        const uint8_t *__restrict__ flags,
                                                        const int *__restrict__ slot_of_row,
                                                        const int *__restrict__ block_offsets, int64_t n_host,
                                                        const int *__restrict__ n_dev, int *tvals,
                                                        int *__restrict__ out_coords, int *__restrict__ unique_index,
                                                        const uint32_t *__restrict__ asc_status) {
  __shared__ int s_cnt[kBlock / 64];
  const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool first = i < n && flags[i];
  const unsigned long long m = __ballot(first);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_cnt[wave] = __popcll(m);
  __syncthreads();
  if (first) {
    int uid = block_offsets[blockIdx.x] + wave_rank(m);
    for (int w = 0; w < wave; ++w) uid += s_cnt[w];
    if (unique_index) unique_index[uid] = (int)i;
    reinterpret_cast<int4 *>(out_coords)[uid] = unpack_key(keys[i]);
    if (!asc_fast(asc_status)) tvals[slot_of_row[i]] = uid;
  }
}

__global__ __launch_bounds__(kBlock) void inverse_kernel(const int *__restrict__ tvals,
                                                         const int *__restrict__ slot_of_row, int64_t n_host,
                                                         const int *__restrict__ n_dev, int *__restrict__ inverse,
                                                         const uint32_t *__restrict__ asc_status) {
  const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) inverse[i] = asc_fast(asc_status) ? (int)i : tvals[slot_of_row[i]];
}

// Level chains: the inverse map of level l and the keys of level l + 1 (made from level l's unique rows, with level l + 1's
// hash map emptied on the way) are independent row-parallel passes behind the same launch (assign of level l): one kernel.
// The last level's call carries the batch count instead (last_out).
__global__ __launch_bounds__(kBlock) void inverse_next_keys_kernel(const int *__restrict__ tvals, const int *__restrict__ slot_of_row,
                                                                   int64_t n_host, const int *__restrict__ n_dev,
                                                                   int *__restrict__ inverse, const void *__restrict__ next_coords,
                                                                   const int *__restrict__ next_n, int next_ts,
                                                                   uint64_t *__restrict__ keys, uint32_t *status,
                                                                   unsigned long long *__restrict__ next_tkeys,
                                                                   int *__restrict__ next_tvals, const int *__restrict__ coords0,
                                                                   const int *__restrict__ n0, int *last_out,
                                                                   const uint32_t *__restrict__ asc_status) {
  {
    const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) inverse[i] = asc_fast(asc_status) ? (int)i : tvals[slot_of_row[i]];
  }
  if (last_out && blockIdx.x == 0 && threadIdx.x == 0) {
    const int n = *n0;
    *last_out = n > 0 ? coords0[4 * (n - 1)] + 1 : 0;  // batch column is non-decreasing (checked by batch_offsets)
  }
  if (next_coords) make_keys_body<1>(next_coords, n_host, next_n, next_ts, keys, status, next_tkeys, next_tvals, 0);
}

__global__ void last_batch_kernel(const int *__restrict__ coords0, const int *__restrict__ n0, int *out) {
  const int n = *n0;
  *out = n > 0 ? coords0[4 * (n - 1)] + 1 : 0;  // batch column is non-decreasing (checked by batch_offsets)
}

// ------------------------------------------------------------------------ kernel map
struct Offsets {
  int d[27 * 3];
};

__global__ __launch_bounds__(kBlock) void kernel_map_kernel(const uint64_t *__restrict__ tkeys,
                                                            const int *__restrict__ tvals, uint64_t mask,
                                                            const int *__restrict__ out_coords, int64_t n_out, int K,
                                                            Offsets off, int *__restrict__ nbr, int *nbr_t) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n_out * K) return;
  const int64_t o = idx / K;
  const int k = (int)(idx - o * K);
  const int4 c = reinterpret_cast<const int4 *>(out_coords)[o];
  uint64_t key;
  int v = -1;
  if (pack_key(c.x, c.y + off.d[3 * k], c.z + off.d[3 * k + 1], c.w + off.d[3 * k + 2], key))
    v = table_find(tkeys, tvals, mask, key);
  nbr[idx] = v;
  if (nbr_t && v >= 0) nbr_t[(int64_t)v * K + k] = (int)o;
}

// ------------------------------------------------------------------------ block index
// Block key of a coordinate at tensor stride ts: cell = coordinate / ts (exact for map rows, floor for safety),
// block = cell >> 2 per axis (on the biased 16-bit value, so negative coordinates floor correctly), packed like a
// voxel key; `local` = position of the cell inside its 4x4x4 block (x fastest).
__device__ __forceinline__ bool block_key(int b, int x, int y, int z, int ts, uint64_t &key, int &local) {
  if (ts > 1) x = floor_to(x, ts) / ts, y = floor_to(y, ts) / ts, z = floor_to(z, ts) / ts;
  const unsigned ux = (unsigned)(x + 32768), uy = (unsigned)(y + 32768), uz = (unsigned)(z + 32768);
  const bool ok = ((unsigned)b <= 65534u) && ux <= 65535u && uy <= 65535u && uz <= 65535u;
  key = ((uint64_t)(unsigned)b << 48) | ((uint64_t)(ux >> 2) << 32) | ((uint64_t)(uy >> 2) << 16) | (uint64_t)(uz >> 2);
  local = (int)((ux & 3u) | ((uy & 3u) << 2) | ((uz & 3u) << 4));
  return ok;
}

// 32-bit multiplicative hash of a block key: the look-up kernels are instruction-bound (splitmix64 alone is ~30 VALU
// instructions in 32-bit hardware), and a block table holds few, well-spread keys
__device__ __forceinline__ uint64_t blk_hash(uint64_t key) {
  unsigned h = (unsigned)key * 0x9E3779B1u ^ (unsigned)(key >> 32) * 0x85EBCA77u;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 13;
  return h;
}

// (A table given too few slots for the blocks -- blk_cap is the caller's promise -- must not hang the card: probing stops after
//  one trip round the table, the row is left out (slot -1: the later passes skip it) and *overflow is cleared from its 0xFF fill.)
__device__ __forceinline__ void blk_insert_body(const int *__restrict__ coords, int64_t n, int ts, unsigned long long *table,
                                                uint64_t mask, int *__restrict__ slot_of_row, int *overflow, int *sticky) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < n;  // (callers drop whole idle workgroups; every lane of a live wave reaches the shuffles)
  const int lane = threadIdx.x & 63;
  const int4 c = reinterpret_cast<const int4 *>(coords)[live ? i : n - 1];
  uint64_t key;
  int local;
  block_key(c.x, c.y, c.z, c.w, ts, key, local);  // rows of a map are always in range
  if (!live) key = kEmptyKey;
  // adjacent lanes in the same block (scan order: the z run of a block, ~4 rows) insert once, with the OR of their cells
  const uint64_t before = shfl_up1_u64(key);
  const bool head = live && (lane == 0 || before != key);
  const unsigned long long head_mask = __ballot(head);
  int s32 = 0;
  {
    // run length of a head = distance to the next head (or to the end of the wave); idle lanes are heads of nothing
    const unsigned long long later = (head_mask | ~__ballot(live)) >> lane >> 1;
    const int len = head ? (later ? __builtin_ctzll(later) + 1 : 64 - lane) : 0;
    unsigned long long bits = 1ull << local;
    int maxlen = len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, d));
    for (int j = 1; j < maxlen; ++j) {
      const int l2 = __shfl(local, min(lane + j, 63));
      if (j < len) bits |= 1ull << l2;
    }
    if (head) {
      uint64_t s = blk_hash(key) & mask;
      bool placed = false;
      for (uint64_t probes = 0; probes <= mask; ++probes) {
        const unsigned long long prev = atomicCAS(&table[2 * s], (unsigned long long)kEmptyKey, (unsigned long long)key);
        if (prev == kEmptyKey || prev == key) {
          placed = true;
          break;
        }
        s = (s + 1) & mask;
      }
      if (placed) {
        atomicAnd(&table[2 * s + 1], ~bits);  // the mask is kept inverted: the 0xFF fill of the table means "empty"
        s32 = (int)s;
      } else {
        *overflow = 0;
        if (sticky) *sticky = 1;  // (mink_set_overflow_sink: a host-visible word nobody refills -- the caller's next batch sees it)
        s32 = -1;
      }
    }
  }
  s32 = __shfl(s32, live ? run_head_lane(head_mask, lane) : lane);
  if (live) slot_of_row[i] = s32;
}

// One row per block (the one in its lowest occupied cell, the "leader") owns the block's run of `rowids`.  Runs are
// laid out in row order of the leaders: pass 1 counts the cells led by each workgroup, pass 2 sums the counts before
// its own workgroup and hands every leader its start.  No atomics (a single counter word takes ~90 adds per
// microsecond: 70 k leaders on it serialise for longer than the rest of the build), deterministic layout.
template <bool ASSIGN>
__device__ __forceinline__ void blk_leader_body(const int *__restrict__ coords, int64_t n, int ts,
                                                const unsigned long long *__restrict__ table,
                                                const int *__restrict__ slot_of_row, int *__restrict__ base,
                                                int *__restrict__ wg_counts) {
  __shared__ int s_wave[kBlock / 64];
  const int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = i0 < n;
  const int64_t i = live ? i0 : n - 1;  // (idle lanes shadow the last row and never lead: whole waves reach the shuffles)
  const int4 c = reinterpret_cast<const int4 *>(coords)[i];
  uint64_t key;
  int local;
  block_key(c.x, c.y, c.z, c.w, ts, key, local);
  const int s = slot_of_row[i];
  if (!live || s < 0) local = -1;
  const unsigned long long m = s < 0 ? 0ull : ~table[2 * (int64_t)s + 1];
  const bool leader = m && local == __builtin_ctzll(m);
  const int cnt = leader ? __popcll(m) : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (!ASSIGN) {
    if (threadIdx.x == 0) wg_counts[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return;
  }
  // this workgroup's offset = sum of the counts of the workgroups before it (a few thousand ints from L2: cheaper than
  // a scan launch between the two passes, and nothing here can be held up by a single-workgroup kernel)
  __shared__ int s_part[kBlock / 64];
  int part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += kBlock) part += wg_counts[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
  if (lane == 0) s_part[wave] = part;
  __syncthreads();
  int before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  for (int w = 0; w < wave; ++w) before += s_wave[w];
  if (leader) base[s] = before + incl - cnt;
}

__device__ __forceinline__ void blk_fill_body(const int *__restrict__ coords, int64_t n, int ts,
                                              const unsigned long long *__restrict__ table, const int *__restrict__ slot_of_row,
                                              const int *__restrict__ base, int *__restrict__ rowids) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4 *>(coords)[i];
  uint64_t key;
  int local;
  block_key(c.x, c.y, c.z, c.w, ts, key, local);
  const int s = slot_of_row[i];
  if (s < 0) return;  // (a row that found no slot: blk_insert_body)
  const unsigned long long m = ~table[2 * (int64_t)s + 1];
  rowids[base[s] + __popcll(m & ((1ull << local) - 1ull))] = (int)i;
}

// The block indices of ALL input maps of a batch's plan are built by four launches, not four per map (blockIdx.y = map):
// the builds are independent of each other, and a prepared batch has six of them.
static int *g_overflow_sink = nullptr;  // mink_set_overflow_sink
constexpr int kMaxBatch = 8;
struct BlkBuild {
  const int *coords;
  int64_t n;
  unsigned long long *table;
  uint64_t mask;
  int *slot, *base, *rowids, *overflow;
  int ts;
};
struct BlkBuildBatch {
  BlkBuild e[kMaxBatch];
  int *sticky;  // process-wide overflow sink (device-visible host word), or nullptr
};
__global__ __launch_bounds__(kBlock) void blk_insert_kernel(BlkBuildBatch b) {
  const BlkBuild &e = b.e[blockIdx.y];
  if ((int64_t)blockIdx.x * kBlock >= e.n) return;
  blk_insert_body(e.coords, e.n, e.ts, e.table, e.mask, e.slot, e.overflow, b.sticky);
}
template <bool ASSIGN>
__global__ __launch_bounds__(kBlock) void blk_leader_kernel(BlkBuildBatch b) {
  const BlkBuild &e = b.e[blockIdx.y];
  if ((int64_t)blockIdx.x * kBlock >= e.n) return;  // (whole workgroups: the barriers below are reached by all or none)
  blk_leader_body<ASSIGN>(e.coords, e.n, e.ts, e.table, e.slot, e.base, e.rowids);  // (workgroup counts live in the head of rowids)
}
__global__ __launch_bounds__(kBlock) void blk_fill_kernel(BlkBuildBatch b) {
  const BlkBuild &e = b.e[blockIdx.y];
  if ((int64_t)blockIdx.x * kBlock >= e.n) return;
  blk_fill_body(e.coords, e.n, e.ts, e.table, e.slot, e.base, e.rowids);
}
// 0xFF fill of up to sixteen int32 buffers in one launch (blockIdx.y = buffer): 16-byte stores over the aligned body, single
// words before and behind it (a hipMemsetAsync per buffer is one launch each, two when the size is not a multiple of 16)
struct FillBatch {
  unsigned *ptr[16];
  int64_t nwords[16];
};
__global__ __launch_bounds__(kBlock) void fill_ff_kernel(FillBatch f) {
  unsigned *p = f.ptr[blockIdx.y];
  const int64_t n = f.nwords[blockIdx.y];
  const int64_t head = min(n, (int64_t)(((16u - (unsigned)((uintptr_t)p & 15u)) & 15u) >> 2));
  const int64_t n16 = (n - head) >> 2, tail0 = head + 4 * n16;
  uint4 *body = reinterpret_cast<uint4 *>(p + head);
  const uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (int64_t)gridDim.x * kBlock) body[i] = v;
  if (blockIdx.x == 0) {
    if ((int64_t)threadIdx.x < head) p[threadIdx.x] = 0xFFFFFFFFu;
    if (tail0 + (int64_t)threadIdx.x < n) p[tail0 + threadIdx.x] = 0xFFFFFFFFu;
  }
}

// Same result as kernel_map_kernel, through the block index.  Thread per (output row, offset): the 27 look-ups of a
// row sit in adjacent lanes and hit the same one to eight block entries.
template <class OFF>
__device__ __forceinline__ void kernel_map_blk_body(const unsigned long long *__restrict__ table, const int *__restrict__ base,
                                                    const int *__restrict__ rowids, uint64_t mask, int ts,
                                                    const int *__restrict__ out_coords, int64_t n_out, int K, const OFF &off,
                                                    int *__restrict__ nbr, int *nbr_t) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n_out * K) return;
  const int64_t o = idx / K;
  const int k = (int)(idx - o * K);
  const int4 c = reinterpret_cast<const int4 *>(out_coords)[o];
  uint64_t key;
  int local, v = -1;
  if (block_key(c.x, c.y + off.d[3 * k], c.z + off.d[3 * k + 1], c.w + off.d[3 * k + 2], ts, key, local)) {
    uint64_t s = blk_hash(key) & mask;
    for (uint64_t probes = 0; probes <= mask; ++probes) {  // (bounded: a table without a free slot must not hang the look-up)
      const ulonglong2 e = reinterpret_cast<const ulonglong2 *>(table)[s];
      if (e.x == key) {
        const unsigned long long m = ~e.y;
        if ((m >> local) & 1ull) v = rowids[base[s] + __popcll(m & ((1ull << local) - 1ull))];
        break;
      }
      if (e.x == kEmptyKey) break;
      s = (s + 1) & mask;
    }
  }
  nbr[idx] = v;
  if (nbr_t && v >= 0) nbr_t[(int64_t)v * K + k] = (int)o;
}

// The 3^3 neighbourhood of one output row per THREAD (unit offsets in input cells: every 3x3x3 convolution of the
// network, strided or not).  The 27 cells lie in at most 2x2x2 blocks -- A = block of (cell - 1), B = block of
// (cell + 1) per axis -- so the thread probes eight block entries and answers all 27 look-ups from their masks with
// bit tests and popcounts: ~20 instructions per neighbour instead of ~300 for one probe chain each.  Results are
// staged in LDS and written as one contiguous run per workgroup.
struct __attribute__((packed, aligned(4))) Row27 {
  int v[27];
};
template <bool HAS_T>
__device__ __forceinline__ void kernel_map_blk27_body(const unsigned long long *__restrict__ table, const int *__restrict__ base,
                                                      const int *__restrict__ rowids, uint64_t mask, int ts,
                                                      const int *__restrict__ out_coords, int64_t n_out, int *__restrict__ nbr,
                                                      int *nbr_t) {
  const int64_t o0 = (int64_t)blockIdx.x * kBlock;
  const int64_t o = o0 + threadIdx.x;
  const bool live = o < n_out;
  const int4 c = reinterpret_cast<const int4 *>(out_coords)[live ? o : n_out - 1];
  int cx = c.y, cy = c.z, cz = c.w;
  if (ts > 1) cx = floor_to(cx, ts) / ts, cy = floor_to(cy, ts) / ts, cz = floor_to(cz, ts) / ts;
  const int u[3] = {cx + 32768, cy + 32768, cz + 32768};  // 0..65535 for rows of a map; neighbours may step outside
  const bool b_ok = (unsigned)c.x <= 65534u;
  // block entries: index = ix + 2 iy + 4 iz, i = 0 -> block of (u - 1), 1 -> block of (u + 1)
  unsigned long long M[8];
  int B[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int nx = u[0] + ((q & 1) ? 1 : -1), ny = u[1] + ((q & 2) ? 1 : -1), nz = u[2] + ((q & 4) ? 1 : -1);
    M[q] = 0ull, B[q] = 0;
    if (b_ok && (unsigned)nx <= 65535u && (unsigned)ny <= 65535u && (unsigned)nz <= 65535u) {
      const uint64_t key = ((uint64_t)(unsigned)c.x << 48) | ((uint64_t)((unsigned)nx >> 2) << 32) |
                           ((uint64_t)((unsigned)ny >> 2) << 16) | (uint64_t)((unsigned)nz >> 2);
      uint64_t s = blk_hash(key) & mask;
      for (uint64_t probes = 0; probes <= mask; ++probes) {
        const ulonglong2 e = reinterpret_cast<const ulonglong2 *>(table)[s];
        if (e.x == key) {
          M[q] = ~e.y, B[q] = base[s];
          break;
        }
        if (e.x == kEmptyKey) break;
        s = (s + 1) & mask;
      }
    }
  }
  // the centre cell of an axis shares block A unless it sits on the low face of its block (then it is in B)
  const bool cB[3] = {(u[0] & 3) == 0, (u[1] & 3) == 0, (u[2] & 3) == 0};
  int out[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int d[3] = {k % 3 - 1, (k / 3) % 3 - 1, k / 9 - 1};
    // select the block entry: per axis bit = 1 (B) for d = +1, 0 (A) for d = -1, cB for d = 0
    unsigned long long m;
    int bs;
    {
      unsigned long long mx[4];
      int bx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // x axis resolved: entries over (iy, iz)
        const bool sel = d[0] > 0 || (d[0] == 0 && cB[0]);
        mx[j] = sel ? M[2 * j + 1] : M[2 * j];
        bx[j] = sel ? B[2 * j + 1] : B[2 * j];
      }
      const bool sy = d[1] > 0 || (d[1] == 0 && cB[1]);
      const unsigned long long my0 = sy ? mx[1] : mx[0], my1 = sy ? mx[3] : mx[2];
      const int by0 = sy ? bx[1] : bx[0], by1 = sy ? bx[3] : bx[2];
      const bool sz = d[2] > 0 || (d[2] == 0 && cB[2]);
      m = sz ? my1 : my0;
      bs = sz ? by1 : by0;
    }
    const int local = ((u[0] + d[0]) & 3) | (((u[1] + d[1]) & 3) << 2) | (((u[2] + d[2]) & 3) << 4);
    int v = -1;
    if ((m >> local) & 1ull) v = rowids[bs + __popcll(m & ((1ull << local) - 1ull))];
    out[k] = v;
    if (HAS_T && live && v >= 0) nbr_t[(int64_t)v * 27 + k] = (int)o;
  }
  // the row's 27 entries (108 contiguous bytes, 4-byte aligned) leave as six 16-byte stores and one of 12: no LDS staging --
  // the 27.6 KB per workgroup it took kept up to 138 KB of a CU's LDS away from the convolution kernels this one runs beside
  if (live) {
    Row27 *dst = reinterpret_cast<Row27 *>(nbr + o * 27);
    Row27 r;
#pragma unroll
    for (int k = 0; k < 27; ++k) r.v[k] = out[k];
    *dst = r;
  }
}

__global__ __launch_bounds__(kBlock) void kernel_map_blk_kernel(const unsigned long long *__restrict__ table,
                                                                const int *__restrict__ base,
                                                                const int *__restrict__ rowids, uint64_t mask, int ts,
                                                                const int *__restrict__ out_coords, int64_t n_out, int K,
                                                                Offsets off, int *__restrict__ nbr, int *nbr_t) {
  kernel_map_blk_body(table, base, rowids, mask, ts, out_coords, n_out, K, off, nbr, nbr_t);
}
struct SmallOffsets {  // up to eight offsets (pooling 2^3, 1x1x1)
  int d[24];
};
struct KMapS {
  const unsigned long long *table;
  const int *base, *rowids, *out_coords;
  uint64_t mask;
  int64_t n_out;
  int *nbr, *nbr_t;
  int ts, K;
  SmallOffsets off;
};
struct KMapSBatch {
  KMapS e[kMaxBatch];
};
__global__ __launch_bounds__(kBlock) void kernel_map_blk_small_kernel(KMapSBatch b) {  // blockIdx.y = table
  const KMapS &e = b.e[blockIdx.y];
  if ((int64_t)blockIdx.x * kBlock >= e.n_out * e.K) return;
  kernel_map_blk_body(e.table, e.base, e.rowids, e.mask, e.ts, e.out_coords, e.n_out, e.K, e.off, e.nbr, e.nbr_t);
}
struct KMap27 {
  const unsigned long long *table;
  const int *base, *rowids, *out_coords;
  uint64_t mask;
  int64_t n_out;
  int *nbr, *nbr_t;
  int ts;
};
struct KMap27Batch {
  KMap27 e[kMaxBatch];
};
template <bool HAS_T>
__global__ __launch_bounds__(kBlock) void kernel_map_blk27_kernel(KMap27Batch b) {  // blockIdx.y = table
  const KMap27 &e = b.e[blockIdx.y];
  if ((int64_t)blockIdx.x * kBlock >= e.n_out) return;
  kernel_map_blk27_body<HAS_T>(e.table, e.base, e.rowids, e.mask, e.ts, e.out_coords, e.n_out, e.nbr, e.nbr_t);
}

// -------------------------------------------------------------------------- rulebook
// Pass 1: pairs per (offset, 256-row chunk); pass 2 (after the scan): fill, ordered by row.
template <bool FILL>
__global__ __launch_bounds__(kBlock) void rulebook_kernel(const int *__restrict__ nbr, int64_t n_out, int K,
                                                          int64_t nchunk, int *__restrict__ chunk_counts,
                                                          const int *__restrict__ chunk_offsets,
                                                          int *__restrict__ pairs_in, int *__restrict__ pairs_out) {
  __shared__ int s_cnt[kBlock / 64][27];
  const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = 0; k < K; ++k) {
    const int v = o < n_out ? nbr[o * K + k] : -1;
    const unsigned long long m = __ballot(v >= 0);
    if (lane == 0) s_cnt[wave][k] = __popcll(m);
    if (FILL) {
      __syncthreads();
      if (v >= 0) {
        int pos = chunk_offsets[(int64_t)k * nchunk + blockIdx.x] + wave_rank(m);
        for (int w = 0; w < wave; ++w) pos += s_cnt[w][k];
        pairs_in[pos] = v;
        pairs_out[pos] = (int)o;
      }
    }
  }
  if (!FILL) {
    __syncthreads();
    if (threadIdx.x < K)
      chunk_counts[(int64_t)threadIdx.x * nchunk + blockIdx.x] =
          s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
  }
}

__global__ void rulebook_counts_kernel(const int *__restrict__ chunk_offsets, int64_t nchunk, int K,
                                       const int *__restrict__ total, int *__restrict__ counts) {
  const int k = threadIdx.x;
  if (k < K) counts[k] = chunk_offsets[(int64_t)k * nchunk];
  if (k == K) counts[K] = *total;
}


// ------------------------------------------------------------------- parity classes
// Rows of a stride-ts map grouped by the parity of (c / ts) per axis (8 classes).  For the
// dgrad of a stride-2 convolution every row of one class can only be reached through the same
// 2^p kernel offsets, so class-pure 128-row tiles skip the other offsets entirely.
// Same count -> scan -> fill scheme as the rulebook; each class segment starts at a multiple of
// `pad` rows in the output permutation (pre-filled with -1).
__device__ __forceinline__ int parity_class(int4 c, int ts) {
  return ((c.y / ts) & 1) | (((c.z / ts) & 1) << 1) | (((c.w / ts) & 1) << 2);
}

template <bool FILL>
__device__ __forceinline__ void class_partition_body(const int *__restrict__ coords, int64_t n, int ts, int64_t nchunk, int pad,
                                                     int *__restrict__ chunk_counts, const int *__restrict__ chunk_offsets,
                                                     const int *__restrict__ total, int *__restrict__ perm) {
  __shared__ int s_cnt[kBlock / 64][8];
  __shared__ int s_shift[8];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cls = i < n ? parity_class(reinterpret_cast<const int4 *>(coords)[i], ts) : -1;
  if (FILL && threadIdx.x == 0) {  // padding inserted in front of each class segment
    int shift = 0;
    for (int c = 0; c < 8; ++c) {
      s_shift[c] = shift;
      const int beg = chunk_offsets[(int64_t)c * nchunk];
      const int end = c < 7 ? chunk_offsets[(int64_t)(c + 1) * nchunk] : *total;
      const int cnt = end - beg;
      shift += (cnt + pad - 1) / pad * pad - cnt;
    }
  }
  int my_rank = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if (lane == 0) s_cnt[wave][c] = __popcll(m);
    if (cls == c) my_rank = wave_rank(m);
  }
  __syncthreads();
  if (FILL) {
    if (cls >= 0) {
      int pos = chunk_offsets[(int64_t)cls * nchunk + blockIdx.x] + s_shift[cls] + my_rank;
      for (int w = 0; w < wave; ++w) pos += s_cnt[w][cls];
      perm[pos] = (int)i;
    }
  } else if (threadIdx.x < 8) {
    chunk_counts[(int64_t)threadIdx.x * nchunk + blockIdx.x] =
        s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
  }
}

// --------------------------------------------------------------------- batch offsets

template <bool FILL>
__global__ __launch_bounds__(kBlock) void class_partition_kernel(const int *__restrict__ coords, int64_t n, int ts,
                                                                 int64_t nchunk, int pad,
                                                                 int *__restrict__ chunk_counts,
                                                                 const int *__restrict__ chunk_offsets,
                                                                 const int *__restrict__ total, int *__restrict__ perm) {
  class_partition_body<FILL>(coords, n, ts, nchunk, pad, chunk_counts, chunk_offsets, total, perm);
}
struct ClassPart {
  const int *coords;
  int64_t n, nchunk;
  int *chunk_counts, *chunk_offsets, *total, *perm;
  int ts, pad;
};
struct ClassPartBatch {
  ClassPart e[8];
};
template <bool FILL>
__global__ __launch_bounds__(kBlock) void class_partition_batch_kernel(ClassPartBatch b) {  // blockIdx.y = map
  const ClassPart &e = b.e[blockIdx.y];
  if ((int64_t)blockIdx.x >= e.nchunk) return;
  class_partition_body<FILL>(e.coords, e.n, e.ts, e.nchunk, e.pad, e.chunk_counts, e.chunk_offsets, e.total, e.perm);
}

__global__ __launch_bounds__(kBlock) void batch_offsets_kernel(const int *__restrict__ coords, int64_t n, int B,
                                                               int *__restrict__ batch_offsets, uint32_t *status) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n && i > 0 && coords[4 * i] < coords[4 * (i - 1)]) atomicOr(status, MINK_STATUS_UNSORTED);
  if (i <= B) {  // lower_bound of batch index i
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (coords[4 * mid] < (int)i) lo = mid + 1;
      else hi = mid;
    }
    batch_offsets[i] = (int)lo;
  }
}

}  // namespace mink

using namespace mink;

// ------------------------------------------------------------------ PeRFception data.npz front-end
// One thread per (voxel, group of four feature columns).  links -> (batch, x, y, z); features =
// the selected columns of [density | sh_q * sh_scale + sh_min (27) | ones] in the caller's order.
// "xyzs" feature of the reference loader (co3d.py:209-214): every point minus the mean of ITS OWN three coordinates (the
// reference reduces over dim=1, not over the points -- kept as it is), divided by the largest norm of the scene.  Every
// product, sum and the division are rounded separately, in the order ((x + y) + z) / 3 and sqrt((dx^2 + dy^2) + dz^2).
__device__ __forceinline__ float xyz_centred(int x, int y, int z, float (&d)[3]) {
  const float fx = (float)x, fy = (float)y, fz = (float)z;
  float s = fx + fy;
  asm volatile("" : "+v"(s));
  s = s + fz;
  asm volatile("" : "+v"(s));
  const float m = __fdiv_rn(s, 3.f);
  d[0] = fx - m, d[1] = fy - m, d[2] = fz - m;
  return m;
}

// per-scene max of the centred norm, as float bits (non-negative floats order like unsigned integers: atomicMax is exact
// and order-independent)
__global__ __launch_bounds__(kBlock) void decode_xyz_maxnorm_kernel(const int *__restrict__ links,
                                                                    const int *__restrict__ scene_offsets, int n_scenes,
                                                                    int64_t n, int ry, int rz,
                                                                    unsigned *__restrict__ scene_maxnorm) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = n_scenes;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)scene_offsets[mid] <= i) lo = mid;
    else hi = mid;
  }
  const int l = links[i];
  const int yz = ry * rz;
  const int x = l / yz, rem = l - x * yz;
  const int y = rem / rz, z = rem - y * rz;
  float d[3];
  xyz_centred(x, y, z, d);
  float a = d[0] * d[0], b2 = d[1] * d[1], c2 = d[2] * d[2];
  asm volatile("" : "+v"(a), "+v"(b2), "+v"(c2));
  float s = a + b2;
  asm volatile("" : "+v"(s));
  s = s + c2;
  asm volatile("" : "+v"(s));
  atomicMax(scene_maxnorm + lo, __float_as_uint(__fsqrt_rn(s)));
}

// The de-quantisation is a separate multiply and add (no fma): bit-identical to numpy's
// `sh.astype(float32) * sh_scale + sh_min` of the reference loader (co3d.py:160-166).
__global__ __launch_bounds__(kBlock) void decode_plenoxel_kernel(
    const int *__restrict__ links, const float *__restrict__ density, const unsigned char *__restrict__ sh_q,
    const int *__restrict__ scene_offsets, int n_scenes, const float *__restrict__ sh_scale,
    const float *__restrict__ sh_min, int64_t n, int ry, int rz, int col_density, int col_sh, int col_ones, int col_xyzs,
    const float *__restrict__ scene_maxnorm, int C, int *__restrict__ coords, float *__restrict__ feats, int ldf, bool vec) {
  const int groups = (C + 3) >> 2;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n * groups) return;
  const int64_t i = t / groups;
  const int gq = (int)(t - i * groups);
  int lo = 0, hi = n_scenes;  // scene b with scene_offsets[b] <= i < scene_offsets[b+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)scene_offsets[mid] <= i) lo = mid;
    else hi = mid;
  }
  const int b = lo;
  const int l = links[i];
  const int yz = ry * rz;
  const int x = l / yz, rem = l - x * yz;
  const int y = rem / rz, z = rem - y * rz;
  if (gq == 0) *reinterpret_cast<int4 *>(coords + 4 * i) = make_int4(b, x, y, z);
  float ctr[3] = {0.f, 0.f, 0.f};
  if (col_xyzs >= 0) {
    const float m = xyz_centred(x, y, z, ctr);
    (void)m;
  }
  float out[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * gq + e;
    if (c >= C) break;
    float v;
    if (col_sh >= 0 && c >= col_sh && c < col_sh + 27) {
      const int j = c - col_sh;
      float prod = (float)sh_q[i * 27 + j] * sh_scale[b * 27 + j];
      asm volatile("" : "+v"(prod));  // keep the product rounded: hipcc contracts a*b+c (and __fmul_rn is a plain multiply)
      v = prod + sh_min[b * 27 + j];
    } else if (c == col_density) {
      v = density[i];
    } else if (col_xyzs >= 0 && c >= col_xyzs && c < col_xyzs + 3) {
      v = __fdiv_rn(ctr[c - col_xyzs], scene_maxnorm[b]);
    } else {  // c == col_ones
      v = 1.f;
    }
    out[e] = v;
  }
  float *dst = feats + i * ldf + 4 * gq;
  if (vec && 4 * gq + 4 <= C) {
    *reinterpret_cast<float4 *>(dst) = make_float4(out[0], out[1], out[2], out[3]);  // one 16-byte store per thread
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * gq + e < C) dst[e] = out[e];
  }
}

extern "C" {

const char *mink_last_error(void) { return g_err; }
int mink_abi_version(void) { return 4; }  // 2: every scratch buffer is passed with its size; 3: MinkStem.xb (bf16 storage); 4: mink_net_* (the whole trunk per call)

int64_t mink_table_capacity(int64_t n) {
  int64_t cap = 64;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

static int64_t unique_nblocks(int64_t n) { return cdiv(n > 0 ? n : 1, kBlock); }

// enqueue the five kernels of insert_and_map; n_dev (optional) holds the row count on the device
static int unique_launch(const uint64_t *keys, int64_t n, const int *n_dev, uint64_t *table_keys, int32_t *table_vals,
                         int64_t cap, int32_t *out_coords, int32_t *unique_index, int32_t *inverse, int32_t *n_unique,
                         void *workspace, hipStream_t st, bool skip_inverse = false, const uint32_t *asc_status = nullptr) {
  const int64_t nb = unique_nblocks(n);
  char *ws = (char *)workspace;
  int *slot_of_row = (int *)ws;
  ws += align_up(4 * n, 256);
  uint8_t *flags = (uint8_t *)ws;
  ws += align_up(n, 256);
  int *block_counts = (int *)ws;
  ws += align_up(4 * nb, 256);
  int *block_offsets = (int *)ws;
  const dim3 grid((unsigned)nb);
  for (int rep = 0; rep < 1 + ((diag_dup() & 8) != 0); ++rep)
    insert_kernel<<<grid, kBlock, 0, st>>>(keys, n, n_dev, (unsigned long long *)table_keys, table_vals,
                                           (uint64_t)cap - 1, slot_of_row, asc_status);
  MINK_CHECK_LAUNCH();
  flag_kernel<<<grid, kBlock, 0, st>>>(table_vals, slot_of_row, n, n_dev, flags, block_counts, asc_status);
  MINK_CHECK_LAUNCH();
  scan_kernel<<<1, kScanBlock, 0, st>>>(block_counts, block_offsets, nb, n_unique);
  MINK_CHECK_LAUNCH();
  assign_kernel<<<grid, kBlock, 0, st>>>(keys, flags, slot_of_row, block_offsets, n, n_dev, table_vals, out_coords,
                                         unique_index, asc_status);
  MINK_CHECK_LAUNCH();
  if (skip_inverse) return MINK_OK;  // (the caller fuses it with the next level's first pass)
  inverse_kernel<<<grid, kBlock, 0, st>>>(table_vals, slot_of_row, n, n_dev, inverse, asc_status);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int64_t mink_unique_workspace_bytes(int64_t n) {
  const int64_t nb = unique_nblocks(n);
  return align_up(4 * n, 256) + align_up(n, 256) + 2 * align_up(4 * nb, 256) + 256;
}

int mink_coords_make_keys(const void *coords, int mode, int64_t n, int32_t out_ts, uint64_t *keys, uint32_t *status,
                          void *stream) {
  MINK_REQUIRE(n >= 0 && out_ts >= 1 && (mode == 0 || mode == 1), "make_keys: bad arguments (n=%lld ts=%d mode=%d)",
               (long long)n, out_ts, mode);
  if (n == 0) return MINK_OK;
  MINK_REQUIRE(coords && keys && status, "make_keys: NULL pointer");
  MINK_REQUIRE(((uintptr_t)coords & 15) == 0, "make_keys: coords must be 16-byte aligned rows of 4");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(n, kBlock));
  if (mode == 0)
    make_keys_kernel<0><<<grid, kBlock, 0, st>>>(coords, n, nullptr, out_ts, keys, status);
  else
    make_keys_kernel<1><<<grid, kBlock, 0, st>>>(coords, n, nullptr, out_ts, keys, status);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_coords_unique(const uint64_t *keys, int64_t n, uint64_t *table_keys, int32_t *table_vals, int64_t cap,
                       int32_t *out_coords, int32_t *unique_index, int32_t *inverse, int32_t *n_unique,
                       void *workspace, int64_t workspace_bytes, void *stream) {
  MINK_REQUIRE(n >= 0 && n < (1ll << 31) - 1, "unique: n out of range");
  MINK_REQUIRE(cap >= mink_table_capacity(n) && (cap & (cap - 1)) == 0, "unique: table capacity %lld too small for n=%lld",
               (long long)cap, (long long)n);
  MINK_REQUIRE(table_keys && table_vals && n_unique, "unique: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  MINK_HIP(hipMemsetAsync(table_keys, 0xFF, cap * sizeof(uint64_t), st));
  MINK_HIP(hipMemsetAsync(table_vals, 0x7F, cap * sizeof(int32_t), st));
  if (n == 0) {
    MINK_HIP(hipMemsetAsync(n_unique, 0, sizeof(int32_t), st));
    return MINK_OK;
  }
  MINK_REQUIRE(keys && out_coords && unique_index && inverse && workspace, "unique: NULL pointer");
  MINK_REQUIRE(workspace_bytes >= mink_unique_workspace_bytes(n), "unique: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
               (long long)(mink_unique_workspace_bytes(n)));
  MINK_REQUIRE(((uintptr_t)out_coords & 15) == 0 && ((uintptr_t)workspace & 255) == 0, "unique: misaligned buffer");
  int rc = unique_launch(keys, n, nullptr, table_keys, table_vals, cap, out_coords, unique_index, inverse, n_unique,
                         workspace, st);
  return rc;
}


/* ---- whole coordinate pyramid in one call ------------------------------------------------- */
int64_t mink_levels_workspace_bytes(int64_t n) { return align_up(8 * (n > 0 ? n : 1), 256) + mink_unique_workspace_bytes(n); }

int mink_coords_build_levels(const void *coords, int mode, int64_t n, int32_t nlev, const int32_t *out_ts_host,
                             uint64_t *const *table_keys, int32_t *const *table_vals, int64_t cap,
                             int32_t *const *out_coords, int32_t *const *index_a, int32_t *const *index_b,
                             int32_t *meta, void *workspace, int64_t workspace_bytes, void *stream) {
  MINK_REQUIRE(n >= 1 && n < (1ll << 31) - 1 && nlev >= 1 && nlev <= 16, "build_levels: bad sizes");
  MINK_REQUIRE(workspace_bytes >= mink_levels_workspace_bytes(n), "build_levels: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
               (long long)(mink_levels_workspace_bytes(n)));
  MINK_REQUIRE(coords && out_ts_host && table_keys && table_vals && out_coords && index_a && index_b && meta && workspace,
               "build_levels: NULL pointer");
  MINK_REQUIRE(cap >= mink_table_capacity(n) && (cap & (cap - 1)) == 0, "build_levels: table capacity too small");
  MINK_REQUIRE(((uintptr_t)coords & 15) == 0 && ((uintptr_t)workspace & 255) == 0, "build_levels: misaligned buffer");
  hipStream_t st = (hipStream_t)stream;
  uint64_t *keys = (uint64_t *)workspace;
  void *uws = (char *)workspace + align_up(8 * n, 256);
  uint32_t *status = (uint32_t *)(meta + nlev);
  MINK_HIP(hipMemsetAsync(meta, 0, sizeof(int32_t) * (nlev + 2), st));
  const dim3 grid((unsigned)cdiv(n, kBlock));
  const int *slot_of_row = (const int *)uws;  // (first region of the unique workspace: unique_launch)
  for (int l = 0; l < nlev; ++l) {
    MINK_REQUIRE(table_keys[l] && table_vals[l] && out_coords[l] && index_b[l] && (l > 0 || index_a[0]),
                 "build_levels: NULL level buffer");
    const void *src = l == 0 ? coords : (const void *)out_coords[l - 1];
    const int *n_dev = l == 0 ? nullptr : meta + (l - 1);
    if (l == 0) {
      // (the level's hash map is emptied by the same launch; levels > 0: by the fused pass at the end of the previous level)
      if (mode == 0)
        make_keys_kernel<0><<<grid, kBlock, 0, st>>>(src, n, n_dev, out_ts_host[l], keys, status, (unsigned long long *)table_keys[l],
                                                    table_vals[l], (uint64_t)cap, true);
      else
        make_keys_kernel<1><<<grid, kBlock, 0, st>>>(src, n, n_dev, out_ts_host[l], keys, status, (unsigned long long *)table_keys[l],
                                                    table_vals[l], (uint64_t)cap, true);
      MINK_CHECK_LAUNCH();
    }
    const uint32_t *asc = l == 0 ? status : nullptr;  // (strictly ascending input rows: level 0 without its hash insert)
    // index_a: first-occurrence rows (optional, kept for level 0), index_b: inverse / in2out
    int rc = unique_launch(keys, n, n_dev, table_keys[l], table_vals[l], cap, out_coords[l],
                           index_a[l], index_b[l], meta + l, uws, st, true, asc);
    if (rc) return rc;
    // inverse of this level + keys (and emptied hash map) of the next; the last level: + batch count
    // (meta[nlev] = status word, meta[nlev+1] = batch index of the last input row + 1)
    const bool last = l + 1 == nlev;
    inverse_next_keys_kernel<<<grid, kBlock, 0, st>>>(
        table_vals[l], slot_of_row, n, n_dev, index_b[l], last ? nullptr : (const void *)out_coords[l], last ? nullptr : meta + l,
        last ? 1 : out_ts_host[l + 1], keys, status, last ? nullptr : (unsigned long long *)table_keys[l + 1],
        last ? nullptr : table_vals[l + 1], (const int *)out_coords[0], meta, last ? meta + nlev + 1 : nullptr, asc);
    MINK_CHECK_LAUNCH();
  }
  return MINK_OK;
}

int mink_kernel_map(const uint64_t *in_table_keys, const int32_t *in_table_vals, int64_t in_cap,
                    const int32_t *out_coords, int64_t n_out, const int32_t *offsets_host, int32_t K, int32_t *nbr,
                    int32_t *nbr_t, void *stream) {
  MINK_REQUIRE(K >= 1 && K <= 27 && offsets_host, "kernel_map: kernel volume %d unsupported (1..27)", K);
  MINK_REQUIRE(in_cap >= 64 && (in_cap & (in_cap - 1)) == 0, "kernel_map: bad table capacity");
  MINK_REQUIRE(n_out >= 0 && n_out * K < (1ll << 31), "kernel_map: n_out*K overflows int32 indexing");
  if (n_out == 0) return MINK_OK;
  MINK_REQUIRE(in_table_keys && in_table_vals && out_coords && nbr, "kernel_map: NULL pointer");
  MINK_REQUIRE(((uintptr_t)out_coords & 15) == 0, "kernel_map: out_coords misaligned");
  Offsets off;
  memset(&off, 0, sizeof off);
  memcpy(off.d, offsets_host, sizeof(int) * 3 * K);
  kernel_map_kernel<<<dim3((unsigned)cdiv(n_out * K, kBlock)), kBlock, 0, (hipStream_t)stream>>>(
      in_table_keys, in_table_vals, (uint64_t)in_cap - 1, out_coords, n_out, K, off, nbr, nbr_t);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_set_overflow_sink(int32_t *host_word) {
  if (!host_word) {
    g_overflow_sink = nullptr;
    return MINK_OK;
  }
  void *dp = nullptr;
  MINK_REQUIRE(hipHostGetDevicePointer(&dp, host_word, 0) == hipSuccess && dp,
               "set_overflow_sink: the word must live in pinned (device-mapped) host memory");
  g_overflow_sink = (int *)dp;
  return MINK_OK;
}

int mink_kernel_map_batch(int32_t n, const MinkKernelMapDesc *d, void *stream) {
  MINK_REQUIRE(n >= 0 && (n == 0 || d), "kernel_map_batch: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  // One batch's plan is ~6 block indices and ~14 tables over independent maps: the 0xFF fills, the four passes of the
  // index builds and the table kernels are each issued as ONE launch over all of them (blockIdx.y = map), in that order.
  // ---- pass 0: validation, fills
  FillBatch fills;
  int n_fill = 0;
  int64_t fill_max = 0;
  auto flush_fills = [&]() -> int {
    if (n_fill == 0) return MINK_OK;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(cdiv(fill_max, kBlock), 2048));
    for (int rep = 0; rep < 1 + ((diag_dup() & 64) != 0); ++rep)
      fill_ff_kernel<<<dim3((unsigned)blocks, (unsigned)n_fill), kBlock, 0, st>>>(fills);
    MINK_CHECK_LAUNCH();
    n_fill = 0, fill_max = 0;
    return MINK_OK;
  };
  auto add_fill = [&](void *ptr, int64_t bytes) -> int {
    if (bytes <= 0) return MINK_OK;
    if (((uintptr_t)ptr & 3) || (bytes & 3)) {  // (not a run of 32-bit words: the runtime's fill)
      MINK_HIP(hipMemsetAsync(ptr, 0xFF, (size_t)bytes, st));
      return MINK_OK;
    }
    fills.ptr[n_fill] = (unsigned *)ptr, fills.nwords[n_fill] = bytes >> 2;
    fill_max = std::max(fill_max, bytes >> 4);
    if (++n_fill == 16) return flush_fills();
    return MINK_OK;
  };
  for (int i = 0; i < n; ++i) {
    const MinkKernelMapDesc &e = d[i];
    if (e.nbr_t && e.n_in > 0) {
      int rc = add_fill(e.nbr_t, (int64_t)sizeof(int32_t) * e.n_in * e.K);
      if (rc) return rc;
    }
    if (!e.blk_table) continue;
    MINK_REQUIRE(e.K >= 1 && e.K <= 27 && e.n_out >= 0 && e.n_in >= 0 && e.n_out * e.K < (1ll << 31) && e.in_ts >= 1,
                 "kernel_map_batch: bad shape in descriptor %d", i);
    MINK_REQUIRE(e.blk_cap >= 64 && (e.blk_cap & (e.blk_cap - 1)) == 0 && e.blk_base && e.blk_slot &&
                     e.blk_rowids && e.blk_counter && e.in_coords && ((uintptr_t)e.blk_table & 15) == 0,
                 "kernel_map_batch: bad block-index buffers in descriptor %d", i);
    MINK_REQUIRE(e.n_out == 0 || (e.out_coords && e.nbr && ((uintptr_t)e.out_coords & 15) == 0), "kernel_map_batch: NULL/misaligned pointer");
    if (e.blk_build) {
      int rc = add_fill(e.blk_table, (int64_t)sizeof(uint64_t) * 2 * e.blk_cap);
      if (!rc) rc = add_fill(e.blk_counter, sizeof(int32_t));  // (0xFFFFFFFF = every block found a slot)
      if (rc) return rc;
    }
  }
  {
    int rc = flush_fills();
    if (rc) return rc;
  }
  // ---- pass 1: block indices (insert, two leader passes, fill)
  for (int i0 = 0; i0 < n;) {
    BlkBuildBatch bb;
    bb.sticky = g_overflow_sink;
    int nb = 0;
    int64_t nmax = 0;
    int i = i0;
    for (; i < n && nb < kMaxBatch; ++i) {
      const MinkKernelMapDesc &e = d[i];
      if (!e.blk_table || !e.blk_build || e.n_in <= 0) continue;
      BlkBuild &q = bb.e[nb++];
      q.coords = e.in_coords, q.n = e.n_in, q.table = (unsigned long long *)e.blk_table, q.mask = (uint64_t)e.blk_cap - 1;
      q.slot = e.blk_slot, q.base = e.blk_base, q.rowids = e.blk_rowids, q.ts = e.in_ts, q.overflow = e.blk_counter;
      nmax = std::max(nmax, e.n_in);
    }
    i0 = i;
    if (nb == 0) continue;
    const dim3 g((unsigned)cdiv(nmax, kBlock), (unsigned)nb);
    const int dup = diag_dup();
    for (int rep = 0; rep < 1 + ((dup & 2) != 0); ++rep) blk_insert_kernel<<<g, kBlock, 0, st>>>(bb);
    MINK_CHECK_LAUNCH();
    for (int rep = 0; rep < 1 + ((dup & 16) != 0); ++rep) blk_leader_kernel<false><<<g, kBlock, 0, st>>>(bb);
    MINK_CHECK_LAUNCH();
    for (int rep = 0; rep < 1 + ((dup & 4) != 0); ++rep) blk_leader_kernel<true><<<g, kBlock, 0, st>>>(bb);
    MINK_CHECK_LAUNCH();
    for (int rep = 0; rep < 1 + ((dup & 32) != 0); ++rep) blk_fill_kernel<<<g, kBlock, 0, st>>>(bb);
    MINK_CHECK_LAUNCH();
  }
  // ---- pass 2: tables.  3x3x3 unit-offset tables (with / without the transposed table) and the small ones (pooling, 1x1x1)
  // in one launch per kind; anything else one launch per table
  KMap27Batch b27[2];
  KMapSBatch bs;
  int n27[2] = {0, 0}, ns = 0;
  int64_t max27[2] = {0, 0}, maxs = 0;
  auto flush27 = [&](int t) -> int {
    if (n27[t] == 0) return MINK_OK;
    const dim3 g((unsigned)cdiv(max27[t], kBlock), (unsigned)n27[t]);
    for (int rep = 0; rep < 1 + ((diag_dup() & 1) != 0); ++rep) {
      if (t) kernel_map_blk27_kernel<true><<<g, kBlock, 0, st>>>(b27[1]);
      else kernel_map_blk27_kernel<false><<<g, kBlock, 0, st>>>(b27[0]);
    }
    MINK_CHECK_LAUNCH();
    n27[t] = 0, max27[t] = 0;
    return MINK_OK;
  };
  auto flush_small = [&]() -> int {
    if (ns == 0) return MINK_OK;
    kernel_map_blk_small_kernel<<<dim3((unsigned)cdiv(maxs, kBlock), (unsigned)ns), kBlock, 0, st>>>(bs);
    MINK_CHECK_LAUNCH();
    ns = 0, maxs = 0;
    return MINK_OK;
  };
  for (int i = 0; i < n; ++i) {
    const MinkKernelMapDesc &e = d[i];
    if (!e.blk_table) {  // per-voxel hash map
      int rc = mink_kernel_map(e.in_table_keys, e.in_table_vals, e.in_cap, e.out_coords, e.n_out, e.offsets, e.K, e.nbr, e.nbr_t,
                               stream);
      if (rc) return rc;
      continue;
    }
    if (e.n_out == 0) continue;
    Offsets off;
    memset(&off, 0, sizeof off);
    memcpy(off.d, e.offsets, sizeof(int) * 3 * e.K);
    bool unit27 = e.K == 27;  // offsets == {-1,0,1}^3 * in_ts, x fastest: the 3x3x3 convolutions
    for (int k = 0; k < 27 && unit27; ++k)
      unit27 = off.d[3 * k] == (k % 3 - 1) * e.in_ts && off.d[3 * k + 1] == ((k / 3) % 3 - 1) * e.in_ts &&
               off.d[3 * k + 2] == (k / 9 - 1) * e.in_ts;
    if (unit27) {
      const int t = e.nbr_t ? 1 : 0;
      KMap27 &q = b27[t].e[n27[t]++];
      q.table = (const unsigned long long *)e.blk_table, q.base = e.blk_base, q.rowids = e.blk_rowids, q.out_coords = e.out_coords;
      q.mask = (uint64_t)e.blk_cap - 1, q.n_out = e.n_out, q.nbr = e.nbr, q.nbr_t = e.nbr_t, q.ts = e.in_ts;
      max27[t] = std::max(max27[t], e.n_out);
      if (n27[t] == kMaxBatch) {
        int rc = flush27(t);
        if (rc) return rc;
      }
      continue;
    }
    if (e.K <= 8) {
      KMapS &q = bs.e[ns++];
      q.table = (const unsigned long long *)e.blk_table, q.base = e.blk_base, q.rowids = e.blk_rowids, q.out_coords = e.out_coords;
      q.mask = (uint64_t)e.blk_cap - 1, q.n_out = e.n_out, q.nbr = e.nbr, q.nbr_t = e.nbr_t, q.ts = e.in_ts, q.K = e.K;
      memset(&q.off, 0, sizeof q.off);
      memcpy(q.off.d, e.offsets, sizeof(int) * 3 * e.K);
      maxs = std::max(maxs, e.n_out * e.K);
      if (ns == kMaxBatch) {
        int rc = flush_small();
        if (rc) return rc;
      }
      continue;
    }
    kernel_map_blk_kernel<<<dim3((unsigned)cdiv(e.n_out * e.K, kBlock)), kBlock, 0, st>>>(
        (const unsigned long long *)e.blk_table, e.blk_base, e.blk_rowids, (uint64_t)e.blk_cap - 1, e.in_ts, e.out_coords, e.n_out,
        e.K, off, e.nbr, e.nbr_t);
    MINK_CHECK_LAUNCH();
  }
  {
    int rc = flush27(0);
    if (!rc) rc = flush27(1);
    if (!rc) rc = flush_small();
    if (rc) return rc;
  }
  return MINK_OK;
}

int64_t mink_rulebook_workspace_bytes(int64_t n_out, int32_t K) {
  const int64_t nchunk = cdiv(n_out > 0 ? n_out : 1, kBlock);
  return 2 * align_up(4 * nchunk * K, 256) + 256;
}

int mink_rulebook(const int32_t *nbr, int64_t n_out, int32_t K, int32_t *counts, int32_t *pairs_in,
                  int32_t *pairs_out, void *workspace, int64_t workspace_bytes, void *stream) {
  MINK_REQUIRE(K >= 1 && K <= 27 && n_out >= 0 && counts, "rulebook: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (n_out == 0) {
    MINK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (K + 1), st));
    return MINK_OK;
  }
  MINK_REQUIRE(nbr && workspace && ((uintptr_t)workspace & 255) == 0, "rulebook: NULL/misaligned pointer");
  MINK_REQUIRE(workspace_bytes >= mink_rulebook_workspace_bytes(n_out, K), "rulebook: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
               (long long)(mink_rulebook_workspace_bytes(n_out, K)));
  const int64_t nchunk = cdiv(n_out, kBlock);
  int *chunk_counts = (int *)workspace;
  int *chunk_offsets = (int *)((char *)workspace + align_up(4 * nchunk * K, 256));
  int *total = chunk_offsets + nchunk * K;  // inside the trailing 256-byte pad
  const dim3 grid((unsigned)nchunk);
  rulebook_kernel<false><<<grid, kBlock, 0, st>>>(nbr, n_out, K, nchunk, chunk_counts, nullptr, nullptr, nullptr);
  MINK_CHECK_LAUNCH();
  scan_kernel<<<1, kScanBlock, 0, st>>>(chunk_counts, chunk_offsets, nchunk * K, total);
  MINK_CHECK_LAUNCH();
  rulebook_counts_kernel<<<1, 64, 0, st>>>(chunk_offsets, nchunk, K, total, counts);
  MINK_CHECK_LAUNCH();
  if (pairs_in && pairs_out) {
    rulebook_kernel<true><<<grid, kBlock, 0, st>>>(nbr, n_out, K, nchunk, nullptr, chunk_offsets, pairs_in, pairs_out);
    MINK_CHECK_LAUNCH();
  }
  return MINK_OK;
}


int64_t mink_class_partition_rows(int64_t n, int32_t pad) { return n + 8 * (int64_t)(pad - 1); }

int64_t mink_class_partition_workspace_bytes(int64_t n) {
  const int64_t nchunk = cdiv(n > 0 ? n : 1, kBlock);
  return 2 * align_up(4 * nchunk * 8, 256) + 256;
}

int mink_class_partition(const int32_t *coords, int64_t n, int32_t ts, int32_t pad, int32_t *perm, void *workspace,
                         int64_t workspace_bytes, void *stream) {
  MINK_REQUIRE(n >= 0 && ts >= 1 && pad >= 1 && perm, "class_partition: bad arguments");
  MINK_REQUIRE(n == 0 || workspace_bytes >= mink_class_partition_workspace_bytes(n), "class_partition: workspace of %lld bytes, %lld needed",
               (long long)workspace_bytes, (long long)mink_class_partition_workspace_bytes(n));
  hipStream_t st = (hipStream_t)stream;
  MINK_HIP(hipMemsetAsync(perm, 0xFF, sizeof(int32_t) * mink_class_partition_rows(n, pad), st));
  if (n == 0) return MINK_OK;
  MINK_REQUIRE(coords && workspace && ((uintptr_t)coords & 15) == 0 && ((uintptr_t)workspace & 255) == 0,
               "class_partition: NULL/misaligned pointer");
  const int64_t nchunk = cdiv(n, kBlock);
  int *chunk_counts = (int *)workspace;
  int *chunk_offsets = (int *)((char *)workspace + align_up(4 * nchunk * 8, 256));
  int *total = chunk_offsets + nchunk * 8;
  const dim3 grid((unsigned)nchunk);
  class_partition_kernel<false><<<grid, kBlock, 0, st>>>(coords, n, ts, nchunk, pad, chunk_counts, nullptr, nullptr, nullptr);
  MINK_CHECK_LAUNCH();
  scan_kernel<<<1, kScanBlock, 0, st>>>(chunk_counts, chunk_offsets, nchunk * 8, total);
  MINK_CHECK_LAUNCH();
  class_partition_kernel<true><<<grid, kBlock, 0, st>>>(coords, n, ts, nchunk, pad, nullptr, chunk_offsets, total, perm);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_class_partition_batch(int32_t n_maps, const MinkClassPartitionDesc *d, void *stream) {
  MINK_REQUIRE(n_maps >= 0 && n_maps <= 8 && (n_maps == 0 || d), "class_partition_batch: 0..8 maps per call");
  hipStream_t st = (hipStream_t)stream;
  FillBatch fills;
  ClassPartBatch cb;
  ScanBatch sb;
  int nf = 0, nb = 0;
  int64_t fill_max = 0, chunk_max = 0;
  for (int i = 0; i < n_maps; ++i) {
    const MinkClassPartitionDesc &e = d[i];
    MINK_REQUIRE(e.n >= 0 && e.ts >= 1 && e.pad >= 1 && e.perm, "class_partition_batch: bad arguments in descriptor %d", i);
    MINK_REQUIRE(e.n == 0 || e.workspace_bytes >= mink_class_partition_workspace_bytes(e.n),
                 "class_partition_batch: workspace of %lld bytes, %lld needed", (long long)e.workspace_bytes,
                 (long long)mink_class_partition_workspace_bytes(e.n));
    const int64_t bytes = (int64_t)sizeof(int32_t) * mink_class_partition_rows(e.n, e.pad);
    if (bytes > 0) fills.ptr[nf] = (unsigned *)e.perm, fills.nwords[nf] = bytes >> 2, fill_max = std::max(fill_max, bytes >> 4), ++nf;
    if (e.n == 0) continue;
    MINK_REQUIRE(e.coords && e.workspace && ((uintptr_t)e.coords & 15) == 0 && ((uintptr_t)e.workspace & 255) == 0,
                 "class_partition_batch: NULL/misaligned pointer in descriptor %d", i);
    ClassPart &q = cb.e[nb];
    q.coords = e.coords, q.n = e.n, q.nchunk = cdiv(e.n, kBlock), q.ts = e.ts, q.pad = e.pad, q.perm = e.perm;
    q.chunk_counts = (int *)e.workspace;
    q.chunk_offsets = (int *)((char *)e.workspace + align_up(4 * q.nchunk * 8, 256));
    q.total = q.chunk_offsets + q.nchunk * 8;
    sb.in[nb] = q.chunk_counts, sb.out[nb] = q.chunk_offsets, sb.total[nb] = q.total, sb.n[nb] = q.nchunk * 8;
    chunk_max = std::max(chunk_max, q.nchunk);
    ++nb;
  }
  if (nf) {
    fill_ff_kernel<<<dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(fill_max, kBlock), 2048)), (unsigned)nf), kBlock, 0, st>>>(fills);
    MINK_CHECK_LAUNCH();
  }
  if (nb == 0) return MINK_OK;
  const dim3 grid((unsigned)chunk_max, (unsigned)nb);
  class_partition_batch_kernel<false><<<grid, kBlock, 0, st>>>(cb);
  MINK_CHECK_LAUNCH();
  scan_batch_kernel<<<dim3((unsigned)nb), kScanBlock, 0, st>>>(sb);
  MINK_CHECK_LAUNCH();
  class_partition_batch_kernel<true><<<grid, kBlock, 0, st>>>(cb);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

int mink_batch_offsets(const int32_t *coords, int64_t n, int32_t B, int32_t *batch_offsets, uint32_t *status,
                       void *stream) {
  MINK_REQUIRE(n >= 0 && B >= 0 && batch_offsets && status, "batch_offsets: bad arguments");
  MINK_REQUIRE(n == 0 || coords, "batch_offsets: NULL coords");
  const int64_t work = (n > B + 1) ? n : B + 1;
  batch_offsets_kernel<<<dim3((unsigned)cdiv(work, kBlock)), kBlock, 0, (hipStream_t)stream>>>(coords, n, B,
                                                                                              batch_offsets, status);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}


int mink_decode_plenoxel(const int32_t *links, const float *density, const uint8_t *sh_q, const int32_t *scene_offsets,
                         int32_t n_scenes, const float *sh_scale, const float *sh_min, int64_t n, int32_t reso_y,
                         int32_t reso_z, int32_t col_density, int32_t col_sh, int32_t col_ones, int32_t col_xyzs,
                         float *scene_scratch, int32_t C, int32_t *coords, float *feats, int32_t ldf, void *stream) {
  MINK_REQUIRE(n >= 0 && n_scenes >= 1 && reso_y >= 1 && reso_z >= 1 && C >= 1 && ldf >= C, "decode_plenoxel: bad shape");
  const int want = (col_density >= 0) + 27 * (col_sh >= 0) + (col_ones >= 0) + 3 * (col_xyzs >= 0);
  MINK_REQUIRE(want == C && col_density < C && col_ones < C && (col_sh < 0 || col_sh + 27 <= C) && (col_xyzs < 0 || col_xyzs + 3 <= C),
               "decode_plenoxel: the feature columns (density %d, sh %d, ones %d, xyzs %d) do not tile %d channels", col_density,
               col_sh, col_ones, col_xyzs, C);
  MINK_REQUIRE(col_xyzs < 0 || scene_scratch, "decode_plenoxel: the xyzs feature needs scene_scratch[n_scenes]");
  if (n == 0) return MINK_OK;
  MINK_REQUIRE(links && scene_offsets && coords && feats && (col_density < 0 || density) &&
                   (col_sh < 0 || (sh_q && sh_scale && sh_min)),
               "decode_plenoxel: NULL pointer");
  MINK_REQUIRE(((uintptr_t)coords & 15) == 0, "decode_plenoxel: coords must be 16-byte aligned");
  const int64_t threads = n * ((C + 3) / 4);
  if (col_xyzs >= 0) {  // per-scene largest centred norm first (one extra pass over the links)
    MINK_HIP(hipMemsetAsync(scene_scratch, 0, sizeof(float) * n_scenes, (hipStream_t)stream));
    decode_xyz_maxnorm_kernel<<<dim3((unsigned)cdiv(n, kBlock)), kBlock, 0, (hipStream_t)stream>>>(
        links, scene_offsets, n_scenes, n, reso_y, reso_z, (unsigned *)scene_scratch);
    MINK_CHECK_LAUNCH();
  }
  decode_plenoxel_kernel<<<dim3((unsigned)cdiv(threads, kBlock)), kBlock, 0, (hipStream_t)stream>>>(
      links, density, sh_q, scene_offsets, n_scenes, sh_scale, sh_min, n, reso_y, reso_z, col_density, col_sh, col_ones, col_xyzs,
      scene_scratch, C, coords, feats, ldf, ((uintptr_t)feats & 15) == 0 && (ldf & 3) == 0);
  MINK_CHECK_LAUNCH();
  return MINK_OK;
}

}  // extern "C"
