/*
 * oracle/mink_maps.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the coordinate-structure algorithms of the
 * MinkowskiEngine CPU backend that the reference's hot path calls through
 * `import MinkowskiEngine as ME` (reference call sites:
 * co3d_3d/src/models/mink/resnet.py:163-177, modules/common.py:116-125,
 * models/mink/base_model.py:10-13).  MinkowskiEngine itself is an un-vendored,
 * un-pinned third-party dependency (co3d_3d/README.md:13, install.sh:50-52;
 * effectively v0.5.4) that is absent from /root/reference and not installable
 * here, so this file restates its published algorithm:
 *
 *   A1  quantisation: floor() of float field coordinates, batch column copied
 *   A2  insert_and_map: SEQUENTIAL hash insert, row id = order of first
 *       occurrence; returns unique_index / inverse_mapping
 *   A3  stride map: floor(c / s) * s per spatial dim, de-duplicated
 *   A4/A5 kernel map: for every output row and kernel offset, look the
 *       coordinate (c_out + offset) up in the INPUT map
 *
 * PARITY UNPINNED by the reference: the reference ships no test, fixture or
 * golden vector for this path.  The oracle is pinned
 * instead against an independent dense torch.nn.functional.conv3d identity and
 * brute-force numpy set arithmetic (tests/test_oracle_*.py).
 *
 * Row-order convention (documented in DESIGN.md): ME's CPU stride map takes row
 * ids from robin_hood hash iteration order, which is not reproducible; both this
 * oracle and the HIP path use FIRST-OCCURRENCE order (the order ME itself uses
 * for insert_and_map), and tests additionally compare in canonical
 * (lexicographically sorted) order.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint64_t *keys;
  int32_t *vals;
  uint64_t mask;
} orc_table;

#define ORC_EMPTY 0xFFFFFFFFFFFFFFFFull

/* (b,x,y,z) int32 -> 64-bit key; 16 bits per field, spatial fields biased by 2^15 */
static inline int orc_pack(const int32_t *c, uint64_t *key) {
  int64_t b = c[0], x = (int64_t)c[1] + 32768, y = (int64_t)c[2] + 32768,
          z = (int64_t)c[3] + 32768;
  if (b < 0 || b > 65534 || x < 0 || x > 65535 || y < 0 || y > 65535 || z < 0 ||
      z > 65535)
    return -1;
  *key = ((uint64_t)b << 48) | ((uint64_t)x << 32) | ((uint64_t)y << 16) | (uint64_t)z;
  return 0;
}

static inline uint64_t orc_mix(uint64_t k) { /* splitmix64 finaliser */
  k ^= k >> 30;
  k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27;
  k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return k;
}

static int orc_table_init(orc_table *t, int64_t n) {
  uint64_t cap = 16;
  while (cap < (uint64_t)(2 * n + 1)) cap <<= 1;
  t->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
  t->vals = (int32_t *)malloc(cap * sizeof(int32_t));
  if (!t->keys || !t->vals) return -1;
  memset(t->keys, 0xFF, cap * sizeof(uint64_t));
  t->mask = cap - 1;
  return 0;
}

static void orc_table_free(orc_table *t) {
  free(t->keys);
  free(t->vals);
}

/* insert if absent; returns the value stored for key (existing or new) */
static inline int32_t orc_insert(orc_table *t, uint64_t key, int32_t val) {
  uint64_t s = orc_mix(key) & t->mask;
  for (;;) {
    if (t->keys[s] == ORC_EMPTY) {
      t->keys[s] = key;
      t->vals[s] = val;
      return val;
    }
    if (t->keys[s] == key) return t->vals[s];
    s = (s + 1) & t->mask;
  }
}

static inline int32_t orc_find(const orc_table *t, uint64_t key) {
  uint64_t s = orc_mix(key) & t->mask;
  for (;;) {
    if (t->keys[s] == ORC_EMPTY) return -1;
    if (t->keys[s] == key) return t->vals[s];
    s = (s + 1) & t->mask;
  }
}

/* A1: float (b,x,y,z) rows -> int32 rows, floor per field. */
void orc_quantize(const float *fcoords, int64_t n, int32_t *out) {
  for (int64_t i = 0; i < 4 * n; ++i) out[i] = (int32_t)floorf(fcoords[i]);
}

/* A2: sequential insert_and_map.  unique_index[u] = first input row holding
 * unique coordinate u; inverse[i] = unique row of input row i.
 * Returns the number of unique rows, or -1 on a coordinate outside the
 * packable range / allocation failure. */
int64_t orc_unique(const int32_t *coords, int64_t n, int32_t *unique_index,
                   int32_t *inverse) {
  orc_table t;
  if (orc_table_init(&t, n)) return -1;
  int32_t nu = 0;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t key;
    if (orc_pack(coords + 4 * i, &key)) {
      orc_table_free(&t);
      return -1;
    }
    int32_t u = orc_insert(&t, key, nu);
    if (u == nu) unique_index[nu++] = (int32_t)i;
    inverse[i] = u;
  }
  orc_table_free(&t);
  return nu;
}

static inline int32_t orc_floor_to(int32_t c, int32_t s) {
  /* floor(c / s) * s, correct for negative c (A3: float floor semantics) */
  int32_t q = c / s, r = c % s;
  if (r != 0 && ((r < 0) != (s < 0))) --q;
  return q * s;
}

/* A3: stride map.  out_ts = tensor stride of the OUTPUT map (in_ts * stride).
 * out_coords must have room for n rows; in2out[i] = output row of input row i.
 * Output rows are numbered in first-occurrence order. Returns n_out or -1. */
int64_t orc_stride_map(const int32_t *coords, int64_t n, int32_t out_ts,
                       int32_t *out_coords, int32_t *in2out) {
  orc_table t;
  if (orc_table_init(&t, n)) return -1;
  int32_t no = 0;
  for (int64_t i = 0; i < n; ++i) {
    int32_t c[4] = {coords[4 * i], orc_floor_to(coords[4 * i + 1], out_ts),
                    orc_floor_to(coords[4 * i + 2], out_ts),
                    orc_floor_to(coords[4 * i + 3], out_ts)};
    uint64_t key;
    if (orc_pack(c, &key)) {
      orc_table_free(&t);
      return -1;
    }
    int32_t o = orc_insert(&t, key, no);
    if (o == no) {
      memcpy(out_coords + 4 * (int64_t)no, c, sizeof c);
      ++no;
    }
    in2out[i] = o;
  }
  orc_table_free(&t);
  return no;
}

/* A4/A5: kernel map as a dense neighbour table.
 * offsets: [K][3] int32 spatial offsets already scaled by dilation * in_ts.
 * nbr[o*K + k] = input row at (out_coords[o] + offsets[k]) or -1.
 * The ME-format per-offset lists {k: [2,n]} are the (in,out) pairs with
 * nbr >= 0 in column k, ordered by output row (canonical order). */
int orc_kernel_map(const int32_t *in_coords, int64_t n_in, const int32_t *out_coords,
                   int64_t n_out, const int32_t *offsets, int32_t K, int32_t *nbr) {
  orc_table t;
  if (orc_table_init(&t, n_in)) return -1;
  for (int64_t i = 0; i < n_in; ++i) {
    uint64_t key;
    if (orc_pack(in_coords + 4 * i, &key)) {
      orc_table_free(&t);
      return -1;
    }
    orc_insert(&t, key, (int32_t)i);
  }
#pragma omp parallel for schedule(static)
  for (int64_t o = 0; o < n_out; ++o) {
    const int32_t *c = out_coords + 4 * o;
    for (int32_t k = 0; k < K; ++k) {
      int32_t q[4] = {c[0], c[1] + offsets[3 * k], c[2] + offsets[3 * k + 1],
                      c[3] + offsets[3 * k + 2]};
      uint64_t key;
      nbr[o * K + k] = orc_pack(q, &key) ? -1 : orc_find(&t, key);
    }
  }
  orc_table_free(&t);
  return 0;
}
