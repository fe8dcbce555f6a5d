/*
 * oracle/sanitize_driver.c -- TEST INFRASTRUCTURE (tests/test_oracle_maps.py builds it together with
 * mink_maps.c under -fsanitize=address,undefined and runs it on the CPU).
 *
 * Drives every entry point of the C restatement over seeded inputs that include duplicates, negative
 * coordinates, the edges of the packable range, an out-of-range row and empty inputs, with every buffer
 * allocated at its exact size so that an out-of-bounds access, a signed overflow or a misaligned read is
 * reported by the sanitizers.  Prints a checksum so the run cannot be optimised away; exit code 0 = clean.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void orc_quantize(const float *fcoords, int64_t n, int32_t *out);
int64_t orc_unique(const int32_t *coords, int64_t n, int32_t *unique_index, int32_t *inverse);
int64_t orc_stride_map(const int32_t *coords, int64_t n, int32_t out_ts, int32_t *out_coords, int32_t *in2out);
int orc_kernel_map(const int32_t *in_coords, int64_t n_in, const int32_t *out_coords, int64_t n_out,
                   const int32_t *offsets, int32_t K, int32_t *nbr);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}

int main(void) {
  uint64_t sum = 0;
  for (int round = 0; round < 4; ++round) {
    const int64_t n = round == 3 ? 0 : 5000 + 137 * round;
    const int span = round == 0 ? 12 : 40;  /* small span: many duplicates */
    float *f = (float *)malloc(sizeof(float) * 4 * (n > 0 ? n : 1));
    int32_t *q = (int32_t *)malloc(sizeof(int32_t) * 4 * (n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {
      f[4 * i] = (float)(i * 3 / (n > 0 ? n : 1)); /* sorted batch column 0..2 */
      for (int d = 1; d < 4; ++d) f[4 * i + d] = (float)((int)(rnd() % (2 * span)) - span) + (float)(rnd() % 1000) / 1000.0f;
    }
    if (n > 8 && round == 1) { /* the corners of the packable range */
      f[4 * 5 + 1] = -32768.0f, f[4 * 5 + 2] = 32767.0f, f[4 * 5 + 3] = -32768.0f;
      f[4 * 6 + 1] = 32767.5f, f[4 * 6 + 2] = -32767.5f, f[4 * 6 + 3] = 32767.0f;
    }
    orc_quantize(f, n, q);
    int32_t *uidx = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    int32_t *inv = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    const int64_t nu = orc_unique(q, n, uidx, inv);
    if (nu < 0 || nu > n) return 2;
    int32_t *u = (int32_t *)malloc(sizeof(int32_t) * 4 * (nu > 0 ? nu : 1));
    for (int64_t j = 0; j < nu; ++j)
      for (int d = 0; d < 4; ++d) u[4 * j + d] = q[4 * (int64_t)uidx[j] + d];
    for (int64_t i = 0; i < n; ++i) sum += (uint64_t)inv[i];
    int32_t *cur = u;
    int64_t ncur = nu;
    for (int ts = 2; ts <= 8; ts *= 2) { /* a pyramid of stride maps + kernel maps k=3 (centred) and k=2 ({0,1}) */
      int32_t *oc = (int32_t *)malloc(sizeof(int32_t) * 4 * (ncur > 0 ? ncur : 1));
      int32_t *i2o = (int32_t *)malloc(sizeof(int32_t) * (ncur > 0 ? ncur : 1));
      const int64_t no = orc_stride_map(cur, ncur, ts, oc, i2o);
      if (no < 0 || no > ncur) return 3;
      int32_t off3[27 * 3], off2[8 * 3];
      const int in_ts = ts / 2;
      for (int k = 0; k < 27; ++k) off3[3 * k] = (k % 3 - 1) * in_ts, off3[3 * k + 1] = (k / 3 % 3 - 1) * in_ts, off3[3 * k + 2] = (k / 9 - 1) * in_ts;
      for (int k = 0; k < 8; ++k) off2[3 * k] = (k & 1) * in_ts, off2[3 * k + 1] = (k >> 1 & 1) * in_ts, off2[3 * k + 2] = (k >> 2) * in_ts;
      int32_t *nb3 = (int32_t *)malloc(sizeof(int32_t) * 27 * (no > 0 ? no : 1));
      int32_t *nb2 = (int32_t *)malloc(sizeof(int32_t) * 8 * (no > 0 ? no : 1));
      if (orc_kernel_map(cur, ncur, oc, no, off3, 27, nb3)) return 4;
      if (orc_kernel_map(cur, ncur, oc, no, off2, 8, nb2)) return 5;
      for (int64_t j = 0; j < 27 * no; ++j) sum += (uint64_t)(nb3[j] + 1);
      for (int64_t j = 0; j < 8 * no; ++j) sum += (uint64_t)(nb2[j] + 1);
      for (int64_t i = 0; i < ncur; ++i)
        if (i2o[i] < 0 || i2o[i] >= no) return 6;
      free(nb3), free(nb2), free(i2o);
      if (cur != u) free(cur);
      cur = oc, ncur = no;
    }
    if (cur != u) free(cur);
    if (n > 0) { /* one row outside the packable range must be refused, not wrapped */
      q[1] = 40000;
      if (orc_unique(q, n, uidx, inv) != -1) return 7;
      if (orc_stride_map(q, n, 2, f == NULL ? NULL : (int32_t *)f, inv) != -1) return 8;
    }
    free(u), free(uidx), free(inv), free(q), free(f);
  }
  printf("sanitize_driver ok checksum=%llu\n", (unsigned long long)sum);
  return 0;
}
