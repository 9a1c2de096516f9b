/* Test-only driver: the host side of the library (acm_host.c + acm_flat.c, no HIP) under
 * AddressSanitizer and UBSan -- dictionaries of 1-, 2-, 4- and 8-byte symbols, flat tables, blobs
 * (good and damaged), comparator classes, keyword spellings.  Built and run by
 * tests/test_sanitizers.py; exits 0 when every check held. */
#define _GNU_SOURCE
#include "aho_corasick.h"
#include "acm_gpu.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    if (!(x)) {                                                                                    \
      fprintf (stderr, "check failed: %s (%s:%d)\n", #x, __FILE__, __LINE__);                      \
      exit (1);                                                                                    \
    }                                                                                              \
  } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint64_t
rnd (void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

static int
casecmp8 (const void *a, const void *b, const void *arg) {
  (void)arg;
  unsigned x = *(const unsigned char *)a, y = *(const unsigned char *)b;
  if (x >= 'A' && x <= 'Z')
    x += 32;
  if (y >= 'A' && y <= 'Z')
    y += 32;
  return x > y ? 1 : (x < y ? -1 : 0);
}

static int
casecmp32 (const void *a, const void *b, const void *arg) {
  (void)arg;
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  if (x >= 'A' && x <= 'Z')
    x += 32;
  if (y >= 'A' && y <= 'Z')
    y += 32;
  return x > y ? 1 : (x < y ? -1 : 0);
}

/* builds a machine of n_kw random keywords over `vocab` symbols of sym_bytes bytes */
static ACMachine *
build (size_t sym_bytes, size_t *arg, unsigned n_kw, unsigned vocab, unsigned char **letters_out, CMP_TYPE cmp) {
  *arg = sym_bytes;
  ACMachine *m = acm_create (cmp ? cmp : ACM_CMP_DEFAULT, arg, 0);
  unsigned char *letters = malloc ((size_t)n_kw * 12 * sym_bytes);
  size_t used = 0;
  for (unsigned k = 0; k < n_kw; k++) {
    ACState *s = acm_initiate (m);
    unsigned len = 1 + (unsigned)(rnd () % 11);
    for (unsigned i = 0; i < len; i++) {
      uint64_t v = cmp ? 'A' + rnd () % 58 : rnd () % vocab;
      if (sym_bytes == 8)
        v = v * 0x9E3779B97F4A7C15ull; /* spread over 64 bits */
      memcpy (letters + used, &v, sym_bytes);
      acm_insert_letter_of_keyword (&s, letters + used);
      used += sym_bytes;
    }
    acm_insert_end_of_keyword (&s, 0, 0);
  }
  *letters_out = letters;
  return m;
}

static void
exercise_flat (ACMFlat *flat) {
  ACMFlatInfo info;
  ACMFlatView v;
  acm_flat_info (flat, &info);
  acm_flat_view (flat, &v);
  CHECK (info.n_edges == info.n_states - 1);
  for (uint32_t k = 0; k < info.n_keywords; k += 7) {
    unsigned char buf[16 * 8];
    uint32_t len = 0;
    CHECK (acm_flat_keyword (flat, k, buf, 16, &len) == ACM_GPU_OK);
    CHECK (len >= 1 && len <= 11);
  }
  const size_t bytes = acm_flat_blob_bytes (flat);
  unsigned char *blob = malloc (bytes);
  CHECK (acm_flat_to_blob (flat, blob, bytes) == ACM_GPU_OK);
  ACMFlat *back = NULL;
  CHECK (acm_flat_from_blob (blob, bytes, &back) == ACM_GPU_OK);
  ACMFlatInfo info2;
  acm_flat_info (back, &info2);
  CHECK (memcmp (&info, &info2, sizeof info) == 0);
  acm_flat_release (back);
  /* damaged blobs: truncated, and random byte flips (the checksum or the structure checks refuse) */
  CHECK (acm_flat_from_blob (blob, bytes - 1, &back) == ACM_GPU_E_FORMAT && back == NULL);
  for (int t = 0; t < 50; t++) {
    const size_t at = rnd () % bytes;
    const unsigned char old = blob[at];
    blob[at] ^= (unsigned char)(1 + rnd () % 255);
    int rc = acm_flat_from_blob (blob, bytes, &back);
    CHECK (rc == ACM_GPU_E_FORMAT || (rc == ACM_GPU_OK && at >= 52 && at < 56 && 0)); /* never accepted */
    CHECK (back == NULL);
    blob[at] = old;
  }
  free (blob);
  if (info.sym_bytes == 1 && info.n_edges) {
    const uint32_t rows = info.n_states < 300 ? info.n_states : 300;
    void *dense = malloc ((size_t)rows * info.width * 4);
    CHECK (acm_flat_dense_rows (flat, rows, 4, dense) == ACM_GPU_OK);
    if (info.n_states <= 32768)
      CHECK (acm_flat_dense_rows (flat, rows, 2, dense) == ACM_GPU_OK);
    free (dense);
  }
}

int
main (void) {
  const size_t sizes[4] = { 1, 2, 4, 8 };
  for (int si = 0; si < 4; si++) {
    size_t arg;
    unsigned char *letters;
    ACMachine *m = build (sizes[si], &arg, 400, sizes[si] == 1 ? 26 : 3000, &letters, 0);
    ACMFlat *flat = NULL;
    CHECK (acm_flatten (m, &flat) == ACM_GPU_OK);
    exercise_flat (flat);
    acm_flat_release (flat);
    /* the per-symbol API on the same machine */
    const ACState *cur = acm_initiate (m);
    MatchHolder h;
    acm_matcher_init (&h);
    size_t found = 0;
    for (size_t i = 0; i < 400 * 3; i++) {
      size_t nb = acm_match (&cur, letters + (i % 900) * sizes[si]);
      for (size_t j = 0; j < nb; j++) {
        acm_get_match (cur, j, &h);
        found += h.length;
      }
    }
    CHECK (found > 0);
    acm_matcher_release (&h);
    acm_release (m);
    free (letters);
  }
  { /* comparator classes */
    size_t arg;
    unsigned char *letters;
    ACMachine *m = build (1, &arg, 300, 0, &letters, casecmp8);
    ACMFlat *flat = NULL;
    CHECK (acm_flatten (m, &flat) == ACM_GPU_E_INELIGIBLE);
    CHECK (acm_flatten_classes (m, 1, &flat) == ACM_GPU_OK);
    ACMFlatView v;
    acm_flat_view (flat, &v);
    CHECK (v.class_map && v.class_entries == 256 && v.n_classes == 256 - 26 && v.class_map['a'] == v.class_map['A']);
    exercise_flat (flat);
    acm_flat_release (flat);
    CHECK (acm_flatten_classes (m, 8, &flat) == ACM_GPU_E_ARG);
    acm_release (m);
    free (letters);
  }
  { /* comparator classes of 4-byte symbols: those of the dictionary's own symbols */
    size_t arg;
    unsigned char *letters;
    ACMachine *m = build (4, &arg, 300, 0, &letters, casecmp32);
    ACMFlat *flat = NULL;
    CHECK (acm_flatten_classes (m, 4, &flat) == ACM_GPU_OK);
    ACMFlatView v;
    ACMFlatInfo fi;
    acm_flat_view (flat, &v);
    acm_flat_info (flat, &fi);
    CHECK (v.keys32 && v.keys32_class && v.class_rep32 && v.n_keys32 >= v.n_classes && v.n_classes >= 26 && fi.sym_bytes == 4);
    for (uint32_t i = 0; i < v.n_keys32; i++)
      CHECK (v.keys32_class[i] >= 1 && v.keys32_class[i] <= v.n_classes && (i == 0 || v.keys32[i - 1] < v.keys32[i]));
    for (uint32_t e = 0; e < fi.n_edges; e++)
      CHECK (v.edge_sym[e] >= 1 && v.edge_sym[e] <= v.n_classes);
    CHECK (acm_flat_blob_bytes (flat) > 0);
    void *blob = malloc (acm_flat_blob_bytes (flat));
    CHECK (acm_flat_to_blob (flat, blob, acm_flat_blob_bytes (flat)) == ACM_GPU_E_ARG);
    free (blob);
    acm_flat_release (flat);
    acm_release (m);
    free (letters);
  }
  puts ("sanitizer driver: all checks held");
  return 0;
}
