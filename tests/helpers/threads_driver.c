/* The reference's threading model (README.md:364, aho_corasick.c:81): several threads work on ONE
 * shared machine, each with its own cursor, while another thread keeps inserting keywords.
 * Built with -fsanitize=thread (and once with -fsanitize=address) over acm_host.c + acm_flat.c by
 * tests/test_sanitizers.py.
 *
 * Phase 1 keywords (value != NULL) are complete before the readers start.
 *   mode "disjoint": phase 2 keywords start with a letter the text never holds, so they add no
 *     match and touch no output chain a reader walks: what the readers report must EQUAL a quiet
 *     single-threaded scan whatever the inserter does meanwhile;
 *   mode "shared": phase 2 keywords use the text's alphabet, so failure links are re-pointed and
 *     output counts grow under the readers' feet.  A match call that races with the insertion of
 *     a keyword on its own output chain may see the chain before or after it (as in the
 *     reference, whose acm_match / acm_get_match are not locked either): here the readers only
 *     have to stay memory-safe (the sanitizer's verdict) and the quiet scan afterwards must
 *     report exactly the phase 1 matches again. */
#include "acm.h"
#include "acm_gpu.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h> /* (ThreadSanitizer does not see glibc's thrd_create) */

#define READERS 4
#define TEXT_LEN 400000
#define PHASE1 300
#define PHASE2 4000

static ACMachine *M;
static char *text;
static char (*kw)[16];
static uint64_t expected_matches, expected_sum;
static int shared_alphabet;

static uint64_t
sm (uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

static void
insert_kw (const char *w, void *value) {
  ACState *s = acm_initiate (M);
  for (const char *p = w; *p; p++)
    acm_insert_letter_of_keyword (&s, (void *)p);
  acm_insert_end_of_keyword (&s, value, 0);
}

static void
scan (uint64_t *matches, uint64_t *sum) {
  const ACState *c = acm_initiate (M);
  MatchHolder h;
  acm_matcher_init (&h);
  uint64_t n = 0, sig = 0;
  for (size_t i = 0; i < TEXT_LEN; i++) {
    size_t nb = acm_match (&c, &text[i]);
    for (size_t j = 0; j < nb; j++) {
      acm_get_match (c, j, &h);
      if (h.value) { /* phase 1 keywords only */
        n++;
        sig += sm (i * 1315423911ull ^ (uint64_t)h.length << 40);
      }
    }
  }
  acm_matcher_release (&h);
  *matches = n;
  *sum = sig;
}

static void *
reader (void *arg) {
  (void)arg;
  for (int round = 0; round < 3; round++) {
    uint64_t n, sig;
    scan (&n, &sig);
    if (shared_alphabet ? n == 0 : (n != expected_matches || sig != expected_sum)) {
      fprintf (stderr, "reader: %llu matches (want %llu)\n", (unsigned long long)n, (unsigned long long)expected_matches);
      return (void *)1;
    }
  }
  return 0;
}

static void *
writer (void *arg) {
  (void)arg;
  for (int k = PHASE1; k < PHASE1 + PHASE2; k++)
    insert_kw (kw[k], 0);
  return 0;
}

int
main (int argc, char **argv) {
  shared_alphabet = argc > 1 && strcmp (argv[1], "shared") == 0;
  M = acm_create (ACM_CMP_DEFAULT, &(size_t){ 1 }, 0);
  text = malloc (TEXT_LEN);
  kw = malloc ((size_t)(PHASE1 + PHASE2) * sizeof *kw);
  for (size_t i = 0; i < TEXT_LEN; i++)
    text[i] = (char)('a' + sm (i + 42) % 16); /* a-p only */
  for (int k = 0; k < PHASE1 + PHASE2; k++) {
    const int len = 2 + (int)(sm (1000003ull * (uint64_t)k + 1) % 6);
    for (int j = 0; j < len; j++) {
      const uint64_t h = sm (64ull * (uint64_t)k + (uint64_t)j + 7777);
      /* phase 2: first letter outside the text's alphabet, the rest shared with it, so that the
       * new states hang below the root and get failure links into the phase-1 trie */
      kw[k][j] = (k < PHASE1 || shared_alphabet || j) ? (char)('a' + h % 16) : (char)('q' + h % 10);
    }
    kw[k][len] = 0;
  }
  for (int k = 0; k < PHASE1; k++)
    insert_kw (kw[k], &M);
  scan (&expected_matches, &expected_sum);
  if (expected_matches < 1000) {
    fprintf (stderr, "too few matches for a meaningful test: %llu\n", (unsigned long long)expected_matches);
    return 2;
  }
  pthread_t w, r[READERS];
  for (int i = 0; i < READERS; i++)
    pthread_create (&r[i], 0, reader, 0);
  pthread_create (&w, 0, writer, 0);
  int bad = 0;
  void *res;
  for (int i = 0; i < READERS; i++) {
    pthread_join (r[i], &res);
    bad |= res != 0;
  }
  pthread_join (w, &res);
  bad |= res != 0;
  /* concurrent acm_get_keyword beside nothing else now: every keyword is there */
  if (acm_nb_keywords (M) < PHASE1)
    bad = 1;
  uint64_t n, sig;
  scan (&n, &sig);
  if (n != expected_matches || sig != expected_sum)
    bad = 1;
  acm_release (M);
  free (text);
  free (kw);
  if (bad)
    return 1;
  printf ("threads ok (%s): %llu matches seen by %d readers beside %d insertions\n", shared_alphabet ? "shared" : "disjoint",
          (unsigned long long)expected_matches, READERS, PHASE2);
  return 0;
}
