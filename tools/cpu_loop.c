/* cpu_loop.c -- the reference's caller loop (examples/test.c:17-23: acm_match, then acm_get_match
 * for every index) driven over the DROP-IN's own per-symbol host API (libac75_amd.so, acm_host.c),
 * the way SURVEY.md 8(d) specifies the CPU baseline: T threads, the text sharded with an lmax - 1
 * overlap, one cursor per thread over ONE shared machine (the threading model the reference
 * sanctions, README.md:364).  bench.py times it beside the GPU path; no GPU, no oracle involved.
 *
 * The machine must have been filled with value = (void *)(keyword_id + 1) per keyword (the
 * reference API has no keyword id; the value pointer carries it), so that the loop can form the
 * same order-independent digest as the record set: sum over matches of
 * splitmix64 ((end_pos * 1315423911) ^ (length << 40) ^ (keyword_id + 1)).
 *
 * Built into tools/libcpuloop.so by __graft_entry__.build(). */
#include "acm.h"

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

static inline uint64_t
splitmix64 (uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct job {
  ACMachine *machine;
  const unsigned char *text;
  uint64_t warm, begin, end;
  size_t sym_size;
  uint64_t found, digest;
};

static void *
worker (void *arg) {
  struct job *j = arg;
  const ACState *cursor = acm_initiate (j->machine);
  MatchHolder h;
  acm_matcher_init (&h);
  for (uint64_t i = j->warm; i < j->end; i++) {
    const size_t nb = acm_match (&cursor, j->text + i * j->sym_size);
    if (!nb || i < j->begin)
      continue;
    for (size_t k = 0; k < nb; k++) {
      acm_get_match (cursor, k, &h);
      j->digest += splitmix64 ((i * 1315423911ull) ^ ((uint64_t)h.length << 40) ^ (uint64_t)(uintptr_t)h.value);
      j->found++;
    }
  }
  acm_matcher_release (&h);
  return 0;
}

/* returns the number of matches; *digest as above */
uint64_t
acm_cpu_loop (ACMachine *machine, const void *text, uint64_t n, size_t sym_size, size_t lmax, int threads, uint64_t *digest) {
  if (threads < 1)
    threads = 1;
  struct job *jobs = calloc ((size_t)threads, sizeof *jobs);
  pthread_t *tid = calloc ((size_t)threads, sizeof *tid);
  if (!jobs || !tid)
    abort ();
  const uint64_t overlap = lmax ? lmax - 1 : 0;
  for (int t = 0; t < threads; t++) {
    jobs[t].machine = machine;
    jobs[t].text = text;
    jobs[t].sym_size = sym_size;
    jobs[t].begin = n * (uint64_t)t / (uint64_t)threads;
    jobs[t].end = n * (uint64_t)(t + 1) / (uint64_t)threads;
    jobs[t].warm = jobs[t].begin > overlap ? jobs[t].begin - overlap : 0;
    if (threads == 1)
      worker (&jobs[t]);
    else if (pthread_create (&tid[t], 0, worker, &jobs[t]) != 0)
      abort ();
  }
  uint64_t found = 0, d = 0;
  for (int t = 0; t < threads; t++) {
    if (threads > 1)
      pthread_join (tid[t], 0);
    found += jobs[t].found;
    d += jobs[t].digest;
  }
  free (jobs);
  free (tid);
  if (digest)
    *digest = d;
  return found;
}
