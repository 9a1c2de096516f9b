/*
 * ac_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the scan path of farhiongit/aho-corasick-1975
 * (reference: aho_corasick.h:23-98, aho_corasick.c:39-82,140-482).  It exists to CHECK the
 * product (the HIP scan engine behind include/acm.h + include/acm_gpu.h); nothing under
 * aho-corasick-1975_amd/ may include, link or call it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it.
 *
 * Parity pin: the reference itself is NOT buildable in this image (aho_corasick.c:18 needs
 * map.h from farhiongit/minimaps, which is neither vendored nor pinned: Makefile:3,
 * examples/Makefile:4,20,24).  This restatement is therefore pinned by the reference's own
 * golden output and asserts: README.md:92-93 (stdout of examples/test.c) and
 * examples/aho_corasick_generic_test.c:70,73-99,114,117,211 -- see tests/test_oracle_kat.py.
 *
 * All symbols are prefixed orc_ so the oracle and the product can live in one process.
 */
#ifndef AC_ORACLE_H
#define AC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: aho_corasick.h:23-28 */
typedef struct {
  const void **letters;
  size_t length;
  void *value;
} orc_holder;

typedef struct orc_state orc_state;
typedef struct orc_machine orc_machine;

/* reference: aho_corasick.h:33-34 */
typedef int (*orc_cmp_fn) (const void *a, const void *b, const void *arg);
typedef void (*orc_dtor_fn) (void *);

/* reference: aho_corasick.c:134-138 (memcmp over *(size_t*)arg bytes) */
int orc_cmp_default (const void *a, const void *b, const void *arg);

/* Construction variant, the reference's compile-time NMEYER_85 switch made a run-time one:
 * ORC_MEYER85 = default build (incremental failure maintenance, aho_corasick.c:210-240,253-265,318-338)
 * ORC_AC75    = -DNMEYER_85 build (lazy BFS rebuild, aho_corasick.c:365-418,443-446). */
enum { ORC_MEYER85 = 0, ORC_AC75 = 1 };

orc_machine *orc_create (orc_cmp_fn cmp, void *cmp_arg, orc_dtor_fn dtor, int variant);
void orc_release (orc_machine *m);
orc_state *orc_initiate (orc_machine *m);
void orc_insert_letter_of_keyword (orc_state **cursor, void *letter);
void *orc_insert_end_of_keyword (orc_state **cursor, void *value, orc_dtor_fn dtor);
size_t orc_match (const orc_state **cursor, const void *letter);
void orc_matcher_init (orc_holder *h);
void orc_get_match (const orc_state *state, size_t index, orc_holder *h);
void orc_matcher_release (orc_holder *h);
size_t orc_nb_keywords (const orc_machine *m);
void orc_foreach_keyword (const orc_machine *m, void (*op) (orc_holder));

/* --- introspection used by the table-parity tests (no counterpart in the reference API;
 *     the fields are the reference's struct _ac_state members, aho_corasick.c:44-65) --- */
size_t orc_nb_states (const orc_machine *m);
size_t orc_state_id (const orc_state *s);            /* creation order, :61,101 */
const orc_state *orc_state_fail (const orc_state *s); /* :53 */
size_t orc_state_nb_outputs (const orc_state *s);    /* :55 */
int orc_state_is_end (const orc_state *s);            /* :54 */
size_t orc_state_depth (const orc_state *s);          /* number of previous links, :473 */

/* --- the caller loop (reference: examples/test.c:17-23, generic_test.c:139-158) as a bulk
 *     helper.  Record = SURVEY.md section 8 canonical record. --- */
typedef struct {
  uint64_t end_pos;    /* index i of the symbol whose orc_match call reported the match */
  uint32_t length;     /* holder.length */
  uint32_t keyword_id; /* (uintptr_t)holder.value - 1: callers register value = id + 1 */
} orc_record;

/* Runs `for i: nb = orc_match(); for j < nb: orc_get_match(j)` over text[0..n) starting from
 * orc_initiate(m); appends records in canonical order (end_pos asc, index asc).  Records with
 * end_pos < emit_from are dropped (warm-up for sharded scans).  end_pos is reported as
 * pos_base + i.  Returns the number of records found (may exceed cap; only cap are stored). */
uint64_t orc_scan (orc_machine *m, const void *text, uint64_t n, size_t sym_size,
                   uint64_t pos_base, uint64_t emit_from, orc_record *out, uint64_t cap);

/* Sum of orc_match return values only (reference: generic_test.c:272-273). */
uint64_t orc_count (orc_machine *m, const void *text, uint64_t n, size_t sym_size);

/* Order-independent digest of a record set: sum over records, mod 2^64, of
 * splitmix64((end_pos * 1315423911) ^ ((uint64_t)length << 40) ^ (keyword_id + 1)). */
uint64_t orc_digest (const orc_record *r, uint64_t n);

/* T threads, one cursor each over one shared machine (reference threading model,
 * README.md:364), contiguous shards with lmax-1 symbols of warm-up.  Returns total matches,
 * writes the digest; does not store records.  Used by bench.py's cpu_baseline leg. */
uint64_t orc_scan_mt (orc_machine *m, const void *text, uint64_t n, size_t sym_size,
                      size_t lmax, int threads, uint64_t *digest);
/* the same over a piece of a longer text: global positions in the digest, nothing reported before emit_from */
uint64_t orc_scan_mt_at (orc_machine *m, const void *text, uint64_t n, size_t sym_size, size_t lmax,
                         int threads, uint64_t pos_base, uint64_t emit_from, uint64_t *digest);

#ifdef __cplusplus
}
#endif
#endif
