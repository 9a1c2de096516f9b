/* Internal interface between the host trie (acm_host.c), the flattener (acm_flat.c) and the
 * device side (acm_gpu.hip).  Not installed. */
#ifndef ACM_INTERNAL_H
#define ACM_INTERNAL_H

#include "acm.h"
#include "acm_gpu.h"
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Goto edges of a state: children ordered by the machine comparator on ->letter.  One block that
 * is replaced (never freed before acm_release) when it outgrows its capacity, so that a reader
 * that runs beside an insertion (acm_match is not locked, as in the reference: README.md:364)
 * never touches freed memory; `kver` of the owning state is a sequence lock around in-place
 * insertions (odd while one is under way). */
struct _ac_kidvec {
  uint32_t n, cap;
  struct _ac_state *v[];
};
/* for code that holds the machine lock (writers, the flattener) */
#define ACM_NKIDS(s) ((s)->kids ? (s)->kids->n : 0u)
#define ACM_KID(s, i) ((s)->kids->v[i])

struct _ac_state {
  ACMachine *machine;
  struct _ac_state *parent; /* reference: previous.state, aho_corasick.c:49-52 */
  void *letter;             /* symbol on the edge parent -> this, as handed in by the caller */
  struct _ac_state *fail;   /* f(s); NULL for the root only */
  struct _ac_kidvec *kids;  /* goto edges (NULL: none) */
  struct _ac_state **inv;   /* inverse failure set { x : f(x) == this }, unordered; writers only */
  uint32_t kver;            /* sequence lock of `kids` */
  uint32_t ninv, capinv;
  uint32_t inv_slot;        /* position of this state inside fail->inv (O(1) removal) */
  uint32_t depth;
  uint32_t nb_outputs;      /* keywords ending here = terminal + nb_outputs(fail) */
  uint32_t rank;            /* keyword_id when terminal */
  uint32_t id;              /* creation order, printed by acm_print */
  int terminal;
  void *value;
  void (*value_dtor) (void *);
};

/* dictionary generation counter: bumped whenever a node or a keyword is added; a cached device
 * plan built at another generation is stale. */
uint64_t acm_internal_generation (const ACMachine *m);
ACState *acm_internal_root (const ACMachine *m);
uint32_t acm_internal_nb_states (const ACMachine *m);
/* 0 if the machine uses ACM_CMP_DEFAULT over 1, 2 or 4 byte symbols, else ACM_GPU_E_INELIGIBLE */
int acm_internal_symbol_bytes (const ACMachine *m, uint32_t *sym_bytes);
void acm_internal_comparator (const ACMachine *m, CMP_TYPE *cmp, void **cmp_arg);
/* acm_set_symbol_bytes' value (0: none); the caller loop on the host for machines the GPU cannot take; acm_scan_path's value */
uint32_t acm_internal_declared_symbol_bytes (const ACMachine *m);
int acm_internal_cpu_scan (ACMachine *m, const void *text, uint64_t n_symbols, uint32_t sym_bytes, ACMRecord *records, uint64_t capacity,
                           uint64_t *n_found);
void acm_internal_set_scan_path (ACMachine *m, int path);
/* ACM_NMEYER_85 builds: brings failure links and output counts up to date (no-op otherwise);
 * takes the machine lock itself */
void acm_internal_refresh (ACMachine *m);
void acm_internal_lock (ACMachine *m);
void acm_internal_unlock (ACMachine *m);
/* acm_get_keyword for callers that already hold the machine lock */
int acm_internal_get_keyword (const ACMachine *m, uint32_t keyword_id, MatchHolder *matcher);
/* serialises the users of the machine's cached device plan (acm_scan) */
void acm_internal_plan_lock (ACMachine *m);
void acm_internal_plan_unlock (ACMachine *m);
/* per-machine slot for the cached device plan (owned by acm_gpu.hip) */
void **acm_internal_plan_slot (ACMachine *m);
/* set by the device side once; called by acm_release to drop the cached plan */
extern void (*acm_internal_plan_dropper) (void *plan);

#ifdef __cplusplus
}
#endif
#endif
