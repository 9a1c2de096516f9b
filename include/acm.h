/*
 * acm.h -- C API of the MI355X-native Aho-Corasick engine: the drop-in boundary.
 *
 * Every declaration below has the same name, signature and meaning as the public API of
 * farhiongit/aho-corasick-1975, so a caller of that library can re-link against
 * libac75_amd.so without source changes (include/aho_corasick.h forwards here):
 *
 *   this header                      replaces reference declaration
 *   -------------------------------  ------------------------------------------
 *   MatchHolder                      aho_corasick.h:23-28
 *   ACState / ACMachine (opaque)     aho_corasick.h:30-31
 *   CMP_TYPE / DESTROY_TYPE          aho_corasick.h:33-34
 *   ACM_CMP_DEFAULT                  aho_corasick.h:35   (aho_corasick.c:134-138)
 *   acm_create                       aho_corasick.h:45   (aho_corasick.c:140-151)
 *   acm_initiate                     aho_corasick.h:48   (aho_corasick.c:161-165)
 *   acm_insert_letter_of_keyword     aho_corasick.h:53   (aho_corasick.c:291-316)
 *   acm_insert_end_of_keyword        aho_corasick.h:65   (aho_corasick.c:340-363)
 *   acm_match                        aho_corasick.h:70   (aho_corasick.c:434-448)
 *   acm_matcher_init                 aho_corasick.h:74   (aho_corasick.c:420-424)
 *   acm_get_match                    aho_corasick.h:81   (aho_corasick.c:451-482)
 *   acm_matcher_release              aho_corasick.h:84   (aho_corasick.c:426-431)
 *   acm_nb_keywords                  aho_corasick.h:87   (aho_corasick.c:484-488)
 *   acm_foreach_keyword              aho_corasick.h:90   (aho_corasick.c:521-531)
 *   acm_release                      aho_corasick.h:93   (aho_corasick.c:153-159)
 *   PRINT_TYPE / acm_print           aho_corasick.h:96-97 (aho_corasick.c:583-594)
 *   ACM_INCREMENTAL_STRING_MATCHING  aho_corasick.h:98   (aho_corasick.c:596-600)
 *
 * The per-symbol calls stay host-side, exactly as in the reference (a GPU cannot help a call
 * that hands over one symbol).  The accelerated path is the bulk scan declared in acm_gpu.h,
 * whose result is DEFINED as what the reference's caller loop (examples/test.c:17-23) produces.
 */
#ifndef ACM_H_AMD
#define ACM_H_AMD

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  const void **letters; /* letters[k] -> k-th symbol of the matched keyword, as stored in the dictionary */
  size_t length;        /* number of symbols (not bytes) */
  void *value;          /* value registered with the keyword, or NULL */
} MatchHolder;

typedef struct _ac_state ACState;     /* trie node == insertion cursor == scan cursor */
typedef struct _ac_machine ACMachine;

/* three-way comparison of two symbols; <0, 0, >0 like memcmp */
typedef int (*CMP_TYPE) (const void *letter_a, const void *letter_b, const void *eq_arg);
typedef void (*DESTROY_TYPE) (void *letter);

/* memcmp over *(size_t *)cmp_arg bytes.  The only comparator the GPU path accepts. */
extern const CMP_TYPE ACM_CMP_DEFAULT;

/* cmp is mandatory.  dtor (optional) is applied to every letter the machine no longer needs:
 * at once when the edge already exists, at acm_release for stored ones.  With dtor == NULL the
 * letters passed to acm_insert_letter_of_keyword must outlive the machine (only their addresses
 * are kept). */
ACMachine *acm_create (CMP_TYPE cmp, void *cmp_arg, DESTROY_TYPE dtor);

/* Root state: initial value of an insertion cursor and of a scan cursor. */
ACState *acm_initiate (ACMachine *machine);

/* Advances the insertion cursor by one symbol, creating the node if needed. */
void acm_insert_letter_of_keyword (ACState **state, void *letter);

/* Marks the node under the cursor as a keyword and resets the cursor to the root.  The first
 * non-NULL value registered for a keyword stays attached to it (with its dtor, run at
 * acm_release).  Returns the value already attached, or NULL if there was none -- a non-NULL
 * return means the value passed now was NOT taken over. */
void *acm_insert_end_of_keyword (ACState **state, void *value, void (*dtor) (void *));

/* One automaton step.  Returns how many keywords end at this symbol. */
size_t acm_match (const ACState **state, const void *letter);

void acm_matcher_init (MatchHolder *matcher);

/* index-th keyword ending at the cursor position, longest first; index < last acm_match result.
 * matcher may be NULL. */
void acm_get_match (const ACState *state, size_t index, MatchHolder *matcher);

void acm_matcher_release (MatchHolder *matcher);

size_t acm_nb_keywords (const ACMachine *machine);

/* Calls operator once per keyword, in comparator order of the trie. */
void acm_foreach_keyword (const ACMachine *machine, void (*operator_) (MatchHolder));

void acm_release (ACMachine *machine);

typedef int (*PRINT_TYPE) (FILE *, const void *letter);
/* ASCII drawing of the goto tree with output counts and failure links. */
void acm_print (ACMachine *machine, FILE *stream, PRINT_TYPE printer);

/* 1: failure links are maintained incrementally at every insertion (Meyer 1985). */
extern const int ACM_INCREMENTAL_STRING_MATCHING;

#ifdef __cplusplus
}
#endif
#endif
