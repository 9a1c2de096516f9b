/*
 * acm_host.c -- host side of the drop-in: the acm_* API of include/acm.h.
 *
 * Own design, same observable behaviour as /root/reference/aho_corasick.c:
 *   - states live in slab arenas (stable addresses: an ACState* is the caller's cursor);
 *   - a state's goto edges are a comparator-ordered vector of child pointers (each child carries
 *     the letter of its incoming edge), searched by bisection -- the role minimaps' ordered map
 *     plays at aho_corasick.c:175,299;
 *   - failure links, output counts and the inverse failure sets are kept exact after EVERY
 *     inserted symbol (Meyer 1985, the reference's default build, aho_corasick.c:194-267,318-363),
 *     so matching and insertion interleave freely (generic_test.c:198-229);
 *   - every terminal state also records its first-insertion rank (keyword_id of acm_gpu.h) and
 *     every state its depth, which the flattener (acm_flat.c) ships to the GPU.
 *
 * -DACM_NMEYER_85 (the reference's -DNMEYER_85 build, aho_corasick.c:365-418,443-446; Makefile
 * target `nmeyer85` -> libac75_amd_nmeyer85.so): no incremental maintenance; inserting only marks
 * the failure function stale, and the next acm_match (or flatten, or acm_print) recomputes it for
 * the whole trie by the breadth-first pass of Aho & Corasick 1975, algorithm 3.  The automaton is
 * the same, so the flat tables and every scan are too (tests/test_nmeyer85.py).
 */
#define _GNU_SOURCE
#include "acm_internal.h"

#include <stdlib.h>
#include <string.h>
#include <threads.h>

/* reference error convention, aho_corasick.c:24-36: message on stderr, the calling THREAD exits. */
#define ACM_REQUIRE(cond, msg)                                                                      \
  do {                                                                                              \
    if (!(cond)) {                                                                                  \
      fflush (stdout);                                                                              \
      fprintf (stderr, "FATAL ERROR: A prerequisite is not fulfilled in function %s.\n", __func__); \
      fprintf (stderr, "             %s\n", (msg)[0] ? (msg) : "The condition (" #cond ") is false."); \
      thrd_exit (EXIT_FAILURE);                                                                     \
    }                                                                                               \
  } while (0)

#define SLAB_STATES 4096

struct slab {
  struct slab *next;
  uint32_t used;
  struct _ac_state states[SLAB_STATES];
};

/* an outgrown block kept until acm_release (blocks double, so all of them together are smaller
 * than the live ones) */
struct garbage {
  struct garbage *next;
  void *block;
};

struct _ac_machine {
  struct _ac_state *root;
  size_t nb_keywords;
  uint32_t nb_states;
  uint64_t generation;
  CMP_TYPE cmp;
  void *cmp_arg;
  DESTROY_TYPE letter_dtor;
  struct slab *slabs;
  mtx_t lock;
  mtx_t plan_lock; /* users of the cached device plan (acm_scan), one at a time */
  void *plan; /* cached device plan, see acm_gpu.hip */
  struct garbage *garbage; /* outgrown edge blocks: lock-free readers may still hold them */
  struct _ac_state **keywords; /* keyword_id -> terminal state (acm_get_keyword) */
  size_t keywords_cap;
  int stale; /* ACM_NMEYER_85: failure links and output counts await the breadth-first pass */
  uint32_t declared_sym_bytes; /* acm_set_symbol_bytes: symbol size of a machine with a comparator of its own (0: not said) */
  int scan_path;               /* acm_scan_path: what the last acm_scan on this machine ran */
};

void (*acm_internal_plan_dropper) (void *plan) = 0;

/* ------------------------------------------------------------------ default comparator */
static int
cmp_bytes (const void *a, const void *b, const void *arg) { /* reference :134-138 */
  return memcmp (a, b, *(const size_t *)arg);
}
const CMP_TYPE ACM_CMP_DEFAULT = cmp_bytes;
#ifdef ACM_NMEYER_85
const int ACM_INCREMENTAL_STRING_MATCHING = 0; /* reference :596-597 */
#else
const int ACM_INCREMENTAL_STRING_MATCHING = 1; /* reference :596-597 (default build) */
#endif

/* ------------------------------------------------------------------ internal accessors */
uint64_t
acm_internal_generation (const ACMachine *m) {
  return m->generation;
}
ACState *
acm_internal_root (const ACMachine *m) {
  return m->root;
}
uint32_t
acm_internal_nb_states (const ACMachine *m) {
  return m->nb_states;
}
int
acm_internal_symbol_bytes (const ACMachine *m, uint32_t *sym_bytes) {
  if (m->cmp != ACM_CMP_DEFAULT || !m->cmp_arg)
    return ACM_GPU_E_INELIGIBLE;
  size_t sz = *(const size_t *)m->cmp_arg;
  if (sz != 1 && sz != 2 && sz != 4 && sz != 8)
    return ACM_GPU_E_INELIGIBLE;
  *sym_bytes = (uint32_t)sz;
  return ACM_GPU_OK;
}
/* include/acm_gpu.h: the symbol size of a machine whose comparator is not ACM_CMP_DEFAULT (the
 * library cannot know it: letters are opaque pointers, aho_corasick.h:33-43) -- what acm_scan
 * needs to step through a buffer */
int
acm_set_symbol_bytes (ACMachine *machine, uint32_t sym_bytes) {
  if (!machine || sym_bytes == 0 || sym_bytes > 4096)
    return ACM_GPU_E_ARG;
  uint32_t own = 0;
  if (acm_internal_symbol_bytes (machine, &own) == ACM_GPU_OK && own != sym_bytes)
    return ACM_GPU_E_ARG; /* ACM_CMP_DEFAULT says it itself */
  machine->declared_sym_bytes = sym_bytes;
  return ACM_GPU_OK;
}
uint32_t
acm_internal_declared_symbol_bytes (const ACMachine *m) {
  return m->declared_sym_bytes;
}
int
acm_scan_path (const ACMachine *machine) {
  return machine ? machine->scan_path : ACM_SCAN_PATH_NONE;
}
void
acm_internal_set_scan_path (ACMachine *m, int path) {
  m->scan_path = path;
}

void
acm_internal_comparator (const ACMachine *m, CMP_TYPE *cmp, void **cmp_arg) {
  *cmp = m->cmp;
  *cmp_arg = m->cmp_arg;
}
void
acm_internal_lock (ACMachine *m) {
  ACM_REQUIRE (mtx_lock (&m->lock) == thrd_success, "");
}
void
acm_internal_unlock (ACMachine *m) {
  ACM_REQUIRE (mtx_unlock (&m->lock) == thrd_success, "");
}
void
acm_internal_plan_lock (ACMachine *m) {
  ACM_REQUIRE (mtx_lock (&m->plan_lock) == thrd_success, "");
}
void
acm_internal_plan_unlock (ACMachine *m) {
  ACM_REQUIRE (mtx_unlock (&m->plan_lock) == thrd_success, "");
}
void **
acm_internal_plan_slot (ACMachine *m) {
  return &m->plan;
}

/* Readers (acm_match, acm_get_match) take no lock -- the reference's threading model: many
 * threads, one shared machine, one cursor per thread, insertions in between (README.md:364,
 * aho_corasick.c:81).  Writers hold the machine lock and publish with release stores; readers use
 * acquire loads of the fields a writer may change under them (fail, nb_outputs, terminal, the
 * edge block and its sequence lock). */
#define LOAD(p) __atomic_load_n ((p), __ATOMIC_ACQUIRE)
#define STORE(p, v) __atomic_store_n ((p), (v), __ATOMIC_RELEASE)

/* ------------------------------------------------------------------ states */
static struct _ac_state *
state_alloc (ACMachine *m) {
  struct slab *sl = m->slabs;
  if (!sl || sl->used == SLAB_STATES) {
    sl = malloc (sizeof *sl);
    ACM_REQUIRE (sl, "Out of memory.");
    sl->next = m->slabs;
    sl->used = 0;
    m->slabs = sl;
  }
  struct _ac_state *s = &sl->states[sl->used++];
  memset (s, 0, sizeof *s);
  s->machine = m;
  s->id = m->nb_states++;
  s->rank = UINT32_MAX;
  return s;
}

/* bisection among the children of s; *at = where a missing letter would be inserted.
 * Safe beside a writer: the snapshot is retried if an insertion into this very state ran
 * meanwhile (sequence lock), and a replaced block stays allocated. */
static inline struct _ac_state *
child_find (const struct _ac_state *s, const void *letter, uint32_t *at) {
  const ACMachine *m = s->machine;
  for (;;) {
    const uint32_t v0 = LOAD (&s->kver);
    if (v0 & 1u)
      continue; /* an insertion is shifting the entries right now */
    const struct _ac_kidvec *kv = LOAD (&s->kids);
    uint32_t lo = 0, hi = kv ? __atomic_load_n (&kv->n, __ATOMIC_RELAXED) : 0;
    struct _ac_state *found = 0;
    while (lo < hi) {
      uint32_t mid = lo + (hi - lo) / 2;
      struct _ac_state *k = LOAD (&kv->v[mid]);
      int c = m->cmp (letter, k->letter, m->cmp_arg);
      if (c == 0) {
        found = k;
        break;
      }
      if (c < 0)
        hi = mid;
      else
        lo = mid + 1;
    }
    __atomic_thread_fence (__ATOMIC_ACQUIRE);
    if (__atomic_load_n (&s->kver, __ATOMIC_RELAXED) != v0)
      continue;
    if (!found && at)
      *at = lo;
    return found;
  }
}

/* writer, machine lock held: child k goes to position `at` of n's edges */
static void
child_insert (struct _ac_state *n, uint32_t at, struct _ac_state *k) {
  ACMachine *m = n->machine;
  struct _ac_kidvec *kv = n->kids;
  const uint32_t cnt = kv ? kv->n : 0;
  if (!kv || cnt == kv->cap) {
    /* a fresh block, published whole; the old one goes to the machine's garbage */
    const uint32_t cap = cnt ? 2 * cnt : 2;
    struct _ac_kidvec *nv = malloc (sizeof *nv + cap * sizeof nv->v[0]);
    struct garbage *g = kv ? malloc (sizeof *g) : 0;
    ACM_REQUIRE (nv && (g || !kv), "Out of memory.");
    nv->cap = cap;
    nv->n = cnt + 1;
    if (at)
      memcpy (nv->v, kv->v, at * sizeof nv->v[0]);
    nv->v[at] = k;
    if (cnt > at)
      memcpy (nv->v + at + 1, kv->v + at, (cnt - at) * sizeof nv->v[0]);
    STORE (&n->kids, nv);
    if (kv) {
      g->block = kv;
      g->next = m->garbage;
      m->garbage = g;
    }
    return;
  }
  __atomic_store_n (&n->kver, n->kver + 1, __ATOMIC_RELAXED); /* odd: readers wait */
  __atomic_thread_fence (__ATOMIC_RELEASE);
  for (uint32_t i = cnt; i > at; i--)
    __atomic_store_n (&kv->v[i], kv->v[i - 1], __ATOMIC_RELAXED);
  STORE (&kv->v[at], k);
  __atomic_store_n (&kv->n, cnt + 1, __ATOMIC_RELAXED);
  STORE (&n->kver, n->kver + 1);
}

/* delta(s, letter) of the automaton WITHOUT root self-loops: goto if defined, else down the
 * failure chain; a miss at the root stays there (reference state_goto, :167-192). */
static inline const struct _ac_state *
automaton_step (const struct _ac_state *s, const void *letter) {
  for (;;) {
    const struct _ac_state *k = child_find (s, letter, 0);
    if (k)
      return k;
    const struct _ac_state *f = LOAD (&s->fail);
    if (!f) /* only the root has no failure link */
      return s;
    s = f;
  }
}

static void
inv_add (struct _ac_state *owner, struct _ac_state *x) {
  if (owner->ninv == owner->capinv) {
    owner->capinv = owner->capinv ? 2 * owner->capinv : 4;
    owner->inv = realloc (owner->inv, owner->capinv * sizeof *owner->inv);
    ACM_REQUIRE (owner->inv, "Out of memory.");
  }
  x->inv_slot = owner->ninv;
  owner->inv[owner->ninv++] = x;
}

static void
inv_del (struct _ac_state *owner, struct _ac_state *x) {
  struct _ac_state *last = owner->inv[--owner->ninv];
  owner->inv[x->inv_slot] = last;
  last->inv_slot = x->inv_slot;
}

/* explicit stack for walks over the failure tree (depth can reach the number of states) */
struct walk {
  struct _ac_state **v;
  size_t n, cap;
};
static inline void
walk_push (struct walk *w, struct _ac_state *s) {
  if (w->n == w->cap) {
    w->cap = w->cap ? 2 * w->cap : 64;
    w->v = realloc (w->v, w->cap * sizeof *w->v);
    ACM_REQUIRE (w->v, "Out of memory.");
  }
  w->v[w->n++] = s;
}

/* New leaf `leaf` = child of n on letter c, about to be linked into the goto tree (it becomes
 * reachable only afterwards, complete: child_insert).
 * Failure maintenance (Meyer 1985; reference :194-208, :211-239, :253-265):
 *   f(leaf) = delta(f(n), c), or the root when n is the root;
 *   every existing node x.c whose longest proper suffix in the trie has just become `leaf` is
 *   re-pointed: those are the c-children of the nodes x met by a walk down the failure tree from
 *   n that stops at the first node owning a c-child on each branch.
 * The walk runs on the tree as it was BEFORE any re-pointing (targets are collected first), so
 * it does not depend on container mutation order. */
#ifdef ACM_NMEYER_85
__attribute__ ((unused))
#endif
static void
link_failure_of_new_leaf (struct _ac_state *n, struct _ac_state *leaf) {
  if (n->fail)
    leaf->fail = (struct _ac_state *)automaton_step (n->fail, leaf->letter);
  else
    leaf->fail = n; /* depth-1 states fail to the root */
  leaf->nb_outputs = leaf->fail->nb_outputs; /* leaf is not (yet) a keyword end */

  if (n->ninv) {
    struct walk todo = { 0 }, hits = { 0 };
    for (uint32_t i = 0; i < n->ninv; i++)
      walk_push (&todo, n->inv[i]);
    while (todo.n) {
      struct _ac_state *x = todo.v[--todo.n];
      struct _ac_state *xc = child_find (x, leaf->letter, 0);
      if (xc)
        walk_push (&hits, xc);
      else
        for (uint32_t i = 0; i < x->ninv; i++)
          walk_push (&todo, x->inv[i]);
    }
    for (size_t i = 0; i < hits.n; i++) {
      struct _ac_state *xc = hits.v[i];
      /* old f(xc) == f(leaf): both are the longest suffix shorter than leaf, so nb_outputs(xc)
       * is unchanged by the re-pointing. */
      inv_del (xc->fail, xc);
      STORE (&xc->fail, leaf);
      inv_add (leaf, xc);
    }
    free (todo.v);
    free (hits.v);
  }
  inv_add (leaf->fail, leaf);
}

/* AC-75, algorithm 3 (reference :365-418): failure links and output counts of the whole trie in
 * one breadth-first pass -- f of a depth-1 state is the root; f(child of u on a) = the first goto
 * on a met down u's failure chain, else the root; nb_outputs(s) = [s terminal] + nb_outputs(f(s)).
 * Does nothing in the default (Meyer-85) build, whose links are always current. */
void
acm_internal_refresh (ACMachine *m) {
#ifdef ACM_NMEYER_85
  if (!LOAD (&m->stale))
    return;
  ACM_REQUIRE (mtx_lock (&m->lock) == thrd_success, "");
  if (m->stale) {
    struct _ac_state **queue = malloc ((size_t)m->nb_states * sizeof *queue);
    ACM_REQUIRE (queue, "Out of memory.");
    size_t head = 0, tail = 0;
    queue[tail++] = m->root;
    while (head < tail) {
      struct _ac_state *u = queue[head++];
      for (uint32_t i = 0; i < ACM_NKIDS (u); i++) {
        struct _ac_state *c = ACM_KID (u, i), *target = m->root;
        for (const struct _ac_state *v = u->fail; v; v = v->fail) {
          struct _ac_state *t = child_find (v, c->letter, 0);
          if (t) {
            target = t;
            break;
          }
        }
        STORE (&c->fail, target);
        STORE (&c->nb_outputs, (uint32_t)(c->terminal ? 1 : 0) + target->nb_outputs);
        queue[tail++] = c;
      }
    }
    free (queue);
    STORE (&m->stale, 0);
  }
  ACM_REQUIRE (mtx_unlock (&m->lock) == thrd_success, "");
#else
  (void)m;
#endif
}

/* ------------------------------------------------------------------ public API */
ACMachine *
acm_create (CMP_TYPE cmp, void *cmp_arg, DESTROY_TYPE dtor) {
  ACM_REQUIRE (cmp, "A comparison function should be provided.");
  ACMachine *m = calloc (1, sizeof *m);
  ACM_REQUIRE (m, "Out of memory.");
  m->cmp = cmp;
  m->cmp_arg = cmp_arg;
  m->letter_dtor = dtor;
  m->root = state_alloc (m);
  ACM_REQUIRE (mtx_init (&m->lock, mtx_plain) == thrd_success, "Out of memory.");
  ACM_REQUIRE (mtx_init (&m->plan_lock, mtx_plain) == thrd_success, "Out of memory.");
  return m;
}

void
acm_release (ACMachine *machine) {
  ACM_REQUIRE (machine, "Invalid null machine.");
  if (machine->plan && acm_internal_plan_dropper)
    acm_internal_plan_dropper (machine->plan);
  for (struct slab *sl = machine->slabs; sl;) {
    for (uint32_t i = 0; i < sl->used; i++) {
      struct _ac_state *s = &sl->states[i];
      if (s->parent && machine->letter_dtor) /* stored letters, reference :111-112 */
        machine->letter_dtor (s->letter);
      if (s->value_dtor) /* reference :124-125 */
        s->value_dtor (s->value);
      free (s->kids);
      free (s->inv);
    }
    struct slab *next = sl->next;
    free (sl);
    sl = next;
  }
  for (struct garbage *g = machine->garbage; g;) {
    struct garbage *next = g->next;
    free (g->block);
    free (g);
    g = next;
  }
  mtx_destroy (&machine->lock);
  mtx_destroy (&machine->plan_lock);
  free (machine->keywords);
  free (machine);
}

ACState *
acm_initiate (ACMachine *machine) {
  ACM_REQUIRE (machine, "Invalid null machine.");
  return machine->root;
}

void
acm_insert_letter_of_keyword (ACState **state, void *letter) {
  ACM_REQUIRE (state && *state && letter, "Invalid null state or letter.");
  struct _ac_state *n = *state;
  ACMachine *m = n->machine;
  ACM_REQUIRE (mtx_lock (&m->lock) == thrd_success, "");
  uint32_t at = 0;
  struct _ac_state *k = child_find (n, letter, &at);
  if (k) {
    if (m->letter_dtor) /* the edge exists: this copy of the letter is not kept, reference :306-307 */
      m->letter_dtor (letter);
  } else {
    k = state_alloc (m);
    k->parent = n;
    k->letter = letter;
    k->depth = n->depth + 1;
#ifdef ACM_NMEYER_85
    k->fail = m->root; /* provisional (only the root may have none): acm_internal_refresh sets it */
    m->stale = 1;
#else
    link_failure_of_new_leaf (n, k);
#endif
    child_insert (n, at, k);
    m->generation++;
  }
  *state = k;
  ACM_REQUIRE (mtx_unlock (&m->lock) == thrd_success, "");
}

void *
acm_insert_end_of_keyword (ACState **state, void *value, void (*dtor) (void *)) {
  ACM_REQUIRE (state && *state, "Invalid null state.");
  struct _ac_state *n = *state;
  ACMachine *m = n->machine;
  ACM_REQUIRE (mtx_lock (&m->lock) == thrd_success, "");
  ACM_REQUIRE (n != m->root, "acm_insert_letter_of_keyword should be called first.");
  void *previous = n->value;
  if (!previous) { /* first non-NULL value wins, reference :357-359 */
    n->value_dtor = dtor;
    if (value)
      STORE (&n->value, value);
  }
  if (!n->terminal) {
    if (m->nb_keywords == m->keywords_cap) {
      m->keywords_cap = m->keywords_cap ? 2 * m->keywords_cap : 64;
      m->keywords = realloc (m->keywords, m->keywords_cap * sizeof *m->keywords);
      ACM_REQUIRE (m->keywords, "Out of memory.");
    }
    m->keywords[m->nb_keywords] = n;
    n->rank = (uint32_t)m->nb_keywords++;
    /* the terminal mark first, then the counts: a reader that sees a count sees the terminal
     * states acm_get_match will look for down the failure chain */
    STORE (&n->terminal, 1);
#ifdef ACM_NMEYER_85
    m->stale = 1; /* the output counts come with the next breadth-first pass */
    if (0) {
#else
    {
#endif
    /* one more keyword ends at n and at every state that has n as a suffix, i.e. the whole
     * failure subtree of n (reference enter_output, :330-338) */
    struct walk todo = { 0 };
    walk_push (&todo, n);
    while (todo.n) {
      struct _ac_state *x = todo.v[--todo.n];
      STORE (&x->nb_outputs, x->nb_outputs + 1);
      for (uint32_t i = 0; i < x->ninv; i++)
        walk_push (&todo, x->inv[i]);
    }
    free (todo.v);
    }
    m->generation++;
  }
  *state = m->root;
  ACM_REQUIRE (mtx_unlock (&m->lock) == thrd_success, "");
  return previous;
}

size_t
acm_match (const ACState **state, const void *letter) {
  ACM_REQUIRE (state && *state && letter, "Invalid null state or letter.");
#ifdef ACM_NMEYER_85
  acm_internal_refresh ((*state)->machine); /* reference :443-446 */
#endif
  return LOAD (&(*state = automaton_step (*state, letter))->nb_outputs);
}

void
acm_matcher_init (MatchHolder *matcher) {
  ACM_REQUIRE (matcher, "Invalid null matcher.");
  matcher->letters = 0;
  matcher->length = 0;
  matcher->value = 0;
}

void
acm_matcher_release (MatchHolder *matcher) {
  ACM_REQUIRE (matcher, "Invalid null matcher.");
  free (matcher->letters);
  acm_matcher_init (matcher);
}

void
acm_get_match (const ACState *state, size_t index, MatchHolder *matcher) {
  ACM_REQUIRE (state, "Invalid null state.");
  ACM_REQUIRE (state->parent, "acm_match should be called first and acm_matcher_init called on the MatchHolder.");
  ACM_REQUIRE (index < LOAD (&state->nb_outputs), "Index out of bounds.");
  /* index-th keyword-terminal state along the failure chain, nearest (= longest) first
   * (reference :459-466) */
  const struct _ac_state *t = state;
  for (size_t seen = 0;; t = LOAD (&t->fail)) {
    if (LOAD (&t->terminal) && seen++ == index)
      break;
  }
  if (!matcher)
    return;
  matcher->length = t->depth;
  matcher->letters = realloc (matcher->letters, matcher->length * sizeof *matcher->letters);
  ACM_REQUIRE (matcher->letters || !matcher->length, "Out of memory.");
  size_t k = matcher->length;
  for (const struct _ac_state *s = t; s->parent; s = s->parent)
    matcher->letters[--k] = s->letter;
  matcher->value = LOAD (&t->value);
}

/* What acm_get_match would have put into the holder for a record of the bulk scan: the
 * dictionary's letters, the length and the value of keyword `keyword_id` (include/acm_gpu.h). */
int
acm_get_keyword (const ACMachine *machine, uint32_t keyword_id, MatchHolder *matcher) {
  if (!machine || !matcher)
    return ACM_GPU_E_ARG;
  /* the keyword table may be moved by a concurrent acm_insert_end_of_keyword */
  ACMachine *m = (ACMachine *)machine;
  acm_internal_lock (m);
  const int rc = acm_internal_get_keyword (machine, keyword_id, matcher);
  acm_internal_unlock (m);
  return rc;
}

int
acm_internal_get_keyword (const ACMachine *machine, uint32_t keyword_id, MatchHolder *matcher) {
  if (!machine || !matcher || keyword_id >= machine->nb_keywords)
    return ACM_GPU_E_ARG;
  const struct _ac_state *t = machine->keywords[keyword_id];
  matcher->length = t->depth;
  matcher->letters = realloc (matcher->letters, matcher->length * sizeof *matcher->letters);
  if (!matcher->letters && matcher->length)
    return ACM_GPU_E_NOMEM;
  size_t k = matcher->length;
  for (const struct _ac_state *s = t; s->parent; s = s->parent)
    matcher->letters[--k] = s->letter;
  matcher->value = t->value;
  return ACM_GPU_OK;
}

size_t
acm_nb_keywords (const ACMachine *machine) {
  ACM_REQUIRE (machine, "Invalid null machine.");
  return machine->nb_keywords;
}

/* ------------------------------------------------------------------ enumeration / debug */
struct dfs_frame {
  const struct _ac_state *s;
  uint32_t next_kid;
};

void
acm_foreach_keyword (const ACMachine *machine, void (*operator_) (MatchHolder)) {
  ACM_REQUIRE (machine, "Invalid null machine.");
  if (!operator_)
    return;
  /* iterative pre-order DFS in comparator order (reference :490-519) */
  size_t cap = 16, top = 0;
  struct dfs_frame *st = malloc (cap * sizeof *st);
  const void **letters = malloc (cap * sizeof *letters);
  ACM_REQUIRE (st && letters, "Out of memory.");
  st[top++] = (struct dfs_frame){ machine->root, 0 };
  while (top) {
    struct dfs_frame *f = &st[top - 1];
    if (f->next_kid == 0 && f->s->terminal && f->s->depth) {
      MatchHolder k = { .letters = letters, .length = f->s->depth, .value = f->s->value };
      operator_ (k);
    }
    if (f->next_kid < ACM_NKIDS (f->s)) {
      const struct _ac_state *kid = ACM_KID (f->s, f->next_kid++);
      if (top == cap) {
        cap *= 2;
        st = realloc (st, cap * sizeof *st);
        letters = realloc (letters, cap * sizeof *letters);
        ACM_REQUIRE (st && letters, "Out of memory.");
      }
      letters[kid->depth - 1] = kid->letter;
      st[top++] = (struct dfs_frame){ kid, 0 };
    } else
      top--;
  }
  free (st);
  free (letters);
}

/* Tree drawing, same text as the reference's (:541-594): one edge is
 *   ---<letter>-->(<id>)[+<outputs> if keyword end](v <fail id> if the failure link is not the root)
 * the root id is printed in front of each of its edges; the first child continues the line, the
 * next ones start a new line indented to their parent's column with an 'L' elbow. */
static void
print_subtree (const struct _ac_state *s, FILE *out, int *col, int indent, PRINT_TYPE printer) {
  for (uint32_t i = 0; i < ACM_NKIDS (s); i++) {
    const struct _ac_state *k = ACM_KID (s, i);
    if (indent < *col) {
      *col = 0;
      fprintf (out, "\n");
      if (indent) {
        for (int t = 0; t < indent - 1; t++)
          *col += fprintf (out, " ");
        *col += fprintf (out, "L");
      }
    } else
      while (*col < indent)
        *col += fprintf (out, " ");
    if (!s->parent)
      *col += fprintf (out, "(%03zu)", (size_t)s->id);
    *col += fprintf (out, "---");
    if (printer)
      *col += printer (out, k->letter);
    *col += fprintf (out, "-->(%03zu)", (size_t)k->id);
    if (k->terminal)
      *col += fprintf (out, "[+%zu]", (size_t)k->nb_outputs);
    if (k->fail != s->machine->root)
      *col += fprintf (out, "(v %03zu)", (size_t)k->fail->id);
    print_subtree (k, out, col, *col, printer);
  }
}

void
acm_print (ACMachine *machine, FILE *stream, PRINT_TYPE printer) {
  ACM_REQUIRE (machine, "Invalid null machine.");
  acm_internal_refresh (machine); /* reference :589: also with a null stream */
  if (!stream)
    return;
  int col = 0;
  fprintf (stream, "\n");
  print_subtree (machine->root, stream, &col, 0, printer);
  fprintf (stream, "\n");
}

/* The reference's caller loop (examples/test.c:17-23; acm_match aho_corasick.c:434-448, acm_get_match
 * :451-482) over a buffer of fixed-size symbols, with THIS library's own automaton step and failure
 * chain -- what acm_scan runs for a machine the GPU path cannot take (a comparator of its own over
 * symbols that are not 1, 2 or 4 bytes wide, or one that is no consistent order over all values:
 * SURVEY.md 8b).  Records in the loop's order; *n_found = their total, also beyond `capacity`. */
int
acm_internal_cpu_scan (ACMachine *m, const void *text, uint64_t n_symbols, uint32_t sym_bytes, ACMRecord *records, uint64_t capacity,
                       uint64_t *n_found) {
  if (!m || !n_found || (n_symbols && !text) || (capacity && !records) || !sym_bytes)
    return ACM_GPU_E_ARG;
#ifdef ACM_NMEYER_85
  acm_internal_refresh (m);
#endif
  const unsigned char *t = text;
  const struct _ac_state *s = m->root;
  uint64_t found = 0;
  for (uint64_t i = 0; i < n_symbols; i++) {
    s = automaton_step (s, t + i * sym_bytes);
    uint32_t nb = LOAD (&s->nb_outputs);
    for (const struct _ac_state *o = s; nb; o = LOAD (&o->fail)) { /* nearest (= longest) terminal state first */
      if (!LOAD (&o->terminal))
        continue;
      if (found < capacity) {
        records[found].end_pos = i;
        records[found].length = o->depth;
        records[found].keyword_id = o->rank;
      }
      found++;
      nb--;
    }
  }
  *n_found = found;
  return found > capacity ? ACM_GPU_E_OVERFLOW : ACM_GPU_OK;
}
