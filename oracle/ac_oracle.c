/*
 * ac_oracle.c -- TEST INFRASTRUCTURE ONLY (see ac_oracle.h for the rules and the parity pin).
 *
 * Plain-C restatement of the algorithm of /root/reference/aho_corasick.c for the scan path:
 * pointer trie, per-state ordered edge container searched through the user comparator, failure
 * links, output counts, per-symbol match step and match retrieval.  Each function cites the
 * reference lines it follows.  The external ordered map the reference uses (minimaps, not in
 * tree) is replaced by a sorted pointer vector + binary search: results depend on the map only
 * as "find the element that compares equal" (SURVEY.md 8c).
 */
#define _GNU_SOURCE
#include "ac_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* reference: ACM_ASSERT, aho_corasick.c:24-36 (message format kept, thread exit replaced by
 * abort(): the oracle never runs in a thread the caller wants to survive). */
#define ORC_REQUIRE(cond, msg)                                                                      \
  do {                                                                                              \
    if (!(cond)) {                                                                                  \
      fflush (stdout);                                                                              \
      fprintf (stderr, "FATAL ERROR: A prerequisite is not fulfilled in function %s.\n", __func__); \
      fprintf (stderr, "             %s\n", (msg));                                                 \
      abort ();                                                                                     \
    }                                                                                               \
  } while (0)

/* ---------------------------------------------------------------- small pointer vector */
typedef struct {
  void **v;
  size_t n, cap;
} pvec;

static void
pvec_insert_at (pvec *p, size_t at, void *x) {
  if (p->n == p->cap) {
    p->cap = p->cap ? 2 * p->cap : 2;
    p->v = realloc (p->v, p->cap * sizeof *p->v);
    ORC_REQUIRE (p->v, "Out of memory.");
  }
  memmove (p->v + at + 1, p->v + at, (p->n - at) * sizeof *p->v);
  p->v[at] = x;
  p->n++;
}

static void
pvec_remove_at (pvec *p, size_t at) {
  memmove (p->v + at, p->v + at + 1, (p->n - at - 1) * sizeof *p->v);
  p->n--;
}

/* ---------------------------------------------------------------- data model, :39-82 */
struct edge { /* struct _ac_transition, :39-42 */
  void *letter;
  orc_state *to;
};

struct orc_state { /* struct _ac_state, :44-65 */
  pvec edges;      /* of struct edge*, ordered by the machine comparator on letter */
  struct {
    const void *letter;
    const orc_state *state;
  } previous;
  const orc_state *fail_state;
  int is_end_of_keyword;
  size_t nb_outputs;
  struct {
    void *value;
    orc_dtor_fn dtor;
  } definition;
  orc_machine *machine;
  size_t id;
  pvec inverse_fail; /* Meyer-85 IF[] : set of orc_state* ordered by address, :63,97-98 */
};

struct orc_machine { /* struct _ac_machine, :67-82 (no mutex: the oracle is driven single-writer) */
  orc_state *state_0;
  size_t nb_sequences;
  size_t reconstruct;
  size_t nb_states;
  orc_cmp_fn cmp;
  void *cmp_arg;
  orc_dtor_fn dtor;
  int variant;
};

int
orc_cmp_default (const void *a, const void *b, const void *arg) { /* :134-138 */
  return memcmp (a, b, *(const size_t *)arg);
}

/* ordered lookup of the edge whose letter compares equal; *pos = insertion point. (:175,:299) */
static struct edge *
edge_find (const orc_state *s, const void *letter, size_t *pos) {
  const orc_machine *m = s->machine;
  size_t lo = 0, hi = s->edges.n;
  while (lo < hi) {
    size_t mid = lo + (hi - lo) / 2;
    struct edge *e = s->edges.v[mid];
    int c = m->cmp (letter, e->letter, m->cmp_arg);
    if (c == 0) {
      if (pos)
        *pos = mid;
      return e;
    }
    if (c < 0)
      hi = mid;
    else
      lo = mid + 1;
  }
  if (pos)
    *pos = lo;
  return 0;
}

/* address-ordered set of states (IF[]) */
static size_t
ifs_pos (const pvec *p, const orc_state *x, int *found) {
  size_t lo = 0, hi = p->n;
  while (lo < hi) {
    size_t mid = lo + (hi - lo) / 2;
    if (p->v[mid] == (void *)x) {
      *found = 1;
      return mid;
    }
    if ((uintptr_t)x < (uintptr_t)p->v[mid])
      hi = mid;
    else
      lo = mid + 1;
  }
  *found = 0;
  return lo;
}

static void
ifs_add (pvec *p, orc_state *x) {
  int found;
  size_t at = ifs_pos (p, x, &found);
  ORC_REQUIRE (!found, "IF set already holds the state.");
  pvec_insert_at (p, at, x);
}

static void
ifs_del (pvec *p, orc_state *x) {
  int found;
  size_t at = ifs_pos (p, x, &found);
  ORC_REQUIRE (found, "IF set does not hold the state."); /* :217 requires exactly one removal */
  pvec_remove_at (p, at);
}

static orc_state *
state_new (orc_machine *m) { /* state_create, :90-104 */
  orc_state *s = calloc (1, sizeof *s);
  ORC_REQUIRE (s, "Out of memory.");
  s->machine = m;
  s->id = m->nb_states++;
  return s;
}

static void
state_free (orc_state *s) { /* state_release + transition_release, :107-132 */
  for (size_t i = 0; i < s->edges.n; i++) {
    struct edge *e = s->edges.v[i];
    if (s->machine->dtor)
      s->machine->dtor (e->letter);
    state_free (e->to);
    free (e);
  }
  free (s->edges.v);
  if (s->definition.dtor)
    s->definition.dtor (s->definition.value);
  free (s->inverse_fail.v);
  free (s);
}

orc_machine *
orc_create (orc_cmp_fn cmp, void *cmp_arg, orc_dtor_fn dtor, int variant) { /* :140-151 */
  ORC_REQUIRE (cmp, "A comparison function should be provided.");
  orc_machine *m = calloc (1, sizeof *m);
  ORC_REQUIRE (m, "Out of memory.");
  m->cmp = cmp;
  m->cmp_arg = cmp_arg;
  m->dtor = dtor;
  m->variant = variant;
  m->state_0 = state_new (m);
  return m;
}

void
orc_release (orc_machine *m) { /* :153-159 */
  ORC_REQUIRE (m, "Invalid null machine.");
  state_free (m->state_0);
  free (m);
}

orc_state *
orc_initiate (orc_machine *m) { /* :161-165 */
  ORC_REQUIRE (m, "Invalid null machine.");
  return m->state_0;
}

/* state_goto, :167-192 : follow goto if defined, else failure links until defined or root;
 * a miss at the root stays at the root (:185-186, no LOOP_0 edges are materialised). */
static const orc_state *
step (const orc_state *s, const void *letter) {
  for (;;) {
    struct edge *e = edge_find (s, letter, 0);
    if (e)
      return e->to;
    if (s == s->machine->state_0)
      return s;
    s = s->fail_state;
  }
}

/* complete_fail_state, :194-208 */
static void
complete_fail (orc_state *r, orc_state *s, const void *a) {
  ORC_REQUIRE (r->fail_state || r == r->machine->state_0, "Undefined fail state.");
  ORC_REQUIRE (s != s->machine->state_0, "Incorrect state.");
  if (r->fail_state)
    s->fail_state = step (r->fail_state, a); /* :202 */
  else
    s->fail_state = r; /* children of the root fail to the root, :205 */
  s->nb_outputs += s->fail_state->nb_outputs; /* :207 */
}

/* Meyer-85 maintenance, :211-239.  complete_inverse (x, n', c): if x has a child x' on c then
 * f(x') <- n' (moving x' from IF[old f(x')] to IF[n']); else recurse over IF[x].  The reference
 * traverses live maps while update_fail_state edits other (possibly the same) maps; the
 * restatement visits a snapshot of each IF set, which is the semantics under which the result
 * equals the textbook automaton (checked against ORC_AC75 after every insert in the tests). */
static void
complete_inverse (orc_state *x, struct edge *nprime_edge) {
  struct edge *xe = edge_find (x, nprime_edge->letter, 0);
  if (xe) { /* update_fail_state, :211-222 */
    orc_state *xprime = xe->to;
    ifs_del (&((orc_state *)xprime->fail_state)->inverse_fail, xprime);
    xprime->fail_state = nprime_edge->to;
    ifs_add (&nprime_edge->to->inverse_fail, xprime);
    return;
  }
  size_t n = x->inverse_fail.n; /* :236 */
  if (!n)
    return;
  orc_state **snap = malloc (n * sizeof *snap);
  ORC_REQUIRE (snap, "Out of memory.");
  memcpy (snap, x->inverse_fail.v, n * sizeof *snap);
  for (size_t i = 0; i < n; i++)
    complete_inverse (snap[i], nprime_edge);
  free (snap);
}

/* enter_child, :242-267 */
static orc_state *
enter_child (orc_state *n, void *c, size_t pos) {
  orc_machine *m = n->machine;
  orc_state *nprime = state_new (m);
  struct edge *e = malloc (sizeof *e);
  ORC_REQUIRE (e, "Out of memory.");
  e->letter = c;
  e->to = nprime;
  pvec_insert_at (&n->edges, pos, e);                   /* :250 */
  nprime->previous.state = n;                           /* :252 */
  nprime->previous.letter = c;
  if (m->variant == ORC_MEYER85) {
    complete_fail (n, nprime, c);                       /* :259 */
    ifs_add (&((orc_state *)nprime->fail_state)->inverse_fail, nprime); /* :261 */
    size_t k = n->inverse_fail.n;                       /* :264 */
    if (k) {
      orc_state **snap = malloc (k * sizeof *snap);
      ORC_REQUIRE (snap, "Out of memory.");
      memcpy (snap, n->inverse_fail.v, k * sizeof *snap);
      for (size_t i = 0; i < k; i++)
        complete_inverse (snap[i], e);
      free (snap);
    }
  }
  return nprime;
}

void
orc_insert_letter_of_keyword (orc_state **cursor, void *letter) { /* :291-316 */
  ORC_REQUIRE (cursor && *cursor && letter, "Invalid null state or letter.");
  orc_machine *m = (*cursor)->machine;
  size_t pos = 0;
  struct edge *e = edge_find (*cursor, letter, &pos);
  if (e) {
    *cursor = e->to;
    if (m->dtor)
      m->dtor (letter); /* already known letter is destroyed at once, :306-307 */
  } else
    *cursor = enter_child (*cursor, letter, pos);
}

/* enter_output, :330-338 (+ Meyer recursion over IF[], :321-327,336) */
static void
enter_output (orc_state *n) {
  n->nb_outputs += 1;
  if (n->machine->variant == ORC_MEYER85)
    for (size_t i = 0; i < n->inverse_fail.n; i++)
      enter_output (n->inverse_fail.v[i]);
}

void *
orc_insert_end_of_keyword (orc_state **cursor, void *value, orc_dtor_fn dtor) { /* :340-363 */
  ORC_REQUIRE (cursor && *cursor, "Invalid null state.");
  orc_state *s = *cursor;
  orc_machine *m = s->machine;
  ORC_REQUIRE (s != m->state_0, "acm_insert_letter_of_keyword should be called first.");
  if (!s->is_end_of_keyword) {
    enter_output (s);
    s->is_end_of_keyword = 1;
    m->nb_sequences++;
    if (!++m->reconstruct) /* :353-354 */
      m->reconstruct = 1;
  }
  void *ret = s->definition.value; /* :357 */
  if (s->definition.value == 0) {  /* first non-null value wins, :358-359 */
    s->definition.value = value;
    s->definition.dtor = dtor;
  }
  *cursor = m->state_0; /* :360 */
  return ret;
}

/* state_fail_state_construct + transition_fail_state_construct, :367-417 : BFS over the goto
 * tree, resetting nb_outputs from is_end_of_keyword (:381) then complete_fail (:382). */
static void
rebuild_bfs (orc_machine *m) {
  if (!m->reconstruct)
    return;
  orc_state **queue = malloc (m->nb_states * sizeof *queue);
  ORC_REQUIRE (queue, "Out of memory.");
  size_t head = 0, tail = 0;
  queue[tail++] = m->state_0;
  while (head < tail) {
    orc_state *r = queue[head++];
    for (size_t i = 0; i < r->edges.n; i++) {
      struct edge *e = r->edges.v[i];
      orc_state *s = e->to;
      queue[tail++] = s;
      s->nb_outputs = s->is_end_of_keyword ? 1 : 0;
      complete_fail (r, s, e->letter);
    }
  }
  free (queue);
  m->reconstruct = 0;
}

size_t
orc_match (const orc_state **cursor, const void *letter) { /* :434-448 */
  ORC_REQUIRE (cursor && *cursor && letter, "Invalid null state or letter.");
  if ((*cursor)->machine->variant == ORC_AC75)
    rebuild_bfs ((*cursor)->machine); /* :443-446 */
  return (*cursor = step (*cursor, letter))->nb_outputs;
}

void
orc_matcher_init (orc_holder *h) { /* :420-424 */
  ORC_REQUIRE (h, "Invalid null matcher.");
  memset (h, 0, sizeof *h);
}

void
orc_matcher_release (orc_holder *h) { /* :426-431 */
  ORC_REQUIRE (h, "Invalid null matcher.");
  free (h->letters);
  orc_matcher_init (h);
}

/* walk the failure chain to the index-th terminal state, :457-466 */
static const orc_state *
nth_output_state (const orc_state *state, size_t index) {
  size_t i = 0;
  for (; state; state = state->fail_state, i++) {
    while (!state->is_end_of_keyword)
      state = state->fail_state;
    if (i == index)
      break;
  }
  return state;
}

void
orc_get_match (const orc_state *state, size_t index, orc_holder *h) { /* :451-482 */
  ORC_REQUIRE (state, "Invalid null state.");
  ORC_REQUIRE (state != state->machine->state_0, "acm_match should be called first and acm_matcher_init called on the MatchHolder.");
  ORC_REQUIRE (index < state->nb_outputs, "Index out of bounds.");
  state = nth_output_state (state, index);
  if (h) {
    h->length = 0; /* :472-474 */
    for (const orc_state *s = state; s && s->previous.state; s = s->previous.state)
      h->length++;
    h->letters = realloc (h->letters, h->length * sizeof *h->letters); /* :476 */
    ORC_REQUIRE (h->letters, "Out of memory.");
    size_t i = 0; /* :477-479 */
    for (const orc_state *s = state; s && s->previous.state; s = s->previous.state, i++)
      h->letters[h->length - i - 1] = s->previous.letter;
    h->value = state->definition.value; /* :480 */
  }
}

size_t
orc_nb_keywords (const orc_machine *m) { /* :484-488 */
  ORC_REQUIRE (m, "Invalid null machine.");
  return m->nb_sequences;
}

/* foreach_keyword, :490-531 : DFS in container order, operator called on every terminal node */
static void
foreach_rec (const orc_state *s, const void ***letters, size_t *cap, size_t depth, void (*op) (orc_holder)) {
  if (s->is_end_of_keyword && depth) {
    orc_holder k = { .letters = *letters, .length = depth, .value = s->definition.value };
    op (k);
  }
  for (size_t i = 0; i < s->edges.n; i++) {
    struct edge *e = s->edges.v[i];
    if (depth >= *cap) {
      *cap = depth + 1;
      *letters = realloc (*letters, *cap * sizeof **letters);
      ORC_REQUIRE (*letters, "Out of memory.");
    }
    (*letters)[depth] = e->letter;
    foreach_rec (e->to, letters, cap, depth + 1, op);
  }
}

void
orc_foreach_keyword (const orc_machine *m, void (*op) (orc_holder)) {
  ORC_REQUIRE (m, "Invalid null machine.");
  if (!op)
    return;
  const void **letters = 0;
  size_t cap = 0;
  foreach_rec (m->state_0, &letters, &cap, 0, op);
  free (letters);
}

/* ---------------------------------------------------------------- introspection */
size_t
orc_nb_states (const orc_machine *m) {
  return m->nb_states;
}
size_t
orc_state_id (const orc_state *s) {
  return s->id;
}
const orc_state *
orc_state_fail (const orc_state *s) {
  if (s->machine->variant == ORC_AC75)
    rebuild_bfs (s->machine);
  return s->fail_state;
}
size_t
orc_state_nb_outputs (const orc_state *s) {
  if (s->machine->variant == ORC_AC75)
    rebuild_bfs (s->machine);
  return s->nb_outputs;
}
int
orc_state_is_end (const orc_state *s) {
  return s->is_end_of_keyword;
}
size_t
orc_state_depth (const orc_state *s) {
  size_t d = 0;
  for (; s && s->previous.state; s = s->previous.state)
    d++;
  return d;
}

/* ---------------------------------------------------------------- the caller loop, bulk form */
static inline uint64_t
splitmix64 (uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

static inline uint64_t
record_hash (uint64_t end_pos, uint32_t length, uint32_t id) {
  return splitmix64 ((end_pos * 1315423911ull) ^ ((uint64_t)length << 40) ^ ((uint64_t)id + 1));
}

uint64_t
orc_digest (const orc_record *r, uint64_t n) {
  uint64_t d = 0;
  for (uint64_t i = 0; i < n; i++)
    d += record_hash (r[i].end_pos, r[i].length, r[i].keyword_id);
  return d;
}

/* reference loop: examples/test.c:17-23, generic_test.c:139-158 (index order 0..nb-1) */
uint64_t
orc_scan (orc_machine *m, const void *text, uint64_t n, size_t sym_size, uint64_t pos_base,
          uint64_t emit_from, orc_record *out, uint64_t cap) {
  const unsigned char *p = text;
  const orc_state *cur = orc_initiate (m);
  orc_holder h;
  orc_matcher_init (&h);
  uint64_t found = 0;
  for (uint64_t i = 0; i < n; i++) {
    size_t nb = orc_match (&cur, p + i * sym_size);
    if (!nb || i < emit_from)
      continue;
    for (size_t j = 0; j < nb; j++) {
      orc_get_match (cur, j, &h);
      if (out && found < cap) {
        out[found].end_pos = pos_base + i;
        out[found].length = (uint32_t)h.length;
        out[found].keyword_id = (uint32_t)((uintptr_t)h.value - 1);
      }
      found++;
    }
  }
  orc_matcher_release (&h);
  return found;
}

uint64_t
orc_count (orc_machine *m, const void *text, uint64_t n, size_t sym_size) { /* generic_test.c:272-273 */
  const unsigned char *p = text;
  const orc_state *cur = orc_initiate (m);
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; i++)
    total += orc_match (&cur, p + i * sym_size);
  return total;
}

struct mt_job {
  orc_machine *m;
  const unsigned char *text;
  uint64_t begin, end, warm;
  size_t sym_size;
  uint64_t pos_base; /* added to a record's position in the digest (a piece of a longer text) */
  uint64_t found, digest;
};

static void *
mt_worker (void *arg) {
  struct mt_job *j = arg;
  const orc_state *cur = orc_initiate (j->m);
  orc_holder h;
  orc_matcher_init (&h);
  for (uint64_t i = j->warm; i < j->end; i++) {
    size_t nb = orc_match (&cur, j->text + i * j->sym_size);
    if (!nb || i < j->begin)
      continue;
    for (size_t k = 0; k < nb; k++) {
      orc_get_match (cur, k, &h);
      j->digest += record_hash (j->pos_base + i, (uint32_t)h.length, (uint32_t)((uintptr_t)h.value - 1));
      j->found++;
    }
  }
  orc_matcher_release (&h);
  return 0;
}

/* A piece text[0 .. n) of a longer text whose first symbol has global index pos_base: matches that
 * end before emit_from are not counted (the piece starts with an overlap of lmax - 1 symbols of
 * its predecessor, scanned as a warm-up only), the digest is taken over global positions -- it is
 * a sum over records, so the digests of the pieces of a text add up to the digest of the whole. */
uint64_t
orc_scan_mt_at (orc_machine *m, const void *text, uint64_t n, size_t sym_size, size_t lmax,
                int threads, uint64_t pos_base, uint64_t emit_from, uint64_t *digest) {
  if (threads < 1)
    threads = 1;
  if (m->variant == ORC_AC75)
    rebuild_bfs (m); /* keep the lazy rebuild out of the worker threads */
  struct mt_job *jobs = calloc ((size_t)threads, sizeof *jobs);
  pthread_t *tid = calloc ((size_t)threads, sizeof *tid);
  ORC_REQUIRE (jobs && tid, "Out of memory.");
  uint64_t overlap = lmax ? lmax - 1 : 0;
  for (int t = 0; t < threads; t++) {
    jobs[t].m = m;
    jobs[t].text = text;
    jobs[t].sym_size = sym_size;
    if (emit_from > n)
      emit_from = n;
    jobs[t].pos_base = pos_base;
    jobs[t].begin = emit_from + (n - emit_from) * (uint64_t)t / (uint64_t)threads;
    jobs[t].end = emit_from + (n - emit_from) * (uint64_t)(t + 1) / (uint64_t)threads;
    jobs[t].warm = jobs[t].begin > overlap ? jobs[t].begin - overlap : 0;
    if (threads == 1)
      mt_worker (&jobs[t]);
    else
      ORC_REQUIRE (pthread_create (&tid[t], 0, mt_worker, &jobs[t]) == 0, "pthread_create failed.");
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

uint64_t
orc_scan_mt (orc_machine *m, const void *text, uint64_t n, size_t sym_size, size_t lmax,
             int threads, uint64_t *digest) {
  return orc_scan_mt_at (m, text, n, sym_size, lmax, threads, 0, 0, digest);
}
