/*
 * acm_flat.c -- flattens the host trie into the arrays the GPU walks (include/acm_gpu.h, ACMFlat).
 *
 * What is flattened is exactly what the reference's scan reads per node (aho_corasick.c:44-65):
 * goto edges (:47), failure link (:53), keyword-end flag (:54), output count (:55) and the depth
 * that acm_get_match recomputes through `previous` (:472-474).  States are renumbered
 * breadth-first, children in ascending symbol value (== memcmp order for byte alphabets).
 */
#define _GNU_SOURCE
#include "acm_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct ACMFlat {
  ACMFlatInfo info;
  uint32_t *row_ptr, *edge_sym, *edge_next, *fail, *depth, *nb_outputs, *term_kw, *out_link;
  uint32_t *depth_start, *kw_state;
  /* machines flattened over comparator classes (acm_flatten_classes): edge_sym holds class ids */
  uint16_t *class_map;   /* [class_entries] symbol value -> class id; NULL for ACM_CMP_DEFAULT machines */
  uint32_t *edge_letter; /* [n_edges] the dictionary's own symbol on each edge */
  uint32_t class_entries, n_classes;
  /* 8-byte symbols: edge_sym holds 1 + the rank of the symbol among the distinct symbols of the
   * dictionary (0 is kept for every other symbol of a text); keys64 are those symbols, ascending */
  uint64_t *keys64;
  uint32_t n_keys64;
  /* 4-byte symbols flattened over comparator classes (acm_flatten_classes, sym_bytes 4): the
   * 2^32 symbol values cannot be enumerated, so the classes are those of the dictionary's OWN
   * symbols -- class 1 .. n_classes in comparator order, 0 for a symbol that compares equal to
   * none of them -- and a text symbol is classified when it is first met (acm_gpu.hip) */
  uint32_t *keys32, *keys32_class; /* the dictionary's distinct symbols by value, and their classes */
  uint32_t n_keys32;
  uint32_t *class_rep32;           /* [n_classes] one symbol of each class, in comparator order */
};

static uint32_t
symbol_value (const void *letter, uint32_t sym_bytes) {
  const unsigned char *p = letter;
  uint32_t v = 0;
  for (uint32_t i = 0; i < sym_bytes; i++)
    v |= (uint32_t)p[i] << (8 * i);
  return v;
}

static uint64_t
symbol_value64 (const void *letter) {
  const unsigned char *p = letter;
  uint64_t v = 0;
  for (uint32_t i = 0; i < 8; i++)
    v |= (uint64_t)p[i] << (8 * i);
  return v;
}

static int
u64_cmp (const void *a, const void *b) {
  const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : x > y;
}

/* 1 + rank of v among the n sorted keys (it is one of them) */
static uint32_t
key_rank1 (const uint64_t *keys, uint32_t n, uint64_t v) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (keys[mid] <= v)
      lo = mid;
    else
      hi = mid;
  }
  return lo + 1;
}

struct row_item {
  uint32_t sym;
  struct _ac_state *node;
};

static int
row_item_cmp (const void *a, const void *b) {
  uint32_t x = ((const struct row_item *)a)->sym, y = ((const struct row_item *)b)->sym;
  return x < y ? -1 : x > y;
}

/* children of one state occupy consecutive new ids, so sorting (sym, node) pairs in place keeps
 * edge_next[] = first_id + i valid */
static int
sort_row_by_value (uint32_t *syms, struct _ac_state **nodes, uint32_t n) {
  struct row_item *tmp = malloc ((size_t)n * sizeof *tmp);
  if (!tmp)
    return -1;
  for (uint32_t i = 0; i < n; i++)
    tmp[i] = (struct row_item){ syms[i], nodes[i] };
  qsort (tmp, n, sizeof *tmp, row_item_cmp);
  for (uint32_t i = 0; i < n; i++) {
    syms[i] = tmp[i].sym;
    nodes[i] = tmp[i].node;
  }
  free (tmp);
  return 0;
}

void
acm_flat_release (ACMFlat *f) {
  if (!f)
    return;
  free (f->row_ptr);
  free (f->edge_sym);
  free (f->edge_next);
  free (f->fail);
  free (f->depth);
  free (f->nb_outputs);
  free (f->term_kw);
  free (f->out_link);
  free (f->depth_start);
  free (f->kw_state);
  free (f->class_map);
  free (f->edge_letter);
  free (f->keys64);
  free (f->keys32);
  free (f->keys32_class);
  free (f->class_rep32);
  free (f);
}

/* class_map != NULL: symbols are replaced by their comparator class (ownership of class_map
 * passes to the flat tables on success) */
/* class of symbol v among the n sorted keys; UINT32_MAX if it is not one of them (a keyword that
 * came in between the walk that collected the keys and the snapshot that asks: flatten_classes32
 * starts over) */
static uint32_t
key32_class (const uint32_t *keys, const uint32_t *classes, uint32_t n, uint32_t v) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (keys[mid] <= v)
      lo = mid;
    else
      hi = mid;
  }
  return n && keys[lo] == v ? classes[lo] : UINT32_MAX;
}
#define ACM_FLAT_STALE_KEYS (-1000) /* internal: flatten_impl met a symbol the key table does not hold */

struct keys32_arg {
  uint32_t *keys, *classes, *reps;
  uint32_t n, n_classes;
};

static int
flatten_impl (ACMachine *machine, uint32_t sym_bytes, uint16_t *class_map, uint32_t class_entries, uint32_t n_classes,
              ACMFlat **out, const struct keys32_arg *k32) {

  acm_internal_refresh (machine);
  acm_internal_lock (machine); /* writers are excluded while the snapshot is taken */
  const uint32_t n = acm_internal_nb_states (machine);
  ACMFlat *f = calloc (1, sizeof *f);
  struct _ac_state **order = malloc ((size_t)n * sizeof *order); /* BFS order: new id -> node */
  uint32_t *newid = malloc ((size_t)n * sizeof *newid);          /* creation id -> new id */
  if (!f || !order || !newid)
    goto nomem;
  f->row_ptr = malloc (((size_t)n + 1) * sizeof (uint32_t));
  f->edge_sym = malloc ((size_t)(n ? n : 1) * sizeof (uint32_t));
  f->edge_next = malloc ((size_t)(n ? n : 1) * sizeof (uint32_t));
  f->fail = malloc ((size_t)n * sizeof (uint32_t));
  f->depth = malloc ((size_t)n * sizeof (uint32_t));
  f->nb_outputs = malloc ((size_t)n * sizeof (uint32_t));
  f->term_kw = malloc ((size_t)n * sizeof (uint32_t));
  f->out_link = malloc ((size_t)n * sizeof (uint32_t));
  if (!f->row_ptr || !f->edge_sym || !f->edge_next || !f->fail || !f->depth || !f->nb_outputs || !f->term_kw || !f->out_link)
    goto nomem;
  if (class_map || k32) {
    f->edge_letter = malloc ((size_t)(n ? n : 1) * sizeof (uint32_t));
    if (!f->edge_letter)
      goto nomem;
  }

  /* breadth-first renumbering; the queue is `order` itself */
  uint32_t head = 0, tail = 0, edges = 0, lmax = 0, nkw = 0, max_out = 0;
  int stale_keys = 0;
  order[tail++] = acm_internal_root (machine);
  newid[order[0]->id] = 0;
  if (sym_bytes == 8) {
    /* intern the symbols: a first walk collects the distinct ones (the kernels then see 4-byte
     * ids; a text is mapped through the same table on the device before it is walked) */
    uint64_t *keys = malloc ((size_t)(n ? n : 1) * sizeof *keys);
    if (!keys)
      goto nomem;
    f->keys64 = keys;
    uint32_t nk = 0;
    for (uint32_t h = 0, t = 1; h < t; h++) {
      struct _ac_state *st = order[h];
      for (uint32_t i = 0; i < ACM_NKIDS (st); i++) {
        keys[nk++] = symbol_value64 (ACM_KID (st, i)->letter);
        order[t++] = ACM_KID (st, i);
      }
    }
    qsort (keys, nk, sizeof *keys, u64_cmp);
    uint32_t u = 0;
    for (uint32_t i = 0; i < nk; i++)
      if (i == 0 || keys[i] != keys[i - 1])
        keys[u++] = keys[i];
    f->n_keys64 = u;
  }
  while (head < tail) {
    struct _ac_state *s = order[head];
    f->row_ptr[head] = edges;
    for (uint32_t i = 0; i < ACM_NKIDS (s); i++) {
      struct _ac_state *k = ACM_KID (s, i);
      newid[k->id] = tail;
      f->edge_sym[edges] = sym_bytes == 8 ? key_rank1 (f->keys64, f->n_keys64, symbol_value64 (k->letter))
                                          : symbol_value (k->letter, sym_bytes);
      if (class_map) {
        f->edge_letter[edges] = f->edge_sym[edges];
        f->edge_sym[edges] = class_map[f->edge_sym[edges]];
      } else if (k32) {
        f->edge_letter[edges] = f->edge_sym[edges];
        f->edge_sym[edges] = key32_class (k32->keys, k32->classes, k32->n, f->edge_sym[edges]);
        stale_keys |= f->edge_sym[edges] == UINT32_MAX;
      }
      f->edge_next[edges] = tail;
      edges++;
      order[tail++] = k;
    }
    if (sym_bytes > 1 && ACM_NKIDS (s) > 1 && !class_map && !k32) { /* class ids already ascend in comparator order */
      /* the comparator orders multi-byte symbols by memcmp; the device bisects rows by numeric
       * value, so re-order this row (and the ids just handed out) by value */
      if (sort_row_by_value (f->edge_sym + f->row_ptr[head], order + f->edge_next[f->row_ptr[head]], ACM_NKIDS (s)))
        goto nomem;
      for (uint32_t i = 0; i < ACM_NKIDS (s); i++)
        newid[order[f->edge_next[f->row_ptr[head]] + i]->id] = f->edge_next[f->row_ptr[head]] + i;
    }
    head++;
  }
  f->row_ptr[n] = edges;
  if (stale_keys) { /* never hand out tables in which a symbol took a neighbour's class */
    acm_internal_unlock (machine);
    free (order);
    free (newid);
    acm_flat_release (f);
    return ACM_FLAT_STALE_KEYS;
  }

  uint32_t lo = UINT32_MAX, hi = 0;
  for (uint32_t e = 0; e < edges; e++) {
    if (f->edge_sym[e] < lo)
      lo = f->edge_sym[e];
    if (f->edge_sym[e] > hi)
      hi = f->edge_sym[e];
  }
  for (uint32_t i = 0; i < n; i++) {
    struct _ac_state *s = order[i];
    f->fail[i] = s->fail ? newid[s->fail->id] : 0;
    f->depth[i] = s->depth;
    f->nb_outputs[i] = s->nb_outputs;
    f->term_kw[i] = s->terminal ? s->rank : UINT32_MAX;
    if (s->depth > lmax)
      lmax = s->depth;
    if (s->terminal)
      nkw++;
    if (s->nb_outputs > max_out)
      max_out = s->nb_outputs;
  }
  /* out_link: BFS order guarantees f(i) < i is final when i is reached */
  for (uint32_t i = 0; i < n; i++) {
    uint32_t p = f->fail[i];
    f->out_link[i] = (i == 0 || p == 0) ? 0 : (f->term_kw[p] != UINT32_MAX ? p : f->out_link[p]);
  }
  f->depth_start = malloc (((size_t)lmax + 2) * sizeof (uint32_t));
  f->kw_state = malloc ((size_t)(nkw ? nkw : 1) * sizeof (uint32_t));
  if (!f->depth_start || !f->kw_state)
    goto nomem;
  for (uint32_t d = 0, i = 0; d <= lmax + 1; d++) {
    while (i < n && f->depth[i] < d)
      i++;
    f->depth_start[d] = i;
  }
  for (uint32_t i = 0; i < n; i++)
    if (f->term_kw[i] != UINT32_MAX)
      f->kw_state[f->term_kw[i]] = i;

  f->info.sym_bytes = sym_bytes;
  f->info.n_states = n;
  f->info.n_keywords = nkw;
  f->info.n_edges = edges;
  f->info.lmax = lmax;
  f->info.max_outputs = max_out;
  if (edges && sym_bytes == 1) {
    f->info.alpha_lo = lo;
    f->info.alpha_span = hi - lo + 1;
    f->info.width = f->info.alpha_span == 256 ? 256 : f->info.alpha_span + 1;
  } else {
    f->info.alpha_lo = edges ? lo : 0;
    f->info.alpha_span = 0;
    f->info.width = sym_bytes == 1 ? 1 : 0; /* empty byte machine: the single "other" class */
  }
  acm_internal_unlock (machine);
  free (order);
  free (newid);
  f->class_map = class_map;
  f->class_entries = class_map ? class_entries : 0;
  f->n_classes = class_map ? n_classes : 0;
  if (k32) { /* ownership passes to the flat tables */
    f->keys32 = k32->keys;
    f->keys32_class = k32->classes;
    f->class_rep32 = k32->reps;
    f->n_keys32 = k32->n;
    f->n_classes = k32->n_classes;
  }
  *out = f;
  return ACM_GPU_OK;

nomem:
  acm_internal_unlock (machine);
  free (order);
  free (newid);
  acm_flat_release (f);
  return ACM_GPU_E_NOMEM;
}

int
acm_flatten (ACMachine *machine, ACMFlat **out) {
  if (!machine || !out)
    return ACM_GPU_E_ARG;
  uint32_t sym_bytes = 0;
  int rc = acm_internal_symbol_bytes (machine, &sym_bytes);
  if (rc)
    return rc;
  return flatten_impl (machine, sym_bytes, NULL, 0, 0, out, 0);
}

/* ------------------------------------------------------------------ comparator classes
 * A machine built with another comparator than ACM_CMP_DEFAULT (the reference's flagship "any
 * ordered alphabet", README.md:43-44; e.g. the case-insensitive alphacmp of generic_test.c:48-54)
 * cannot tell some symbols apart.  For 1- and 2-byte symbols all 256 / 65,536 values are sorted
 * with the machine's own comparator; runs of values that compare equal are the classes, numbered
 * in comparator order.  The automaton is flattened over class ids and the device maps the text
 * through the same table before it walks it (classmap kernel), so the walk compares the way the
 * comparator does, symbol for symbol. */
struct class_sort_ctx {
  CMP_TYPE cmp;
  void *arg;
  uint32_t sym_bytes;
};

static int
class_sort_cmp (const void *a, const void *b, void *ctxp) {
  const struct class_sort_ctx *ctx = ctxp;
  /* the comparator sees what a caller's letters look like: sym_bytes bytes of the value in memory */
  const uint32_t va = *(const uint32_t *)a, vb = *(const uint32_t *)b;
  unsigned char la[2] = { (unsigned char)va, (unsigned char)(va >> 8) };
  unsigned char lb[2] = { (unsigned char)vb, (unsigned char)(vb >> 8) };
  const int c = ctx->cmp (la, lb, ctx->arg);
  if (c)
    return c;
  return va < vb ? -1 : va > vb; /* keep the sort deterministic inside a class */
}

/* comparator classes of the dictionary's own 4-byte symbols (ACMFlat::keys32) */
struct sym32_ctx {
  CMP_TYPE cmp;
  void *arg;
};
static int
sym32_sort_cmp (const void *a, const void *b, void *ctxp) {
  const struct sym32_ctx *ctx = ctxp;
  const int c = ctx->cmp (a, b, ctx->arg); /* the values as they lie in memory: what a caller's letters look like */
  if (c)
    return c;
  const uint32_t va = *(const uint32_t *)a, vb = *(const uint32_t *)b;
  return va < vb ? -1 : va > vb;
}
struct key_class {
  uint32_t key, cls;
};
static int
key_class_cmp (const void *a, const void *b) {
  const uint32_t x = ((const struct key_class *)a)->key, y = ((const struct key_class *)b)->key;
  return x < y ? -1 : x > y;
}

static int
flatten_classes32_once (ACMachine *machine, ACMFlat **out) {
  struct sym32_ctx ctx;
  acm_internal_comparator (machine, &ctx.cmp, &ctx.arg);
  /* every letter of the dictionary: one per state but the root (duplicates go away below) */
  acm_internal_lock (machine);
  const uint32_t n_states = acm_internal_nb_states (machine);
  uint32_t *letters = malloc ((size_t)(n_states ? n_states : 1) * sizeof *letters);
  struct _ac_state **stack = malloc ((size_t)(n_states ? n_states : 1) * sizeof *stack);
  uint32_t nl = 0;
  if (letters && stack) {
    uint32_t top = 0;
    stack[top++] = acm_internal_root (machine);
    while (top) {
      struct _ac_state *st = stack[--top];
      for (uint32_t i = 0; i < ACM_NKIDS (st); i++) {
        struct _ac_state *k = ACM_KID (st, i);
        letters[nl++] = symbol_value (k->letter, 4);
        stack[top++] = k;
      }
    }
  }
  acm_internal_unlock (machine);
  const int walked = letters && stack;
  free (stack);
  if (!walked) {
    free (letters);
    return ACM_GPU_E_NOMEM;
  }
  qsort_r (letters, nl, sizeof *letters, sym32_sort_cmp, &ctx);
  struct key_class *kc = malloc ((size_t)(nl ? nl : 1) * sizeof *kc);
  uint32_t *reps = malloc ((size_t)(nl ? nl : 1) * sizeof *reps);
  if (!kc || !reps) {
    free (letters);
    free (kc);
    free (reps);
    return ACM_GPU_E_NOMEM;
  }
  uint32_t nk = 0, n_classes = 0;
  int consistent = 1;
  for (uint32_t i = 0; i < nl && consistent; i++) {
    if (i && letters[i] == letters[i - 1])
      continue; /* the same symbol again */
    int fresh = 1;
    if (nk) {
      const int ab = ctx.cmp (&reps[n_classes - 1], &letters[i], ctx.arg), ba = ctx.cmp (&letters[i], &reps[n_classes - 1], ctx.arg);
      if (ab > 0 || ba < 0 || (ab == 0) != (ba == 0))
        consistent = 0; /* not an order: sorted neighbours out of order either way */
      fresh = ab != 0;
    }
    if (fresh)
      reps[n_classes++] = letters[i];
    kc[nk].key = letters[i];
    kc[nk].cls = n_classes; /* classes 1 .. n_classes; 0 is "equal to no symbol of the dictionary" */
    nk++;
  }
  /* the sort only compared neighbours: the representatives against one another too (all pairs up
   * to 1024 classes, pairs at power-of-two distances beyond) */
  for (uint32_t i = 0; i < n_classes && consistent; i++)
    for (uint32_t step = 1; i + step < n_classes && consistent; step = n_classes <= 1024 ? step + 1 : step * 2)
      consistent = ctx.cmp (&reps[i], &reps[i + step], ctx.arg) < 0 && ctx.cmp (&reps[i + step], &reps[i], ctx.arg) > 0;
  free (letters);
  if (!consistent) {
    free (kc);
    free (reps);
    return ACM_GPU_E_INELIGIBLE;
  }
  qsort (kc, nk, sizeof *kc, key_class_cmp);
  struct keys32_arg k32;
  k32.keys = malloc ((size_t)(nk ? nk : 1) * sizeof (uint32_t));
  k32.classes = malloc ((size_t)(nk ? nk : 1) * sizeof (uint32_t));
  k32.reps = reps;
  k32.n = nk;
  k32.n_classes = n_classes;
  if (!k32.keys || !k32.classes) {
    free (kc);
    free (reps);
    free (k32.keys);
    free (k32.classes);
    return ACM_GPU_E_NOMEM;
  }
  for (uint32_t i = 0; i < nk; i++) {
    k32.keys[i] = kc[i].key;
    k32.classes[i] = kc[i].cls;
  }
  free (kc);
  /* (a keyword inserted between the walk above and the snapshot below can bring a symbol without a
   * class: flatten_impl notices and the caller below starts over) */
  int rc = flatten_impl (machine, 4, NULL, 0, 0, out, &k32);
  if (rc) {
    free (k32.keys);
    free (k32.classes);
    free (reps);
  }
  return rc;
}

static int
flatten_classes32 (ACMachine *machine, ACMFlat **out) {
  /* the symbols are collected under one hold of the machine's lock and the snapshot is taken under
   * the next: an insertion in between that brings a new symbol makes the snapshot stale -- again,
   * a few times; a dictionary that keeps growing new symbols faster than that is not flattened */
  for (int attempt = 0; attempt < 8; attempt++) {
    const int rc = flatten_classes32_once (machine, out);
    if (rc != ACM_FLAT_STALE_KEYS)
      return rc;
  }
  return ACM_GPU_E_INELIGIBLE;
}

int
acm_flatten_classes (ACMachine *machine, uint32_t sym_bytes, ACMFlat **out) {
  if (machine && out && sym_bytes == 4)
    return flatten_classes32 (machine, out);
  if (!machine || !out || (sym_bytes != 1 && sym_bytes != 2))
    return ACM_GPU_E_ARG;
  struct class_sort_ctx ctx;
  ctx.sym_bytes = sym_bytes;
  acm_internal_comparator (machine, &ctx.cmp, &ctx.arg);
  const uint32_t V = sym_bytes == 1 ? 256u : 65536u;
  uint32_t *order = malloc ((size_t)V * sizeof *order);
  uint16_t *class_map = malloc ((size_t)V * sizeof *class_map);
  if (!order || !class_map) {
    free (order);
    free (class_map);
    return ACM_GPU_E_NOMEM;
  }
  for (uint32_t v = 0; v < V; v++)
    order[v] = v;
  qsort_r (order, V, sizeof *order, class_sort_cmp, &ctx);
  /* classes = runs of equal neighbours; a comparator that is not a consistent order over all
   * values (sorted neighbours out of order either way) cannot be canonicalised */
  uint32_t n_classes = 0;
  int consistent = 1;
  for (uint32_t i = 0; i < V && consistent; i++) {
    if (i == 0)
      class_map[order[0]] = 0;
    else {
      const uint32_t a = order[i - 1], b = order[i];
      unsigned char la[2] = { (unsigned char)a, (unsigned char)(a >> 8) };
      unsigned char lb[2] = { (unsigned char)b, (unsigned char)(b >> 8) };
      const int ab = ctx.cmp (la, lb, ctx.arg), ba = ctx.cmp (lb, la, ctx.arg);
      if (ab > 0 || ba < 0 || (ab == 0) != (ba == 0))
        consistent = 0;
      if (ab < 0)
        n_classes++;
      class_map[b] = (uint16_t)n_classes;
    }
  }
  n_classes++;
  /* the sort only looked at neighbours: check the class representatives against one another too
   * (all pairs up to 1024 classes, pairs at power-of-two distances beyond), and every value
   * against the representative of its class */
  if (consistent) {
    uint32_t *rep = malloc ((size_t)n_classes * sizeof *rep);
    if (!rep) {
      free (order);
      free (class_map);
      return ACM_GPU_E_NOMEM;
    }
    for (uint32_t i = V; i-- > 0;)
      rep[class_map[order[i]]] = order[i];
    for (uint32_t v = 0; v < V && consistent; v++) {
      const uint32_t r = rep[class_map[v]];
      unsigned char lv[2] = { (unsigned char)v, (unsigned char)(v >> 8) };
      unsigned char lr[2] = { (unsigned char)r, (unsigned char)(r >> 8) };
      consistent = ctx.cmp (lv, lr, ctx.arg) == 0 && ctx.cmp (lr, lv, ctx.arg) == 0;
    }
    for (uint32_t i = 0; i < n_classes && consistent; i++)
      for (uint32_t step = 1; i + step < n_classes && consistent; step = n_classes <= 1024 ? step + 1 : step * 2) {
        const uint32_t a = rep[i], b = rep[i + step];
        unsigned char la[2] = { (unsigned char)a, (unsigned char)(a >> 8) };
        unsigned char lb[2] = { (unsigned char)b, (unsigned char)(b >> 8) };
        consistent = ctx.cmp (la, lb, ctx.arg) < 0 && ctx.cmp (lb, la, ctx.arg) > 0;
      }
    free (rep);
  }
  free (order);
  if (!consistent) {
    free (class_map);
    return ACM_GPU_E_INELIGIBLE;
  }
  int rc = flatten_impl (machine, sym_bytes, class_map, V, n_classes, out, 0);
  if (rc)
    free (class_map);
  return rc;
}

void
acm_flat_info (const ACMFlat *f, ACMFlatInfo *info) {
  *info = f->info;
}

void
acm_flat_view (const ACMFlat *f, ACMFlatView *v) {
  v->row_ptr = f->row_ptr;
  v->edge_sym = f->edge_sym;
  v->edge_next = f->edge_next;
  v->fail = f->fail;
  v->depth = f->depth;
  v->nb_outputs = f->nb_outputs;
  v->term_kw = f->term_kw;
  v->out_link = f->out_link;
  v->depth_start = f->depth_start;
  v->kw_state = f->kw_state;
  v->class_map = f->class_map;
  v->edge_letter = f->edge_letter;
  v->class_entries = f->class_entries;
  v->n_classes = f->n_classes;
  v->keys64 = f->keys64;
  v->n_keys64 = f->n_keys64;
  v->keys32 = f->keys32;
  v->keys32_class = f->keys32_class;
  v->n_keys32 = f->n_keys32;
  v->class_rep32 = f->class_rep32;
}

/* Failure-resolved rows.  Row 0: goto or stay at the root.  Row s > 0: copy of row f(s) (already
 * final: f(s) < s in BFS order), overwritten by s's own goto edges -- the textbook
 * delta(s, a) = g(s, a) if defined else delta(f(s), a), without root self-loop edges in g
 * (reference :167-192). */
int
acm_flat_dense_rows (const ACMFlat *f, uint32_t n_rows, uint32_t entry_bytes, void *out) {
  const ACMFlatInfo *in = &f->info;
  if (in->sym_bytes != 1 || !out || n_rows > in->n_states || (entry_bytes != 2 && entry_bytes != 4))
    return ACM_GPU_E_ARG;
  if (entry_bytes == 2 && in->n_states > 32768)
    return ACM_GPU_E_ARG;
  const uint32_t W = in->width, lo = in->alpha_lo;
  const uint32_t flag = entry_bytes == 2 ? 0x8000u : 0x80000000u;
  /* rows are needed for every state on some failure chain of a requested row: all ids < n_rows
   * (f(s) < s), so building the first n_rows rows in order is self-contained. */
  uint32_t *rows = malloc ((size_t)n_rows * W * sizeof *rows);
  if (!rows && n_rows)
    return ACM_GPU_E_NOMEM;
  for (uint32_t s = 0; s < n_rows; s++) {
    uint32_t *row = rows + (size_t)s * W;
    if (s == 0)
      memset (row, 0, W * sizeof *row);
    else
      memcpy (row, rows + (size_t)f->fail[s] * W, W * sizeof *row);
    for (uint32_t e = f->row_ptr[s]; e < f->row_ptr[s + 1]; e++)
      row[f->edge_sym[e] - lo] = f->edge_next[e];
  }
  for (size_t i = 0; i < (size_t)n_rows * W; i++) {
    uint32_t nx = rows[i];
    uint32_t ent = nx | (f->nb_outputs[nx] ? flag : 0);
    if (entry_bytes == 2)
      ((uint16_t *)out)[i] = (uint16_t)ent;
    else
      ((uint32_t *)out)[i] = ent;
  }
  free (rows);
  return ACM_GPU_OK;
}

/* ------------------------------------------------------------------ serialised form
 * The reference has no on-disk form of a built machine (it is rebuilt from its keywords at each
 * start).  A blob is the flat tables verbatim:
 *
 *   header, 80 bytes: "AC75FLAT" | u32 version (1) | u32 0x01020304 (byte-order probe) |
 *                     ACMFlatInfo (9 x u32) | u32 class-table entries (0, 256 or 65536) |
 *                     u64 payload bytes | u64 FNV-1a-64 of payload | u64 number of 8-byte symbols (keys64)
 *   payload, u32 arrays in this order: row_ptr[n+1] edge_sym[E] edge_next[E] fail[n] depth[n]
 *                     nb_outputs[n] term_kw[n] out_link[n] depth_start[lmax+2] kw_state[K]
 *                     and for comparator-class machines: class_map (u16, entries / 2 words)
 *                     edge_letter[E]; for 8-byte symbols: keys64 (2 words each)
 *
 * Loading trusts nothing: the goto function (row_ptr, edge_sym, term_kw) is checked for shape,
 * every other array -- the failure function included -- is recomputed from it and compared.  A
 * blob that loads therefore holds exactly what acm_flatten would have produced, and cannot send a
 * device walker out of its tables. */
#define BLOB_MAGIC "AC75FLAT"
#define BLOB_VERSION 1u
#define BLOB_HEADER 80u

static uint64_t
fnv1a64 (const unsigned char *p, size_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; i++)
    h = (h ^ p[i]) * 0x100000001b3ull;
  return h;
}

static size_t
blob_payload_words (const ACMFlatInfo *in, uint32_t class_entries, uint64_t n_keys64) {
  const size_t n = in->n_states, e = in->n_edges;
  return (n + 1) + 2 * e + 5 * n + ((size_t)in->lmax + 2) + in->n_keywords + (class_entries ? class_entries / 2 + e : 0) +
         2 * (size_t)n_keys64;
}

size_t
acm_flat_blob_bytes (const ACMFlat *f) {
  return f ? BLOB_HEADER + 4 * blob_payload_words (&f->info, f->class_entries, f->n_keys64) : 0;
}

int
acm_flat_to_blob (const ACMFlat *f, void *out, size_t capacity) {
  if (!f || !out || capacity < acm_flat_blob_bytes (f) || f->keys32 /* classes of 4-byte symbols need their comparator */)
    return ACM_GPU_E_ARG;

  const ACMFlatInfo *in = &f->info;
  unsigned char *b = out, *p = b + BLOB_HEADER;
#define PUT(arr, cnt)                                                                              \
  do {                                                                                             \
    memcpy (p, (arr), (size_t)(cnt) * 4);                                                          \
    p += (size_t)(cnt) * 4;                                                                        \
  } while (0)
  PUT (f->row_ptr, (size_t)in->n_states + 1);
  PUT (f->edge_sym, in->n_edges);
  PUT (f->edge_next, in->n_edges);
  PUT (f->fail, in->n_states);
  PUT (f->depth, in->n_states);
  PUT (f->nb_outputs, in->n_states);
  PUT (f->term_kw, in->n_states);
  PUT (f->out_link, in->n_states);
  PUT (f->depth_start, (size_t)in->lmax + 2);
  PUT (f->kw_state, in->n_keywords);
  if (f->class_entries) {
    PUT (f->class_map, f->class_entries / 2);
    PUT (f->edge_letter, in->n_edges);
  }
  if (f->n_keys64)
    PUT (f->keys64, 2 * (size_t)f->n_keys64);
#undef PUT
  const uint64_t payload = (uint64_t)(p - (b + BLOB_HEADER));
  const uint64_t sum = fnv1a64 (b + BLOB_HEADER, (size_t)payload);
  const uint32_t version = BLOB_VERSION, probe = 0x01020304u, zero = f->class_entries;
  const uint64_t zero64 = f->n_keys64;
  memcpy (b, BLOB_MAGIC, 8);
  memcpy (b + 8, &version, 4);
  memcpy (b + 12, &probe, 4);
  memcpy (b + 16, in, sizeof *in); /* 36 bytes */
  memcpy (b + 52, &zero, 4);
  memcpy (b + 56, &payload, 8);
  memcpy (b + 64, &sum, 8);
  memcpy (b + 72, &zero64, 8);
  return ACM_GPU_OK;
}

/* goto edge of state s on symbol c in a CSR with rows in ascending symbol order; UINT32_MAX if none */
static uint32_t
csr_child (const uint32_t *row_ptr, const uint32_t *edge_sym, uint32_t s, uint32_t c) {
  uint32_t lo = row_ptr[s], hi = row_ptr[s + 1];
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (edge_sym[mid] < c)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo < row_ptr[s + 1] && edge_sym[lo] == c ? lo + 1 : UINT32_MAX; /* edge e leads to state e + 1 */
}

int
acm_flat_from_blob (const void *blob, size_t bytes, ACMFlat **out) {
  if (!blob || !out)
    return ACM_GPU_E_ARG;
  *out = NULL;
  const unsigned char *b = blob;
  uint32_t version, probe, class_entries;
  uint64_t payload, sum;
  ACMFlatInfo in;
  if (bytes < BLOB_HEADER || memcmp (b, BLOB_MAGIC, 8) != 0)
    return ACM_GPU_E_FORMAT;
  memcpy (&version, b + 8, 4);
  memcpy (&probe, b + 12, 4);
  memcpy (&in, b + 16, sizeof in);
  memcpy (&class_entries, b + 52, 4);
  uint64_t n_keys64;
  memcpy (&n_keys64, b + 72, 8);
  memcpy (&payload, b + 56, 8);
  memcpy (&sum, b + 64, 8);
  if (version != BLOB_VERSION || probe != 0x01020304u)
    return ACM_GPU_E_FORMAT;
  if (in.n_states == 0 || in.n_edges != in.n_states - 1 || in.n_keywords > in.n_states ||
      in.lmax >= in.n_states + 1 || (in.sym_bytes != 1 && in.sym_bytes != 2 && in.sym_bytes != 4 && in.sym_bytes != 8))
    return ACM_GPU_E_FORMAT;
  if (n_keys64 > in.n_edges || (in.sym_bytes != 8 && n_keys64) || (in.sym_bytes == 8 && (n_keys64 != 0) != (in.n_edges != 0)))
    return ACM_GPU_E_FORMAT;
  if (class_entries != 0 && !(class_entries == 256 && in.sym_bytes == 1) && !(class_entries == 65536 && in.sym_bytes == 2))
    return ACM_GPU_E_FORMAT;
  if (payload != 4ull * blob_payload_words (&in, class_entries, n_keys64) || bytes - BLOB_HEADER < payload ||
      fnv1a64 (b + BLOB_HEADER, (size_t)payload) != sum)
    return ACM_GPU_E_FORMAT;

  const uint32_t n = in.n_states, E = in.n_edges;
  ACMFlat *f = calloc (1, sizeof *f);
  if (!f)
    return ACM_GPU_E_NOMEM;
  const unsigned char *p = b + BLOB_HEADER;
  int bad = 0;
#define GET(field, cnt)                                                                            \
  do {                                                                                             \
    const size_t c_ = (size_t)(cnt);                                                               \
    f->field = malloc ((c_ ? c_ : 1) * 4);                                                         \
    if (!f->field)                                                                                 \
      bad = 1;                                                                                     \
    else                                                                                           \
      memcpy (f->field, p, c_ * 4);                                                                \
    p += c_ * 4;                                                                                   \
  } while (0)
  GET (row_ptr, (size_t)n + 1);
  GET (edge_sym, E);
  GET (edge_next, E);
  GET (fail, n);
  GET (depth, n);
  GET (nb_outputs, n);
  GET (term_kw, n);
  GET (out_link, n);
  GET (depth_start, (size_t)in.lmax + 2);
  GET (kw_state, in.n_keywords);
  if (class_entries) {
    f->class_map = malloc ((size_t)class_entries * 2);
    if (!f->class_map)
      bad = 1;
    else
      memcpy (f->class_map, p, (size_t)class_entries * 2);
    p += (size_t)class_entries * 2;
    GET (edge_letter, E);
  }
  if (n_keys64) {
    f->keys64 = malloc ((size_t)n_keys64 * 8);
    if (!f->keys64)
      bad = 1;
    else
      memcpy (f->keys64, p, (size_t)n_keys64 * 8);
    p += (size_t)n_keys64 * 8;
    f->n_keys64 = (uint32_t)n_keys64;
  }
#undef GET
  if (bad) {
    acm_flat_release (f);
    return ACM_GPU_E_NOMEM;
  }
  f->info = in;
  f->class_entries = class_entries;

  /* ---- shape of the goto function: breadth-first numbering means edge e leads to state e + 1 */
  int ok = f->row_ptr[0] == 0 && f->row_ptr[n] == E;
  for (uint32_t s = 0; ok && s < n; s++)
    ok = f->row_ptr[s] <= f->row_ptr[s + 1] && f->row_ptr[s + 1] <= E;
  const uint64_t sym_limit = in.sym_bytes == 8 ? n_keys64 + 1 : (in.sym_bytes == 4 ? (1ull << 32) : (1ull << (8 * in.sym_bytes)));
  for (uint64_t k = 1; ok && k < n_keys64; k++) /* 8-byte symbols: strictly ascending table, ids from 1 */
    ok = f->keys64[k - 1] < f->keys64[k];
  uint32_t lo = UINT32_MAX, hi = 0;
  for (uint32_t s = 0; ok && s < n; s++)
    for (uint32_t e = f->row_ptr[s]; ok && e < f->row_ptr[s + 1]; e++) {
      ok = f->edge_next[e] == e + 1 && f->edge_sym[e] < sym_limit && (in.sym_bytes != 8 || f->edge_sym[e] >= 1) &&
           (e == f->row_ptr[s] || f->edge_sym[e - 1] < f->edge_sym[e]);
      if (f->edge_sym[e] < lo)
        lo = f->edge_sym[e];
      if (f->edge_sym[e] > hi)
        hi = f->edge_sym[e];
    }
  /* ---- everything else is a function of (row_ptr, edge_sym, term_kw): recompute and compare */
  uint32_t *parent = ok ? malloc ((size_t)n * 4) : NULL;
  if (ok && !parent) {
    acm_flat_release (f);
    return ACM_GPU_E_NOMEM;
  }
  uint32_t lmax = 0, nkw = 0, max_out = 0;
  if (ok) {
    parent[0] = 0;
    for (uint32_t s = 0; s < n; s++)
      for (uint32_t e = f->row_ptr[s]; e < f->row_ptr[s + 1]; e++)
        parent[e + 1] = s;
    ok = f->depth[0] == 0 && f->fail[0] == 0 && f->out_link[0] == 0;
    for (uint32_t s = 1; ok && s < n; s++) {
      const uint32_t par = parent[s], c = f->edge_sym[s - 1];
      ok = par < s && f->depth[s] == f->depth[par] + 1;
      uint32_t fs = 0;
      if (ok && par != 0) { /* f(s) = delta(f(parent), c); children of the root fail to the root */
        uint32_t t = f->fail[par];
        for (;;) {
          const uint32_t ch = csr_child (f->row_ptr, f->edge_sym, t, c);
          if (ch != UINT32_MAX) {
            fs = ch;
            break;
          }
          if (t == 0)
            break;
          t = f->fail[t];
        }
      }
      ok = ok && f->fail[s] == fs && fs < s;
      if (ok) {
        const uint32_t want_link = fs == 0 ? 0 : (f->term_kw[fs] != UINT32_MAX ? fs : f->out_link[fs]);
        ok = f->out_link[s] == want_link;
      }
    }
    for (uint32_t s = 0; ok && s < n; s++) {
      const uint32_t term = f->term_kw[s] != UINT32_MAX;
      ok = (s != 0 || !term) && f->nb_outputs[s] == term + (s ? f->nb_outputs[f->fail[s]] : 0) &&
           (!term || f->term_kw[s] < in.n_keywords);
      if (s && f->depth[s] < f->depth[s - 1])
        ok = 0;
      if (f->depth[s] > lmax)
        lmax = f->depth[s];
      nkw += term;
      if (f->nb_outputs[s] > max_out)
        max_out = f->nb_outputs[s];
    }
    ok = ok && lmax == in.lmax && nkw == in.n_keywords && max_out == in.max_outputs;
    /* keyword ids are a permutation of [0, K) and kw_state is its inverse */
    for (uint32_t k = 0; ok && k < in.n_keywords; k++)
      ok = f->kw_state[k] < n && f->term_kw[f->kw_state[k]] == k;
    for (uint32_t d = 0, i = 0; ok && d <= lmax + 1; d++) {
      while (i < n && f->depth[i] < d)
        i++;
      ok = f->depth_start[d] == i;
    }
    if (ok && E && in.sym_bytes == 1)
      ok = in.alpha_lo == lo && in.alpha_span == hi - lo + 1 &&
           in.width == (in.alpha_span == 256 ? 256u : in.alpha_span + 1);
    else if (ok)
      ok = in.alpha_lo == (E ? lo : 0) && in.alpha_span == 0 && in.width == (in.sym_bytes == 1 ? 1u : 0u);
  }
  free (parent);
  if (ok && class_entries) {
    /* class ids are dense from 0, and every edge carries the class of its own letter */
    uint32_t top = 0;
    for (uint32_t v = 0; v < class_entries; v++)
      if (f->class_map[v] > top)
        top = f->class_map[v];
    f->n_classes = top + 1;
    unsigned char *seen = calloc (f->n_classes, 1);
    if (!seen) {
      acm_flat_release (f);
      return ACM_GPU_E_NOMEM;
    }
    for (uint32_t v = 0; v < class_entries; v++)
      seen[f->class_map[v]] = 1;
    for (uint32_t c = 0; ok && c < f->n_classes; c++)
      ok = seen[c];
    free (seen);
    for (uint32_t e = 0; ok && e < E; e++)
      ok = f->edge_letter[e] < class_entries && f->class_map[f->edge_letter[e]] == f->edge_sym[e];
  }
  if (!ok) {
    acm_flat_release (f);
    return ACM_GPU_E_FORMAT;
  }
  *out = f;
  return ACM_GPU_OK;
}

int
acm_flat_save (const ACMFlat *f, const char *path) {
  if (!f || !path)
    return ACM_GPU_E_ARG;
  const size_t bytes = acm_flat_blob_bytes (f);
  void *buf = malloc (bytes);
  if (!buf)
    return ACM_GPU_E_NOMEM;
  int rc = acm_flat_to_blob (f, buf, bytes);
  if (rc == ACM_GPU_OK) {
    FILE *fp = fopen (path, "wb");
    if (!fp || fwrite (buf, 1, bytes, fp) != bytes)
      rc = ACM_GPU_E_IO;
    if (fp && fclose (fp) != 0)
      rc = ACM_GPU_E_IO;
  }
  free (buf);
  return rc;
}

int
acm_flat_load (const char *path, ACMFlat **out) {
  if (!path || !out)
    return ACM_GPU_E_ARG;
  *out = NULL;
  FILE *fp = fopen (path, "rb");
  if (!fp)
    return ACM_GPU_E_IO;
  int rc = ACM_GPU_OK;
  void *buf = NULL;
  long len = -1;
  if (fseek (fp, 0, SEEK_END) != 0 || (len = ftell (fp)) < 0 || fseek (fp, 0, SEEK_SET) != 0)
    rc = ACM_GPU_E_IO;
  if (rc == ACM_GPU_OK && !(buf = malloc (len ? (size_t)len : 1)))
    rc = ACM_GPU_E_NOMEM;
  if (rc == ACM_GPU_OK && fread (buf, 1, (size_t)len, fp) != (size_t)len)
    rc = ACM_GPU_E_IO;
  fclose (fp);
  if (rc == ACM_GPU_OK)
    rc = acm_flat_from_blob (buf, (size_t)len, out);
  free (buf);
  return rc;
}

/* Spelling of keyword kw_id read back from the tables alone (each state but the root has one
 * incoming goto edge: edge e leads to state e + 1): `symbols` receives min(length, capacity)
 * symbols of sym_bytes bytes each, front to back; *length the keyword's length.  This is what
 * acm_get_match's letters[] spell (reference :472-479), for consumers that only hold a blob. */
int
acm_flat_keyword (const ACMFlat *f, uint32_t kw_id, void *symbols, uint32_t capacity, uint32_t *length) {
  if (!f || kw_id >= f->info.n_keywords || (!symbols && capacity))
    return ACM_GPU_E_ARG;
  uint32_t s = f->kw_state[kw_id];
  const uint32_t len = f->depth[s], sb = f->info.sym_bytes;
  if (length)
    *length = len;
  if (f->keys64) { /* 8-byte symbols: edge_sym is 1 + the rank in keys64 */
    for (uint32_t d = len; d > 0; d--) {
      const uint32_t e = s - 1;
      if (d - 1 < capacity) {
        const uint64_t v = f->keys64[f->edge_sym[e] - 1];
        unsigned char *o = (unsigned char *)symbols + (size_t)(d - 1) * 8;
        for (uint32_t i = 0; i < 8; i++)
          o[i] = (unsigned char)(v >> (8 * i));
      }
      uint32_t lo = 0, hi = f->info.n_states;
      while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (f->row_ptr[mid] <= e)
          lo = mid;
        else
          hi = mid;
      }
      s = lo;
    }
    return ACM_GPU_OK;
  }
  for (uint32_t d = len; d > 0; d--) {
    /* parent of s: the row that contains edge s - 1 */
    const uint32_t e = s - 1;
    if (d - 1 < capacity) {
      const uint32_t v = f->edge_letter ? f->edge_letter[e] : f->edge_sym[e]; /* the dictionary's own spelling */
      unsigned char *o = (unsigned char *)symbols + (size_t)(d - 1) * sb;
      for (uint32_t i = 0; i < sb; i++)
        o[i] = (unsigned char)(v >> (8 * i));
    }
    uint32_t lo = 0, hi = f->info.n_states; /* last s' with row_ptr[s'] <= e */
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (f->row_ptr[mid] <= e)
        lo = mid;
      else
        hi = mid;
    }
    s = lo;
  }
  return ACM_GPU_OK;
}
