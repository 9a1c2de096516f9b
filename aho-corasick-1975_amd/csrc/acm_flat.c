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

#include <stdlib.h>
#include <string.h>

struct ACMFlat {
  ACMFlatInfo info;
  uint32_t *row_ptr, *edge_sym, *edge_next, *fail, *depth, *nb_outputs, *term_kw, *out_link;
  uint32_t *depth_start, *kw_state;
};

static uint32_t
symbol_value (const void *letter, uint32_t sym_bytes) {
  const unsigned char *p = letter;
  uint32_t v = 0;
  for (uint32_t i = 0; i < sym_bytes; i++)
    v |= (uint32_t)p[i] << (8 * i);
  return v;
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
  free (f);
}

int
acm_flatten (ACMachine *machine, ACMFlat **out) {
  if (!machine || !out)
    return ACM_GPU_E_ARG;
  uint32_t sym_bytes = 0;
  int rc = acm_internal_symbol_bytes (machine, &sym_bytes);
  if (rc)
    return rc;

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

  /* breadth-first renumbering; the queue is `order` itself */
  uint32_t head = 0, tail = 0, edges = 0, lmax = 0, nkw = 0, max_out = 0;
  order[tail++] = acm_internal_root (machine);
  newid[order[0]->id] = 0;
  while (head < tail) {
    struct _ac_state *s = order[head];
    f->row_ptr[head] = edges;
    for (uint32_t i = 0; i < s->nkids; i++) {
      struct _ac_state *k = s->kids[i];
      newid[k->id] = tail;
      f->edge_sym[edges] = symbol_value (k->letter, sym_bytes);
      f->edge_next[edges] = tail;
      edges++;
      order[tail++] = k;
    }
    if (sym_bytes > 1 && s->nkids > 1) {
      /* the comparator orders multi-byte symbols by memcmp; the device bisects rows by numeric
       * value, so re-order this row (and the ids just handed out) by value */
      if (sort_row_by_value (f->edge_sym + f->row_ptr[head], order + f->edge_next[f->row_ptr[head]], s->nkids))
        goto nomem;
      for (uint32_t i = 0; i < s->nkids; i++)
        newid[order[f->edge_next[f->row_ptr[head]] + i]->id] = f->edge_next[f->row_ptr[head]] + i;
    }
    head++;
  }
  f->row_ptr[n] = edges;

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
  *out = f;
  return ACM_GPU_OK;

nomem:
  acm_internal_unlock (machine);
  free (order);
  free (newid);
  acm_flat_release (f);
  return ACM_GPU_E_NOMEM;
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
