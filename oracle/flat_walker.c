/*
 * flat_walker.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU walkers over the flattened tables of include/acm_gpu.h (ACMFlatView and
 * acm_flat_dense_rows).  They let the CPU test-suite check the tables the GPU kernels consume
 * against the oracle without a GPU.  They take raw array pointers so that nothing here links
 * against the product library.  Emission order = the reference's acm_get_match index order
 * (aho_corasick.c:459-466): the state itself if terminal, then successive terminal states down
 * the failure chain.
 */
#include "ac_oracle.h"

static uint64_t
emit_chain (uint32_t s, uint64_t pos, const uint32_t *term_kw, const uint32_t *out_link,
            const uint32_t *depth, orc_record *out, uint64_t cap, uint64_t found) {
  uint32_t t = term_kw[s] != 0xFFFFFFFFu ? s : out_link[s];
  while (t) {
    if (out && found < cap) {
      out[found].end_pos = pos;
      out[found].length = depth[t];
      out[found].keyword_id = term_kw[t];
    }
    found++;
    t = out_link[t];
  }
  return found;
}

static inline uint32_t
load_sym (const unsigned char *p, uint64_t i, uint32_t sym_bytes) {
  uint32_t v = 0;
  for (uint32_t b = 0; b < sym_bytes; b++)
    v |= (uint32_t)p[i * sym_bytes + b] << (8 * b);
  return v;
}

/* goto / failure walk over the CSR arrays (reference state_goto, aho_corasick.c:167-192) */
uint64_t
flatwalk_csr (const uint32_t *row_ptr, const uint32_t *edge_sym, const uint32_t *edge_next,
              const uint32_t *fail, const uint32_t *nb_outputs, const uint32_t *term_kw,
              const uint32_t *out_link, const uint32_t *depth, const void *text, uint64_t n,
              uint32_t sym_bytes, uint64_t emit_from, uint64_t pos_base, orc_record *out, uint64_t cap) {
  uint32_t s = 0;
  uint64_t found = 0;
  for (uint64_t i = 0; i < n; i++) {
    uint32_t c = load_sym (text, i, sym_bytes);
    for (;;) {
      uint32_t e = row_ptr[s], end = row_ptr[s + 1], nx = 0xFFFFFFFFu;
      for (; e < end; e++)
        if (edge_sym[e] == c) {
          nx = edge_next[e];
          break;
        }
      if (nx != 0xFFFFFFFFu) {
        s = nx;
        break;
      }
      if (s == 0)
        break;
      s = fail[s];
    }
    if (nb_outputs[s] && i >= emit_from)
      found = emit_chain (s, pos_base + i, term_kw, out_link, depth, out, cap, found);
  }
  return found;
}

/* one table lookup per symbol over failure-resolved rows (acm_flat_dense_rows) */
uint64_t
flatwalk_dense (const void *rows, uint32_t entry_bytes, uint32_t width, uint32_t lo, uint32_t span,
                const uint32_t *term_kw, const uint32_t *out_link, const uint32_t *depth,
                const unsigned char *text, uint64_t n, uint64_t emit_from, uint64_t pos_base,
                orc_record *out, uint64_t cap) {
  uint32_t s = 0;
  uint64_t found = 0;
  const uint32_t flag = entry_bytes == 2 ? 0x8000u : 0x80000000u;
  for (uint64_t i = 0; i < n; i++) {
    uint32_t cls = (uint32_t)text[i] - lo;
    if (cls > span)
      cls = span; /* unsigned wrap: below lo also lands here */
    if (cls >= width)
      cls = width - 1;
    uint32_t e = entry_bytes == 2 ? ((const uint16_t *)rows)[(uint64_t)s * width + cls]
                                  : ((const uint32_t *)rows)[(uint64_t)s * width + cls];
    s = e & (flag - 1);
    if ((e & flag) && i >= emit_from)
      found = emit_chain (s, pos_base + i, term_kw, out_link, depth, out, cap, found);
  }
  return found;
}
