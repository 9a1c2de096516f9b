/*
 * acm_gpu.hip -- the bulk scan (include/acm_gpu.h) for MI355X / gfx950: device plan, launches,
 * streaming, canonical sort and the C ABI; the kernels are in the dev_*.h files included below
 * (one translation unit).
 *
 * What runs on the device is the reference's caller loop (examples/test.c:17-23): per input
 * symbol one automaton step (acm_match -> state_goto, aho_corasick.c:434-448,167-192) and, when
 * the new state has outputs, the walk over the keyword-terminal states of its failure chain
 * (acm_get_match, aho_corasick.c:459-466), one 16-byte record per match.
 *
 * Kernels
 *   dev_dense.h   scan_dense_kernel   byte alphabets, <= 32,768 states: failure-resolved rows of the
 *                                     hot states in LDS, one ds_read_u16 per symbol, continuation
 *                                     items for the rest (config 2: the headline kernel); sticky
 *                                     mode with rows in HBM for bigger dictionaries
 *   dev_gram.h    scan_gram_kernel    byte alphabets, big dictionaries of keywords >= 4 symbols:
 *                                     4-gram bit table in LDS, every position tested on its own
 *   dev_starts.h  scan_starts_kernel  2- and 4-byte symbols: root table by symbol value in LDS,
 *                                     every position tested on its own; walk_starts, hit parking
 *   dev_sparse.h  scan_sparse_kernel  2- and 4-byte symbols, automaton walk (ACM_GPU_SPARSE=walk)
 *   dev_csr.h     scan_csr_kernel     any width and alignment: goto/failure walk over CSR rows
 *   dev_emit.h    queue items -> records: put_outputs, walk_continuation, flush_queue,
 *                 expand_items_kernel (block-wide prefix sum, one atomic per round)
 *   dev_misc.h    classmap (comparator classes), patch (incremental updates), sort keys, synthetic text
 *   dev_order.h   canonical order by position buckets + LDS sorts (three passes instead of a radix sort's eight)
 * No MFMA anywhere: this is byte/integer table walking bound by LDS lookups and HBM reads.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "acm_internal.h"

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      fprintf (stderr, "acm_gpu: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString (_e), __FILE__, \
               __LINE__);                                                                          \
      return ACM_GPU_E_HIP;                                                                        \
    }                                                                                              \
  } while (0)

#include "dev_all.h"

/* ====================================================================== host side */
/* Host mirror of the start-parallel tables of a plan that takes incremental updates
 * (acm_gpu_plan_update).  The start-parallel kernel walks the goto function only -- no failure
 * links, no output counts -- so a new keyword touches just the states on its own path: the new
 * ones are appended (ids in order of creation after the breadth-first ids of the plan's build),
 * the state they hang off gets one more edge, and the root table / pair table entries of at most
 * two symbols change.  The changed words are collected as patches and written by patch_kernel on
 * the stream of the next scan, in front of it. */
enum { PT_REC = 0, PT_EDGE = 1, PT_LUT = 2, PT_PAIRS = 3, PT_OINFO = 4 };
struct StartsMirror {
  std::vector<uint32_t> tab[5];           /* rec: 8 per state, edge: 2 per slot, lut, pairs: 2 per state, oinfo: 4 per state */
  std::vector<uint32_t> parent, parent_sym, depth; /* host only, per state */
  uint32_t *dev[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
  size_t cap[5] = { 0, 0, 0, 0, 0 };      /* device capacity in words */
  bool own = false;                       /* device arrays of their own (else still inside the plan's blob) */
  bool full_upload = false;
  std::vector<uint4> patches;
  uint4 *d_patches = nullptr;
  size_t cap_patches = 0;
  uint32_t n_states = 0, lmax = 0, n_keywords = 0, n_edges = 0;

  void
  set (int t, size_t idx, uint32_t v) {
    if (idx >= tab[t].size ())
      tab[t].resize (idx + 1, 0);
    if (tab[t][idx] == v)
      return;
    tab[t][idx] = v;
    if (idx >= cap[t] || !own)
      full_upload = true; /* beyond what the device holds (or still in the blob): everything goes up again */
    else if (!full_upload)
      patches.push_back (make_uint4 ((uint32_t)t, (uint32_t)idx, v, 0));
  }
};

struct ACMPlan {
  int device = 0;
  ACMFlatInfo finfo{};
  ACMPlanInfo info{};
  void *blob = nullptr; /* one device allocation holding every table */
  size_t blob_bytes = 0;
  /* CSR kernel (breadth-first numbering) */
  CsrTables csr{};
  const uint4 *d_oinfo = nullptr;
  const uint32_t *d_kw4 = nullptr;
  const uint16_t *d_cont_dh = nullptr;
  const uint4 *d_chain = nullptr; /* EmitCtx::chain */
  /* sparse kernel (2- and 4-byte symbols) */
  SparseK SK{};
  StartsK TK{};
  bool starts = false; /* start-parallel kernel instead of the sparse walk */
  StartsMirror *mir = nullptr; /* starts plans: what acm_gpu_plan_update edits */
  GramK GK{};
  bool gram = false; /* 4-gram sieve kernel instead of the sticky dense walk */
  bool gram_shorts = false, gram_wide = false;
  bool gram2 = false; /* scan_gram2_kernel (dev_gram2.h) instead of scan_gram_kernel */
  bool short_pass = false; /* narrow alphabets: the keywords of 1-3 symbols in a pass of their own (scan_short_kernel, dev_short.h) */
  uint32_t short_lds_bytes = 0, short_lds_count_bytes = 0; /* scan_short_kernel: with records / count-only (the nibbles alone) */
  bool short_ids_lds = false; /* scan_short_kernel: the keyword ids are in LDS (else read by rank from the image in HBM) */
  uint32_t short_blocks_per_cu = 1; /* scan_short_kernel, count-only: two blocks a CU when its LDS allows (8 waves a SIMD) */
  uint32_t gram_lds_bytes = 0;
  uint32_t class_sym_bytes = 0; /* comparator-class plans: the symbol size they were made for */
  bool sparse = false, sparse_lut_lds = false, starts_lut_lds = false;
  uint32_t sparse_lds_bytes = 0, starts_lds_bytes = 0;
  /* dense kernel (breadth-first numbering too: the LDS rows are a breadth-first prefix) */
  DenseK K{};
  const void *d_dense = nullptr;     /* failure-resolved rows of every state */
  const void *d_lds_image = nullptr; /* what every workgroup copies into LDS */
  const uint32_t *d_wrows = nullptr;
  const uint32_t *d_dstart = nullptr;
  uint32_t lds_image_bytes = 0;
  uint32_t entry_bytes = 0, streams = ACM_DENSE_S, chunk = 64;
  /* item buffer of the dense kernel: regions x region_items items of 8 B, one region per wave */
  void *d_items = nullptr;
  uint32_t *d_fill = nullptr;   /* per region, zero between launches */
  unsigned long long *d_total = nullptr; /* [0] running total of a scan, [1] low word = expand ticket; zero between scans */
  unsigned int *d_pool_ctr = nullptr;    /* 2 x POOL_CLASSES tile-pool counters (same allocation), alternating per launch */
  uint32_t launch_seq = 0;
  /* comparator-class plans: the text is mapped to class ids into d_remap before every scan */
  uint16_t *d_classlut = nullptr;
  /* 8-byte symbols: hash table {key, id} of the dictionary's symbols; the text is interned to
   * 4-byte ids into d_remap and finfo.sym_bytes is 4 (what the kernels walk) */
  uint4 *d_intern = nullptr;
  uint32_t intern_mask = 0;
  uint32_t text_sym_bytes = 0; /* symbol size of the caller's text (== finfo.sym_bytes unless interned) */
  /* comparator classes of 4-byte symbols (ACMFlatView::keys32): every symbol classified so far ->
   * class, on the host and as the device's table {symbol, class + 1}; the text is mapped to class
   * ids into d_remap before every scan and symbols met for the first time are classified on the
   * host with the machine's comparator (cmp32; classify_text32) */
  bool cls32 = false;
  std::unordered_map<uint32_t, uint32_t> cls32_known;
  std::vector<uint32_t> cls32_reps; /* one symbol per class, comparator order: class i + 1 */
  CMP_TYPE cmp32 = nullptr;
  void *cmp32_arg = nullptr;
  unsigned long long *d_cls32 = nullptr;
  uint32_t cls32_slots = 0, cls32_uploaded = 0; /* table size; known symbols it holds */
  uint32_t *d_unknown = nullptr; /* [0] count, [1 .. cls32_cap] symbols */
  /* symbols one pass can hand to the host: doubled (up to CLS32_CAP_MAX) whenever a pass fills the
   * list, so that a text of many distinct symbols takes a few passes, not one per 65,536 of them;
   * the plan refuses texts that bring more than CLS32_KNOWN_MAX distinct symbols in all (the host
   * map and the device table hold every one of them) */
  static constexpr uint32_t CLS32_CAP_MIN = 1u << 16, CLS32_CAP_MAX = 1u << 22, CLS32_KNOWN_MAX = 1u << 25;
  uint32_t cls32_cap = CLS32_CAP_MIN;
  void *d_remap = nullptr;
  size_t remap_bytes = 0;
  uint32_t regions = 0, region_items = 0;
  /* 4-gram kernel, narrow alphabets (records straight from the scan kernel): one hole descriptor
   * per wave and the spill area, one chunk of records per wave (64 MB on 256 CUs) */
  void *d_holes = nullptr, *d_spill = nullptr;
  uint32_t spill_chunk = 0; /* slots per chunk the spill area was sized for */
  uint32_t holes_waves[2] = { 0, 0 }; /* waves of the widest launch of the scan at hand (4-gram pass, short-keyword pass): the holes to close */
  uint32_t direct_regions = 0;
  /* a tiled scan in progress (acm_gpu_scan_ordered_device -> scan_tiled): the 4-gram kernel writes
   * a TileEntry per tile from tiled_dir[tiled_base] on and links its chunks in tiled_prev */
  void *tiled_dir = nullptr; /* TileEntry[] */
  uint32_t *tiled_prev = nullptr;
  uint32_t tiled_base = 0;

  uint64_t segment = SEGMENT;
  uint64_t generation = 0; /* for the machine-cached plan */
  /* Dictionary growth without a rebuild (acm_gpu_plan_update): the keywords a machine got after
   * this plan was made live in a small second plan, `delta`, scanned right after this one into the
   * same record buffer (the match set of a dictionary is the union of its keywords' match sets;
   * the delta's keyword ids start at kw_base).  A replaced delta waits in `retired` until the
   * stream that may still be scanning with it has passed an event. */
  ACMPlan *delta = nullptr;
  uint64_t delta_scanned = 0;    /* symbols scanned twice (plan, then delta) since this delta's first keyword came */
  uint32_t covered_keywords = 0; /* keywords of the machine this plan and its delta report */
  uint32_t kw_base = 0;          /* added to the keyword ids of this plan's records (delta plans) */
  uint32_t merges = 0;           /* updates that rebuilt everything */
  ACMPlan *items_owner = nullptr; /* delta plans: the plan whose item buffer they park in (scans of the two never overlap) */
  struct Retired {
    ACMPlan *plan;
    hipEvent_t done; /* nullptr until the first scan after the retirement has recorded it */
  };
  std::vector<Retired> retired;
  int cu_count = 0;
  /* timing */
  bool timing = false;
  uint32_t timing_every = 1, timing_seq = 0; /* events around every timing_every-th launch (acm_gpu_plan_timing) */
  struct LaunchEvents {
    hipEvent_t start, scan_done, all_done; /* around the scan kernel; after what follows it (expansion / hole closing) */
  };
  std::vector<LaunchEvents> events;
  size_t events_used = 0;
  double timing_ms = 0, timing_all_ms = 0;
  uint64_t timing_launches = 0;
};

extern "C" const char *
acm_gpu_strerror (int code) {
  switch (code) {
  case ACM_GPU_OK: return "ok";
  case ACM_GPU_E_INELIGIBLE: return "machine not eligible for the GPU path (needs ACM_CMP_DEFAULT over 1/2/4-byte symbols)";
  case ACM_GPU_E_NODEVICE: return "no usable HIP device";
  case ACM_GPU_E_HIP: return "HIP runtime error";
  case ACM_GPU_E_OVERFLOW: return "record buffer too small";
  case ACM_GPU_E_ARG: return "invalid argument";
  case ACM_GPU_E_NOMEM: return "out of memory";
  case ACM_GPU_E_INTERNAL: return "internal consistency check failed on the device";
  case ACM_GPU_E_FORMAT: return "not a valid flat-table blob";
  case ACM_GPU_E_IO: return "file could not be read or written";
  case ACM_GPU_E_COMM: return "librccl.so could not be loaded or an RCCL call failed";
  default: return "unknown error";
  }
}

extern "C" int
acm_gpu_device_count (void) {
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess)
    return 0;
  return n;
}

namespace {

size_t
blob_reserve (size_t &cursor, size_t bytes) {
  cursor = (cursor + 255) & ~(size_t)255;
  size_t at = cursor;
  cursor += bytes;
  return at;
}

template <typename ENTRY, int C, int S, bool CO>
const void *
dense_fn () {
  return reinterpret_cast<const void *> (&scan_dense_kernel<ENTRY, C, S, CO>);
}

/* one geometry is built: 64-byte chunks, 2 streams per lane (the template takes others) */
const void *
dense_kernel_ptr (uint32_t entry_bytes, uint32_t chunk, uint32_t streams, bool count_only) {
  (void)chunk;
  (void)streams;
  if (entry_bytes == 2)
    return count_only ? dense_fn<uint16_t, 64, ACM_DENSE_S, true> () : dense_fn<uint16_t, 64, ACM_DENSE_S, false> ();
  return count_only ? dense_fn<uint32_t, 64, ACM_DENSE_S, true> () : dense_fn<uint32_t, 64, ACM_DENSE_S, false> ();
}

template <typename SYM>
const void *
sparse_fn (bool lut_lds, bool count_only) {
  if (lut_lds)
    return count_only ? reinterpret_cast<const void *> (&scan_sparse_kernel<SYM, true, true>)
                      : reinterpret_cast<const void *> (&scan_sparse_kernel<SYM, true, false>);
  return count_only ? reinterpret_cast<const void *> (&scan_sparse_kernel<SYM, false, true>)
                    : reinterpret_cast<const void *> (&scan_sparse_kernel<SYM, false, false>);
}

const void *
sparse_kernel_ptr (uint32_t sym_bytes, bool lut_lds, bool count_only) {
  return sym_bytes == 2 ? sparse_fn<uint16_t> (lut_lds, count_only) : sparse_fn<uint32_t> (lut_lds, count_only);
}

template <typename SYM>
const void *
starts_fn (bool lut_lds, bool count_only) {
  if (lut_lds)
    return count_only ? reinterpret_cast<const void *> (&scan_starts_kernel<SYM, true, true>)
                      : reinterpret_cast<const void *> (&scan_starts_kernel<SYM, true, false>);
  return count_only ? reinterpret_cast<const void *> (&scan_starts_kernel<SYM, false, true>)
                    : reinterpret_cast<const void *> (&scan_starts_kernel<SYM, false, false>);
}

const void *
short_kernel_ptr (bool count_only, bool ids_lds) {
  if (count_only)
    return reinterpret_cast<const void *> (&scan_short_kernel<true, true>);
  return ids_lds ? reinterpret_cast<const void *> (&scan_short_kernel<false, true>) : reinterpret_cast<const void *> (&scan_short_kernel<false, false>);
}

const void *
gram_kernel_ptr (bool count_only, bool shorts, bool wide, bool tiled = false, bool gram2 = false) {
  if (gram2) {
    if (tiled)
      return reinterpret_cast<const void *> (&scan_gram2_kernel<false, true>);
    return count_only ? reinterpret_cast<const void *> (&scan_gram2_kernel<true, false>) : reinterpret_cast<const void *> (&scan_gram2_kernel<false, false>);
  }
  /* (keywords of 1-3 symbols inside the kernel: hashed windows only -- narrow alphabets give them a
   * pass of their own, scan_short_kernel; the instantiations that did both spilled 140 to 320 vector registers) */
  if (tiled) /* (narrow alphabets, record mode) */
    return reinterpret_cast<const void *> (&scan_gram_kernel<false, false, false, true>);
  if (wide && shorts)
    return count_only ? reinterpret_cast<const void *> (&scan_gram_kernel<true, true, true, false>)
                      : reinterpret_cast<const void *> (&scan_gram_kernel<false, true, true, false>);
  if (wide)
    return count_only ? reinterpret_cast<const void *> (&scan_gram_kernel<true, false, true, false>)
                      : reinterpret_cast<const void *> (&scan_gram_kernel<false, false, true, false>);
  return count_only ? reinterpret_cast<const void *> (&scan_gram_kernel<true, false, false, false>)
                    : reinterpret_cast<const void *> (&scan_gram_kernel<false, false, false, false>);
}

const void *
starts_kernel_ptr (uint32_t sym_bytes, bool lut_lds, bool count_only) {
  return sym_bytes == 2 ? starts_fn<uint16_t> (lut_lds, count_only) : starts_fn<uint32_t> (lut_lds, count_only);
}

void
drop_cached_plan (void *p) {
  acm_gpu_plan_destroy (static_cast<ACMPlan *> (p));
}

} // namespace

namespace {
/* ---- table builders of acm_gpu_plan_create_flat (host images of the device tables) */

/* sparse automaton walk (dev_sparse.h): state records, (symbol, next | out flag) edges, root table */
void
fill_sparse_tables (const ACMFlatView &fv, const ACMFlatInfo &fi, uint32_t lut_size, uint32_t *rec, uint32_t *edge, uint32_t *lut) {
  const uint32_t n = fi.n_states;
  auto entry = [&] (uint32_t e) { return fv.edge_next[e] | (fv.nb_outputs[fv.edge_next[e]] ? 0x80000000u : 0u); };
  for (uint32_t e = 0; e < fi.n_edges; e++) {
    edge[2 * e] = fv.edge_sym[e];
    edge[2 * e + 1] = entry (e);
  }
  for (uint32_t st = 0; st < n; st++) {
    const uint32_t b = fv.row_ptr[st], ne = fv.row_ptr[st + 1] - b;
    uint32_t *r = rec + 8 * (size_t)st;
    r[0] = fv.fail[st];
    r[1] = ne;
    r[2] = b;
    r[3] = 0;
    r[4] = ne >= 1 ? fv.edge_sym[b] : 0;
    r[5] = ne >= 1 ? entry (b) : 0;
    r[6] = ne >= 2 ? fv.edge_sym[b + 1] : 0;
    r[7] = ne >= 2 ? entry (b + 1) : 0;
  }
  for (uint32_t e = 0; e < fv.row_ptr[1]; e++)
    if (fv.edge_sym[e] < lut_size)
      lut[fv.edge_sym[e]] = entry (e);
}

/* start-parallel kernel (dev_starts.h): trie records with terminal flags, plain edges, root table
 * with SECOND / ALWAYS flags, first-two-edge-symbols of the root's children */
void
fill_starts_tables (const ACMFlatView &fv, const ACMFlatInfo &fi, uint32_t lut_size, uint32_t *rec, uint32_t *edge, uint32_t *lut,
                    uint32_t *pairs) {
  const uint32_t n = fi.n_states;
  for (uint32_t e = 0; e < fi.n_edges; e++) {
    edge[2 * e] = fv.edge_sym[e];
    edge[2 * e + 1] = fv.edge_next[e];
  }
  for (uint32_t st = 0; st < n; st++) {
    const uint32_t b = fv.row_ptr[st], ne = fv.row_ptr[st + 1] - b;
    uint32_t *r = rec + 8 * (size_t)st;
    r[0] = 0;
    r[1] = ne;
    r[2] = b;
    r[3] = fv.term_kw[st] != NONE ? 1u : 0u;
    r[4] = ne >= 1 ? fv.edge_sym[b] : 0;
    r[5] = ne >= 1 ? fv.edge_next[b] : 0;
    r[6] = ne >= 2 ? fv.edge_sym[b + 1] : 0;
    r[7] = ne >= 2 ? fv.edge_next[b + 1] : 0;
  }
  const uint32_t root_edges = fv.row_ptr[1];
  {
    /* the root's row in the edge table: only the symbols the root table cannot hold (the tail
     * of the sorted row; usually nothing) -- the table itself answers for the others */
    uint32_t beyond = 0;
    while (beyond < root_edges && fv.edge_sym[root_edges - 1 - beyond] >= lut_size)
      beyond++;
    rec[1] = beyond;
    rec[2] = root_edges - beyond;
  }
  for (uint32_t st = 0; st <= root_edges; st++) {
    const uint32_t b = fv.row_ptr[st], ne = fv.row_ptr[st + 1] - b;
    pairs[2 * st] = ne >= 1 ? fv.edge_sym[b] : 0;
    pairs[2 * st + 1] = ne >= 2 ? fv.edge_sym[b + 1] : pairs[2 * st];
  }
  for (uint32_t e = 0; e < root_edges; e++)
    if (fv.edge_sym[e] < lut_size) {
      const uint32_t child = fv.edge_next[e];
      const uint32_t ne = fv.row_ptr[child + 1] - fv.row_ptr[child];
      /* keyword by itself, more edges than the pair shows, or a pair that cannot be told
       * from "no edge" (symbol 0 twice): never sieved out */
      const bool always = fv.term_kw[child] != NONE || ne > 2 || ne == 0;
      lut[fv.edge_sym[e]] = child | (always ? ST_ALWAYS : 0u);
    }
  /* second symbols: the edges that leave the root's children (states 1 .. root_edges) */
  for (uint32_t e = fv.row_ptr[1]; e < fv.row_ptr[root_edges + 1]; e++)
    if (fv.edge_sym[e] < lut_size)
      lut[fv.edge_sym[e]] |= ST_SECOND;
}

/* 4-gram sieve kernel (dev_gram.h): where its tables go and how they are keyed */
struct GramImage {
  bool wide, shorts;
  uint32_t W, bloom_log2, wtab_log2, stab_log2;
  uint32_t *bits, *g4, *rec, *edge, *g4gid; /* first-stage bits, second-stage records, trie records (depth-first), their edges, depth-4 state -> record */
  unsigned char *nib;                      /* narrow alphabets: nibble per 3-gram */
  uint32_t *prefix, *entry;                /* narrow alphabets: set bits before each word of `bits`; by rank {children mask | terminal << 31, first child's state id, keyword id} */
  uint32_t *peek;                          /* narrow alphabets: per depth-5 state {its record, the symbol of its only edge or GRAM_NO_PEEK} */
  uint32_t *bloom;                         /* narrow alphabets: Bloom bits, terminal 4-grams then 5-grams (GramK::bloom5_bits; NULL: none) */
  uint32_t bloomT_bits, bloom5_bits, lo, kw_base;
  bool kw_inline;                          /* keyword ids fit a hit's word (HIT_KW) */
  bool peek_packed;                        /* 4 bytes per depth-5 state (fewer than 2^23 records) instead of 8 */
  uint32_t *g3, *stab;                     /* short keywords: prefix states per 3-gram (narrow) / table of tagged windows (wide) */
  uint32_t *tab2;                          /* scan_gram2_kernel: two bits per 4-gram (GramK::tab2); NULL: not made */
  uint32_t *rows2, *over2;                 /* the second stage's entries by row of 16 4-grams (GramK::rows2), the rows' 8th and later entries */
};

void
fill_gram_tables (const ACMFlatView &fv, const ACMFlatInfo &fi, const GramImage &G) {
  const uint32_t n = fi.n_states;
  uint32_t *bits = G.bits, *g4 = G.g4, *rec = G.rec, *edge = G.edge, *g4gid = G.g4gid;
  {
    /* records of the states of depth >= 4 in depth-first (preorder) order, subtree after
     * subtree of the depth-4 states: the tail of a keyword is a run of consecutive 32-byte
     * records, so a walk touches one or two cache lines instead of one per symbol */
    std::vector<uint32_t> gid (n, 0), stack;
    uint32_t next_gid = 0;
    for (uint32_t root4 = fv.depth_start[4]; root4 < fv.depth_start[5]; root4++) {
      g4gid[root4 - fv.depth_start[4]] = next_gid;
      stack.push_back (root4);
      while (!stack.empty ()) {
        const uint32_t st = stack.back ();
        stack.pop_back ();
        gid[st] = next_gid++;
        for (uint32_t e = fv.row_ptr[st + 1]; e-- > fv.row_ptr[st];) /* first child on top */
          stack.push_back (fv.edge_next[e]);
      }
    }
    uint32_t slots = 0;
    for (uint32_t st = fv.depth_start[4]; st < n; st++) {
      const uint32_t b0 = fv.row_ptr[st], ne = fv.row_ptr[st + 1] - b0;
      uint32_t *r = rec + 8 * (size_t)gid[st];
      r[0] = st;
      r[1] = ne | (fv.depth[st] << 16); /* (byte alphabet: at most 256 edges; depth < 4,096: the dense eligibility test) */
      r[2] = slots;
      r[3] = fv.term_kw[st] != NONE ? fv.term_kw[st] + G.kw_base + 1u : 0u; /* terminal: keyword id + 1 (a record written on the spot needs it) */
      r[4] = ne >= 1 ? fv.edge_sym[b0] : 0;
      r[5] = ne >= 1 ? gid[fv.edge_next[b0]] : 0;
      r[6] = ne >= 2 ? fv.edge_sym[b0 + 1] : 0;
      r[7] = ne >= 2 ? gid[fv.edge_next[b0 + 1]] : 0;
      for (uint32_t e = b0; e < b0 + ne; e++) {
        edge[2 * (size_t)slots] = fv.edge_sym[e];
        edge[2 * (size_t)slots + 1] = gid[fv.edge_next[e]];
        slots++;
      }
    }
    /* what a walk that starts at a depth-5 state asks first (GramK::g5peek): where its record is,
     * and -- when the state is no keyword's end and has one way on -- the symbol that way takes,
     * so that 25 of 26 candidates end on 8 bytes that stay in L2 instead of a 32-byte record
     * from the 16 MB of them */
    const uint32_t d5_end = fi.lmax >= 5 ? fv.depth_start[6 <= fi.lmax + 1 ? 6 : fi.lmax + 1] : fv.depth_start[5];
    for (uint32_t st = fv.depth_start[5]; G.peek && fi.lmax >= 5 && st < d5_end; st++) {
      const uint32_t b0 = fv.row_ptr[st], ne = fv.row_ptr[st + 1] - b0;
      const uint32_t sym = (ne == 1 && fv.term_kw[st] == NONE) ? fv.edge_sym[b0] : GRAM_NO_PEEK;
      if (G.peek_packed) /* record | symbol << 23 | "look at the record" << 31 */
        G.peek[st - fv.depth_start[5]] = gid[st] | (sym == GRAM_NO_PEEK ? 0x80000000u : sym << 23);
      else {
        G.peek[2 * (size_t)(st - fv.depth_start[5])] = gid[st];
        G.peek[2 * (size_t)(st - fv.depth_start[5]) + 1] = sym;
      }
    }
  }
  /* base-W number of the path of every state down to depth 4 (parents come first in
   * breadth-first order); the depth-4 states are the 4-grams some keyword starts with */
  std::vector<uint32_t> path (fv.depth_start[5 <= fi.lmax + 1 ? 5 : fi.lmax + 1], 0);
  for (uint32_t st = 0; st < fv.depth_start[4]; st++)
    for (uint32_t e = fv.row_ptr[st]; e < fv.row_ptr[st + 1]; e++)
      path[fv.edge_next[e]] = G.wide ? path[st] | (fv.edge_sym[e] << (8 * fv.depth[st])) /* the 4 bytes as the text holds them */
                                        : path[st] * G.W + (fv.edge_sym[e] - fi.alpha_lo);
  for (uint32_t st = fv.depth_start[4]; G.wide && st < fv.depth_start[5]; st++) {
    const uint32_t win = path[st];
    const uint32_t hb = (win * WIDE_H1) >> (32 - G.bloom_log2);
    bits[hb >> 5] |= 1u << (hb & 31);
    uint32_t slot = (win * WIDE_H2) >> (32 - G.wtab_log2);
    while (g4[2 * (size_t)slot + 1])
      slot = (slot + 1) & ((1u << G.wtab_log2) - 1);
    g4[2 * (size_t)slot] = win;
    g4[2 * (size_t)slot + 1] = st | (fv.term_kw[st] != NONE ? WT_TERM : 0u) | (fv.row_ptr[st + 1] > fv.row_ptr[st] ? WT_KIDS : 0u);
  }
  uint32_t n_over2 = 0, row_over = 0;
  std::vector<uint8_t> row_fill (G.rows2 ? (G.W * G.W * G.W * G.W + 15) / 16 : 0, 0); /* (a row has 16 4-grams) */
  for (uint32_t st = fv.depth_start[4]; !G.wide && st < fv.depth_start[5]; st++) {
    const uint32_t idx = path[st];
    uint32_t mask = fv.term_kw[st] != NONE ? 0x80000000u : 0u;
    for (uint32_t e = fv.row_ptr[st]; e < fv.row_ptr[st + 1]; e++)
      mask |= 1u << (fv.edge_sym[e] - fi.alpha_lo);
    bits[idx >> 5] |= 1u << (idx & 31);
    if (G.rows2) {
      /* the entry of this 4-gram: slot r of its row when it is the r-th (< 7) 4-gram of the row that
       * exists (the depth-4 states come in ascending index order), else the next entry of the
       * overflow array, whose first index for the row stands in slot 7 */
      const uint32_t row = idx >> 4;
      const uint32_t r = row_fill[row]++;
      const bool term = fv.term_kw[st] != NONE, kids = fv.row_ptr[st + 1] > fv.row_ptr[st];
      uint32_t e0 = (mask & 0x3FFFFFFFu) | (term ? 0x80000000u : 0u), e1;
      if (term && kids) { /* a keyword of 4 symbols that others go on from: left to the walk, at its own record */
        e0 |= 0x40000000u;
        e1 = g4gid[st - fv.depth_start[4]];
      } else
        e1 = term ? fv.term_kw[st] + G.kw_base : fv.edge_next[fv.row_ptr[st]];
      if (r < 8) { /* (slot 7: the 8th entry itself while there is no 9th) */
        G.rows2[16 * (size_t)row + 2 * r] = e0;
        G.rows2[16 * (size_t)row + 2 * r + 1] = e1;
      }
      if (r >= 7) {
        if (r == 7)
          row_over = n_over2;
        if (r == 8) { /* nine or more: slot 7 says where the 8th and later entries are */
          G.rows2[16 * (size_t)row + 14] = 0x20000000u | row_over;
          G.rows2[16 * (size_t)row + 15] = 0;
        }
        G.over2[2 * (size_t)n_over2] = e0;
        G.over2[2 * (size_t)n_over2 + 1] = e1;
        n_over2++;
      }
    }
    if (G.tab2) {
      /* T of this 4-gram; H of it when it is a keyword; H of the tails of the 5-grams below it */
      const bool term = fv.term_kw[st] != NONE;
      G.tab2[idx >> 4] |= (term ? 3u : 1u) << (2 * (idx & 15));
      const uint32_t W3 = G.W * G.W * G.W;
      for (uint32_t e = fv.row_ptr[st]; e < fv.row_ptr[st + 1]; e++) {
        const uint32_t tail = (idx % W3) * G.W + (fv.edge_sym[e] - fi.alpha_lo);
        G.tab2[tail >> 4] |= 2u << (2 * (tail & 15));
      }
    }
    g4[2 * (size_t)idx] = mask;
    g4[2 * (size_t)idx + 1] = st;
    G.entry[3 * (size_t)(st - fv.depth_start[4])] = mask;
    G.entry[3 * (size_t)(st - fv.depth_start[4]) + 1] = fv.row_ptr[st + 1] > fv.row_ptr[st] ? fv.edge_next[fv.row_ptr[st]] : 0u;
    G.entry[3 * (size_t)(st - fv.depth_start[4]) + 2] = fv.term_kw[st] == NONE ? NONE : fv.term_kw[st] + G.kw_base;
    if (G.bloom) {
      auto set = [&] (uint32_t slot) { G.bloom[slot >> 5] |= 1u << (slot & 31); };
      if (fv.term_kw[st] != NONE) {
        set (gram_bloom_slot (gram_bloom_hash (idx, 0), G.bloomT_bits, 0));
        set (gram_bloom_slot (gram_bloom_hash (idx, 0), G.bloomT_bits, 1));
      }
      for (uint32_t e = fv.row_ptr[st]; e < fv.row_ptr[st + 1]; e++) {
        const uint32_t c5 = fv.edge_sym[e] - G.lo;
        set (G.bloomT_bits + gram_bloom_slot (gram_bloom_hash5 (idx, c5, 0), G.bloom5_bits, 0));
        set (G.bloomT_bits + gram_bloom_slot (gram_bloom_hash5 (idx, c5, 0), G.bloom5_bits, 1));
      }
    }
  }
  if (!G.wide) {
    const uint32_t words = (G.W * G.W * G.W * G.W + 31) / 32;
    uint32_t acc = 0;
    for (uint32_t w = 0; w < words; w++) {
      G.prefix[w] = acc;
      acc += (uint32_t)__builtin_popcount (bits[w]);
    }
  }
  if (G.shorts && G.wide) {
    uint32_t *stab = G.stab;
    for (uint32_t st = 1; st < fv.depth_start[4]; st++) {
      if (fv.term_kw[st] == NONE)
        continue;
      const uint32_t key = path[st] | (fv.depth[st] << 24);
      const uint32_t hb = (key * WIDE_H1) >> (32 - G.bloom_log2);
      bits[hb >> 5] |= 1u << (hb & 31);
      uint32_t slot = (key * WIDE_H2) >> (32 - G.stab_log2);
      while (stab[2 * (size_t)slot + 1])
        slot = (slot + 1) & ((1u << G.stab_log2) - 1);
      stab[2 * (size_t)slot] = key;
      stab[2 * (size_t)slot + 1] = G.kw_inline ? (fv.term_kw[st] + G.kw_base) | (fv.depth[st] << 28) | HIT_KW : st;
    }
  }
  if (G.shorts && !G.wide) {
    /* a keyword of d < 4 symbols with path p covers the 3-gram indices [p * W^(3-d), (p+1) * W^(3-d)) */
    unsigned char *nib = G.nib;
    uint32_t *g3 = G.g3;
    for (uint32_t st = 1; st < fv.depth_start[4]; st++) {
      if (fv.term_kw[st] == NONE)
        continue;
      const uint32_t d = fv.depth[st];
      uint32_t width = 1;
      for (uint32_t k = d; k < 3; k++)
        width *= G.W;
      for (uint32_t i3 = path[st] * width; i3 < (path[st] + 1) * width; i3++) {
        nib[i3 >> 1] |= (unsigned char)((1u << (d - 1)) << ((i3 & 1) * 4));
        g3[4 * (size_t)i3 + (d - 1)] = fv.term_kw[st] + G.kw_base; /* the keyword's id: its record is written on the spot */
      }
    }
  }
}

} // namespace

namespace {
int plan_create_flat_kw (const ACMFlat *flat, int device, uint32_t kw_base, ACMPlan **out);
}

extern "C" int
acm_gpu_plan_create_flat (const ACMFlat *flat, int device, ACMPlan **out) {
  return plan_create_flat_kw (flat, device, 0, out);
}

namespace {
/* kw_base: added to every keyword id the plan reports (the delta plans of acm_gpu_plan_update) */
int
plan_create_flat_kw (const ACMFlat *flat, int device, uint32_t kw_base, ACMPlan **out) {
  if (!flat || !out)
    return ACM_GPU_E_ARG;
  int ndev = 0;
  if (hipGetDeviceCount (&ndev) != hipSuccess || ndev <= 0)
    return ACM_GPU_E_NODEVICE;
  if (device < 0 || device >= ndev)
    return ACM_GPU_E_ARG;
  HIP_TRY (hipSetDevice (device));
  /* (asked once per device: the call takes a good part of a millisecond, and the delta plans of
   * acm_gpu_plan_update are made at every dictionary change) */
  static std::mutex prop_mutex;
  static std::vector<std::pair<int, hipDeviceProp_t>> prop_cache;
  hipDeviceProp_t prop;
  {
    std::lock_guard<std::mutex> g (prop_mutex);
    bool have = false;
    for (const auto &c : prop_cache)
      if (c.first == device) {
        prop = c.second;
        have = true;
      }
    if (!have) {
      HIP_TRY (hipGetDeviceProperties (&prop, device));
      prop_cache.emplace_back (device, prop);
    }
  }

  ACMFlatInfo fi;
  ACMFlatView fv;
  acm_flat_info (flat, &fi);
  acm_flat_view (flat, &fv);
  if (fi.lmax >= (1u << 24))
    return ACM_GPU_E_INELIGIBLE;
  const bool interned = fi.sym_bytes == 8;
  if (interned) {
    if (!fv.keys64 && fi.n_edges)
      return ACM_GPU_E_ARG;
    fi.sym_bytes = 4; /* from here on: a machine over 4-byte ids */
  }

  ACMPlan *p = new (std::nothrow) ACMPlan ();
  if (!p)
    return ACM_GPU_E_NOMEM;
  p->device = device;
  p->finfo = fi;
  p->kw_base = kw_base;
  p->covered_keywords = fi.n_keywords;
  p->text_sym_bytes = interned ? 8 : fi.sym_bytes;
  p->cu_count = prop.multiProcessorCount;

  if (const char *e = getenv ("ACM_GPU_SEGMENT_LOG2")) {
    const int lg = atoi (e);
    if (lg >= 12 && lg <= 31)
      p->segment = 1ull << lg;
  }
  /* failure-resolved rows for byte alphabets whenever the whole DFA fits comfortably in HBM */
  const uint32_t n = fi.n_states;
  const uint32_t entry_bytes = n <= 32768 ? 2 : 4;
  const bool dense = fi.sym_bytes == 1 && fi.n_edges > 0 && (uint64_t)n * fi.width < (1ull << 31) &&
                     (uint64_t)n * fi.width * entry_bytes <= (8ull << 30) && fi.lmax >= 1 && fi.lmax - 1 <= 16u * 255;
  const bool cont = dense && entry_bytes == 2; /* continuation mode, see scan_dense_kernel */
  const uint32_t rowbytes = fi.width * entry_bytes;
  const uint32_t queue_bytes = (DENSE_THREADS / WAVE) * QCAP * 8;

  /* ---- LDS budget: rows of a breadth-first prefix [0, HD); continuation mode also keeps
   *      hotfail(s) (2 bytes) for every other state */
  uint32_t HD = 0;
  if (dense) {
    const uint32_t lds_total = (uint32_t)prop.maxSharedMemoryPerMultiProcessor >= 160 * 1024 ? 160 * 1024 : 64 * 1024;
    const uint32_t budget = lds_total - queue_bytes - 512;
    if (cont) {
      /* HD * rowbytes + (n - HD) * 2 <= budget */
      const uint64_t fixed = (uint64_t)n * 2;
      HD = fixed >= budget ? 1 : (uint32_t)((budget - fixed) / (rowbytes - 2));
    } else
      HD = budget / rowbytes;
    if (HD > n)
      HD = n;
    if (HD < 1)
      HD = 1;
  }
  std::vector<uint16_t> hotfail;
  if (cont) {
    hotfail.resize (n);
    for (uint32_t s = 0; s < n; s++) /* f(s) < s: one pass in breadth-first order */
      hotfail[s] = (uint16_t)(s < HD ? s : hotfail[fv.fail[s]]);
  }

  /* ---- device blob layout */
  size_t cur = 0;
  const size_t o_row = blob_reserve (cur, ((size_t)n + 1) * 4);
  const size_t o_sym = blob_reserve (cur, (size_t)(fi.n_edges ? fi.n_edges : 1) * 4);
  const size_t o_next = blob_reserve (cur, (size_t)(fi.n_edges ? fi.n_edges : 1) * 4);
  const size_t o_fail = blob_reserve (cur, (size_t)n * 4);
  const size_t o_cnbo = blob_reserve (cur, (size_t)n * 4);
  const size_t o_oinfo = blob_reserve (cur, (size_t)n * 16);
  const size_t dense_bytes = dense ? (size_t)n * rowbytes : 0;
  const size_t o_dense = blob_reserve (cur, dense_bytes + 16);
  const size_t o_contdh = blob_reserve (cur, cont ? (size_t)n * 2 : 0);
  const size_t o_wrows = blob_reserve (cur, cont ? (size_t)n * fi.width * 4 : 0);
  const size_t o_chain = blob_reserve (cur, cont ? (size_t)(n - HD) * 16 + 16 : 0);
  const size_t o_dstart = blob_reserve (cur, ((size_t)fi.lmax + 2) * 4);
  const uint32_t rows_lds = dense ? ((HD * rowbytes + 15) & ~15u) : 0;
  const uint32_t image_bytes = dense ? ((rows_lds + (cont ? (n - HD) * 2 : 0) + 15) & ~15u) : 0;
  const size_t o_image = blob_reserve (cur, image_bytes + 16);
  /* sparse kernel tables: state records, (symbol, next) edges, root table by symbol value */
  const bool sparse = !dense && fi.sym_bytes >= 2 && fi.n_edges > 0 && n < 0x7FFFFFFFu;
  uint32_t lut_size = 0;
  if (sparse) {
    const uint32_t root_edges = fv.row_ptr[1];
    const uint64_t top = root_edges ? (uint64_t)fv.edge_sym[root_edges - 1] + 1 : 0; /* rows are in ascending order */
    lut_size = (uint32_t)(top < (1ull << 22) ? top : (1ull << 22));
    lut_size = (lut_size + 3) & ~3u;
  }
  const size_t o_srec = blob_reserve (cur, sparse ? (size_t)n * 32 : 0);
  const size_t o_sedge = blob_reserve (cur, sparse ? (size_t)fi.n_edges * 8 : 0);
  const size_t o_lut = blob_reserve (cur, sparse ? (size_t)lut_size * 4 + 16 : 0);
  /* the same three tables for the start-parallel kernel (flags mean something else there) */
  const bool starts = sparse && n < 0x40000000u;
  const size_t o_trec = blob_reserve (cur, starts ? (size_t)n * 32 : 0);
  const size_t o_tedge = blob_reserve (cur, starts ? (size_t)fi.n_edges * 8 : 0);
  const size_t o_tlut = blob_reserve (cur, starts ? (size_t)lut_size * 4 + 16 : 0);
  const size_t o_tpairs = blob_reserve (cur, starts ? (size_t)n * 8 : 0); /* by state id; filled for the root's children */
  /* 4-gram sieve kernel: byte dictionaries that the LDS scheme of the dense kernel does not serve
   * well.  That is every automaton of more than 32,768 states, and the smaller ones whose hot set
   * outgrows LDS: the share of a uniform text's positions that land in a state without an LDS row
   * is estimated as the sum over those states of span^-depth; above 0.1 % the continuation items
   * swamp the dense kernel (measured on a-z, ms per GiB, dense against 4-gram: 1,100 keywords
   * 0.06 % -> 0.38 / 0.41; 1,250 keywords 0.13 % -> 0.46 / 0.42; 1,500 keywords 0.26 % -> 0.58 /
   * 0.43; 3,000 keywords -> 9.0 / 0.48). */
  const char *gram_env = getenv ("ACM_GPU_GRAM"); /* 0: never; 2: whenever the dictionary qualifies (experiments) */
  const int gram_mode = gram_env ? atoi (gram_env) : 1;
  double rowless_share = 0;
  if (cont && fi.alpha_span > 1) {
    for (uint32_t s = HD; s < n; s++)
      rowless_share += pow ((double)fi.alpha_span, -(double)fv.depth[s]);
  }
  const bool gram_narrow = fi.width <= 30 && fi.width == fi.alpha_span + 1 && gram_mode != 3; /* 3: hashed windows always (experiments) */
  bool any_short = false;
  for (uint32_t k = 0; k < fi.n_keywords; k++)
    any_short |= fv.depth[fv.kw_state[k]] < 4;
  const bool gram_big = dense && (entry_bytes == 4 || gram_mode >= 2 || rowless_share > 0.001) && gram_mode != 0 &&
                        fi.lmax >= 4 && n < 0x40000000u;
  bool gram_shorts = gram_big && any_short; /* keywords of 1-3 symbols: a nibble per 3-gram + their ids (narrow alphabets: a pass of their own, scan_short_kernel; hashed windows: the kernel's third queue) */
  /* wide alphabets: hashed 4-byte windows instead of the exact base-W index */
  const bool gram_wide = gram_big && !gram_narrow;
  bool gram = gram_big;
  if (!gram)
    gram_shorts = false;
  const uint32_t gW = fi.width;
  const uint32_t bloom_log2 = 19; /* wide: 64 KB of Bloom bits */
  const uint32_t n_depth4 = gram ? fv.depth_start[5] - fv.depth_start[4] : 0;
  uint32_t wtab_log2 = 4;
  while (gram_wide && (1u << wtab_log2) < 2 * n_depth4 + 2)
    wtab_log2++;
  uint32_t n_short = 0, short_lens = 0; /* wide alphabets: keywords of 1-3 symbols */
  for (uint32_t k = 0; k < fi.n_keywords && gram_wide && gram_shorts; k++)
    if (fv.depth[fv.kw_state[k]] < 4) {
      n_short++;
      short_lens |= 1u << (fv.depth[fv.kw_state[k]] - 1);
    }
  uint32_t stab_log2 = 4;
  while ((1u << stab_log2) < 2 * n_short + 2)
    stab_log2++;
  const uint32_t gW4 = !gram ? 0 : (gram_wide ? 1u << wtab_log2 : gW * gW * gW * gW); /* 8-byte records of the second stage */
  const uint32_t g4words = gram ? (gram_wide ? (1u << bloom_log2) / 32 : (gW4 + 31) / 32) : 0;
  const uint32_t gW3 = gram && !gram_wide ? gW * gW * gW : 0;
  const uint32_t g3_off = (g4words * 4 + 15) & ~15u;                 /* nibble table right after the 4-gram bits */
  const uint32_t g3_bytes = gram_shorts && !gram_wide ? ((gW3 + 1) / 2 + 15) & ~15u : 0;
  /* narrow alphabets: what LDS has left after the bits and the queues goes to the two Bloom filters
   * of the record gather (GramK::bloom5_bits): 8 to 16 bits per terminal 4-gram, the rest for the
   * 5-grams; not worth it below 4 bits per 5-gram or for dictionaries of a few hundred keywords */
  uint32_t bloom_off = 0, bloomT_bits = 0, bloom5_bits = 0;
  const char *bloom_env = getenv ("ACM_GPU_BLOOM"); /* 0: no Bloom filters (experiments) */
  if (gram && !gram_wide && n_depth4 >= 2048 && !(bloom_env && atoi (bloom_env) == 0)) {
    const uint32_t lds_cap = (uint32_t)prop.maxSharedMemoryPerMultiProcessor >= 160 * 1024 ? 160 * 1024 : 64 * 1024;
    const uint32_t gq_bytes = (SPARSE_THREADS / WAVE) * ((gram_shorts && gram_wide ? QCAP : 0u) + GRAM_Q1 + GRAM_Q2 + (gram_wide ? HITS_STRIDE : 0u)) * 8;
    bloom_off = (g3_off + g3_bytes + 15) & ~15u;
    const uint64_t used = (uint64_t)bloom_off + gq_bytes + WALK_CTX_BYTES + 64;
    uint32_t n_term4 = 0, n_5 = 0;
    for (uint32_t st = fv.depth_start[4]; st < fv.depth_start[5]; st++) {
      n_term4 += fv.term_kw[st] != NONE ? 1u : 0u;
      n_5 += fv.row_ptr[st + 1] - fv.row_ptr[st];
    }
    if (used < lds_cap) {
      const uint64_t free_bits = ((uint64_t)lds_cap - used) / 16 * 16 * 8;
      uint64_t tb = (uint64_t)n_term4 * 12 + 1024;
      if (tb > free_bits / 3)
        tb = free_bits / 3;
      tb = tb / 128 * 128;
      const uint64_t fb = (free_bits - tb) / 128 * 128;
      if (tb >= 1024 && fb >= (uint64_t)n_5 * 2 && fb < (1u << 24)) {
        bloomT_bits = (uint32_t)tb;
        bloom5_bits = (uint32_t)fb;
      }
    }
  }
  const uint32_t bloom_bytes = bloom5_bits ? (bloomT_bits + bloom5_bits) / 8 : 0;
  /* scan_gram2_kernel (dev_gram2.h: lane-local sieve on two bits per 4-gram, no queue push per
   * position, no Bloom filters): narrow alphabets without keywords of 1-3 symbols whose table fits
   * LDS beside the waves' areas -- up to 26 symbols + "other".  ACM_GPU_GRAM2=0: scan_gram_kernel. */
  const uint32_t tab2_words = gram && !gram_wide ? (gW4 + 15) / 16 : 0;
  const uint32_t g2_off = (tab2_words * 4 + 15) & ~15u;
  const char *gram2_env = getenv ("ACM_GPU_GRAM2");
  const bool gram2 = gram && !gram_wide && !(gram2_env && atoi (gram2_env) == 0) &&
                     (uint64_t)g2_off + G2_LDS_FIXED <= ((uint32_t)prop.maxSharedMemoryPerMultiProcessor >= 160 * 1024 ? 160u * 1024 : 64u * 1024);
  const size_t o_g4bits = blob_reserve (cur, gram ? (size_t)(bloom5_bits ? bloom_off + bloom_bytes : g3_off + g3_bytes) + 16 : 0);
  const size_t o_g3rec = blob_reserve (cur, gram_shorts && !gram_wide ? (size_t)gW3 * 16 : 0);
  const size_t o_stab = blob_reserve (cur, gram_wide && gram_shorts ? ((size_t)8 << stab_log2) : 0);
  /* scan_short_kernel's LDS image: the nibbles, the number of set nibble bits in front of every 8
   * 3-grams, the keyword ids in the order of those bits (a keyword of d symbols is the id of W^(3-d) 3-grams) */
  uint32_t sh_ids = 0;
  for (uint32_t st = 1; gram_shorts && !gram_wide && st < fv.depth_start[4 <= fi.lmax + 1 ? 4 : fi.lmax + 1]; st++)
    if (fv.term_kw[st] != NONE)
      sh_ids += fv.depth[st] == 1 ? gW * gW : (fv.depth[st] == 2 ? gW : 1u);
  const uint32_t sh_nib_bytes = g3_bytes, sh_base_bytes = gram_shorts && !gram_wide ? (((gW3 + 7) / 8) * 4 + 15) & ~15u : 0;
  const uint32_t sh_ids_bytes = (sh_ids * 4 + 15) & ~15u;
  const size_t o_shimg = blob_reserve (cur, gram_shorts && !gram_wide ? (size_t)sh_nib_bytes + sh_base_bytes + sh_ids_bytes + 16 : 0);
  const size_t o_g4rec = blob_reserve (cur, gram ? (size_t)gW4 * 8 : 0);
  const size_t o_grec = blob_reserve (cur, gram ? (size_t)n * 32 : 0);
  const size_t o_gedge = blob_reserve (cur, gram ? (size_t)fi.n_edges * 8 : 0);
  const size_t o_g4gid = blob_reserve (cur, gram ? (size_t)(fv.depth_start[5 <= fi.lmax + 1 ? 5 : fi.lmax + 1] - fv.depth_start[4]) * 4 + 16 : 0);
  const size_t o_kw4 = blob_reserve (cur, gram ? (size_t)n_depth4 * 4 + 16 : 0);
  const size_t o_g4prefix = blob_reserve (cur, gram && !gram_wide ? (size_t)g4words * 4 + 16 : 0);
  const size_t o_g4entry = blob_reserve (cur, gram && !gram_wide ? (size_t)n_depth4 * 12 + 16 : 0);
  const uint32_t n_depth5 = gram && fi.lmax >= 5 ? fv.depth_start[6 <= fi.lmax + 1 ? 6 : fi.lmax + 1] - fv.depth_start[5] : 0;
  /* peek entries of 4 bytes while a record index fits 23 bits (ACM_GPU_PEEK8=1: 8 bytes anyway -- tests) */
  const bool peek_packed = n < (1u << 23) && !(getenv ("ACM_GPU_PEEK8") && atoi (getenv ("ACM_GPU_PEEK8")) == 1);
  const size_t o_g5peek = blob_reserve (cur, gram && !gram_wide ? (size_t)n_depth5 * (peek_packed ? 4 : 8) + 16 : 0);
  const size_t o_tab2 = blob_reserve (cur, gram2 ? (size_t)g2_off + 16 : 0);
  const bool rows2 = gram2; /* (entry words: classes are below 30, bits 30 and 31 are flags) */
  const size_t o_rows2 = blob_reserve (cur, rows2 ? (size_t)tab2_words * 64 + 16 : 0);
  const size_t o_over2 = blob_reserve (cur, rows2 ? (size_t)n_depth4 * 8 + 16 : 0);
  p->blob_bytes = cur;

  std::vector<unsigned char> host (cur, 0);
  memcpy (&host[o_row], fv.row_ptr, ((size_t)n + 1) * 4);
  memcpy (&host[o_sym], fv.edge_sym, (size_t)fi.n_edges * 4);
  memcpy (&host[o_next], fv.edge_next, (size_t)fi.n_edges * 4);
  memcpy (&host[o_fail], fv.fail, (size_t)n * 4);
  memcpy (&host[o_cnbo], fv.nb_outputs, (size_t)n * 4);
  {
    uint32_t *oi = reinterpret_cast<uint32_t *> (&host[o_oinfo]);
    for (uint32_t s = 0; s < n; s++) {
      const uint32_t nb = fv.nb_outputs[s];
      const uint32_t t0 = fv.term_kw[s] != NONE ? s : fv.out_link[s];
      oi[4 * s + 0] = nb;
      oi[4 * s + 1] = nb ? fv.out_link[t0] : 0;
      oi[4 * s + 2] = nb ? fv.depth[t0] : 0;
      oi[4 * s + 3] = nb ? fv.term_kw[t0] + kw_base : 0;
    }
  }
  memcpy (&host[o_dstart], fv.depth_start, ((size_t)fi.lmax + 2) * 4);
  if (sparse)
    fill_sparse_tables (fv, fi, lut_size, reinterpret_cast<uint32_t *> (&host[o_srec]), reinterpret_cast<uint32_t *> (&host[o_sedge]),
                        reinterpret_cast<uint32_t *> (&host[o_lut]));
  if (starts)
    fill_starts_tables (fv, fi, lut_size, reinterpret_cast<uint32_t *> (&host[o_trec]), reinterpret_cast<uint32_t *> (&host[o_tedge]),
                        reinterpret_cast<uint32_t *> (&host[o_tlut]), reinterpret_cast<uint32_t *> (&host[o_tpairs]));
  if (gram) {
    GramImage G{};
    G.wide = gram_wide;
    G.shorts = gram_shorts;
    G.W = gW;
    G.bloom_log2 = bloom_log2;
    G.wtab_log2 = wtab_log2;
    G.stab_log2 = stab_log2;
    G.bits = reinterpret_cast<uint32_t *> (&host[o_g4bits]);
    G.g4 = reinterpret_cast<uint32_t *> (&host[o_g4rec]);
    G.rec = reinterpret_cast<uint32_t *> (&host[o_grec]);
    G.edge = reinterpret_cast<uint32_t *> (&host[o_gedge]);
    G.g4gid = reinterpret_cast<uint32_t *> (&host[o_g4gid]);
    G.nib = &host[o_g4bits + g3_off];
    G.g3 = reinterpret_cast<uint32_t *> (&host[o_g3rec]);
    G.stab = reinterpret_cast<uint32_t *> (&host[o_stab]);
    for (uint32_t st = fv.depth_start[4]; st < fv.depth_start[5]; st++)
      reinterpret_cast<uint32_t *> (&host[o_kw4])[st - fv.depth_start[4]] = fv.term_kw[st] == NONE ? NONE : fv.term_kw[st] + kw_base;
    G.prefix = reinterpret_cast<uint32_t *> (&host[o_g4prefix]);
    G.entry = reinterpret_cast<uint32_t *> (&host[o_g4entry]);
    G.peek = gram_wide ? nullptr : reinterpret_cast<uint32_t *> (&host[o_g5peek]);
    G.bloom = bloom5_bits ? reinterpret_cast<uint32_t *> (&host[o_g4bits + bloom_off]) : nullptr;
    G.bloomT_bits = bloomT_bits;
    G.bloom5_bits = bloom5_bits;
    G.lo = fi.alpha_lo;
    G.kw_base = kw_base;
    G.kw_inline = (uint64_t)fi.n_keywords + kw_base <= HIT_KW_ID;
    G.peek_packed = peek_packed;
    G.tab2 = gram2 ? reinterpret_cast<uint32_t *> (&host[o_tab2]) : nullptr;
    G.rows2 = rows2 ? reinterpret_cast<uint32_t *> (&host[o_rows2]) : nullptr;
    G.over2 = rows2 ? reinterpret_cast<uint32_t *> (&host[o_over2]) : nullptr;
    fill_gram_tables (fv, fi, G);
    if (gram_shorts && !gram_wide) {
      unsigned char *img = &host[o_shimg];
      memcpy (img, G.nib, (gW3 + 1) / 2);
      uint32_t *base = reinterpret_cast<uint32_t *> (img + sh_nib_bytes), *ids = reinterpret_cast<uint32_t *> (img + sh_nib_bytes + sh_base_bytes);
      uint32_t r = 0;
      for (uint32_t i3 = 0; i3 < gW3; i3++) {
        if ((i3 & 7) == 0)
          base[i3 >> 3] = r;
        const uint32_t nb = (G.nib[i3 >> 1] >> ((i3 & 1) * 4)) & 7u;
        for (uint32_t d = 0; d < 3; d++)
          if ((nb >> d) & 1u)
            ids[r++] = G.g3[4 * (size_t)i3 + d];
      }
      if (r != sh_ids) { /* (the two counts are of the same keywords) */
        delete p;
        return ACM_GPU_E_ARG;
      }
    }
  }
  if (dense) {
    int rc = acm_flat_dense_rows (flat, n, entry_bytes, &host[o_dense]);
    if (rc) {
      delete p;
      return rc;
    }
    /* LDS image: the first HD rows, then hotfail of the states [HD, n) */
    memcpy (&host[o_image], &host[o_dense], (size_t)HD * rowbytes);
    if (cont) {
      const uint16_t *r16 = reinterpret_cast<const uint16_t *> (&host[o_dense]);
      uint32_t *wr = reinterpret_cast<uint32_t *> (&host[o_wrows]);
      for (size_t i = 0; i < (size_t)n * fi.width; i++)
        wr[i] = r16[i] | (fv.depth[r16[i] & 0x7FFFu] << 16);
      uint16_t *cdh = reinterpret_cast<uint16_t *> (&host[o_contdh]);
      for (uint32_t s = 0; s < n; s++)
        cdh[s] = (uint16_t)fv.depth[hotfail[s]];
      memcpy (&host[o_image + rows_lds], hotfail.data () + HD, (size_t)(n - HD) * 2);
      /* Chain records (EmitCtx::chain).  A continuation item says: walk on from rowless state s and
       * report what is longer than j + depth (hotfail (s)) after j more symbols.  When f(s) has a
       * row (hotfail (s) = f(s)), every failure transition out of the trie below s lands no deeper
       * than that bound, so the walk can only ever report along the goto path; and when that path
       * is a single chain of r <= 8 symbols to a leaf t with no keyword ending on the way, the
       * whole walk is one comparison of the next r text bytes: 2 independent loads instead of 4-8
       * dependent ones in expand_items_once_kernel. */
      uint32_t *ch = reinterpret_cast<uint32_t *> (&host[o_chain]);
      for (uint32_t s0 = HD; s0 < n; s0++) {
        uint32_t *r = ch + 4 * (size_t)(s0 - HD);
        const uint32_t dh = fv.depth[hotfail[s0]];
        if (fv.fail[s0] >= HD || dh >= 4000)
          continue;
        uint64_t syms = 0;
        uint32_t len = 0, st = s0;
        bool ok = true;
        while (fv.row_ptr[st + 1] > fv.row_ptr[st]) { /* until a leaf */
          if (fv.row_ptr[st + 1] - fv.row_ptr[st] != 1 || len == 8 || (st != s0 && fv.term_kw[st] != NONE)) {
            ok = false;
            break;
          }
          syms |= (uint64_t)(fv.edge_sym[fv.row_ptr[st]] & 0xFFu) << (8 * len);
          st = fv.edge_next[fv.row_ptr[st]];
          len++;
        }
        if (!ok)
          continue;
        r[0] = (len ? len : 15u) | (dh << 4);
        r[1] = st;
        r[2] = (uint32_t)syms;
        r[3] = (uint32_t)(syms >> 32);
      }
    }
  }
  if (hipMalloc (&p->blob, cur) != hipSuccess) {
    delete p;
    return ACM_GPU_E_NOMEM;
  }
  if (hipMemcpy (p->blob, host.data (), cur, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree (p->blob);
    delete p;
    return ACM_GPU_E_HIP;
  }
  const size_t ctl_bytes = 16 + 2 * POOL_CLASSES * POOL_CTR_STRIDE * sizeof (unsigned int);
  if (hipMalloc (reinterpret_cast<void **> (&p->d_total), ctl_bytes) != hipSuccess ||
      hipMemset (p->d_total, 0, ctl_bytes) != hipSuccess) {
    (void)hipFree (p->blob);
    delete p;
    return ACM_GPU_E_NOMEM;
  }
  p->d_pool_ctr = reinterpret_cast<unsigned int *> (p->d_total + 2);
  unsigned char *b = static_cast<unsigned char *> (p->blob);
  auto u32p = [&] (size_t off) { return reinterpret_cast<const uint32_t *> (b + off); };
  p->csr.row_ptr = u32p (o_row);
  p->csr.edge_sym = u32p (o_sym);
  p->csr.edge_next = u32p (o_next);
  p->csr.fail = u32p (o_fail);
  p->csr.nb_outputs = u32p (o_cnbo);
  p->csr.lmax = fi.lmax;
  p->d_oinfo = reinterpret_cast<const uint4 *> (b + o_oinfo);
  p->entry_bytes = entry_bytes;
  if (dense) {
    p->d_dense = b + o_dense;
    p->d_lds_image = b + o_image;
    p->d_cont_dh = cont ? reinterpret_cast<const uint16_t *> (b + o_contdh) : nullptr;
    p->d_chain = cont ? reinterpret_cast<const uint4 *> (b + o_chain) : nullptr;
    p->d_wrows = cont ? reinterpret_cast<const uint32_t *> (b + o_wrows) : nullptr;
    p->lds_image_bytes = image_bytes;
    DenseK &K = p->K;
    K.W = fi.width;
    K.rowbytes = rowbytes;
    K.lo = fi.alpha_lo;
    K.span = fi.alpha_span;
    K.HD = HD;
    K.aux_off = rows_lds;
    K.queue_off = image_bytes;
    K.wub = fi.lmax > 1 ? (fi.lmax - 1 + 15) / 16 : 0;
    K.lmax = fi.lmax;
    K.stream_stride = WAVE * p->chunk;
  }
  p->d_dstart = u32p (o_dstart);
  if (gram) {
    const uint32_t lds_total = (uint32_t)prop.maxSharedMemoryPerMultiProcessor >= 160 * 1024 ? 160 * 1024 : 64 * 1024;
    const uint32_t gq = (SPARSE_THREADS / WAVE) * ((gram_shorts && gram_wide ? QCAP : 0u) + GRAM_Q1 + GRAM_Q2 + (gram_wide ? HITS_STRIDE : 0u)) * 8;
    const uint32_t bits_bytes = bloom5_bits ? bloom_off + bloom_bytes : g3_off + g3_bytes;
    if ((uint64_t)bits_bytes + gq + WALK_CTX_BYTES <= lds_total) {
      p->d_kw4 = u32p (o_kw4);
      p->GK.g4prefix = u32p (o_g4prefix);
      p->GK.g4entry = u32p (o_g4entry);
      p->GK.kw_inline = (uint64_t)fi.n_keywords + kw_base <= HIT_KW_ID ? 1u : 0u;
      p->GK.g5peek = u32p (o_g5peek);
      p->GK.peek_packed = peek_packed ? 1u : 0u;
      p->GK.d5_begin = fv.depth_start[5 <= fi.lmax + 1 ? 5 : fi.lmax + 1];
      p->GK.d5_rel = peek_packed ? 1u : 0u; /* (fewer than 2^23 states in all: a record index leaves bits 24-29 free too) */
      p->GK.bloom_off = bloom_off;
      p->GK.bloomT_bits = bloomT_bits;
      p->GK.bloom5_bits = bloom5_bits;
      p->gram = true;
      p->GK.g4rec = reinterpret_cast<const uint2 *> (b + o_g4rec);
      p->GK.g4bits = u32p (o_g4bits);
      p->GK.srec = reinterpret_cast<const uint4 *> (b + o_grec);
      p->GK.sedge = reinterpret_cast<const uint2 *> (b + o_gedge);
      p->GK.g4gid = u32p (o_g4gid);
      p->GK.d4_begin = fv.depth_start[4];
      p->GK.g4words = g4words;
      p->GK.g3rec = reinterpret_cast<const uint4 *> (b + o_g3rec);
      p->GK.g3_off = g3_off;
      p->GK.g3_bytes = g3_bytes;
      p->GK.wtab = reinterpret_cast<const uint2 *> (b + o_g4rec);
      p->GK.bloom_log2 = bloom_log2;
      p->GK.wtab_log2 = wtab_log2;
      p->GK.stab = reinterpret_cast<const uint2 *> (b + o_stab);
      p->GK.stab_log2 = stab_log2;
      p->GK.short_lens = short_lens;
      p->gram_shorts = gram_shorts && gram_wide; /* (the kernel's own short-keyword path: hashed windows only) */
      p->short_pass = gram_shorts && !gram_wide;
      p->GK.sh_img = u32p (o_shimg);
      p->GK.sh_nib_bytes = sh_nib_bytes;
      p->GK.sh_base_bytes = sh_base_bytes;
      p->GK.sh_ids_bytes = sh_ids_bytes;
      /* the ids in LDS too while they fit (29 K of them beside the tables and the waves' areas) */
      p->short_ids_lds = (uint64_t)sh_nib_bytes + sh_base_bytes + sh_ids_bytes + SH_LDS_FIXED <= lds_total;
      p->short_lds_bytes = sh_nib_bytes + sh_base_bytes + (p->short_ids_lds ? sh_ids_bytes : 0u) + SH_LDS_FIXED;
      p->short_lds_count_bytes = sh_nib_bytes + SH_LDS_FIXED;
      p->short_blocks_per_cu = 2u * (sh_nib_bytes + SH_LDS_FIXED) <= 160u * 1024u ? 2u : 1u; /* (count-only: the nibbles alone) */
      if (const char *e = getenv ("ACM_GPU_SHORT_BLOCKS"))
        if (atoi (e) == 1)
          p->short_blocks_per_cu = 1;
      p->gram_wide = gram_wide;
      p->GK.W = gW;
      p->GK.lo = fi.alpha_lo;
      p->GK.span = fi.alpha_span;
      p->GK.W4 = gW4;
      p->GK.queue_off = bits_bytes;
      p->gram_lds_bytes = bits_bytes + gq + WALK_CTX_BYTES;
      if (rows2) {
        p->GK.rows2 = reinterpret_cast<const uint2 *> (b + o_rows2);
        p->GK.over2 = reinterpret_cast<const uint2 *> (b + o_over2);
      }
      if (gram2) {
        p->gram2 = true;
        p->GK.tab2 = u32p (o_tab2);

        p->GK.tab2_words = tab2_words;
        p->GK.g2_off = g2_off;
        p->gram_lds_bytes = g2_off + G2_LDS_FIXED;
      }
    }
  }
  if (sparse) {
    const uint32_t tps = 128 / fi.sym_bytes;
    const uint32_t warm = fi.lmax > 1 ? fi.lmax - 1 : 0;
    const uint32_t sparse_queue_bytes = (SPARSE_THREADS / WAVE) * QCAP * 8;
    const uint32_t lds_total = (uint32_t)prop.maxSharedMemoryPerMultiProcessor >= 160 * 1024 ? 160 * 1024 : 64 * 1024;
    p->sparse = true;
    p->sparse_lut_lds = (uint64_t)lut_size * 4 + sparse_queue_bytes + 512 <= lds_total;
    p->SK.srec = reinterpret_cast<const uint4 *> (b + o_srec);
    p->SK.sedge = reinterpret_cast<const uint2 *> (b + o_sedge);
    p->SK.lut = u32p (o_lut);
    p->SK.lut_size = lut_size;
    p->SK.warm_subs = (warm + tps - 1) / tps;
    p->SK.warm_skip = p->SK.warm_subs * tps - warm;
    p->SK.queue_off = p->sparse_lut_lds ? lut_size * 4 : 0;
    p->SK.R = 0; /* per launch */
    p->sparse_lds_bytes = p->SK.queue_off + sparse_queue_bytes + 16;
    const char *mode = getenv ("ACM_GPU_SPARSE");
    p->starts = starts && !(mode && strcmp (mode, "walk") == 0);
    if (starts) {
      p->TK.srec = reinterpret_cast<const uint4 *> (b + o_trec);
      p->TK.sedge = reinterpret_cast<const uint2 *> (b + o_tedge);
      p->TK.lut = u32p (o_tlut);
      p->TK.pairs = reinterpret_cast<const uint2 *> (b + o_tpairs);
      p->TK.lut_size = lut_size;
      const uint32_t starts_queue_bytes = (SPARSE_THREADS / WAVE) * (QCAP + HITS_STRIDE) * 8;
      p->starts_lut_lds = (uint64_t)lut_size * 4 + starts_queue_bytes + WALK_CTX_BYTES <= lds_total;
      p->TK.queue_off = p->starts_lut_lds ? lut_size * 4 : 0;
      p->TK.R = 0;
      p->starts_lds_bytes = p->TK.queue_off + starts_queue_bytes + WALK_CTX_BYTES;
      if (p->starts) {
        StartsMirror *M = new (std::nothrow) StartsMirror ();
        if (M) {
          auto words = [&] (size_t off, size_t cnt) {
            const uint32_t *w = reinterpret_cast<const uint32_t *> (&host[off]);
            return std::vector<uint32_t> (w, w + cnt);
          };
          M->tab[PT_REC] = words (o_trec, (size_t)n * 8);
          M->tab[PT_EDGE] = words (o_tedge, (size_t)fi.n_edges * 2);
          M->tab[PT_LUT] = words (o_tlut, lut_size);
          M->tab[PT_PAIRS] = words (o_tpairs, (size_t)n * 2);
          M->tab[PT_OINFO] = words (o_oinfo, (size_t)n * 4);
          M->parent.assign (n, 0);
          M->parent_sym.assign (n, 0);
          M->depth.assign (fv.depth, fv.depth + n);
          for (uint32_t st = 0; st < n; st++) {
            M->tab[PT_REC][8 * (size_t)st] = st ? fv.row_ptr[st + 1] - fv.row_ptr[st] : M->tab[PT_REC][1]; /* row capacity = its size */
            for (uint32_t e = fv.row_ptr[st]; e < fv.row_ptr[st + 1]; e++) {
              M->parent[fv.edge_next[e]] = st;
              M->parent_sym[fv.edge_next[e]] = fv.edge_sym[e];
            }
          }
          M->dev[PT_REC] = reinterpret_cast<uint32_t *> (b + o_trec);
          M->dev[PT_EDGE] = reinterpret_cast<uint32_t *> (b + o_tedge);
          M->dev[PT_LUT] = reinterpret_cast<uint32_t *> (b + o_tlut);
          M->dev[PT_PAIRS] = reinterpret_cast<uint32_t *> (b + o_tpairs);
          M->dev[PT_OINFO] = reinterpret_cast<uint32_t *> (b + o_oinfo);
          M->cap[PT_LUT] = lut_size;
          M->n_states = n;
          M->n_edges = fi.n_edges;
          M->n_keywords = fi.n_keywords;
          M->lmax = fi.lmax;
          p->mir = M;
        }
      }
    }
  }

  ACMPlanInfo &I = p->info;
  I.device = device;
  I.kernel = p->gram ? 5 : (dense ? 1 : (sparse ? (p->starts ? 4 : 3) : 2));
  I.entry_bytes = dense ? entry_bytes : 0;
  I.width = fi.width;
  I.dense_rows = dense ? n : 0;
  I.lds_rows = HD;
  I.lds_hotfail = cont ? n - HD : 0;
  I.lds_bytes = dense ? image_bytes + queue_bytes + 16 /* tile counter */ : QCAP * 8;
  I.block_threads = dense ? DENSE_THREADS : WAVE;
  I.grid_blocks = dense ? (uint32_t)p->cu_count : (uint32_t)p->cu_count * 16;
  I.chunk_bytes = p->chunk;
  I.streams = p->streams;
  I.table_bytes = cur;

/* a failure from here on must not leak the plan built so far */
#define PLAN_TRY(expr)                                                                             \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      fprintf (stderr, "acm_gpu: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString (_e), __FILE__, \
               __LINE__);                                                                          \
      acm_gpu_plan_destroy (p);                                                                    \
      return ACM_GPU_E_HIP;                                                                        \
    }                                                                                              \
  } while (0)
  if (const char *e = getenv ("ACM_GPU_GRID_BLOCKS")) {
    /* tests: fewer workgroups than CUs, so that one wave sees many tiles of a small text (the
     * overflow path of the item regions fires hundreds of times per wave) */
    const int g = atoi (e);
    if (g >= 1 && g < p->cu_count)
      p->cu_count = g;
    I.grid_blocks = dense ? (uint32_t)p->cu_count : (uint32_t)p->cu_count * 16;
  }
  if (p->gram) {
    I.lds_bytes = p->gram_lds_bytes;
    I.block_threads = SPARSE_THREADS;
    I.grid_blocks = (uint32_t)p->cu_count;
    I.streams = 1;
    I.chunk_bytes = 16;
  }
  if (sparse) {
    I.lds_bytes = p->starts ? p->starts_lds_bytes : p->sparse_lds_bytes;
    I.block_threads = SPARSE_THREADS;
    I.grid_blocks = (uint32_t)p->cu_count;
    I.streams = p->starts ? 1 : SPARSE_S;
    I.chunk_bytes = 128;
    for (int co = 0; co < 2; co++) {
      PLAN_TRY (hipFuncSetAttribute (sparse_kernel_ptr (fi.sym_bytes, p->sparse_lut_lds, co != 0),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->sparse_lds_bytes));
      if (p->starts)
        PLAN_TRY (hipFuncSetAttribute (starts_kernel_ptr (fi.sym_bytes, p->starts_lut_lds, co != 0),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->starts_lds_bytes));
    }
  }
  if (p->gram) {
    for (int co = 0; co < 2; co++)
      PLAN_TRY (hipFuncSetAttribute (gram_kernel_ptr (co != 0, p->gram_shorts, p->gram_wide, false, p->gram2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)p->gram_lds_bytes));
    if (!p->gram_wide)
      PLAN_TRY (hipFuncSetAttribute (gram_kernel_ptr (false, p->gram_shorts, false, true, p->gram2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)p->gram_lds_bytes));
    if (p->short_pass)
      for (int co = 0; co < 2; co++)
        PLAN_TRY (hipFuncSetAttribute (short_kernel_ptr (co != 0, p->short_ids_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(co ? p->short_lds_count_bytes : p->short_lds_bytes)));
  }
  if (dense && !p->gram) {
    for (int co = 0; co < 2; co++)
      PLAN_TRY (hipFuncSetAttribute (dense_kernel_ptr (entry_bytes, p->chunk, p->streams, co != 0),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)I.lds_bytes));
  }
  if (interned) {
    uint32_t cap = 16;
    while (cap < 2 * (fv.n_keys64 + 1))
      cap <<= 1;
    std::vector<uint32_t> tab ((size_t)cap * 4, 0);
    auto mix = [] (uint64_t x) {
      x ^= x >> 30;
      x *= 0xBF58476D1CE4E5B9ull;
      x ^= x >> 27;
      x *= 0x94D049BB133111EBull;
      return x ^ (x >> 31);
    };
    for (uint32_t k = 0; k < fv.n_keys64; k++) {
      const uint64_t key = fv.keys64[k];
      uint32_t h = (uint32_t)mix (key) & (cap - 1);
      while (tab[4 * (size_t)h + 2])
        h = (h + 1) & (cap - 1);
      tab[4 * (size_t)h] = (uint32_t)key;
      tab[4 * (size_t)h + 1] = (uint32_t)(key >> 32);
      tab[4 * (size_t)h + 2] = k + 1;
    }
    if (hipMalloc (reinterpret_cast<void **> (&p->d_intern), (size_t)cap * 16) != hipSuccess ||
        hipMemcpy (p->d_intern, tab.data (), (size_t)cap * 16, hipMemcpyHostToDevice) != hipSuccess) {
      acm_gpu_plan_destroy (p);
      return ACM_GPU_E_NOMEM;
    }
    p->intern_mask = cap - 1;
  }
  if (fv.keys32 || (fi.sym_bytes == 4 && fv.class_rep32)) {
    p->cls32 = true;
    for (uint32_t i = 0; i < fv.n_keys32; i++)
      p->cls32_known[fv.keys32[i]] = fv.keys32_class[i];
    p->cls32_reps.assign (fv.class_rep32, fv.class_rep32 + fv.n_classes);
    if (hipMalloc (reinterpret_cast<void **> (&p->d_unknown), (size_t)(p->cls32_cap + 1) * 4) != hipSuccess) {
      acm_gpu_plan_destroy (p);
      return ACM_GPU_E_NOMEM;
    }
  }
  if (fv.class_map) {
    /* 65,536 entries either way: 2-byte symbols directly, bytes in pairs (see classmap_kernel) */
    std::vector<uint16_t> lut (65536);
    if (fi.sym_bytes == 1) {
      for (uint32_t v = 0; v < 65536; v++)
        lut[v] = (uint16_t)((fv.class_map[v >> 8] << 8) | (fv.class_map[v & 255] & 255));
    } else
      memcpy (lut.data (), fv.class_map, 65536 * 2);
    if (hipMalloc (reinterpret_cast<void **> (&p->d_classlut), 65536 * 2) != hipSuccess ||
        hipMemcpy (p->d_classlut, lut.data (), 65536 * 2, hipMemcpyHostToDevice) != hipSuccess) {
      acm_gpu_plan_destroy (p);
      return ACM_GPU_E_NOMEM;
    }
    PLAN_TRY (hipFuncSetAttribute (reinterpret_cast<const void *> (&classmap_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 65536 * 2));
  }
#undef PLAN_TRY
  *out = p;
  return ACM_GPU_OK;
}
} // namespace

extern "C" int
acm_gpu_plan_create (ACMachine *machine, int device, ACMPlan **out) {
  ACMFlat *flat = nullptr;
  int rc = acm_flatten (machine, &flat);
  if (rc)
    return rc;
  rc = acm_gpu_plan_create_flat (flat, device, out);
  acm_flat_release (flat);
  return rc;
}

extern "C" int
acm_gpu_plan_create_classes (ACMachine *machine, uint32_t sym_bytes, int device, ACMPlan **out) {
  ACMFlat *flat = nullptr;
  int rc = acm_flatten_classes (machine, sym_bytes, &flat);
  if (rc)
    return rc;
  rc = acm_gpu_plan_create_flat (flat, device, out);
  acm_flat_release (flat);
  if (!rc) {
    (*out)->class_sym_bytes = sym_bytes;
    if ((*out)->cls32) /* symbols a text brings are classified with the machine's own comparator */
      acm_internal_comparator (machine, &(*out)->cmp32, &(*out)->cmp32_arg);
  }
  return rc;
}

extern "C" void
acm_gpu_plan_destroy (ACMPlan *plan) {
  if (!plan)
    return;
  (void)hipSetDevice (plan->device);
  if (plan->delta || !plan->retired.empty ())
    (void)hipDeviceSynchronize (); /* scans with a retired delta may still be in flight */
  for (auto &r : plan->retired) {
    if (r.done)
      (void)hipEventDestroy (r.done);
    acm_gpu_plan_destroy (r.plan);
  }
  plan->retired.clear ();
  if (plan->delta)
    acm_gpu_plan_destroy (plan->delta);
  plan->delta = nullptr;
  for (auto &ev : plan->events) {
    (void)hipEventDestroy (ev.start);
    (void)hipEventDestroy (ev.scan_done);
    (void)hipEventDestroy (ev.all_done);
  }
  if (plan->blob)
    (void)hipFree (plan->blob);
  if (plan->d_items && !plan->items_owner)
    (void)hipFree (plan->d_items);
  if (plan->d_fill && !plan->items_owner)
    (void)hipFree (plan->d_fill);
  if (plan->d_holes)
    (void)hipFree (plan->d_holes);
  if (plan->d_spill)
    (void)hipFree (plan->d_spill);
  if (plan->d_total)
    (void)hipFree (plan->d_total);
  if (plan->mir) {
    if (plan->mir->own)
      for (int t = 0; t < 5; t++)
        if (t != PT_LUT && plan->mir->dev[t])
          (void)hipFree (plan->mir->dev[t]);
    if (plan->mir->d_patches)
      (void)hipFree (plan->mir->d_patches);
    delete plan->mir;
  }
  if (plan->d_intern)
    (void)hipFree (plan->d_intern);
  if (plan->d_cls32)
    (void)hipFree (plan->d_cls32);
  if (plan->d_unknown)
    (void)hipFree (plan->d_unknown);
  if (plan->d_classlut)
    (void)hipFree (plan->d_classlut);
  if (plan->d_remap)
    (void)hipFree (plan->d_remap);
  delete plan;
}

extern "C" void
acm_gpu_plan_info (const ACMPlan *plan, ACMPlanInfo *info) {
  *info = plan->info;
  info->delta_keywords = plan->delta ? plan->delta->finfo.n_keywords : 0;
  info->merges = plan->merges;
  info->records_direct = ((plan->gram && !plan->gram_wide) || plan->info.kernel == 2) ? 1u : 0u;
  info->variant = (plan->gram2 ? 2u : 0u) | (plan->short_pass ? 4u : 0u) | (plan->short_pass && !plan->short_ids_lds ? 8u : 0u);
}

extern "C" int
acm_gpu_plan_status (ACMPlan *plan) {
  if (!plan)
    return ACM_GPU_E_ARG;
  HIP_TRY (hipSetDevice (plan->device));
  HIP_TRY (hipDeviceSynchronize ());
  if (!plan->d_total)
    return ACM_GPU_OK;
  unsigned int words[4] = { 0, 0, 0, 0 };
  HIP_TRY (hipMemcpy (words, plan->d_total, sizeof words, hipMemcpyDeviceToHost));
  if (words[3])
    return ACM_GPU_E_INTERNAL;
  return plan->delta ? acm_gpu_plan_status (plan->delta) : ACM_GPU_OK;
}

extern "C" int
acm_gpu_plan_timing (ACMPlan *plan, int enable) {
  if (!plan)
    return ACM_GPU_E_ARG;
  plan->timing = enable != 0;
  plan->timing_every = enable > 1 ? (uint32_t)enable : 1u;
  plan->timing_seq = 0;
  plan->events_used = 0;
  plan->timing_ms = 0;
  plan->timing_all_ms = 0;
  plan->timing_launches = 0;
  return ACM_GPU_OK;
}

extern "C" int
acm_gpu_plan_timing_read_all (ACMPlan *plan, double *scan_ms, double *all_ms, uint64_t *launches) {
  if (!plan)
    return ACM_GPU_E_ARG;
  HIP_TRY (hipSetDevice (plan->device));
  for (size_t i = 0; i < plan->events_used; i++) {
    HIP_TRY (hipEventSynchronize (plan->events[i].all_done));
    float ms = 0;
    HIP_TRY (hipEventElapsedTime (&ms, plan->events[i].start, plan->events[i].scan_done));
    plan->timing_ms += ms;
    HIP_TRY (hipEventElapsedTime (&ms, plan->events[i].start, plan->events[i].all_done));
    plan->timing_all_ms += ms;
    plan->timing_launches++;
  }
  plan->events_used = 0;
  if (scan_ms)
    *scan_ms = plan->timing_ms;
  if (all_ms)
    *all_ms = plan->timing_all_ms;
  if (launches)
    *launches = plan->timing_launches;
  return ACM_GPU_OK;
}

extern "C" int
acm_gpu_plan_timing_read (ACMPlan *plan, double *total_ms, uint64_t *launches) {
  return acm_gpu_plan_timing_read_all (plan, total_ms, nullptr, launches);
}

namespace {

int
timing_begin (ACMPlan *p, hipStream_t st, hipEvent_t *stop, hipEvent_t *stop_all) {
  *stop = nullptr;
  *stop_all = nullptr;
  if (!p->timing || p->timing_seq++ % p->timing_every != 0)
    return ACM_GPU_OK;
  if (p->events_used == p->events.size ()) {
    if (p->events.size () >= 4096) { /* fold what is recorded so far */
      int rc = acm_gpu_plan_timing_read (p, nullptr, nullptr);
      if (rc)
        return rc;
    } else {
      ACMPlan::LaunchEvents ev;
      HIP_TRY (hipEventCreate (&ev.start));
      HIP_TRY (hipEventCreate (&ev.scan_done));
      HIP_TRY (hipEventCreate (&ev.all_done));
      p->events.push_back (ev);
    }
  }
  auto &ev = p->events[p->events_used++];
  HIP_TRY (hipEventRecord (ev.start, st));
  *stop = ev.scan_done;
  *stop_all = ev.all_done;
  return ACM_GPU_OK;
}

template <bool COUNT_ONLY>
int
launch_sparse (ACMPlan *p, const EmitCtx &E, Launch a, hipStream_t st) {
  const uint32_t tps = 128 / p->finfo.sym_bytes;
  const uint32_t nsubs = (uint32_t)(((uint64_t)a.n + tps - 1) / tps);
  uint32_t grid = p->info.grid_blocks;
  const uint64_t lane_streams = (uint64_t)grid * (SPARSE_THREADS / WAVE) * WAVE * SPARSE_S;
  /* sub-chunks per chunk: about four tiles per wave, the warm-up at most a quarter of a chunk */
  uint64_t R = nsubs / (lane_streams * 4);
  const uint64_t rmin = 4ull * p->SK.warm_subs > 4 ? 4ull * p->SK.warm_subs : 4;
  if (R < rmin)
    R = rmin;
  if (R > 64 && R > rmin)
    R = 64 > rmin ? 64 : rmin;
  SparseK K = p->SK;
  K.R = (uint32_t)R;
  a.range_begin = 0;
  a.range_end = (uint32_t)((nsubs + R - 1) / R);
  const uint32_t tiles = (a.range_end + WAVE * SPARSE_S - 1) / (WAVE * SPARSE_S);
  const uint32_t wpb = SPARSE_THREADS / WAVE;
  if ((tiles + wpb - 1) / wpb < grid)
    grid = (tiles + wpb - 1) / wpb;
  void *args[] = { &K, const_cast<EmitCtx *> (&E), &a, &a.text };
  HIP_TRY (hipLaunchKernel (sparse_kernel_ptr (p->finfo.sym_bytes, p->sparse_lut_lds, COUNT_ONLY), dim3 (grid),
                            dim3 (SPARSE_THREADS), args, p->sparse_lds_bytes, st));
  return ACM_GPU_OK;
}

/* the last sixteenth of the tiles [range_begin, range_end) becomes the dynamic pool (none for
 * launches of a few tiles per wave); counters alternate from launch to launch (TileShare) */
void
set_tile_pool (ACMPlan *p, Launch &a, uint32_t waves) {
  const uint32_t tiles = a.range_end - a.range_begin;
  const uint32_t pool = tiles >= waves * 8 ? tiles / 16 : 0;
  a.static_end = a.range_end - pool;
  const uint32_t blocks = waves / (SPARSE_THREADS / WAVE) > 0 ? waves / (SPARSE_THREADS / WAVE) : 1;
  a.pool_classes = blocks < POOL_CLASSES ? blocks : POOL_CLASSES; /* (grids of fewer blocks than parts: every part must have a block) */
  a.pool_class_tiles = (pool + a.pool_classes - 1) / a.pool_classes;
  a.pool_ctr = p->d_pool_ctr + (p->launch_seq & 1) * POOL_CLASSES * POOL_CTR_STRIDE;
  a.pool_reset = p->d_pool_ctr + ((p->launch_seq & 1) ^ 1) * POOL_CLASSES * POOL_CTR_STRIDE;
  p->launch_seq++;
}

/* records of the hits parked by `regions_used` waves of the start-parallel / 4-gram kernels */
void
launch_expand_hits (ACMPlan *p, const EmitCtx &E, uint32_t regions_used, hipStream_t st) {
  /* 1024 threads per 16 regions (config 3, 4 GiB: 1.10 ms; 8, 4, 2 regions or smaller blocks: 1.25-1.33) */
  hipLaunchKernelGGL ((expand_hits_kernel<1024, 16>), dim3 ((regions_used + 15) / 16), dim3 (1024), 0, st, E,
                      static_cast<const uint2 *> (p->d_items), p->region_items, p->d_fill, regions_used);
}

template <bool COUNT_ONLY>
int
launch_starts (ACMPlan *p, const EmitCtx &E, Launch a, hipStream_t st, hipEvent_t stop) {
  const uint32_t group = WAVE * (16 / p->finfo.sym_bytes); /* symbols per 1 KiB group */
  const uint32_t ngroups = (uint32_t)(((uint64_t)a.n + group - 1) / group);
  uint32_t grid = p->info.grid_blocks;
  const uint32_t wpb = SPARSE_THREADS / WAVE;
  /* a match that ends at emit_from or later starts no earlier than emit_from - (lmax - 1) */
  const uint32_t back = p->finfo.lmax > 1 ? p->finfo.lmax - 1 : 0;
  const uint32_t first_group = (a.emit_from > back ? a.emit_from - back : 0) / group;
  /* groups per tile: about sixteen tiles per wave, 4 to 64 KiB each, a multiple of 4 */
  uint64_t R = (ngroups - first_group) / ((uint64_t)grid * wpb * 16) & ~3ull;
  if (R < 4)
    R = 4;
  if (R > 64)
    R = 64;
  StartsK K = p->TK;
  K.R = (uint32_t)R;
  a.range_begin = first_group / (uint32_t)R;
  a.range_end = (uint32_t)((ngroups + R - 1) / R);
  const uint32_t tiles = a.range_end - a.range_begin;
  if ((tiles + wpb - 1) / wpb < grid)
    grid = (tiles + wpb - 1) / wpb;
  set_tile_pool (p, a, grid * wpb);
  void *items = COUNT_ONLY ? nullptr : p->d_items;
  uint32_t *fill = COUNT_ONLY ? nullptr : p->d_fill;
  void *args[] = { &K, const_cast<EmitCtx *> (&E), &a, &a.text, &items, &p->region_items, &fill };
  HIP_TRY (hipLaunchKernel (starts_kernel_ptr (p->finfo.sym_bytes, p->starts_lut_lds, COUNT_ONLY), dim3 (grid),
                            dim3 (SPARSE_THREADS), args, p->starts_lds_bytes, st));
  if (stop) /* the timing brackets the scan kernel alone, as for the dense kernel */
    HIP_TRY (hipEventRecord (stop, st));
  if (!COUNT_ONLY)
    launch_expand_hits (p, E, grid * wpb, st);
  return ACM_GPU_OK;
}

/* the tiles of one launch of the 4-gram kernel over a segment of n symbols: R groups of 1,024
 * symbols each, tiles [begin, end) */
struct GramTiling {
  uint32_t R, begin, end;
};
constexpr uint32_t GRAM_R_MIN = 4; /* groups per tile, at least (tiled_layout's bound counts on it) */
GramTiling
gram_tiling (const ACMPlan *p, uint32_t n, uint32_t emit_from) {
  const uint32_t group = WAVE * 16;
  const uint32_t ngroups = (uint32_t)(((uint64_t)n + group - 1) / group);
  const uint32_t wpb = SPARSE_THREADS / WAVE;
  const uint32_t back = p->finfo.lmax > 1 ? p->finfo.lmax - 1 : 0;
  const uint32_t first_group = (emit_from > back ? emit_from - back : 0) / group;
  uint64_t R = (ngroups - first_group) / ((uint64_t)p->cu_count * wpb * 16);
  if (R < GRAM_R_MIN)
    R = GRAM_R_MIN;
  if (R > 64)
    R = 64;
  static const int r_env = getenv ("ACM_GPU_GRAM_R") ? atoi (getenv ("ACM_GPU_GRAM_R")) : 0; /* experiments: groups per tile (a multiple of 4, up to 64) */
  if (r_env >= 4 && r_env <= 64 && r_env % 4 == 0)
    R = (uint64_t)r_env;
  GramTiling t;
  t.R = (uint32_t)R;
  t.begin = first_group / (uint32_t)R;
  t.end = (uint32_t)((ngroups + R - 1) / R);
  return t;
}

/* ACM_GPU_CLOSE_SORT=network: close_holes_kernel sorts its descriptors with the bitonic network whatever
 * their spread (its fallback for crowded buckets; tests) */
uint32_t
close_network_only () {
  const char *e = getenv ("ACM_GPU_CLOSE_SORT");
  return e && strcmp (e, "network") == 0 ? 1u : 0u;
}

template <bool COUNT_ONLY>
int
launch_gram (ACMPlan *p, const EmitCtx &E, Launch a, hipStream_t st, hipEvent_t stop, bool first_segment, bool last_segment) {
  uint32_t grid = (uint32_t)p->cu_count;
  const uint32_t wpb = SPARSE_THREADS / WAVE;
  const GramTiling T = gram_tiling (p, a.n, a.emit_from);
  GramK K = p->GK;
  K.R = T.R;
  a.range_begin = T.begin;
  a.range_end = T.end;
  const uint32_t tiles = a.range_end - a.range_begin;
  if ((tiles + wpb - 1) / wpb < grid)
    grid = (tiles + wpb - 1) / wpb;
  set_tile_pool (p, a, grid * wpb);
  /* narrow alphabets: the kernel writes the records itself (no item buffer, no expansion; the holes
   * its waves leave in their last chunks are closed right behind it); hashed windows: hits parked
   * per wave and expanded as in the start-parallel kernel */
  const bool direct = !p->gram_wide;
  void *items = (COUNT_ONLY || direct) ? nullptr : p->d_items;
  uint32_t *fill = (COUNT_ONLY || direct) ? nullptr : p->d_fill;
  void *holes = (!COUNT_ONLY && direct) ? p->d_holes : nullptr;
  /* the segments of one scan share the waves' chunks of records: a wave picks up in segment k + 1
   * the chunk it was filling at the end of segment k (its hole descriptor says where), and the
   * holes are closed once, behind the last segment */
  uint32_t resume = first_segment ? 0u : 1u;
  if (holes && first_segment)
    HIP_TRY (hipMemsetAsync (holes, 0, (size_t)p->direct_regions * sizeof (RecHole), st));
  if (first_segment || grid * wpb > p->holes_waves[0])
    p->holes_waves[0] = grid * wpb;
  /* a tiled scan: a directory entry per tile, the chunks linked, the holes left alone (dev_tiles.h) */
  TileEntry *dir = (!COUNT_ONLY && direct) ? static_cast<TileEntry *> (p->tiled_dir) : nullptr;
  uint32_t dir_base = p->tiled_base;
  if (dir)
    p->tiled_base += tiles;
  void *args[] = { &K, const_cast<EmitCtx *> (&E), &a, &a.text, &items, &p->region_items, &fill, &holes, &resume, &dir, &dir_base };
  HIP_TRY (hipLaunchKernel (gram_kernel_ptr (COUNT_ONLY, p->gram_shorts, p->gram_wide, dir != nullptr, p->gram2), dim3 (grid), dim3 (SPARSE_THREADS), args,
                            p->gram_lds_bytes, st));
  if (stop)
    HIP_TRY (hipEventRecord (stop, st));
  if (!COUNT_ONLY) {
    if (direct && dir) {
      /* (nothing: tile_gather_kernel reads the records where they lie) */
    } else if (direct && last_segment) {
      /* (the waves of the scan's widest launch: an earlier segment may have had more blocks than this
       * one; a short text has few -- every block of the kernel sorts all the descriptors it is given) */
      const uint32_t n_waves = p->holes_waves[0];
      uint32_t npow = 64;
      while (npow < n_waves)
        npow <<= 1;
      const uint32_t blocks = n_waves / 16 > 0 ? n_waves / 16 : 1; /* 16 holes per block */
      hipLaunchKernelGGL (close_holes_kernel, dim3 (blocks), dim3 (CLOSE_THREADS), npow * 16, st, E, static_cast<const RecHole *> (p->d_holes),
                          n_waves, npow, reinterpret_cast<unsigned int *> (p->d_total + 1), close_network_only ());
      HIP_TRY (hipGetLastError ());
    } else if (!direct)
      launch_expand_hits (p, E, grid * wpb, st);
  }
  return ACM_GPU_OK;
}

/* the keywords of 1-3 symbols of a 4-gram plan over a narrow alphabet: a pass of their own over the
 * segment (dev_short.h), into the same record buffer; its waves' holes have descriptors of their own */
template <bool COUNT_ONLY>
int
launch_short (ACMPlan *p, const EmitCtx &E, Launch a, hipStream_t st, hipEvent_t stop, bool first_segment, bool last_segment) {
  /* (count-only: 61 registers, two blocks a CU while LDS allows; with records one) */
  uint32_t grid = (uint32_t)p->cu_count * (COUNT_ONLY ? p->short_blocks_per_cu : 1u);
  const uint32_t wpb = SPARSE_THREADS / WAVE;
  const uint32_t short_regions = p->direct_regions;
  /* a match of 1-3 symbols that ends at emit_from or later starts no earlier than emit_from - 2 */
  const uint32_t group = WAVE * 16;
  const uint32_t ngroups = (uint32_t)(((uint64_t)a.n + group - 1) / group);
  const uint32_t first_group = (a.emit_from > 2 ? a.emit_from - 2 : 0) / group;
  uint64_t R = (ngroups - first_group) / ((uint64_t)grid * wpb * 16);
  R = R < 4 ? 4 : (R > 64 ? 64 : R & ~3ull); /* (a multiple of 4: the kernel takes a tile's groups four at a time) */
  GramK K = p->GK;
  K.R = (uint32_t)R;
  a.range_begin = first_group / (uint32_t)R;
  a.range_end = (uint32_t)((ngroups + R - 1) / R);
  const uint32_t tiles = a.range_end - a.range_begin;
  if ((tiles + wpb - 1) / wpb < grid)
    grid = (tiles + wpb - 1) / wpb;
  set_tile_pool (p, a, grid * wpb);
  RecHole *holes = COUNT_ONLY ? nullptr : static_cast<RecHole *> (p->d_holes) + p->direct_regions;
  uint32_t resume = first_segment ? 0u : 1u;
  if (holes && first_segment)
    HIP_TRY (hipMemsetAsync (holes, 0, (size_t)short_regions * sizeof (RecHole), st));
  if (first_segment || grid * wpb > p->holes_waves[1])
    p->holes_waves[1] = grid * wpb;
  void *args[] = { &K, const_cast<EmitCtx *> (&E), &a, &a.text, &holes, &resume };
  HIP_TRY (hipLaunchKernel (short_kernel_ptr (COUNT_ONLY, p->short_ids_lds), dim3 (grid), dim3 (SPARSE_THREADS), args,
                            COUNT_ONLY ? p->short_lds_count_bytes : p->short_lds_bytes, st));
  if (stop)
    HIP_TRY (hipEventRecord (stop, st));
  if (!COUNT_ONLY && last_segment) {
    const uint32_t n_waves = p->holes_waves[1];
    uint32_t npow = 64;
    while (npow < n_waves)
      npow <<= 1;
    const uint32_t blocks = n_waves / 16 > 0 ? n_waves / 16 : 1;
    hipLaunchKernelGGL (close_holes_kernel, dim3 (blocks), dim3 (CLOSE_THREADS), npow * 16, st, E, holes, n_waves, npow,
                        reinterpret_cast<unsigned int *> (p->d_total + 1), close_network_only ());
    HIP_TRY (hipGetLastError ());
  }
  return ACM_GPU_OK;
}

template <bool COUNT_ONLY>
int
launch_csr (ACMPlan *p, const EmitCtx &E, Launch a, hipStream_t st) {
  if (a.range_end <= a.range_begin)
    return ACM_GPU_OK;
  const uint64_t len = a.range_end - a.range_begin;
  /* chunk length: enough chunks to fill the chip, long enough to amortise the warm-up */
  const uint64_t target_lanes = (uint64_t)p->cu_count * 16 * WAVE;
  uint64_t chunk = (len + target_lanes - 1) / target_lanes;
  const uint64_t minchunk = 64 > 8ull * p->finfo.lmax ? 64 : 8ull * p->finfo.lmax;
  if (chunk < minchunk)
    chunk = minchunk;
  if (chunk > 4096 && chunk > minchunk)
    chunk = 4096 > minchunk ? 4096 : minchunk;
  const uint64_t nchunks = (len + chunk - 1) / chunk;
  uint64_t blocks = (nchunks + WAVE - 1) / WAVE;
  const uint64_t maxblocks = (uint64_t)p->cu_count * 32;
  if (blocks > maxblocks)
    blocks = maxblocks;
  dim3 g ((uint32_t)blocks), b (WAVE);
  switch (p->finfo.sym_bytes) {
  case 1: hipLaunchKernelGGL ((scan_csr_kernel<uint8_t, COUNT_ONLY>), g, b, 0, st, p->csr, E, a, (uint32_t)chunk); break;
  case 2: hipLaunchKernelGGL ((scan_csr_kernel<uint16_t, COUNT_ONLY>), g, b, 0, st, p->csr, E, a, (uint32_t)chunk); break;
  default: hipLaunchKernelGGL ((scan_csr_kernel<uint32_t, COUNT_ONLY>), g, b, 0, st, p->csr, E, a, (uint32_t)chunk); break;
  }
  HIP_TRY (hipGetLastError ());
  return ACM_GPU_OK;
}

/* (re)allocate the item buffer for segments of up to n symbols: room for one item per 256
 * symbols, at least 256 per wave; denser matches are expanded in the kernel itself */
int
ensure_item_buffer (ACMPlan *user, uint64_t n, uint32_t symbols_per_item = 256, uint32_t min_items = 256) {
  /* a delta plan parks in the buffer of the plan it belongs to: their scans run one after the
   * other on one stream, and a fresh 140 MB buffer (with the waits it takes to set one up) for
   * every delta made every dictionary change cost more than the delta itself */
  ACMPlan *p = user->items_owner ? user->items_owner : user;
  /* one region per wave of the kernels that park items (dense, 4-gram, start-parallel: one block of
   * 16 waves per CU) -- of the plan that SCANS, not of the buffer's owner: an owner of the CSR kind
   * (a plan made from an empty machine) has cu_count * 16 single-wave blocks, which sized the
   * buffer of a dense delta at 65,536 regions x 4,352 items = 2.3 GB */
  uint32_t regions = (uint32_t)user->cu_count * (DENSE_THREADS / WAVE);
  if (p->d_items && p->regions > regions)
    regions = p->regions; /* (shared by a plan and its delta: never shrink what the other one uses) */
  uint64_t per = (n / symbols_per_item + regions - 1) / regions;
  per = (per + 63) / 64 * 64;
  if (per < min_items)
    per = min_items;
  if (per > (1u << 20))
    per = 1u << 20;
  if (!(p->d_items && p->regions == regions && p->region_items >= per)) {
    if (p->d_items || p->d_fill)
      HIP_TRY (hipDeviceSynchronize ()); /* earlier scans (pieces of a stream) may still be parking items */
    if (p->d_items)
      HIP_TRY (hipFree (p->d_items));
    if (p->d_fill)
      HIP_TRY (hipFree (p->d_fill));
    p->d_items = nullptr;
    p->d_fill = nullptr;
    if (hipMalloc (&p->d_items, (size_t)regions * per * 8) != hipSuccess)
      return ACM_GPU_E_NOMEM;
    if (hipMalloc (reinterpret_cast<void **> (&p->d_fill), (size_t)regions * 4) != hipSuccess)
      return ACM_GPU_E_NOMEM;
    HIP_TRY (hipMemset (p->d_fill, 0, (size_t)regions * 4));
    /* the memset runs on the null stream; the scans may run on streams that do not wait for it */
    HIP_TRY (hipDeviceSynchronize ());
    p->regions = regions;
    p->region_items = (uint32_t)per;
  }
  user->d_items = p->d_items;
  user->d_fill = p->d_fill;
  user->regions = p->regions;
  user->region_items = p->region_items;
  return ACM_GPU_OK;
}

/* expand_items_once_kernel: 1024 threads per 16 regions (one block per CU on config 2), the items
 * of 4 rounds in registers, one atomic per block (measured against one atomic per round of 1,024
 * items, smaller blocks and fewer rounds: DESIGN.md 4.2) */
/* hole descriptors and spill area of a plan whose scan kernel writes the records itself */
int
ensure_direct_buffers (ACMPlan *p, uint32_t rec_chunk) {
  const uint32_t regions = (uint32_t)p->cu_count * (SPARSE_THREADS / WAVE);
  if (p->d_holes && p->d_spill && p->direct_regions >= regions && p->spill_chunk >= rec_chunk)
    return ACM_GPU_OK;
  if (p->d_holes || p->d_spill)
    HIP_TRY (hipDeviceSynchronize ());
  if (p->d_holes)
    HIP_TRY (hipFree (p->d_holes));
  if (p->d_spill)
    HIP_TRY (hipFree (p->d_spill));
  p->d_holes = p->d_spill = nullptr;
  if (hipMalloc (&p->d_holes, (size_t)regions * 2 * sizeof (RecHole)) != hipSuccess || /* (the 4-gram pass's and the short-keyword pass's) */
      hipMalloc (&p->d_spill, (size_t)regions * rec_chunk * 16) != hipSuccess) /* (one chunk per wave: 64 MB, 256 MB with big chunks) */
    return ACM_GPU_E_NOMEM;
  p->direct_regions = regions;
  p->spill_chunk = rec_chunk;
  return ACM_GPU_OK;
}

template <bool CONT, bool COUNT_ONLY>
void
launch_expand_cfg (ACMPlan *p, const EmitCtx &E, uint32_t regions_used, const ExpandTail &tail, hipStream_t st) {
  const dim3 g ((regions_used + 15) / 16);
  hipLaunchKernelGGL ((expand_items_once_kernel<CONT, COUNT_ONLY, 1024, 16, 4>), g, dim3 (1024), 0, st, E,
                      static_cast<const uint2 *> (p->d_items), p->region_items, p->d_fill, tail);
}

/* scan kernel, then the expansion of what it parked; the caller's counter is written by the
 * expansion of the last segment (no memsets: both kernels leave their bookkeeping zeroed) */
template <bool COUNT_ONLY>
int
launch_dense (ACMPlan *p, const EmitCtx &E, Launch a, hipStream_t st, hipEvent_t stop, uint64_t *d_count, bool last_segment) {
  const uint32_t TILE = WAVE * p->streams * p->chunk;
  a.range_begin = 0;
  a.range_end = (uint32_t)(((uint64_t)a.n + TILE - 1) / TILE);
  uint32_t grid = p->info.grid_blocks;
  const uint32_t wpb = DENSE_THREADS / WAVE;
  const uint32_t blocks_needed = (a.range_end + wpb - 1) / wpb;
  if (blocks_needed < grid)
    grid = blocks_needed;
  /* the last 1/16 of the tiles is the dynamic pool (none for inputs of a few tiles per wave) */
  const uint32_t pool = a.range_end >= grid * wpb * 8 ? a.range_end / 16 : 0;
  a.static_end = a.range_end - pool;
  a.pool_classes = grid < POOL_CLASSES ? grid : POOL_CLASSES;
  a.pool_class_tiles = (pool + a.pool_classes - 1) / a.pool_classes;
  a.pool_ctr = p->d_pool_ctr + (p->launch_seq & 1) * POOL_CLASSES * POOL_CTR_STRIDE;
  a.pool_reset = p->d_pool_ctr + ((p->launch_seq & 1) ^ 1) * POOL_CLASSES * POOL_CTR_STRIDE;
  p->launch_seq++;
  void *args[] = { &p->K, const_cast<EmitCtx *> (&E), &a, &p->d_dense, &p->d_lds_image, &p->lds_image_bytes, &a.text,
                   &p->d_items, &p->region_items, &p->d_fill, &p->d_dstart };
  HIP_TRY (hipLaunchKernel (dense_kernel_ptr (p->entry_bytes, p->chunk, p->streams, COUNT_ONLY), dim3 (grid), dim3 (DENSE_THREADS), args,
                            p->info.lds_bytes, st));
  if (stop)
    HIP_TRY (hipEventRecord (stop, st));
  ExpandTail tail;
  tail.user_count = reinterpret_cast<unsigned long long *> (d_count);
  tail.ticket = reinterpret_cast<unsigned int *> (p->d_total + 1);
  tail.last_segment = last_segment ? 1 : 0;
  if (p->entry_bytes == 2)
    launch_expand_cfg<true, COUNT_ONLY> (p, E, grid * wpb, tail, st);
  else
    launch_expand_cfg<false, COUNT_ONLY> (p, E, grid * wpb, tail, st);
  HIP_TRY (hipGetLastError ());
  return ACM_GPU_OK;
}

/* ---- incremental updates of a start-parallel plan */
/* brings the device tables in line with the mirror, on the stream of the scan that follows */
int
starts_flush (ACMPlan *p, hipStream_t st) {
  StartsMirror &M = *p->mir;
  if (M.full_upload) {
    /* (re)allocate with headroom and send everything; scans already enqueued still read the
     * old arrays, so wait for them before those are freed */
    HIP_TRY (hipDeviceSynchronize ());
    uint32_t *fresh[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    size_t want[5];
    for (int t = 0; t < 5; t++) {
      if (t == PT_LUT)
        continue; /* fixed size, stays in the plan's blob */
      want[t] = M.tab[t].size () * 2 + 4096;
      const bool keep = M.own && M.cap[t] >= M.tab[t].size ();
      if (keep)
        fresh[t] = M.dev[t];
      else if (hipMalloc (reinterpret_cast<void **> (&fresh[t]), want[t] * 4) != hipSuccess ||
               hipMemset (fresh[t], 0, want[t] * 4) != hipSuccess) { /* words never set are 0 in the mirror too */
        for (int u = 0; u <= t; u++)
          if (fresh[u] && fresh[u] != M.dev[u])
            (void)hipFree (fresh[u]);
        return ACM_GPU_E_NOMEM;
      }
    }
    for (int t = 0; t < 5; t++) {
      if (t == PT_LUT) {
        HIP_TRY (hipMemcpy (M.dev[t], M.tab[t].data (), M.tab[t].size () * 4, hipMemcpyHostToDevice));
        continue;
      }
      HIP_TRY (hipMemcpy (fresh[t], M.tab[t].data (), M.tab[t].size () * 4, hipMemcpyHostToDevice));
      if (fresh[t] != M.dev[t]) {
        if (M.own && M.dev[t])
          HIP_TRY (hipFree (M.dev[t]));
        M.dev[t] = fresh[t];
        M.cap[t] = want[t];
      }
    }
    M.own = true;
    M.full_upload = false;
    M.patches.clear ();
    p->TK.srec = reinterpret_cast<const uint4 *> (M.dev[PT_REC]);
    p->TK.sedge = reinterpret_cast<const uint2 *> (M.dev[PT_EDGE]);
    p->TK.pairs = reinterpret_cast<const uint2 *> (M.dev[PT_PAIRS]);
    p->d_oinfo = reinterpret_cast<const uint4 *> (M.dev[PT_OINFO]);
    return ACM_GPU_OK;
  }
  if (M.patches.empty ())
    return ACM_GPU_OK;
  /* a word may have been set several times since the last flush and the patch kernel writes in
   * no particular order: every patch carries the word's final value */
  for (uint4 &q : M.patches)
    q.z = M.tab[q.x][q.y];
  const size_t np = M.patches.size ();
  if (M.cap_patches < np) {
    if (M.d_patches) {
      HIP_TRY (hipStreamSynchronize (st));
      HIP_TRY (hipFree (M.d_patches));
      M.d_patches = nullptr;
    }
    M.cap_patches = np * 2 + 1024;
    if (hipMalloc (reinterpret_cast<void **> (&M.d_patches), M.cap_patches * sizeof (uint4)) != hipSuccess) {
      M.cap_patches = 0;
      return ACM_GPU_E_NOMEM;
    }
  }
  /* (blocking copy ordered on the stream: the staging vector can be reused at once, and an
   * earlier patch kernel that reads d_patches has finished before it is overwritten) */
  HIP_TRY (hipMemcpyWithStream (M.d_patches, M.patches.data (), np * sizeof (uint4), hipMemcpyHostToDevice, st));
  PatchTables T;
  for (int t = 0; t < 5; t++)
    T.t[t] = M.dev[t];
  hipLaunchKernelGGL (patch_kernel, dim3 ((uint32_t)((np + 255) / 256)), dim3 (256), 0, st, T, M.d_patches, (uint32_t)np);
  HIP_TRY (hipGetLastError ());
  M.patches.clear ();
  return ACM_GPU_OK;
}

/* goto edge of mirror state s on symbol c (NONE if there is none) */
uint32_t
mirror_child (const StartsMirror &M, uint32_t s, uint32_t c) {
  const uint32_t *r = &M.tab[PT_REC][8 * (size_t)s];
  const uint32_t ne = r[1];
  if (s == 0 && c < M.tab[PT_LUT].size ()) /* the root table answers for the symbols it holds */
    return (M.tab[PT_LUT][c] & ST_STATE) ? (M.tab[PT_LUT][c] & ST_STATE) : NONE;
  if (ne <= 2 && s != 0) {
    if (ne >= 1 && r[4] == c)
      return r[5];
    if (ne >= 2 && r[6] == c)
      return r[7];
    return NONE;
  }
  uint32_t lo = r[2], hi = r[2] + ne;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (M.tab[PT_EDGE][2 * (size_t)mid] < c)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo < r[2] + ne && M.tab[PT_EDGE][2 * (size_t)lo] == c ? M.tab[PT_EDGE][2 * (size_t)lo + 1] : NONE;
}

/* root-table entry of the root's child `child` (its flags follow from its record) */
void
mirror_refresh_root_child (StartsMirror &M, uint32_t child) {
  const uint32_t sym = M.parent_sym[child];
  const uint32_t *r = &M.tab[PT_REC][8 * (size_t)child];
  const uint32_t ne = r[1];
  const bool always = r[3] != 0 || ne > 2 || ne == 0;
  M.set (PT_PAIRS, 2 * (size_t)child, ne >= 1 ? r[4] : 0);
  M.set (PT_PAIRS, 2 * (size_t)child + 1, ne >= 2 ? r[6] : (ne >= 1 ? r[4] : 0));
  if (sym < M.tab[PT_LUT].size ())
    M.set (PT_LUT, sym, (M.tab[PT_LUT][sym] & ST_SECOND) | child | (always ? ST_ALWAYS : 0u));
}

/* adds the edge s --c--> nx: rows stay sorted; states with more than two edges (and the root)
 * keep theirs in the edge table, in a slot range with room to grow */
void
mirror_add_edge (StartsMirror &M, uint32_t s, uint32_t c, uint32_t nx) {
  if (s == 0 && c < M.tab[PT_LUT].size ()) { /* the root-table entry (mirror_refresh_root_child) is the edge */
    M.n_edges++;
    return;
  }
  const size_t R = 8 * (size_t)s;
  const uint32_t ne = M.tab[PT_REC][R + 1];
  std::vector<std::pair<uint32_t, uint32_t>> row;
  row.reserve (ne + 1);
  if (ne <= 2 && s != 0) {
    if (ne >= 1)
      row.emplace_back (M.tab[PT_REC][R + 4], M.tab[PT_REC][R + 5]);
    if (ne >= 2)
      row.emplace_back (M.tab[PT_REC][R + 6], M.tab[PT_REC][R + 7]);
  } else {
    const uint32_t b = M.tab[PT_REC][R + 2];
    for (uint32_t e = 0; e < ne; e++)
      row.emplace_back (M.tab[PT_EDGE][2 * (size_t)(b + e)], M.tab[PT_EDGE][2 * (size_t)(b + e) + 1]);
  }
  row.insert (std::upper_bound (row.begin (), row.end (), std::make_pair (c, 0u),
                                [] (const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) { return a.first < b.first; }),
              std::make_pair (c, nx));
  const uint32_t nn = ne + 1;
  if (nn > 2 || s == 0) {
    uint32_t begin = M.tab[PT_REC][R + 2], capacity = M.tab[PT_REC][R + 0];
    if (capacity < nn || (ne <= 2 && s != 0)) { /* no slots yet, or no room: a new range at the end */
      capacity = nn * 2 > 4 ? nn * 2 : 4;
      begin = (uint32_t)(M.tab[PT_EDGE].size () / 2);
      M.tab[PT_EDGE].resize ((size_t)(begin + capacity) * 2, 0);
      if ((size_t)(begin + capacity) * 2 > M.cap[PT_EDGE])
        M.full_upload = true;
      M.set (PT_REC, R + 0, capacity);
      M.set (PT_REC, R + 2, begin);
    }
    for (uint32_t e = 0; e < nn; e++) {
      M.set (PT_EDGE, 2 * (size_t)(begin + e), row[e].first);
      M.set (PT_EDGE, 2 * (size_t)(begin + e) + 1, row[e].second);
    }
  }
  M.set (PT_REC, R + 1, nn);
  M.set (PT_REC, R + 4, row[0].first);
  M.set (PT_REC, R + 5, row[0].second);
  M.set (PT_REC, R + 6, nn >= 2 ? row[1].first : 0);
  M.set (PT_REC, R + 7, nn >= 2 ? row[1].second : 0);
  M.n_edges++;
}

/* one new keyword: symbols[0 .. len), keyword id kw */
void
mirror_insert (StartsMirror &M, const uint32_t *symbols, uint32_t len, uint32_t kw) {
  uint32_t s = 0;
  for (uint32_t i = 0; i < len; i++) {
    const uint32_t c = symbols[i];
    uint32_t nx = mirror_child (M, s, c);
    if (nx == NONE) {
      nx = M.n_states++;
      M.parent.push_back (s);
      M.parent_sym.push_back (c);
      M.depth.push_back (i + 1);
      for (int w = 0; w < 8; w++)
        M.set (PT_REC, 8 * (size_t)nx + w, 0);
      if (M.tab[PT_REC].size () < 8 * (size_t)(nx + 1))
        M.tab[PT_REC].resize (8 * (size_t)(nx + 1), 0);
      if (M.tab[PT_PAIRS].size () < 2 * (size_t)(nx + 1))
        M.tab[PT_PAIRS].resize (2 * (size_t)(nx + 1), 0);
      if (M.tab[PT_OINFO].size () < 4 * (size_t)(nx + 1))
        M.tab[PT_OINFO].resize (4 * (size_t)(nx + 1), 0);
      if (8 * (size_t)(nx + 1) > M.cap[PT_REC] || 2 * (size_t)(nx + 1) > M.cap[PT_PAIRS] || 4 * (size_t)(nx + 1) > M.cap[PT_OINFO])
        M.full_upload = true;
      mirror_add_edge (M, s, c, nx);
      if (s == 0)
        mirror_refresh_root_child (M, nx); /* a new child of the root: its root-table entry */
      else if (M.parent[s] == 0 && s != 0) {
        /* s is a child of the root and got another edge: c is now a second symbol, and the
         * child's pair / ALWAYS flag may have changed */
        if (c < M.tab[PT_LUT].size ())
          M.set (PT_LUT, c, M.tab[PT_LUT][c] | ST_SECOND);
        mirror_refresh_root_child (M, s);
      }
    }
    s = nx;
  }
  /* terminal: what a record of this keyword carries (length, keyword id) */
  M.set (PT_REC, 8 * (size_t)s + 3, 1);
  M.set (PT_OINFO, 4 * (size_t)s + 0, 1);
  M.set (PT_OINFO, 4 * (size_t)s + 2, len);
  M.set (PT_OINFO, 4 * (size_t)s + 3, kw);
  if (M.parent[s] == 0 && s != 0)
    mirror_refresh_root_child (M, s);
  if (len > M.lmax)
    M.lmax = len;
  M.n_keywords++;
}

/* maps n symbols of d_text to class ids into the plan's own buffer (grown as needed) */
int
ensure_remap_buffer (ACMPlan *p, size_t bytes, hipStream_t st) {
  if (p->remap_bytes < bytes + 16) {
    if (p->d_remap) {
      HIP_TRY (hipStreamSynchronize (st)); /* an earlier scan may still read the old buffer */
      HIP_TRY (hipFree (p->d_remap));
      p->d_remap = nullptr;
      p->remap_bytes = 0;
    }
    const size_t want = bytes + bytes / 8 + 4096;
    if (hipMalloc (&p->d_remap, want) != hipSuccess)
      return ACM_GPU_E_NOMEM;
    p->remap_bytes = want;
  }
  return ACM_GPU_OK;
}

int
classmap_text (ACMPlan *p, const void *d_text, uint64_t n, hipStream_t st) {
  const uint32_t sb = p->finfo.sym_bytes;
  const size_t bytes = (size_t)n * sb;
  int rc0 = ensure_remap_buffer (p, bytes, st);
  if (rc0)
    return rc0;
  const bool aligned = (reinterpret_cast<uintptr_t> (d_text) & 15) == 0;
  const uint64_t blocks16 = aligned ? bytes / 16 : 0;
  if (blocks16) {
    const uint64_t want_blocks = (blocks16 + 1023) / 1024;
    const uint32_t grid = (uint32_t)(want_blocks < (uint64_t)p->cu_count ? want_blocks : (uint64_t)p->cu_count);
    hipLaunchKernelGGL (classmap_kernel, dim3 (grid), dim3 (1024), 65536 * 2, st, static_cast<const uint4 *> (d_text),
                        static_cast<uint4 *> (p->d_remap), blocks16, p->d_classlut);
  }
  const uint64_t done = blocks16 * 16 / sb;
  if (done < n) {
    const uint64_t left = n - done;
    const uint32_t grid = (uint32_t)((left + 255) / 256 < 4096 ? (left + 255) / 256 : 4096);
    if (sb == 1)
      hipLaunchKernelGGL ((classmap_tail_kernel<uint8_t>), dim3 (grid), dim3 (256), 0, st, static_cast<const uint8_t *> (d_text),
                          static_cast<uint8_t *> (p->d_remap), done, n, p->d_classlut);
    else
      hipLaunchKernelGGL ((classmap_tail_kernel<uint16_t>), dim3 (grid), dim3 (256), 0, st, static_cast<const uint16_t *> (d_text),
                          static_cast<uint16_t *> (p->d_remap), done, n, p->d_classlut);
  }
  HIP_TRY (hipGetLastError ());
  return ACM_GPU_OK;
}

/* comparator classes of 4-byte symbols: class of a symbol the tables have not seen, by bisection
 * among the class representatives with the machine's comparator (0: equal to none of them) */
uint32_t
classify_symbol32 (const ACMPlan *p, uint32_t sym) {
  uint32_t lo = 0, hi = (uint32_t)p->cls32_reps.size ();
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    const int c = p->cmp32 (&sym, &p->cls32_reps[mid], p->cmp32_arg);
    if (c == 0)
      return mid + 1;
    if (c < 0)
      hi = mid;
    else
      lo = mid + 1;
  }
  return 0;
}

/* the device table from everything classified so far (grown to stay at most a quarter full) */
int
upload_cls32_table (ACMPlan *p, hipStream_t st) {
  uint32_t slots = p->cls32_slots ? p->cls32_slots : 1u << 12;
  while ((uint64_t)slots < 4ull * (p->cls32_known.size () + p->cls32_cap))
    slots <<= 1;
  if (slots != p->cls32_slots) {
    if (p->d_cls32) {
      HIP_TRY (hipStreamSynchronize (st));
      HIP_TRY (hipFree (p->d_cls32));
      p->d_cls32 = nullptr;
    }
    if (hipMalloc (reinterpret_cast<void **> (&p->d_cls32), (size_t)slots * 8) != hipSuccess)
      return ACM_GPU_E_NOMEM;
    p->cls32_slots = slots;
  }
  std::vector<unsigned long long> tab (slots, 0ull);
  auto mix = [] (uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
  };
  for (const auto &kv : p->cls32_known) {
    uint32_t h = (uint32_t)mix (kv.first) & (slots - 1);
    while (tab[h])
      h = (h + 1) & (slots - 1);
    tab[h] = ((unsigned long long)(kv.second + 1) << 32) | kv.first;
  }
  HIP_TRY (hipMemcpyWithStream (p->d_cls32, tab.data (), (size_t)slots * 8, hipMemcpyHostToDevice, st));
  p->cls32_uploaded = (uint32_t)p->cls32_known.size ();
  return ACM_GPU_OK;
}

/* maps n 4-byte symbols of d_text to class ids into the plan's own buffer.  Symbols the plan has
 * not met yet come back on a list, are classified here with the comparator and the pass is
 * repeated: the call waits for the stream (once when the text brings nothing new). */
int
classify_text32 (ACMPlan *p, const void *d_text, uint64_t n, hipStream_t st) {
  if (reinterpret_cast<uintptr_t> (d_text) & 3)
    return ACM_GPU_E_ARG;
  int rc = ensure_remap_buffer (p, (size_t)n * 4, st);
  if (rc)
    return rc;
  if (!p->d_cls32 || p->cls32_uploaded != p->cls32_known.size ()) {
    rc = upload_cls32_table (p, st);
    if (rc)
      return rc;
  }
  const uint64_t want_blocks = (n + 255) / 256;
  const uint32_t grid = (uint32_t)(want_blocks < (uint64_t)p->cu_count * 32 ? want_blocks : (uint64_t)p->cu_count * 32);
  std::vector<uint32_t> fresh;
  /* every pass but the last classifies a full list: at most KNOWN_MAX / CAP_MIN passes with a list
   * that never grows, far fewer with one that doubles */
  for (uint32_t round = 0; round < ACMPlan::CLS32_KNOWN_MAX / ACMPlan::CLS32_CAP_MIN + 64; round++) {
    HIP_TRY (hipMemsetAsync (p->d_unknown, 0, 4, st));
    hipLaunchKernelGGL (classify32_kernel, dim3 (grid), dim3 (256), 0, st, static_cast<const uint32_t *> (d_text),
                        static_cast<uint32_t *> (p->d_remap), n, p->d_cls32, p->cls32_slots - 1, p->d_unknown + 1, p->d_unknown,
                        p->cls32_cap);
    HIP_TRY (hipGetLastError ());
    uint32_t cnt = 0;
    HIP_TRY (hipMemcpyWithStream (&cnt, p->d_unknown, 4, hipMemcpyDeviceToHost, st));
    if (cnt == 0)
      return ACM_GPU_OK;
    if (!p->cmp32)
      return ACM_GPU_E_INELIGIBLE; /* tables without their machine: nothing to classify new symbols with */
    const uint32_t listed = cnt < p->cls32_cap ? cnt : p->cls32_cap;
    if (p->cls32_known.size () + listed > ACMPlan::CLS32_KNOWN_MAX) {
      fprintf (stderr, "acm_gpu: the text brings more than %u distinct symbols to a comparator-class plan\n", ACMPlan::CLS32_KNOWN_MAX);
      return ACM_GPU_E_INELIGIBLE;
    }
    fresh.resize (listed);
    HIP_TRY (hipMemcpyWithStream (fresh.data (), p->d_unknown + 1, (size_t)listed * 4, hipMemcpyDeviceToHost, st));
    for (uint32_t sym : fresh)
      if (!p->cls32_known.count (sym))
        p->cls32_known[sym] = classify_symbol32 (p, sym);
    if (cnt >= p->cls32_cap && p->cls32_cap < ACMPlan::CLS32_CAP_MAX) {
      /* the list was full: the text has more to bring -- a longer list for the next pass */
      HIP_TRY (hipStreamSynchronize (st));
      HIP_TRY (hipFree (p->d_unknown));
      p->d_unknown = nullptr;
      p->cls32_cap *= 2;
      if (hipMalloc (reinterpret_cast<void **> (&p->d_unknown), (size_t)(p->cls32_cap + 1) * 4) != hipSuccess)
        return ACM_GPU_E_NOMEM;
    }
    rc = upload_cls32_table (p, st); /* also clears the slots claimed for symbols that did not fit the list */
    if (rc)
      return rc;
  }
  return ACM_GPU_E_INTERNAL;
}

template <bool COUNT_ONLY>
int
scan_impl (ACMPlan *p, const void *d_text, uint64_t n, uint64_t emit_from, uint64_t pos_base, ACMRecord *d_records,
           uint64_t capacity, uint64_t *d_count, hipStream_t st, bool accumulate = false, unsigned long long *shared_total = nullptr) {
  /* shared_total (with accumulate): the running total of ANOTHER plan to add to -- a delta plan
   * appends its records to those of the plan it belongs to */
  /* accumulate (streaming): records are appended after those of earlier calls -- the plan's
   * running total keeps counting and is handed over by acm_gpu_stream_finish, not here */
  HIP_TRY (hipSetDevice (p->device));
  const uint32_t sb = p->finfo.sym_bytes;
  /* (a comparator-class plan walks its own aligned copy of the text) */
  const bool use_dense = p->info.kernel == 1 && (p->d_classlut || (reinterpret_cast<uintptr_t> (d_text) & 15) == 0);
  if (!accumulate && (n == 0 || p->finfo.n_edges == 0 || emit_from >= n || !use_dense))
    HIP_TRY (hipMemsetAsync (d_count, 0, sizeof (uint64_t), st));
  if (n == 0 || p->finfo.n_edges == 0 || emit_from >= n)
    return ACM_GPU_OK;
  if (p->mir && (p->mir->full_upload || !p->mir->patches.empty ())) {
    int rc = starts_flush (p, st); /* dictionary updates since the last scan */
    if (rc)
      return rc;
  }
  if (p->d_intern) {
    /* 8-byte symbols: the kernels walk the 4-byte ids of the text */
    if (reinterpret_cast<uintptr_t> (d_text) & 7)
      return ACM_GPU_E_ARG;
    int rc = ensure_remap_buffer (p, (size_t)n * 4, st);
    if (rc)
      return rc;
    const uint64_t want_blocks = (n + 255) / 256;
    const uint32_t grid = (uint32_t)(want_blocks < (uint64_t)p->cu_count * 32 ? want_blocks : (uint64_t)p->cu_count * 32);
    hipLaunchKernelGGL (intern_kernel, dim3 (grid), dim3 (256), 0, st, static_cast<const uint64_t *> (d_text),
                        static_cast<uint32_t *> (p->d_remap), n, p->d_intern, p->intern_mask);
    HIP_TRY (hipGetLastError ());
    d_text = p->d_remap;
  } else if (p->cls32) {
    int rc = classify_text32 (p, d_text, n, st);
    if (rc)
      return rc;
    d_text = p->d_remap;
  } else if (p->d_classlut) {
    /* comparator-class plan: walk the class ids of the text (our own, aligned, copy) */
    int rc = classmap_text (p, d_text, n, st);
    if (rc)
      return rc;
    d_text = p->d_remap;
  } else if ((p->starts || p->gram) && (reinterpret_cast<uintptr_t> (d_text) & 15) != 0) {
    /* start-parallel / 4-gram plan, buffer not 16-byte aligned: scan an aligned copy (the CSR walk that
     * would take it as it is runs 20x slower, and knows nothing of incremental updates) */
    int rc = ensure_remap_buffer (p, (size_t)n * sb, st);
    if (rc)
      return rc;
    HIP_TRY (hipMemcpyAsync (p->d_remap, d_text, (size_t)n * sb, hipMemcpyDeviceToDevice, st));
    d_text = p->d_remap;
  }
  const bool direct = p->gram && !p->gram_wide; /* records straight from the scan kernel: no item buffer */
  if (use_dense || (!COUNT_ONLY && !direct && (p->gram || p->starts))) {
    /* 4-gram plans over hashed windows see dense matches: room for one hit per 16 symbols, per 8
     * when the dictionary has keywords of 1-3 symbols; past that a wave reserves records 64 at a time */
    int rc = ensure_item_buffer (p, n < p->segment ? n : p->segment, p->gram ? (p->gram_shorts ? 8 : 16) : 256,
                                 use_dense ? DENSE_MIN_REGION_ITEMS : 256);
    if (rc)
      return rc;
  }
  const bool tiled = !COUNT_ONLY && direct && p->tiled_dir != nullptr;
  /* slots per chunk of records (EmitCtx::rec_chunk): big chunks for long texts, when no directory
   * of tiles counts in chunks (ACM_GPU_REC_CHUNK=1024 / 4096: one or the other anyway -- tests) */
  uint32_t rec_chunk = (!tiled && n >= (64ull << 20)) ? REC_CHUNK_BIG : REC_CHUNK;
  if (const char *e = getenv ("ACM_GPU_REC_CHUNK"))
    if (!tiled && (atoi (e) == (int)REC_CHUNK || atoi (e) == (int)REC_CHUNK_BIG))
      rec_chunk = (uint32_t)atoi (e);
  if (!COUNT_ONLY && direct) {
    int rc = ensure_direct_buffers (p, rec_chunk);
    if (rc)
      return rc;
  }
  EmitCtx E{};
  E.oinfo = p->d_oinfo;
  E.records = d_records;
  E.count = shared_total ? shared_total : ((use_dense || accumulate || tiled) ? p->d_total : reinterpret_cast<unsigned long long *> (d_count));
  E.capacity = COUNT_ONLY ? 0 : capacity;
  E.wrows = p->d_wrows;
  E.cont_dh = p->d_cont_dh;
  E.W = p->K.W;
  E.lo = p->K.lo;
  E.span = p->K.span;
  E.chunk = p->chunk;
  E.n_states = p->finfo.n_states;
  E.kw4 = p->d_kw4;
  E.chain = p->d_chain;
  E.chain_base = p->K.HD;
  E.spill = static_cast<uint4 *> (p->d_spill);
  E.spill_slots = (!COUNT_ONLY && direct && !tiled) ? (uint64_t)p->direct_regions * rec_chunk : 0;
  E.rec_chunk = rec_chunk;
  E.chunk_prev = tiled ? p->tiled_prev : nullptr;
  E.error = p->d_total ? reinterpret_cast<unsigned int *> (p->d_total) + 3 : nullptr;

  /* segments of at most SEGMENT symbols; each restarts from the root `halo` symbols early
   * (a multiple of 16 bytes so that the dense kernel keeps its alignment) */
  const uint64_t halo = p->finfo.lmax > 1 ? (((uint64_t)p->finfo.lmax - 1 + 15) / 16) * 16 : 0;
  const uint64_t SEG = p->segment;
  const uint64_t first_seg = emit_from / SEG * SEG; /* earlier segments have nothing to report */
  for (uint64_t seg = first_seg; seg < n; seg += SEG) {
    const uint64_t seg_end = seg + SEG < n ? seg + SEG : n;
    const uint64_t read_begin = seg > halo ? seg - halo : 0;
    Launch a{};
    a.text = static_cast<const unsigned char *> (d_text) + read_begin * sb;
    a.n = (uint32_t)(seg_end - read_begin);
    const uint64_t ef = emit_from > seg ? emit_from : seg;
    a.emit_from = (uint32_t)(ef - read_begin);
    E.pos_base = pos_base + read_begin;
    E.text = a.text;
    E.n = a.n;
    E.emit_from = a.emit_from;
    hipEvent_t stop, stop_all;
    int rc = timing_begin (p, st, &stop, &stop_all);
    if (rc)
      return rc;
    if (use_dense)
      rc = launch_dense<COUNT_ONLY> (p, E, a, st, stop, d_count, seg_end == n && !accumulate);
    else {
      a.range_begin = 0;
      a.range_end = a.n;
      if (p->gram)
        rc = launch_gram<COUNT_ONLY> (p, E, a, st, stop, seg == first_seg, seg_end == n);
      else if (p->starts && (reinterpret_cast<uintptr_t> (a.text) & 15) == 0)
        rc = launch_starts<COUNT_ONLY> (p, E, a, st, stop);
      else {
        if (p->sparse && (reinterpret_cast<uintptr_t> (a.text) & 15) == 0)
          rc = launch_sparse<COUNT_ONLY> (p, E, a, st);
        else
          rc = launch_csr<COUNT_ONLY> (p, E, a, st);
        if (!rc && stop)
          HIP_TRY (hipEventRecord (stop, st));
      }
    }
    if (!rc && stop_all) /* behind the expansion / hole closing the launch functions enqueue after their scan kernel */
      HIP_TRY (hipEventRecord (stop_all, st));
    if (rc) {
      /* earlier segments may have left a partial running total and expand ticket behind */
      if (use_dense && p->d_total)
        (void)hipMemsetAsync (p->d_total, 0, 16, st);
      return rc;
    }
  }
  /* narrow alphabets: the keywords of 1-3 symbols, a pass of their own over the same segments
   * (dev_short.h); their records follow the 4-gram pass' in the same buffer */
  for (uint64_t seg = first_seg; p->gram && p->short_pass && !use_dense && seg < n; seg += SEG) {
    const uint64_t seg_end = seg + SEG < n ? seg + SEG : n;
    const uint64_t read_begin = seg > 16 ? seg - 16 : 0; /* (a halo of 2 symbols would do: 16 keeps the alignment) */
    Launch a{};
    a.text = static_cast<const unsigned char *> (d_text) + read_begin * sb;
    a.n = (uint32_t)(seg_end - read_begin);
    const uint64_t ef = emit_from > seg ? emit_from : seg;
    a.emit_from = (uint32_t)(ef - read_begin);
    E.pos_base = pos_base + read_begin;
    E.text = a.text;
    E.n = a.n;
    E.emit_from = a.emit_from;
    hipEvent_t stop, stop_all;
    int rc = timing_begin (p, st, &stop, &stop_all);
    if (!rc)
      rc = launch_short<COUNT_ONLY> (p, E, a, st, stop, seg == first_seg, seg_end == n);
    if (!rc && stop_all)
      HIP_TRY (hipEventRecord (stop_all, st));
    if (rc)
      return rc;
  }
  return ACM_GPU_OK;
}

/* a plan and, if it has one, its delta (acm_gpu_plan_update): both scans append to the same record
 * buffer through the plan's running total, handed to the caller's counter at the end */
template <bool COUNT_ONLY>
int
scan_plan (ACMPlan *p, const void *d_text, uint64_t n, uint64_t emit_from, uint64_t pos_base, ACMRecord *d_records,
           uint64_t capacity, uint64_t *d_count, hipStream_t st, bool accumulate = false) {
  if (!p->retired.empty ()) {
    HIP_TRY (hipSetDevice (p->device));
    /* deltas replaced since the last scan: whatever may still use them is on this stream, in front of this mark */
    for (auto &r : p->retired)
      if (!r.done) {
        HIP_TRY (hipEventCreateWithFlags (&r.done, hipEventDisableTiming));
        HIP_TRY (hipEventRecord (r.done, st));
      }
  }
  if (!p->delta)
    return scan_impl<COUNT_ONLY> (p, d_text, n, emit_from, pos_base, d_records, capacity, d_count, st, accumulate);
  int rc = scan_impl<COUNT_ONLY> (p, d_text, n, emit_from, pos_base, d_records, capacity, d_count, st, true);
  if (!rc)
    rc = scan_impl<COUNT_ONLY> (p->delta, d_text, n, emit_from, pos_base, d_records, capacity, d_count, st, true, p->d_total);
  p->delta_scanned += n; /* what acm_gpu_plan_update weighs against the cost of one plan of everything */
  if (!accumulate) { /* (also after a failure: the total must not leak into the next scan) */
    hipLaunchKernelGGL (finish_count_kernel, dim3 (1), dim3 (64), 0, st, p->d_total, reinterpret_cast<unsigned long long *> (d_count));
    HIP_TRY (hipGetLastError ());
  }
  return rc;
}

} // namespace

extern "C" int
acm_gpu_scan_device (ACMPlan *plan, const void *d_text, uint64_t n_symbols, uint64_t emit_from, uint64_t pos_base,
                     ACMRecord *d_records, uint64_t capacity, uint64_t *d_count, void *stream) {
  if (!plan || !d_count || (n_symbols && !d_text) || (capacity && !d_records))
    return ACM_GPU_E_ARG;
  return scan_plan<false> (plan, d_text, n_symbols, emit_from, pos_base, d_records, capacity, d_count,
                           static_cast<hipStream_t> (stream));
}

extern "C" int
acm_gpu_count_device (ACMPlan *plan, const void *d_text, uint64_t n_symbols, uint64_t emit_from, uint64_t *d_count,
                      void *stream) {
  if (!plan || !d_count || (n_symbols && !d_text))
    return ACM_GPU_E_ARG;
  return scan_plan<true> (plan, d_text, n_symbols, emit_from, 0, nullptr, 0, d_count, static_cast<hipStream_t> (stream));
}

/* ------------------------------------------------------------------ streaming scan
 * The reference's callers read their text piece by piece (generic_test.c:191 uses fgetwc) and the
 * scan is resumable by construction: the cursor depends on the last lmax symbols only.  A stream
 * keeps two device slots; piece k is copied into slot k % 2 on a copy stream while piece k - 1 is
 * scanned on the compute stream; the last `halo` symbols of the stream so far are placed in front
 * of every piece so that matches straddling pieces are found (reported once: emit_from = halo). */
struct ACMStream {
  ACMPlan *plan = nullptr;
  uint32_t sb = 1;
  uint64_t halo = 0, max_piece = 0;
  unsigned char *slot[2] = { nullptr, nullptr }; /* (halo + max_piece) symbols each */
  hipStream_t copy = nullptr, compute = nullptr;
  hipEvent_t copied[2] = { nullptr, nullptr }, scanned[2] = { nullptr, nullptr }, tail_read[2] = { nullptr, nullptr };
  ACMRecord *d_records = nullptr;
  uint64_t capacity = 0;
  uint64_t position = 0; /* symbols fed so far */
  uint64_t pieces = 0;
  uint64_t prev_valid = 0; /* context + piece symbols held, contiguously, by the previous slot */
  uint64_t last_piece = 0; /* length of the previous piece */
  uint64_t prev_piece_len () const { return last_piece; }
};

extern "C" int
acm_gpu_stream_open (ACMPlan *plan, uint64_t max_piece_symbols, uint64_t record_capacity, ACMStream **out) {
  if (!plan || !out || max_piece_symbols == 0)
    return ACM_GPU_E_ARG;
  HIP_TRY (hipSetDevice (plan->device));
  ACMStream *s = new (std::nothrow) ACMStream ();
  if (!s)
    return ACM_GPU_E_NOMEM;
  s->plan = plan;
  s->sb = plan->text_sym_bytes; /* of the caller's text (8-byte symbols are interned inside the scan) */
  s->halo = plan->finfo.lmax > 1 ? (((uint64_t)plan->finfo.lmax - 1 + 15) / 16) * 16 : 0;
  s->max_piece = (max_piece_symbols + 15) / 16 * 16;
  s->capacity = record_capacity;
  const size_t slot_bytes = (size_t)(s->halo + s->max_piece) * s->sb + 16;
  bool ok = hipMalloc (reinterpret_cast<void **> (&s->slot[0]), slot_bytes) == hipSuccess &&
            hipMalloc (reinterpret_cast<void **> (&s->slot[1]), slot_bytes) == hipSuccess &&
            hipMalloc (reinterpret_cast<void **> (&s->d_records), (record_capacity ? record_capacity : 1) * sizeof (ACMRecord)) == hipSuccess &&
            hipStreamCreateWithFlags (&s->copy, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags (&s->compute, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; ok && i < 2; i++)
    ok = hipEventCreateWithFlags (&s->copied[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags (&s->scanned[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags (&s->tail_read[i], hipEventDisableTiming) == hipSuccess;
  /* a stream owns the plan's running total while it is open: start from zero.  Earlier scans of
   * the plan (on whatever stream) must have drained, and the memset (null stream) must have landed
   * before the stream's own non-blocking streams touch the total: without the waits a warm
   * process lost the first piece's count now and then */
  ok = ok && hipDeviceSynchronize () == hipSuccess && hipMemset (plan->d_total, 0, 8) == hipSuccess &&
       hipDeviceSynchronize () == hipSuccess;
  if (!ok) {
    acm_gpu_stream_close (s);
    return ACM_GPU_E_NOMEM;
  }
  *out = s;
  return ACM_GPU_OK;
}

/* Feeds the next n_symbols of the stream from HOST memory (pinned memory makes the copy truly
 * asynchronous).  Returns as soon as the copy and the scan are enqueued; `text` must stay
 * untouched until the second next feed or acm_gpu_stream_finish has returned: before a slot is
 * filled again the host waits for the copy that filled it two pieces ago, so by the time feed k
 * returns the buffers of the pieces up to k - 2 have been read. */
extern "C" int
acm_gpu_stream_feed (ACMStream *s, const void *text, uint64_t n_symbols) {
  if (!s || (n_symbols && !text))
    return ACM_GPU_E_ARG;
  ACMPlan *p = s->plan;
  HIP_TRY (hipSetDevice (p->device));
  const unsigned char *src = static_cast<const unsigned char *> (text);
  while (n_symbols) {
    const uint64_t n = n_symbols < s->max_piece ? n_symbols : s->max_piece;
    const int cur = (int)(s->pieces & 1), prev = cur ^ 1;
    unsigned char *piece_at = s->slot[cur] + s->halo * s->sb;
    /* the slot is free once the scan that used it two pieces ago is done and the previous piece
     * has taken its context from the slot's tail */
    if (s->pieces >= 2) {
      HIP_TRY (hipEventSynchronize (s->copied[cur])); /* the caller may now reuse that piece's buffer */
      HIP_TRY (hipStreamWaitEvent (s->copy, s->scanned[cur], 0));
    }
    if (s->pieces >= 1)
      HIP_TRY (hipStreamWaitEvent (s->copy, s->tail_read[cur], 0));
    HIP_TRY (hipMemcpyAsync (piece_at, src, (size_t)n * s->sb, hipMemcpyHostToDevice, s->copy));
    HIP_TRY (hipEventRecord (s->copied[cur], s->copy));
    /* context: the last `ctx` symbols the previous slot holds, end-aligned in front of the piece */
    const uint64_t ctx = s->prev_valid < s->halo ? s->prev_valid : s->halo;
    HIP_TRY (hipStreamWaitEvent (s->compute, s->copied[cur], 0));
    if (ctx) {
      const unsigned char *tail = s->slot[prev] + (s->halo + s->prev_piece_len () - ctx) * s->sb;
      HIP_TRY (hipMemcpyAsync (piece_at - ctx * s->sb, tail, (size_t)ctx * s->sb, hipMemcpyDeviceToDevice, s->compute));
    }
    HIP_TRY (hipEventRecord (s->tail_read[prev], s->compute));
    int rc = scan_plan<false> (p, piece_at - ctx * s->sb, ctx + n, ctx, s->position - ctx, s->d_records, s->capacity, nullptr,
                               s->compute, true);
    if (rc)
      return rc;
    HIP_TRY (hipEventRecord (s->scanned[cur], s->compute));
    s->prev_valid = ctx + n;
    s->last_piece = n;
    s->position += n;
    s->pieces++;
    src += (size_t)n * s->sb;
    n_symbols -= n;
  }
  return ACM_GPU_OK;
}

/* Waits for everything fed so far, puts the records in canonical order and copies them to the
 * host.  *n_found = matches of the whole stream so far (ACM_GPU_E_OVERFLOW if more than the
 * stream's record capacity or than `capacity`).  The stream stays open and keeps accumulating. */
extern "C" int
acm_gpu_stream_finish (ACMStream *s, ACMRecord *records, uint64_t capacity, uint64_t *n_found) {
  if (!s || !n_found)
    return ACM_GPU_E_ARG;
  ACMPlan *p = s->plan;
  HIP_TRY (hipSetDevice (p->device));
  HIP_TRY (hipStreamSynchronize (s->copy));
  HIP_TRY (hipStreamSynchronize (s->compute));
  unsigned long long total = 0;
  HIP_TRY (hipMemcpy (&total, p->d_total, 8, hipMemcpyDeviceToHost));
  *n_found = total;
  if (total > s->capacity || total > capacity)
    return ACM_GPU_E_OVERFLOW;
  if (total > 1) {
    const size_t tb = acm_gpu_order_tmp_bytes (p, total, s->position);
    void *tmp = nullptr;
    if (hipMalloc (&tmp, tb) != hipSuccess)
      return ACM_GPU_E_NOMEM;
    int rc = acm_gpu_order_records_device (p, s->d_records, total, 0, s->position, tmp, tb, s->compute);
    if (!rc && hipStreamSynchronize (s->compute) != hipSuccess)
      rc = ACM_GPU_E_HIP;
    (void)hipFree (tmp);
    if (rc)
      return rc;
  }
  if (total && records)
    HIP_TRY (hipMemcpy (records, s->d_records, total * sizeof (ACMRecord), hipMemcpyDeviceToHost));
  return acm_gpu_plan_status (p);
}

extern "C" void
acm_gpu_stream_close (ACMStream *s) {
  if (!s)
    return;
  (void)hipSetDevice (s->plan->device);
  if (s->copy)
    (void)hipStreamSynchronize (s->copy);
  if (s->compute)
    (void)hipStreamSynchronize (s->compute);
  (void)hipMemset (s->plan->d_total, 0, 8); /* hand the plan back with a clean running total */
  (void)hipDeviceSynchronize ();
  for (int i = 0; i < 2; i++) {
    if (s->slot[i]) (void)hipFree (s->slot[i]);
    if (s->copied[i]) (void)hipEventDestroy (s->copied[i]);
    if (s->scanned[i]) (void)hipEventDestroy (s->scanned[i]);
    if (s->tail_read[i]) (void)hipEventDestroy (s->tail_read[i]);
  }
  if (s->d_records) (void)hipFree (s->d_records);
  if (s->copy) (void)hipStreamDestroy (s->copy);
  if (s->compute) (void)hipStreamDestroy (s->compute);
  delete s;
}

/* ------------------------------------------------------------------ canonical order */
namespace {
size_t
cub_sort_bytes (uint64_t n) {
  size_t tmp = 0;
  hipcub::DoubleBuffer<uint64_t> k (nullptr, nullptr);
  hipcub::DoubleBuffer<Rec16> v (nullptr, nullptr);
  (void)hipcub::DeviceRadixSort::SortPairs (nullptr, tmp, k, v, (int)n, 0, 64, nullptr);
  return tmp;
}
size_t
align256 (size_t x) {
  return (x + 255) & ~(size_t)255;
}
} // namespace

extern "C" size_t
acm_gpu_sort_tmp_bytes (uint64_t n) {
  if (n == 0)
    return 256;
  return align256 (n * 8) * 2 + align256 (n * 16) + align256 (cub_sort_bytes (n)) + 256;
}

namespace {
int radix_sort_records (ACMPlan *plan, ACMRecord *d_records, uint64_t n, void *d_tmp, size_t tmp_bytes, void *stream, uint64_t pos_lo, int key_bits);
}

extern "C" int
acm_gpu_sort_records_device (ACMPlan *plan, ACMRecord *d_records, uint64_t n, void *d_tmp, size_t tmp_bytes, void *stream) {
  return radix_sort_records (plan, d_records, n, d_tmp, tmp_bytes, stream, 0, 64);
}

namespace {
/* key = (end_pos - pos_lo) << len_bits | (max - length); only the low key_bits bits are sorted on
 * (a caller that knows the range of the positions saves the radix passes over bits that are zero) */
int
radix_sort_records (ACMPlan *plan, ACMRecord *d_records, uint64_t n, void *d_tmp, size_t tmp_bytes, void *stream, uint64_t pos_lo, int key_bits) {
  if (!plan || (n && (!d_records || !d_tmp)))
    return ACM_GPU_E_ARG;
  if (n <= 1)
    return ACM_GPU_OK;
  if (n >= (1ull << 31) || tmp_bytes < acm_gpu_sort_tmp_bytes (n))
    return ACM_GPU_E_ARG;
  HIP_TRY (hipSetDevice (plan->device));
  hipStream_t st = static_cast<hipStream_t> (stream);
  unsigned char *t = static_cast<unsigned char *> (d_tmp);
  uint64_t *k0 = reinterpret_cast<uint64_t *> (t);
  uint64_t *k1 = reinterpret_cast<uint64_t *> (t + align256 (n * 8));
  Rec16 *v1 = reinterpret_cast<Rec16 *> (t + 2 * align256 (n * 8));
  void *cub_tmp = t + 2 * align256 (n * 8) + align256 (n * 16);
  size_t cub_bytes = cub_sort_bytes (n);
  uint32_t len_bits = 1;
  while ((1u << len_bits) <= plan->finfo.lmax)
    len_bits++;
  /* key = end_pos in the high bits, (max - length) below: 64 - len_bits bits remain for positions */
  hipLaunchKernelGGL (make_keys_kernel, dim3 ((uint32_t)((n + 255) / 256)), dim3 (256), 0, st, d_records, n, len_bits, pos_lo, k0);
  HIP_TRY (hipGetLastError ());
  hipcub::DoubleBuffer<uint64_t> keys (k0, k1);
  hipcub::DoubleBuffer<Rec16> vals (reinterpret_cast<Rec16 *> (d_records), v1);
  HIP_TRY (hipcub::DeviceRadixSort::SortPairs (cub_tmp, cub_bytes, keys, vals, (int)n, 0, key_bits, st));
  if (vals.Current () != reinterpret_cast<Rec16 *> (d_records))
    HIP_TRY (hipMemcpyAsync (d_records, vals.Current (), n * 16, hipMemcpyDeviceToDevice, st));
  return ACM_GPU_OK;
}
} // namespace

/* ---- canonical order of records whose positions lie in [pos_lo, pos_lo + span): dev_order.h */
namespace {
struct OrderPlan {
  uint32_t wlog = 0, n_buckets = 0, len_bits = 1, key_bits = 64;
  bool sparse = false; /* fewer than 8 records per 4,096 positions: pass C goes by windows of buckets, not by bucket */
  size_t o_hist = 0, o_cur = 0, o_rec = 0, o_cub = 0, cub_bytes = 0, total = 0;
  bool ok = false;
};
OrderPlan
order_layout (const ACMPlan *plan, uint64_t n, uint64_t span) {
  OrderPlan L;
  if (n == 0 || span == 0 || n >= (1ull << 31))
    return L;
  uint32_t span_bits = 1, len_bits = 1;
  while ((1ull << span_bits) < span && span_bits < 63)
    span_bits++;
  const uint32_t lmax = plan->finfo.lmax > (plan->delta ? plan->delta->finfo.lmax : 0) ? plan->finfo.lmax : (plan->delta ? plan->delta->finfo.lmax : 0);
  while ((1u << len_bits) <= lmax)
    len_bits++;
  L.len_bits = len_bits;
  L.key_bits = span_bits + len_bits < 64 ? (int)(span_bits + len_bits) : 64;
  /* buckets of ORDER_POSITIONS positions (fewer when the whole range is shorter): whatever a
   * bucket holds, order_count_role has a counter per position for it */
  uint32_t wlog = 0;
  while ((2u << wlog) <= ORDER_POSITIONS && (1ull << wlog) < span)
    wlog++;
  L.sparse = order_is_sparse (n, span);
  const uint64_t nb = (span >> wlog) + 1;
  if (nb >= (1ull << 28) || span_bits + len_bits > 63 || wlog + len_bits > 31) /* (a bucket's keys are 32-bit) */
    return L;
  L.wlog = wlog;
  L.n_buckets = (uint32_t)nb;
  size_t cub = 0;
  (void)hipcub::DeviceScan::ExclusiveSum (nullptr, cub, static_cast<uint32_t *> (nullptr), static_cast<uint32_t *> (nullptr), (int)(nb + 1), nullptr);
  L.cub_bytes = cub;
  size_t cur = 0;
  L.o_hist = blob_reserve (cur, (nb + 1) * 4 + (nb + 2) * 4); /* counts, then (o_cur, right behind them) a zero word and the sums */
  L.o_cur = L.o_hist + (nb + 1) * 4;
  L.o_rec = blob_reserve (cur, n * sizeof (ACMRecord));
  L.o_cub = blob_reserve (cur, cub + 16);
  L.total = cur + 256;
  L.ok = true;
  return L;
}
} // namespace

extern "C" size_t
acm_gpu_order_tmp_bytes (const ACMPlan *plan, uint64_t n, uint64_t span) {
  if (!plan)
    return 0;
  const OrderPlan L = order_layout (plan, n, span);
  const size_t radix = acm_gpu_sort_tmp_bytes (n);
  return L.ok && L.total > radix ? L.total : radix; /* (room for the fallback either way) */
}

namespace {
/* what is cleared in front of pass A: the counts and cur[0] -- rounded up to whole 256 bytes (one
 * fill kernel instead of a body and a tail; what lies behind is the sums' own space) */
size_t
order_zero_bytes (const OrderPlan &L) {
  return (((size_t)L.n_buckets + 2) * 4 + 255) & ~(size_t)255;
}

/* n_dev == nullptr: n records.  Else: the record count is the scan's, in device memory, and n the
 * capacity of d_records (OrderK::n_dev) -- nothing here waits for the host. */
bool
order_by_buckets (const ACMPlan *plan, const OrderPlan &L) {
  const char *env = getenv ("ACM_GPU_ORDER"); /* radix: always the radix sort (experiments, tests) */
  (void)plan;
  return L.ok && !(env && strcmp (env, "radix") == 0);
}

int
order_records (ACMPlan *plan, ACMRecord *d_records, uint64_t n, const unsigned long long *n_dev, uint64_t pos_lo, uint64_t span, void *d_tmp,
               size_t tmp_bytes, void *stream) {
  if (tmp_bytes < acm_gpu_order_tmp_bytes (plan, n, span))
    return ACM_GPU_E_ARG;
  const OrderPlan L = order_layout (plan, n, span);
  if (!order_by_buckets (plan, L)) {
    if (n_dev)
      return ACM_GPU_E_ARG; /* (the caller asks order_by_buckets first) */
    return radix_sort_records (plan, d_records, n, d_tmp, tmp_bytes, stream, pos_lo, L.key_bits);
  }
  HIP_TRY (hipSetDevice (plan->device));
  hipStream_t st = static_cast<hipStream_t> (stream);
  unsigned char *t = static_cast<unsigned char *> (d_tmp);
  uint32_t *hist = reinterpret_cast<uint32_t *> (t + L.o_hist), *cur = reinterpret_cast<uint32_t *> (t + L.o_cur);
  ACMRecord *bucketed = reinterpret_cast<ACMRecord *> (t + L.o_rec);
  OrderK K{};
  K.in = d_records;
  K.n = n;
  K.pos_lo = pos_lo;
  K.wlog = L.wlog;
  K.n_buckets = L.n_buckets;
  K.len_bits = L.len_bits;
  K.error = plan->d_total ? reinterpret_cast<unsigned int *> (plan->d_total) + 3 : nullptr;
  K.n_dev = n_dev;
  K.span = span;
  K.mode = n_dev ? 2u : (L.sparse ? 1u : 0u);
  /* `hist`: the buckets' counts (pass A).  `cur`: one word that stays 0, then the buckets' exclusive
   * prefix sums -- where each bucket begins, the cursors pass B advances; when it is done cur[1 + b]
   * is where bucket b ENDS, so that P = cur reads P[b] = begin, P[b + 1] = end for pass C (no copy
   * of the sums is kept) */
  const uint64_t pieces = (n + ORDER_PIECE - 1) / ORDER_PIECE, pblocks = (pieces + ORDER_THREADS / WAVE - 1) / (ORDER_THREADS / WAVE);
  const uint32_t grid = (uint32_t)(pblocks < (uint64_t)plan->cu_count * 8 ? pblocks : (uint64_t)plan->cu_count * 8);
  HIP_TRY (hipMemsetAsync (hist, 0, order_zero_bytes (L), st)); /* (the counts and cur[0], which lies right behind them) */
  hipLaunchKernelGGL (order_bucket_kernel<false>, dim3 (grid), dim3 (ORDER_THREADS), 0, st, K, hist, static_cast<ACMRecord *> (nullptr));
  HIP_TRY (hipGetLastError ());
  size_t cub = L.cub_bytes;
  HIP_TRY (hipcub::DeviceScan::ExclusiveSum (t + L.o_cub, cub, hist, cur + 1, (int)(L.n_buckets + 1), st));
  hipLaunchKernelGGL (order_bucket_kernel<true>, dim3 (grid), dim3 (ORDER_THREADS), 0, st, K, cur + 1, bucketed);
  HIP_TRY (hipGetLastError ());
  /* pass C: buckets (dense record sets) or windows of buckets (sparse ones) of up to 256 records by
   * a wave each, crowded buckets by a block each (each role skips the others' buckets) */
  uint32_t wgrid = 0, sgrid = 0;
  if (n_dev || L.sparse) { /* (with the count on the device both are there: the one whose kind of set it is not returns at once) */
    const uint64_t windows = (n + ORDER_WINDOW - 1) / ORDER_WINDOW, wblocks = (windows + 3) / 4;
    wgrid = (uint32_t)(wblocks < (uint64_t)plan->cu_count * 16 ? wblocks : (uint64_t)plan->cu_count * 16);
  }
  if (n_dev || !L.sparse)
    sgrid = (uint32_t)((L.n_buckets + 3) / 4 < (uint32_t)plan->cu_count * 16 ? (L.n_buckets + 3) / 4 : (uint32_t)plan->cu_count * 16);
  const uint32_t cblocks = (L.n_buckets + ORDER_COUNT_THREADS - 1) / ORDER_COUNT_THREADS;
  const uint32_t cgrid = cblocks < (uint32_t)plan->cu_count * 8 ? cblocks : (uint32_t)plan->cu_count * 8;
  hipLaunchKernelGGL (order_finish_kernel, dim3 (wgrid + sgrid + cgrid), dim3 (256), 0, st, K, cur, bucketed, d_records, wgrid, sgrid);
  HIP_TRY (hipGetLastError ());
  return ACM_GPU_OK;
}
} // namespace

extern "C" int
acm_gpu_order_records_device (ACMPlan *plan, ACMRecord *d_records, uint64_t n, uint64_t pos_lo, uint64_t span, void *d_tmp,
                              size_t tmp_bytes, void *stream) {
  if (!plan || (n && (!d_records || !d_tmp)))
    return ACM_GPU_E_ARG;
  if (n <= 1)
    return ACM_GPU_OK;
  return order_records (plan, d_records, n, nullptr, pos_lo, span, d_tmp, tmp_bytes, stream);
}

/* ---- tiled scans (dev_tiles.h): 4-gram plans over narrow alphabets */
namespace {
struct TiledPlan {
  bool ok = false;
  uint32_t n_tiles = 0;
  uint64_t raw_slots = 0;
  uint32_t len_bits = 1, nsub = 2;
  size_t o_raw = 0, o_prev = 0, o_dir = 0, o_size = 0, o_begin = 0, o_crowded = 0, o_over = 0, o_cub = 0, cub_bytes = 0, total = 0;
};

/* the tiles of all the launches scan_impl makes for this text (the same walk over the segments).
 * bound: an upper bound of the layout over every emit_from (acm_gpu_scan_ordered_tmp_bytes): the
 * tiles are counted at the smallest R gram_tiling ever picks.  (Round 3 sized the scratch with
 * emit_from = 0 "for the most tiles" -- but R = clamp (groups behind emit_from / (16 per wave), 4,
 * 64), so a later emit_from can LOWER R and give more tiles than emit_from = 0: 1 GiB on 256 CUs
 * is 16,384 tiles at emit_from = 0 and 20,480 when 81,919 groups are left.) */
TiledPlan
tiled_layout (const ACMPlan *p, uint64_t capacity, uint64_t n, uint64_t emit_from, bool bound = false) {
  TiledPlan L;
  const char *env = getenv ("ACM_GPU_ORDER"); /* radix / buckets: not this way (experiments, tests) */
  if (env && (strcmp (env, "radix") == 0 || strcmp (env, "buckets") == 0))
    return L;
  if (!p->gram || p->gram_wide || p->short_pass || p->delta || p->finfo.lmax > WAVE * 16 || p->finfo.n_edges == 0)
    return L; /* (a second pass' records do not lie tile by tile: the general order passes) */
  if (n == 0 || emit_from >= n || capacity == 0 || capacity >= (1ull << 31))
    return L;
  const uint64_t halo = p->finfo.lmax > 1 ? (((uint64_t)p->finfo.lmax - 1 + 15) / 16) * 16 : 0;
  const uint64_t SEG = p->segment;
  uint64_t tiles = 0;
  for (uint64_t seg = emit_from / SEG * SEG; seg < n; seg += SEG) {
    const uint64_t seg_end = seg + SEG < n ? seg + SEG : n;
    const uint64_t read_begin = seg > halo ? seg - halo : 0;
    const uint64_t ef = emit_from > seg ? emit_from : seg;
    GramTiling T = gram_tiling (p, (uint32_t)(seg_end - read_begin), (uint32_t)(ef - read_begin));
    if (bound) { /* every group of the segment, GRAM_R_MIN groups per tile */
      const uint32_t ngroups = (uint32_t)((seg_end - read_begin + WAVE * 16 - 1) / (WAVE * 16));
      T.begin = 0;
      T.end = (ngroups + GRAM_R_MIN - 1) / GRAM_R_MIN;
    }
    tiles += T.end - T.begin;
    if (T.R * (WAVE * 16) / (1u << TILE_BUCKET_LOG2) + 1 > L.nsub)
      L.nsub = T.R * (WAVE * 16) / (1u << TILE_BUCKET_LOG2) + 1;
  }
  if (tiles == 0 || tiles >= (1ull << 30))
    return L;
  L.n_tiles = (uint32_t)tiles;
  /* whole chunks: the records and what every wave may leave unused of its last chunk */
  L.raw_slots = (capacity + REC_CHUNK - 1) / REC_CHUNK * REC_CHUNK + ((uint64_t)p->cu_count * (SPARSE_THREADS / WAVE) + 1) * REC_CHUNK;
  while ((1u << L.len_bits) <= p->finfo.lmax)
    L.len_bits++;
  size_t cub = 0;
  (void)hipcub::DeviceScan::ExclusiveSum (nullptr, cub, static_cast<uint32_t *> (nullptr), static_cast<uint32_t *> (nullptr), (int)(L.n_tiles + 1), nullptr);
  L.cub_bytes = cub;
  size_t cur = 0;
  L.o_raw = blob_reserve (cur, L.raw_slots * sizeof (ACMRecord));
  L.o_prev = blob_reserve (cur, (L.raw_slots / REC_CHUNK) * 4);
  L.o_dir = blob_reserve (cur, (size_t)L.n_tiles * sizeof (TileEntry));
  L.o_size = blob_reserve (cur, ((size_t)L.n_tiles + 1) * 4);
  L.o_begin = blob_reserve (cur, ((size_t)L.n_tiles + 1) * 4);
  L.o_crowded = blob_reserve (cur, ((size_t)L.n_tiles + 1) * 4);
  L.o_over = blob_reserve (cur, 8);
  L.o_cub = blob_reserve (cur, cub + 16);
  L.total = cur + 256;
  L.ok = true;
  return L;
}

int
scan_tiled (ACMPlan *plan, const TiledPlan &L, const void *d_text, uint64_t n_symbols, uint64_t emit_from, uint64_t pos_base, ACMRecord *d_records,
            uint64_t capacity, uint64_t *d_count, void *d_tmp, hipStream_t st) {
  HIP_TRY (hipSetDevice (plan->device));
  unsigned char *t = static_cast<unsigned char *> (d_tmp);
  plan->tiled_dir = t + L.o_dir;
  plan->tiled_prev = reinterpret_cast<uint32_t *> (t + L.o_prev);
  plan->tiled_base = 0;
  int rc = scan_impl<false> (plan, d_text, n_symbols, emit_from, pos_base, reinterpret_cast<ACMRecord *> (t + L.o_raw), L.raw_slots, d_count, st);
  const uint32_t written = plan->tiled_base;
  plan->tiled_dir = nullptr;
  plan->tiled_prev = nullptr;
  plan->tiled_base = 0;
  if (rc || written != L.n_tiles) {
    (void)hipMemsetAsync (plan->d_total, 0, 8, st); /* (the scan's running total must not leak into the next one) */
    return rc ? rc : ACM_GPU_E_INTERNAL;
  }
  TileK K{};
  K.raw = reinterpret_cast<const ACMRecord *> (t + L.o_raw);
  K.chunk_prev = reinterpret_cast<const uint32_t *> (t + L.o_prev);
  K.dir = reinterpret_cast<const TileEntry *> (t + L.o_dir);
  K.n_tiles = L.n_tiles;
  K.size = reinterpret_cast<uint32_t *> (t + L.o_size);
  K.begin = reinterpret_cast<const uint32_t *> (t + L.o_begin);
  K.out = d_records;
  K.capacity = capacity;
  K.d_count = reinterpret_cast<unsigned long long *> (d_count);
  K.reserved = plan->d_total;
  K.len_bits = L.len_bits;
  K.nsub = L.nsub;
  K.crowded = reinterpret_cast<uint32_t *> (t + L.o_crowded);
  K.raw_slots = L.raw_slots;
  K.over_total = reinterpret_cast<unsigned long long *> (t + L.o_over);
  HIP_TRY (hipMemsetAsync (K.over_total, 0, 8, st));
  K.error = reinterpret_cast<unsigned int *> (plan->d_total) + 3;
  const uint32_t sblocks = (L.n_tiles + 1 + 3) / 4;
  hipLaunchKernelGGL (tile_size_kernel, dim3 (sblocks < (uint32_t)plan->cu_count * 16 ? sblocks : (uint32_t)plan->cu_count * 16), dim3 (256), 0, st, K);
  HIP_TRY (hipGetLastError ());
  size_t cub = L.cub_bytes;
  HIP_TRY (hipcub::DeviceScan::ExclusiveSum (t + L.o_cub, cub, K.size, reinterpret_cast<uint32_t *> (t + L.o_begin), (int)(L.n_tiles + 1), st));
  const uint32_t gblocks = L.n_tiles < (uint32_t)plan->cu_count * 16 ? L.n_tiles : (uint32_t)plan->cu_count * 16;
  hipLaunchKernelGGL (tile_gather_kernel, dim3 (gblocks), dim3 (TILE_THREADS), tile_lds_bytes (L.nsub), st, K);
  HIP_TRY (hipGetLastError ());
  /* the tiles it found crowded (dense matches; none on ordinary texts: the kernel returns at once) */
  const uint32_t cblocks = L.n_tiles < (uint32_t)plan->cu_count * 4 ? L.n_tiles : (uint32_t)plan->cu_count * 4;
  hipLaunchKernelGGL (tile_crowded_kernel, dim3 (cblocks), dim3 (TILE_THREADS), tile_lds_bytes (L.nsub), st, K);
  HIP_TRY (hipGetLastError ());
  return ACM_GPU_OK;
}
} // namespace

/* Scan and canonical order in one call, nothing but kernel launches on `stream`: the order passes
 * take the number of records from *d_count on the device.  A scan that overflows `capacity` leaves
 * the total in *d_count as acm_gpu_scan_device does and nothing in order (the caller repeats it with
 * room).  4-gram plans over narrow alphabets scan in tiles and order in one pass (dev_tiles.h).
 * Record sets the bucket passes do not take (2^31 records or more, positions past 2^63 /
 * lengths): the count comes to the host and acm_gpu_order_records_device's fallback runs. */
extern "C" size_t
acm_gpu_scan_ordered_tmp_bytes (const ACMPlan *plan, uint64_t capacity, uint64_t n_symbols) {
  if (!plan)
    return 0;
  const size_t general = acm_gpu_order_tmp_bytes (plan, capacity, n_symbols);
  const TiledPlan L = tiled_layout (plan, capacity, n_symbols, 0, true); /* (an upper bound over every emit_from) */
  return L.ok && L.total > general ? L.total : general;
}

extern "C" int
acm_gpu_scan_ordered_device (ACMPlan *plan, const void *d_text, uint64_t n_symbols, uint64_t emit_from, uint64_t pos_base,
                             ACMRecord *d_records, uint64_t capacity, uint64_t *d_count, void *d_tmp, size_t tmp_bytes, void *stream) {
  if (!plan || !d_count || (n_symbols && !d_text) || (capacity && (!d_records || !d_tmp)))
    return ACM_GPU_E_ARG;
  if (capacity && tmp_bytes < acm_gpu_scan_ordered_tmp_bytes (plan, capacity, n_symbols))
    return ACM_GPU_E_ARG;
  const TiledPlan T = tiled_layout (plan, capacity, n_symbols, emit_from);
  /* (the layout of THIS emit_from must fit what the caller gave -- the query is an upper bound, so
   * it does; a buffer sized some other way takes the general passes, never a write past its end) */
  if (T.ok && T.total <= tmp_bytes)
    return scan_tiled (plan, T, d_text, n_symbols, emit_from, pos_base, d_records, capacity, d_count, d_tmp, static_cast<hipStream_t> (stream));
  const OrderPlan L = order_layout (plan, capacity, n_symbols);
  int rc = acm_gpu_scan_device (plan, d_text, n_symbols, emit_from, pos_base, d_records, capacity, d_count, stream);
  if (rc || capacity == 0 || n_symbols == 0)
    return rc;
  if (order_by_buckets (plan, L))
    return order_records (plan, d_records, capacity, reinterpret_cast<const unsigned long long *> (d_count), pos_base, n_symbols, d_tmp, tmp_bytes, stream);
  uint64_t found = 0;
  HIP_TRY (hipMemcpyAsync (&found, d_count, 8, hipMemcpyDeviceToHost, static_cast<hipStream_t> (stream)));
  HIP_TRY (hipStreamSynchronize (static_cast<hipStream_t> (stream)));
  if (found <= 1 || found > capacity)
    return ACM_GPU_OK;
  return order_records (plan, d_records, found, nullptr, pos_base, n_symbols, d_tmp, tmp_bytes, stream);
}

/* ------------------------------------------------------------------ host-buffer convenience */
extern "C" int
acm_gpu_scan_host (ACMPlan *plan, const void *text, uint64_t n_symbols, uint64_t emit_from, uint64_t pos_base,
                   ACMRecord *records, uint64_t capacity, uint64_t *n_found) {
  if (!plan || !n_found || (n_symbols && !text) || (capacity && !records))
    return ACM_GPU_E_ARG;
  HIP_TRY (hipSetDevice (plan->device));
  const size_t tbytes = (size_t)n_symbols * plan->text_sym_bytes;
  void *d_text = nullptr, *d_rec = nullptr, *d_tmp = nullptr;
  uint64_t *d_count = nullptr;
  int rc = ACM_GPU_OK;
  uint64_t found = 0;
  auto cleanup = [&] () {
    if (d_text) (void)hipFree (d_text);
    if (d_rec) (void)hipFree (d_rec);
    if (d_tmp) (void)hipFree (d_tmp);
    if (d_count) (void)hipFree (d_count);
  };
#define HOST_TRY(expr)                                                                             \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      fprintf (stderr, "acm_gpu: %s failed: %s\n", #expr, hipGetErrorString (_e));                  \
      cleanup ();                                                                                  \
      return _e == hipErrorOutOfMemory ? ACM_GPU_E_NOMEM : ACM_GPU_E_HIP;                          \
    }                                                                                              \
  } while (0)
  HOST_TRY (hipMalloc (&d_text, tbytes ? tbytes : 16));
  HOST_TRY (hipMalloc (reinterpret_cast<void **> (&d_count), 8));
  HOST_TRY (hipMalloc (&d_rec, capacity ? capacity * 16 : 16));
  if (tbytes)
    HOST_TRY (hipMemcpy (d_text, text, tbytes, hipMemcpyHostToDevice));
  rc = acm_gpu_scan_device (plan, d_text, n_symbols, emit_from, pos_base, static_cast<ACMRecord *> (d_rec), capacity, d_count, nullptr);
  if (rc) {
    cleanup ();
    return rc;
  }
  HOST_TRY (hipMemcpy (&found, d_count, 8, hipMemcpyDeviceToHost));
  *n_found = found;
  if (found > capacity) {
    cleanup ();
    return ACM_GPU_E_OVERFLOW;
  }
  if (found > 1) {
    size_t tb = acm_gpu_order_tmp_bytes (plan, found, n_symbols);
    HOST_TRY (hipMalloc (&d_tmp, tb));
    rc = acm_gpu_order_records_device (plan, static_cast<ACMRecord *> (d_rec), found, pos_base, n_symbols, d_tmp, tb, nullptr);
    if (rc) {
      cleanup ();
      return rc;
    }
  }
  if (found)
    HOST_TRY (hipMemcpy (records, d_rec, found * 16, hipMemcpyDeviceToHost));
  HOST_TRY (hipDeviceSynchronize ());
  cleanup ();
  return ACM_GPU_OK;
#undef HOST_TRY
}

/* ------------------------------------------------------------------ records on the wire (include/acm_gpu.h) */
extern "C" int
acm_gpu_wire_bits (const ACMPlan *plan, uint64_t span, uint32_t *pos_bits, uint32_t *len_bits, uint32_t *kw_bits) {
  if (!plan || !pos_bits || !len_bits || !kw_bits)
    return ACM_GPU_E_ARG;
  auto bits = [] (uint64_t v) { /* bits that hold every value in [0, v] */
    uint32_t b = 1;
    while (b < 64 && (v >> b))
      b++;
    return b;
  };
  const uint64_t kw_max = (uint64_t)plan->covered_keywords + plan->finfo.n_keywords + plan->kw_base; /* (an upper bound: a delta's ids follow the plan's) */
  *pos_bits = bits (span ? span - 1 : 0);
  *len_bits = bits (plan->finfo.lmax);
  *kw_bits = bits (kw_max);
  return *pos_bits + *len_bits + *kw_bits <= 64 ? ACM_GPU_OK : ACM_GPU_E_INELIGIBLE;
}

extern "C" int
acm_gpu_pack_records_device (const ACMRecord *d_records, uint64_t n, uint64_t pos_lo, uint32_t pos_bits, uint32_t len_bits, uint64_t *d_packed,
                             void *stream) {
  if ((n && (!d_records || !d_packed)) || pos_bits == 0 || pos_bits + len_bits >= 64)
    return ACM_GPU_E_ARG;
  if (n == 0)
    return ACM_GPU_OK;
  const uint64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL (pack_records_kernel, dim3 ((uint32_t)(blocks < 16384 ? blocks : 16384)), dim3 (256), 0, static_cast<hipStream_t> (stream), d_records, n,
                      pos_lo, pos_bits, len_bits, d_packed);
  HIP_TRY (hipGetLastError ());
  return ACM_GPU_OK;
}

extern "C" int
acm_gpu_unpack_records_device (const uint64_t *d_packed, uint64_t n, uint64_t pos_lo, uint32_t pos_bits, uint32_t len_bits, ACMRecord *d_records,
                               void *stream) {
  if ((n && (!d_records || !d_packed)) || pos_bits == 0 || pos_bits + len_bits >= 64)
    return ACM_GPU_E_ARG;
  if (n == 0)
    return ACM_GPU_OK;
  const uint64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL (unpack_records_kernel, dim3 ((uint32_t)(blocks < 16384 ? blocks : 16384)), dim3 (256), 0, static_cast<hipStream_t> (stream), d_packed, n,
                      pos_lo, pos_bits, len_bits, d_records);
  HIP_TRY (hipGetLastError ());
  return ACM_GPU_OK;
}

/* ------------------------------------------------------------------ several GPUs, one process (include/acm_gpu.h)
 * contiguous shards, an lmax - 1 halo, tables replicated, no collective on the data
 * path; the one exchange step is the gather of the ordered records on devices[0] by direct peer
 * copies (reference model: one shared machine, one cursor per worker, README.md:364). */
struct ACMMulti {
  std::vector<int> dev;        /* per shard */
  std::vector<int> distinct;   /* the devices in use, devices[0] first */
  std::vector<ACMPlan *> plan; /* per distinct device */
  std::vector<hipStream_t> stream;
  uint32_t lmax = 0, sym_bytes = 1;
  /* what a shard's scan needs on its device, kept from call to call (grow-only): hipMalloc and
   * hipFree wait for the device, and a scan of config 4 wants 7 GB of records + twice that of
   * scratch per shard -- allocated once, sized by what the call before found */
  struct ShardBuf {
    ACMRecord *rec = nullptr;
    uint64_t rec_cap = 0;
    uint64_t *cnt = nullptr;
    void *tmp = nullptr;
    size_t tmp_cap = 0;
    uint64_t last_found = 0, last_span = 0;
  };
  std::vector<ShardBuf> buf; /* per shard */
  uint64_t *h_found = nullptr; /* pinned, one count per shard */
  /* records on the wire: a shard of another device packs its ordered records to 8 bytes each
   * (acm_gpu_pack_records_device), sends them to this staging area on devices[0], and they are
   * unpacked into their place there -- half the bytes over the shard's one link to the root */
  uint64_t *stage0 = nullptr;
  uint64_t stage0_cap = 0; /* in 8-byte words */
  std::vector<hipEvent_t> arrived; /* per shard: its packed records have landed on devices[0] */
  int slot_of (int device) const {
    for (size_t i = 0; i < distinct.size (); i++)
      if (distinct[i] == device)
        return (int)i;
    return -1;
  }
};

extern "C" void
acm_gpu_multi_destroy (ACMMulti *mu) {
  if (!mu)
    return;
  for (size_t i = 0; i < mu->distinct.size (); i++) {
    (void)hipSetDevice (mu->distinct[i]);
    if (i < mu->stream.size () && mu->stream[i]) {
      (void)hipStreamSynchronize (mu->stream[i]);
      (void)hipStreamDestroy (mu->stream[i]);
    }
    if (i < mu->plan.size () && mu->plan[i])
      acm_gpu_plan_destroy (mu->plan[i]);
  }
  for (size_t r = 0; r < mu->buf.size (); r++) {
    (void)hipSetDevice (mu->dev[r]);
    if (mu->buf[r].rec) (void)hipFree (mu->buf[r].rec);
    if (mu->buf[r].cnt) (void)hipFree (mu->buf[r].cnt);
    if (mu->buf[r].tmp) (void)hipFree (mu->buf[r].tmp);
  }
  if (mu->h_found)
    (void)hipHostFree (mu->h_found);
  if (mu->stage0) {
    (void)hipSetDevice (mu->dev[0]);
    (void)hipFree (mu->stage0);
  }
  for (size_t r = 0; r < mu->arrived.size (); r++)
    if (mu->arrived[r]) {
      (void)hipSetDevice (mu->dev[r]);
      (void)hipEventDestroy (mu->arrived[r]);
    }
  delete mu;
}

extern "C" int
acm_gpu_multi_create (ACMachine *machine, const int *devices, int n_shards, ACMMulti **out) {
  if (!machine || !devices || n_shards < 1 || !out)
    return ACM_GPU_E_ARG;
  int ndev = 0;
  if (hipGetDeviceCount (&ndev) != hipSuccess || ndev <= 0)
    return ACM_GPU_E_NODEVICE;
  ACMMulti *mu = new (std::nothrow) ACMMulti ();
  if (!mu)
    return ACM_GPU_E_NOMEM;
  for (int r = 0; r < n_shards; r++) {
    if (devices[r] < 0 || devices[r] >= ndev) {
      delete mu;
      return ACM_GPU_E_ARG;
    }
    mu->dev.push_back (devices[r]);
    if (mu->slot_of (devices[r]) < 0)
      mu->distinct.push_back (devices[r]);
  }
  /* one snapshot of the dictionary for every device: the same tables everywhere */
  ACMFlat *flat = nullptr;
  int rc = acm_flatten (machine, &flat);
  if (rc) {
    delete mu;
    return rc;
  }
  ACMFlatInfo fi;
  acm_flat_info (flat, &fi);
  mu->lmax = fi.lmax;
  mu->sym_bytes = fi.sym_bytes;
  mu->plan.assign (mu->distinct.size (), nullptr);
  mu->stream.assign (mu->distinct.size (), nullptr);
  mu->buf.assign (mu->dev.size (), ACMMulti::ShardBuf ());
  if (hipHostMalloc (reinterpret_cast<void **> (&mu->h_found), mu->dev.size () * sizeof (uint64_t), hipHostMallocDefault) != hipSuccess) {
    mu->h_found = nullptr;
    acm_flat_release (flat);
    delete mu;
    return ACM_GPU_E_NOMEM;
  }
  for (size_t i = 0; i < mu->distinct.size () && !rc; i++) {
    rc = acm_gpu_plan_create_flat (flat, mu->distinct[i], &mu->plan[i]);
    if (!rc && (hipSetDevice (mu->distinct[i]) != hipSuccess || hipStreamCreateWithFlags (&mu->stream[i], hipStreamNonBlocking) != hipSuccess))
      rc = ACM_GPU_E_HIP;
    /* direct peer copies into devices[0] where the hardware offers them (otherwise the runtime stages the copy) */
    if (!rc && i > 0) {
      int can = 0;
      if (hipDeviceCanAccessPeer (&can, mu->distinct[i], mu->distinct[0]) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess (mu->distinct[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          (void)hipGetLastError ();
      }
    }
  }
  acm_flat_release (flat);
  if (rc) {
    acm_gpu_multi_destroy (mu);
    return rc;
  }
  *out = mu;
  return ACM_GPU_OK;
}

extern "C" int
acm_gpu_multi_shard_bounds (const ACMMulti *mu, uint64_t n, int shard, uint64_t *read_begin, uint64_t *own_begin, uint64_t *own_end) {
  if (!mu || shard < 0 || shard >= (int)mu->dev.size ())
    return ACM_GPU_E_ARG;
  const uint64_t R = mu->dev.size ();
  /* the rule of sharded.shard_bounds (rank r owns [r N / R, (r + 1) N / R)), with the halo grown to
   * the next 16-byte boundary of the text so that a shard's buffer can keep the text's alignment */
  const uint64_t b = n / R * (uint64_t)shard + n % R * (uint64_t)shard / R, e = n / R * (uint64_t)(shard + 1) + n % R * (uint64_t)(shard + 1) / R;
  const uint64_t per16 = 16 / mu->sym_bytes ? 16 / mu->sym_bytes : 1;
  const uint64_t warm = mu->lmax > 1 ? mu->lmax - 1 : 0;
  uint64_t rb = b > warm ? b - warm : 0;
  rb = rb / per16 * per16;
  if (read_begin)
    *read_begin = rb;
  if (own_begin)
    *own_begin = b;
  if (own_end)
    *own_end = e;
  return ACM_GPU_OK;
}

namespace {
/* shards scanned, ordered and gathered: d_text[r] on dev[r] holds [read_begin_r, own_end_r) */
int
multi_scan (ACMMulti *mu, const void *const *d_text, uint64_t n, ACMRecord *d_out, uint64_t capacity, uint64_t *n_found) {
  const size_t R = mu->dev.size ();
  struct Shard {
    uint64_t rb = 0, b = 0, e = 0;
    int slot = 0;
  };
  std::vector<Shard> sh (R);
  int rc = ACM_GPU_OK;
#define MULTI_TRY(expr)                                                                            \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      fprintf (stderr, "acm_gpu: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString (_e), __FILE__, __LINE__); \
      for (size_t i_ = 0; i_ < mu->distinct.size (); i_++) {                                       \
        (void)hipSetDevice (mu->distinct[i_]);                                                     \
        (void)hipStreamSynchronize (mu->stream[i_]);                                               \
      }                                                                                            \
      return _e == hipErrorOutOfMemory ? ACM_GPU_E_NOMEM : ACM_GPU_E_HIP;                          \
    }                                                                                              \
  } while (0)
  /* a shard's record buffer and scratch: at least `want` records (grow-only, kept by the handle) */
  auto ensure = [&] (size_t r, uint64_t want) -> int {
    ACMMulti::ShardBuf &B = mu->buf[r];
    const Shard &s = sh[r];
    if (!B.cnt && hipMalloc (reinterpret_cast<void **> (&B.cnt), 8) != hipSuccess)
      return ACM_GPU_E_NOMEM;
    if (B.rec_cap < want) {
      if (B.rec)
        (void)hipFree (B.rec);
      B.rec = nullptr;
      B.rec_cap = 0;
      if (hipMalloc (reinterpret_cast<void **> (&B.rec), want * sizeof (ACMRecord)) != hipSuccess)
        return ACM_GPU_E_NOMEM;
      B.rec_cap = want;
    }
    const size_t tb = s.e > s.b ? acm_gpu_scan_ordered_tmp_bytes (mu->plan[s.slot], B.rec_cap, s.e - s.rb) : 0;
    if (B.tmp_cap < tb) {
      if (B.tmp)
        (void)hipFree (B.tmp);
      B.tmp = nullptr;
      B.tmp_cap = 0;
      if (hipMalloc (&B.tmp, tb) != hipSuccess)
        return ACM_GPU_E_NOMEM;
      B.tmp_cap = tb;
    }
    return ACM_GPU_OK;
  };
  /* first pass: every shard into a record buffer sized by what the call before found on a text of
   * this length (+ an eighth), or by a guess (one match per 32 symbols); the counts tell which
   * shards need a second pass with the exact size */
  for (int pass = 0; pass < 2; pass++) {
    bool any = false;
    for (size_t r = 0; r < R; r++) {
      Shard &s = sh[r];
      ACMMulti::ShardBuf &B = mu->buf[r];
      if (pass == 0) {
        s.slot = mu->slot_of (mu->dev[r]);
        (void)acm_gpu_multi_shard_bounds (mu, n, (int)r, &s.rb, &s.b, &s.e);
      } else if (mu->h_found[r] <= B.rec_cap)
        continue; /* the first pass had room for everything it found */
      any = true;
      MULTI_TRY (hipSetDevice (mu->dev[r]));
      uint64_t want = mu->h_found[r];
      if (pass == 0) {
        const uint64_t span = s.e - s.b;
        want = B.last_found && B.last_span == span ? B.last_found + B.last_found / 8 + 4096 : span / 32 + 4096;
        if (want < B.rec_cap)
          want = B.rec_cap;
      }
      rc = ensure (r, want);
      if (rc) {
        for (size_t i = 0; i < mu->distinct.size (); i++) {
          (void)hipSetDevice (mu->distinct[i]);
          (void)hipStreamSynchronize (mu->stream[i]);
        }
        return rc;
      }
      /* (shards of one device share its stream: their scans run one after the other.)  Scan and
       * canonical order in one call: nothing waits for the host between them, and 4-gram plans scan
       * in tiles and order in one pass */
      if (s.e > s.b) {
        rc = acm_gpu_scan_ordered_device (mu->plan[s.slot], d_text[r], s.e - s.rb, s.b - s.rb, s.rb, B.rec, B.rec_cap, B.cnt, B.tmp, B.tmp_cap, mu->stream[s.slot]);
        if (rc) {
          for (size_t i = 0; i < mu->distinct.size (); i++) {
            (void)hipSetDevice (mu->distinct[i]);
            (void)hipStreamSynchronize (mu->stream[i]);
          }
          return rc;
        }
      } else
        MULTI_TRY (hipMemsetAsync (B.cnt, 0, 8, mu->stream[s.slot]));
      MULTI_TRY (hipMemcpyAsync (&mu->h_found[r], B.cnt, 8, hipMemcpyDeviceToHost, mu->stream[s.slot]));
    }
    if (!any)
      break;
    for (size_t i = 0; i < mu->distinct.size (); i++) {
      MULTI_TRY (hipSetDevice (mu->distinct[i]));
      MULTI_TRY (hipStreamSynchronize (mu->stream[i]));
    }
  }
  uint64_t total = 0;
  for (size_t r = 0; r < R; r++) {
    total += mu->h_found[r];
    mu->buf[r].last_found = mu->h_found[r];
    mu->buf[r].last_span = sh[r].e - sh[r].b;
  }
  *n_found = total;
  if (total > capacity)
    return ACM_GPU_E_OVERFLOW;
  /* (every shard's records are in canonical order where they are:) each shard's run into its place
   * on devices[0] -- from another device as 8-byte words when the fields fit (acm_gpu_wire_bits):
   * packed into the shard's scratch, sent to the staging area, unpacked there behind an event */
  const char *wire_env = getenv ("ACM_GPU_WIRE"); /* 0: 16-byte records over the links; 2: the wire form for shards of devices[0] too (tests on one GPU) */
  const bool wire_off = wire_env && atoi (wire_env) == 0, wire_all = wire_env && atoi (wire_env) == 2;
  uint64_t remote = 0;
  for (size_t r = 0; r < R; r++)
    if (mu->dev[r] != mu->dev[0] || wire_all)
      remote += mu->h_found[r];
  if (remote > mu->stage0_cap && !wire_off) {
    MULTI_TRY (hipSetDevice (mu->dev[0]));
    if (mu->stage0)
      MULTI_TRY (hipFree (mu->stage0));
    mu->stage0 = nullptr;
    mu->stage0_cap = 0;
    MULTI_TRY (hipMalloc (reinterpret_cast<void **> (&mu->stage0), remote * 8));
    mu->stage0_cap = remote;
  }
  if (mu->arrived.size () < R)
    mu->arrived.resize (R, nullptr);
  uint64_t off = 0, soff = 0;
  for (size_t r = 0; r < R; r++) {
    const Shard &s = sh[r];
    const uint64_t found = mu->h_found[r];
    MULTI_TRY (hipSetDevice (mu->dev[r]));
    hipStream_t st = mu->stream[s.slot];
    if (found) {
      uint32_t pb = 0, lb = 0, kb = 0;
      if (mu->dev[r] == mu->dev[0] && !wire_all)
        MULTI_TRY (hipMemcpyAsync (d_out + off, mu->buf[r].rec, found * sizeof (ACMRecord), hipMemcpyDeviceToDevice, st));
      else if (!wire_off && acm_gpu_wire_bits (mu->plan[s.slot], s.e - s.rb, &pb, &lb, &kb) == ACM_GPU_OK && found * 8 <= mu->buf[r].tmp_cap) {
        uint64_t *packed = static_cast<uint64_t *> (mu->buf[r].tmp); /* (the scan's scratch: it is done with it) */
        rc = acm_gpu_pack_records_device (mu->buf[r].rec, found, s.rb, pb, lb, packed, st);
        if (!rc) {
          MULTI_TRY (hipMemcpyPeerAsync (mu->stage0 + soff, mu->dev[0], packed, mu->dev[r], found * 8, st));
          if (!mu->arrived[r])
            MULTI_TRY (hipEventCreateWithFlags (&mu->arrived[r], hipEventDisableTiming));
          MULTI_TRY (hipEventRecord (mu->arrived[r], st));
          MULTI_TRY (hipSetDevice (mu->dev[0]));
          MULTI_TRY (hipStreamWaitEvent (mu->stream[0], mu->arrived[r], 0));
          rc = acm_gpu_unpack_records_device (mu->stage0 + soff, found, s.rb, pb, lb, d_out + off, mu->stream[0]);
        }
        if (rc) {
          for (size_t i = 0; i < mu->distinct.size (); i++) {
            (void)hipSetDevice (mu->distinct[i]);
            (void)hipStreamSynchronize (mu->stream[i]);
          }
          return rc;
        }
        soff += found;
      } else
        MULTI_TRY (hipMemcpyPeerAsync (d_out + off, mu->dev[0], mu->buf[r].rec, mu->dev[r], found * sizeof (ACMRecord), st));
    }
    off += found;
  }
  for (size_t i = 0; i < mu->distinct.size (); i++) {
    MULTI_TRY (hipSetDevice (mu->distinct[i]));
    MULTI_TRY (hipStreamSynchronize (mu->stream[i]));
  }
  for (size_t i = 0; i < mu->distinct.size () && !rc; i++)
    rc = acm_gpu_plan_status (mu->plan[i]);
  return rc;
#undef MULTI_TRY
}
} // namespace

extern "C" int
acm_gpu_multi_scan_device (ACMMulti *mu, const void *const *d_shard_text, uint64_t n, ACMRecord *d_records, uint64_t capacity,
                           uint64_t *n_found) {
  if (!mu || !n_found || (n && !d_shard_text) || (capacity && !d_records))
    return ACM_GPU_E_ARG;
  return multi_scan (mu, d_shard_text, n, d_records, capacity, n_found);
}

extern "C" int
acm_gpu_multi_scan_host (ACMMulti *mu, const void *text, uint64_t n, ACMRecord *records, uint64_t capacity, uint64_t *n_found) {
  if (!mu || !n_found || (n && !text) || (capacity && !records))
    return ACM_GPU_E_ARG;
  const size_t R = mu->dev.size ();
  std::vector<void *> d_text (R, nullptr);
  ACMRecord *d_out = nullptr;
  int rc = ACM_GPU_OK;
  auto cleanup = [&] () {
    for (size_t r = 0; r < R; r++)
      if (d_text[r]) {
        (void)hipSetDevice (mu->dev[r]);
        (void)hipFree (d_text[r]);
      }
    if (d_out) {
      (void)hipSetDevice (mu->dev[0]);
      (void)hipFree (d_out);
    }
  };
  for (size_t r = 0; r < R && !rc; r++) {
    uint64_t rb, b, e;
    (void)acm_gpu_multi_shard_bounds (mu, n, (int)r, &rb, &b, &e);
    const size_t bytes = (size_t)(e - rb) * mu->sym_bytes;
    const int slot = mu->slot_of (mu->dev[r]);
    if (hipSetDevice (mu->dev[r]) != hipSuccess || hipMalloc (&d_text[r], bytes + 16) != hipSuccess)
      rc = ACM_GPU_E_NOMEM;
    else if (bytes && hipMemcpyAsync (d_text[r], static_cast<const unsigned char *> (text) + (size_t)rb * mu->sym_bytes, bytes, hipMemcpyHostToDevice,
                                      mu->stream[slot]) != hipSuccess)
      rc = ACM_GPU_E_HIP;
  }
  if (!rc && (hipSetDevice (mu->dev[0]) != hipSuccess || hipMalloc (reinterpret_cast<void **> (&d_out), (capacity ? capacity : 1) * sizeof (ACMRecord)) != hipSuccess))
    rc = ACM_GPU_E_NOMEM;
  if (!rc)
    rc = multi_scan (mu, d_text.data (), n, d_out, capacity, n_found);
  if (!rc && *n_found) {
    if (hipSetDevice (mu->dev[0]) != hipSuccess || hipMemcpy (records, d_out, *n_found * sizeof (ACMRecord), hipMemcpyDeviceToHost) != hipSuccess)
      rc = ACM_GPU_E_HIP;
  }
  cleanup ();
  return rc;
}

/* the reference's dictionaries grow while they are used (README.md:352-356,
 * generic_test.c:214-229).  Brings `plan` up to date with `machine` (the machine it was made from,
 * later): plans of the start-parallel kernel take the new keywords as edits of a few table words
 * (StartsMirror); every other plan keeps its tables and gets the new keywords as a small delta
 * plan scanned beside it (below), merged into one plan again only now and then. */
extern "C" int
acm_gpu_plan_update (ACMPlan *plan, ACMachine *machine) {
  if (!plan || !machine)
    return ACM_GPU_E_ARG;
  const uint64_t gen = acm_internal_generation (machine);
  uint32_t sym_bytes = 0;
  const bool plain = acm_internal_symbol_bytes (machine, &sym_bytes) == ACM_GPU_OK;
  if (plan->mir && plan->starts && plain && !plan->class_sym_bytes && sym_bytes == plan->finfo.sym_bytes) {
    StartsMirror &M = *plan->mir;
    acm_internal_lock (machine);
    const uint32_t nk = (uint32_t)acm_nb_keywords (machine);
    if (nk < M.n_keywords) {
      acm_internal_unlock (machine);
      return ACM_GPU_E_ARG; /* not the machine this plan came from */
    }
    MatchHolder h;
    acm_matcher_init (&h);
    std::vector<uint32_t> sym;
    int rc = ACM_GPU_OK;
    for (uint32_t k = M.n_keywords; k < nk && rc == ACM_GPU_OK; k++) {
      rc = acm_internal_get_keyword (machine, k, &h); /* the machine lock is held */
      if (rc)
        break;
      sym.resize (h.length);
      for (size_t i = 0; i < h.length; i++) {
        const unsigned char *l = static_cast<const unsigned char *> (h.letters[i]);
        uint32_t v = 0;
        for (uint32_t bb = 0; bb < sym_bytes; bb++)
          v |= (uint32_t)l[bb] << (8 * bb);
        sym[i] = v;
      }
      if (M.n_states + h.length >= ST_STATE)
        rc = ACM_GPU_E_INELIGIBLE;
      else
        mirror_insert (M, sym.data (), (uint32_t)h.length, k);
    }
    acm_matcher_release (&h);
    acm_internal_unlock (machine);
    if (rc)
      return rc;
    plan->finfo.n_states = M.n_states;
    plan->finfo.n_edges = M.n_edges;
    plan->finfo.n_keywords = M.n_keywords;
    plan->covered_keywords = M.n_keywords;
    plan->finfo.lmax = M.lmax;
    plan->generation = gen;
    return ACM_GPU_OK;
  }
  /* Every other kind of plan has tables that are functions of the whole dictionary (a new keyword
   * changes a column of up to every failure-resolved row, the rank of every later 4-gram ...).
   * They are left alone: the keywords the machine got since the plan was made go into a small
   * delta plan of their own, rebuilt from just those keywords at every update -- a cost that does
   * not depend on the size of the dictionary -- and scanned after the plan itself (scan_plan).
   * Only when the delta has grown past an eighth of the dictionary (at least 256 keywords) is
   * everything rebuilt into one plan again.  (`gen` was read before the snapshots: keywords
   * inserted while they are taken leave the plan stale, never wrongly fresh.) */
  HIP_TRY (hipSetDevice (plan->device));
  for (size_t i = 0; i < plan->retired.size ();) { /* deltas no scan can be using any more */
    ACMPlan::Retired &r = plan->retired[i];
    if (r.done && hipEventQuery (r.done) == hipSuccess) {
      (void)hipEventDestroy (r.done);
      acm_gpu_plan_destroy (r.plan);
      plan->retired.erase (plan->retired.begin () + (long)i);
    } else
      i++;
  }
  const uint32_t base_kw = plan->finfo.n_keywords;
  acm_internal_lock (machine);
  const uint32_t nk = (uint32_t)acm_nb_keywords (machine);
  if (nk < plan->covered_keywords) {
    acm_internal_unlock (machine);
    return ACM_GPU_E_ARG; /* not the machine this plan came from */
  }
  /* A delta is a second pass over every text: on 1 GiB of config 2's text 0.57 ms per scan against
   * 0.32 without one (tools/exp_delta_big.py: a dense delta pass runs at the dense kernel's rate
   * whatever its size), while one plan of everything costs ~3.5 ms per 1,000 keywords once.  So the
   * delta is also given up when the texts scanned with it add up to more than that is worth:
   * 14 Gi symbols per 1,000 keywords of the dictionary (acm_scan asks after every scan). */
  const bool scanned_enough = plan->delta && plan->delta_scanned > (uint64_t)(base_kw > 1000 ? base_kw : 1000) * (14ull << 20);
  if (nk == plan->covered_keywords && !scanned_enough) {
    acm_internal_unlock (machine);
    plan->generation = gen;
    return ACM_GPU_OK;
  }
  const uint32_t threshold = base_kw / 8 > 256 ? base_kw / 8 : 256;
  const char *delta_env = getenv ("ACM_GPU_DELTA"); /* 0: always rebuild (experiments) */
  if (nk - base_kw <= threshold && !scanned_enough && !(delta_env && atoi (delta_env) == 0)) {
    /* the new keywords into a machine of their own: same comparator, the main machine's letters */
    CMP_TYPE cmp;
    void *cmp_arg;
    acm_internal_comparator (machine, &cmp, &cmp_arg);
    ACMachine *tm = acm_create (cmp, cmp_arg, 0);
    MatchHolder h;
    acm_matcher_init (&h);
    int rc = ACM_GPU_OK;
    uint32_t lmax = plan->finfo.lmax;
    for (uint32_t k = base_kw; k < nk && rc == ACM_GPU_OK; k++) {
      rc = acm_internal_get_keyword (machine, k, &h); /* the machine lock is held */
      if (rc)
        break;
      ACState *cur = acm_initiate (tm);
      for (size_t i = 0; i < h.length; i++)
        acm_insert_letter_of_keyword (&cur, const_cast<void *> (h.letters[i]));
      (void)acm_insert_end_of_keyword (&cur, 0, 0);
      if (h.length > lmax)
        lmax = (uint32_t)h.length;
    }
    acm_matcher_release (&h);
    acm_internal_unlock (machine);
    ACMFlat *flat = nullptr;
    if (!rc)
      rc = plan->class_sym_bytes ? acm_flatten_classes (tm, plan->class_sym_bytes, &flat) : acm_flatten (tm, &flat);
    acm_release (tm);
    ACMPlan *fresh = nullptr;
    if (!rc) {
      rc = plan_create_flat_kw (flat, plan->device, base_kw, &fresh);
      acm_flat_release (flat);
    }
    if (rc)
      return rc;
    fresh->class_sym_bytes = plan->class_sym_bytes;
    fresh->cmp32 = plan->cmp32;
    fresh->cmp32_arg = plan->cmp32_arg;
    fresh->segment = plan->segment;
    fresh->items_owner = plan;
    if (plan->delta)
      plan->retired.push_back (ACMPlan::Retired{ plan->delta, nullptr });
    plan->delta = fresh;
    plan->covered_keywords = nk;
    plan->finfo.lmax = lmax; /* halos and sort keys go by the longest keyword of both */
    plan->generation = gen;
    return ACM_GPU_OK;
  }
  acm_internal_unlock (machine);
  /* the delta has outgrown its share: one plan of everything, behind the same handle */
  ACMPlan *fresh = nullptr;
  int rc = plan->class_sym_bytes ? acm_gpu_plan_create_classes (machine, plan->class_sym_bytes, plan->device, &fresh)
                                 : acm_gpu_plan_create (machine, plan->device, &fresh);
  if (rc)
    return rc;
  HIP_TRY (hipDeviceSynchronize ()); /* scans in flight still read the old tables */
  const bool timing = plan->timing;
  const uint64_t segment = plan->segment;
  const uint32_t merges = plan->merges + 1;
  std::swap (*plan, *fresh);
  acm_gpu_plan_destroy (fresh); /* the old tables, their delta and what was retired */
  plan->segment = segment;
  plan->merges = merges;
  plan->generation = gen;
  if (timing)
    (void)acm_gpu_plan_timing (plan, 1);
  return ACM_GPU_OK;
}

namespace {
/* acm_release drops the cached plan through this hook: set once, when the library is loaded */
struct PlanDropperInit {
  PlanDropperInit () { acm_internal_plan_dropper = drop_cached_plan; }
} plan_dropper_init;
} // namespace

/* The reference lets many threads work on one shared machine (README.md:364); the cached plan and
 * its scratch buffers serve one scan at a time, so concurrent acm_scan calls on one machine queue
 * up on the machine's plan lock (threads that want to scan in parallel make a plan each). */
extern "C" int
acm_scan (ACMachine *machine, const void *text, uint64_t n_symbols, ACMRecord *records, uint64_t capacity, uint64_t *n_found) {
  if (!machine || !n_found)
    return ACM_GPU_E_ARG;
  /* which path (include/acm_gpu.h): ACM_CMP_DEFAULT over 1/2/4/8 bytes -> GPU; another comparator
   * with its symbol size declared -> GPU over its classes (1/2/4 bytes) or the caller loop on the host */
  uint32_t own_bytes = 0;
  const bool plain = acm_internal_symbol_bytes (machine, &own_bytes) == ACM_GPU_OK;
  uint32_t said = acm_internal_declared_symbol_bytes (machine);
  if (!plain && said == 0) {
    /* (ACM_CMP_DEFAULT over symbols of another size says the size itself: memcmp's length) */
    CMP_TYPE cmp = nullptr;
    void *cmp_arg = nullptr;
    acm_internal_comparator (machine, &cmp, &cmp_arg);
    if (cmp == ACM_CMP_DEFAULT && cmp_arg && *static_cast<const size_t *> (cmp_arg) > 0 && *static_cast<const size_t *> (cmp_arg) <= 4096)
      said = (uint32_t)*static_cast<const size_t *> (cmp_arg);
  }
  if (!plain && said == 0)
    return ACM_GPU_E_INELIGIBLE;
  bool classes = !plain && (said == 1 || said == 2 || said == 4);
  {
    CMP_TYPE cmp = nullptr;
    void *cmp_arg = nullptr;
    acm_internal_comparator (machine, &cmp, &cmp_arg);
    if (cmp == ACM_CMP_DEFAULT)
      classes = false; /* (memcmp over 3, 5, ... bytes: no classes to enumerate, the loop itself) */
  }
  acm_internal_plan_lock (machine);
  int rc = ACM_GPU_OK;
  if (!plain && !classes) {
    rc = acm_internal_cpu_scan (machine, text, n_symbols, said, records, capacity, n_found);
    acm_internal_set_scan_path (machine, ACM_SCAN_PATH_CPU_LOOP);
    acm_internal_plan_unlock (machine);
    return rc;
  }
  void **slot = acm_internal_plan_slot (machine);
  ACMPlan *plan = static_cast<ACMPlan *> (*slot);
  if (plan && (plan->generation != acm_internal_generation (machine) ||
               (plan->delta && plan->delta_scanned > (uint64_t)(plan->finfo.n_keywords > 1000 ? plan->finfo.n_keywords : 1000) * (14ull << 20))))
    rc = acm_gpu_plan_update (plan, machine); /* new keywords, or a delta that has cost more second passes than one plan of everything */
  if (!rc && !plan) {
    int device = 0;
    if (const char *e = getenv ("ACM_GPU_DEVICE"))
      device = atoi (e);
    /* keywords inserted while the tables are being made are picked up by the next call: the
     * generation is read first */
    const uint64_t gen = acm_internal_generation (machine);
    rc = classes ? acm_gpu_plan_create_classes (machine, said, device, &plan) : acm_gpu_plan_create (machine, device, &plan);
    if (!rc) {
      plan->generation = gen;
      *slot = plan;
    }
  }
  if (rc == ACM_GPU_E_INELIGIBLE && classes) {
    /* the comparator is no consistent order over all symbol values (acm_flatten_classes): the GPU
     * cannot take this machine by its nature -- the loop itself */
    rc = acm_internal_cpu_scan (machine, text, n_symbols, said, records, capacity, n_found);
    acm_internal_set_scan_path (machine, ACM_SCAN_PATH_CPU_LOOP);
    acm_internal_plan_unlock (machine);
    return rc;
  }
  if (!rc) {
    rc = acm_gpu_scan_host (plan, text, n_symbols, 0, 0, records, capacity, n_found);
    acm_internal_set_scan_path (machine, classes ? ACM_SCAN_PATH_GPU_CLASSES : ACM_SCAN_PATH_GPU);
  }
  acm_internal_plan_unlock (machine);
  return rc;
}

#ifdef ACM_DIAG
extern "C" int
acm_gpu_diag_read (unsigned long long *out, unsigned waves) {
  HIP_TRY (hipMemcpyFromSymbol (out, HIP_SYMBOL (g_acm_diag), sizeof (unsigned long long) * 8 * (waves < 8192 ? waves : 8192)));
  return ACM_GPU_OK;
}
#endif

/* ------------------------------------------------------------------ synthetic workload */
extern "C" int
acm_gpu_synth_text (int device, void *d_text, uint64_t n, uint64_t global_begin, uint32_t sym_bytes, uint32_t vocab,
                    const void *d_kw_data, const uint32_t *d_kw_off, uint32_t n_kw, void *stream) {
  if (!d_text || (global_begin & 4095) || (sym_bytes != 1 && sym_bytes != 4) || (sym_bytes == 4 && !vocab))
    return ACM_GPU_E_ARG;
  HIP_TRY (hipSetDevice (device));
  hipStream_t st = static_cast<hipStream_t> (stream);
  if (n == 0)
    return ACM_GPU_OK;
  dim3 g (4096), b (256);
  if (sym_bytes == 1)
    hipLaunchKernelGGL (synth_text_kernel<uint8_t>, g, b, 0, st, static_cast<uint8_t *> (d_text), n, global_begin, vocab,
                        static_cast<const uint8_t *> (d_kw_data), d_kw_off, n_kw);
  else
    hipLaunchKernelGGL (synth_text_kernel<uint32_t>, g, b, 0, st, static_cast<uint32_t *> (d_text), n, global_begin, vocab,
                        static_cast<const uint32_t *> (d_kw_data), d_kw_off, n_kw);
  HIP_TRY (hipGetLastError ());
  return ACM_GPU_OK;
}
