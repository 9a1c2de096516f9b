/*
 * acm_gpu.h -- bulk-scan extension of the acm_* API: the MI355X (gfx950) hot path.
 *
 * The reference has no buffer-level entry point: its hot path is the CALLER's loop
 *
 *     for (i = 0; i < n; i++)                                   examples/test.c:17-23
 *       for (j = 0, nb = acm_match (&cursor, &text[i]); j < nb; j++)   aho_corasick.c:434-448
 *         acm_get_match (cursor, j, &holder);                   aho_corasick.c:451-482
 *
 * Every scan function below is DEFINED as that loop started from acm_initiate(machine): it
 * yields one ACMRecord per (i, j) with end_pos = i, length = holder.length and keyword_id = the
 * 0-based rank of the keyword in order of first acm_insert_end_of_keyword (the value
 * machine->nb_sequences had just before aho_corasick.c:352), in the loop's order
 * (end_pos ascending, then j ascending == length descending).
 *
 * The acm_gpu_* functions run on the GPU as hand-written HIP and have NO CPU fallback: a machine
 * they cannot take (symbol size not in {1,2,4,8}; a comparator other than ACM_CMP_DEFAULT unless
 * the plan is made with acm_gpu_plan_create_classes) or a missing device is reported as an error
 * code.  acm_scan, the call on the machine itself, is total over machines: what the
 * GPU cannot take by its nature runs the loop above on the host and says so (acm_scan_path); a
 * missing device is still an error there, never a silent fallback.
 *
 * Plain C ABI: pointers and sizes only.  `stream` arguments are a hipStream_t passed as void *
 * (NULL = the default stream); `d_` pointers are device memory on the plan's device.
 */
#ifndef ACM_GPU_H_AMD
#define ACM_GPU_H_AMD

#include "acm.h"
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Canonical match record, 16 bytes. */
typedef struct {
  uint64_t end_pos;    /* index of the symbol at which the keyword ends */
  uint32_t length;     /* MatchHolder.length of that match */
  uint32_t keyword_id; /* first-insertion rank of the keyword */
} ACMRecord;

enum {
  ACM_GPU_OK = 0,
  ACM_GPU_E_INELIGIBLE = -1, /* comparator is not ACM_CMP_DEFAULT, or symbol size not 1/2/4/8 bytes */
  ACM_GPU_E_NODEVICE = -2,   /* no usable HIP device */
  ACM_GPU_E_HIP = -3,        /* a HIP call failed (message on stderr) */
  ACM_GPU_E_OVERFLOW = -4,   /* more matches than `capacity`; the count output holds the number needed */
  ACM_GPU_E_ARG = -5,        /* invalid argument */
  ACM_GPU_E_NOMEM = -6,
  ACM_GPU_E_INTERNAL = -7,   /* a device-side consistency check failed (never expected) */
  ACM_GPU_E_FORMAT = -8,     /* not a flat-table blob, wrong version, or its contents do not hold together */
  ACM_GPU_E_IO = -9,         /* a file could not be read or written */
  ACM_GPU_E_COMM = -10       /* librccl.so could not be loaded, or an RCCL call failed (message on stderr) */
};
const char *acm_gpu_strerror (int code);
int acm_gpu_device_count (void);

/* Fills `matcher` (initialised with acm_matcher_init, released with acm_matcher_release) with what
 * acm_get_match would have produced for a record of the bulk scan: letters[] = the dictionary's
 * spelling of keyword `keyword_id` (reference aho_corasick.c:472-479), its length and its value
 * (:480).  Host side, no GPU involved. */
int acm_get_keyword (const ACMachine *machine, uint32_t keyword_id, MatchHolder *matcher);

/* ------------------------------------------------------------------ flattened tables (host)
 * Snapshot of the machine's goto / failure / output functions as flat arrays (what
 * struct _ac_state holds per node in the reference: transitions :47, fail_state :53,
 * is_end_of_keyword :54, nb_outputs :55, previous :49-52).  States are renumbered breadth-first
 * (root = 0, children in memcmp order), so shallow -- hot -- states get the small ids and
 * depth(s) is monotone in s.  Needs no GPU. */
typedef struct ACMFlat ACMFlat;

typedef struct {
  uint32_t sym_bytes;   /* 1, 2, 4 or 8 */
  uint32_t n_states;
  uint32_t n_keywords;
  uint32_t n_edges;     /* = n_states - 1 */
  uint32_t lmax;        /* longest keyword, in symbols */
  uint32_t max_outputs; /* max over states of nb_outputs */
  uint32_t alpha_lo;    /* byte alphabets: smallest symbol used by any keyword */
  uint32_t alpha_span;  /* byte alphabets: largest - smallest + 1 (0 for an empty machine) */
  uint32_t width;       /* byte alphabets: dense row width = span + 1 ("other" class last), or 256 */
} ACMFlatInfo;

typedef struct {
  const uint32_t *row_ptr;     /* [n_states + 1] CSR of the goto function */
  const uint32_t *edge_sym;    /* [n_edges] symbol value (little-endian read of sym_bytes bytes), ascending memcmp order per row */
  const uint32_t *edge_next;   /* [n_edges] g(state, symbol) */
  const uint32_t *fail;        /* [n_states] f(state); f(root) = 0 */
  const uint32_t *depth;       /* [n_states] */
  const uint32_t *nb_outputs;  /* [n_states] what acm_match returns in that state */
  const uint32_t *term_kw;     /* [n_states] keyword_id if a keyword ends here, else 0xFFFFFFFF */
  const uint32_t *out_link;    /* [n_states] nearest keyword-terminal state strictly down the failure chain, 0 if none */
  const uint32_t *depth_start; /* [lmax + 2] first state id of each depth; depth_start[lmax + 1] = n_states */
  const uint32_t *kw_state;    /* [n_keywords] terminal state of each keyword */
  /* comparator-class machines only (acm_flatten_classes), else NULL / 0 */
  const uint16_t *class_map;   /* [class_entries] symbol value -> class id: what edge_sym holds, and what the text is mapped through */
  const uint32_t *edge_letter; /* [n_edges] the dictionary's own symbol on each edge (what MatchHolder.letters[] point at) */
  uint32_t class_entries;      /* 256 or 65536 */
  uint32_t n_classes;
  /* 8-byte symbols only, else NULL / 0: the distinct symbols of the dictionary, ascending; edge_sym
   * holds 1 + the index into this table (0 stands for every other symbol a text may hold) */
  const uint64_t *keys64;
  uint32_t n_keys64;
  /* 4-byte symbols flattened over comparator classes only, else NULL / 0: the dictionary's distinct
   * symbols (ascending by value) and their classes 1 .. n_classes (what edge_sym holds; 0 stands
   * for a symbol that compares equal to none of them), and one symbol of each class in comparator
   * order (a text symbol is classified against these when a scan first meets it) */
  const uint32_t *keys32, *keys32_class;
  uint32_t n_keys32;
  const uint32_t *class_rep32;
} ACMFlatView;

int acm_flatten (ACMachine *machine, ACMFlat **out);
/* Machines created with another comparator than ACM_CMP_DEFAULT (aho_corasick.h:33,45; e.g. the
 * case-insensitive alphacmp of generic_test.c:48-54) over symbols of sym_bytes = 1 or 2 bytes: the
 * comparator is called on all 256 / 65,536 symbol values to find the classes of symbols it cannot
 * tell apart; the tables are built over class ids and a plan made from them maps the text through
 * the class table on the device before walking it.  ACM_GPU_E_INELIGIBLE if the comparator is not
 * a consistent order over all values.  (The symbol size is an argument because a custom
 * comparator's cmp_arg is opaque.)
 * sym_bytes = 4 (the reference's own example: wchar_t with the case-insensitive alphacmp,
 * generic_test.c:48-54,62-164): 2^32 values cannot be enumerated, so the classes are those of the
 * dictionary's own symbols; a plan made from such tables classifies the symbols of a text when it
 * first meets them, with the machine's comparator, on the host (ACMFlatView::keys32).  Such tables
 * have no serialised form (acm_flat_to_blob: ACM_GPU_E_ARG). */
int acm_flatten_classes (ACMachine *machine, uint32_t sym_bytes, ACMFlat **out);
void acm_flat_release (ACMFlat *flat);
void acm_flat_info (const ACMFlat *flat, ACMFlatInfo *info);
void acm_flat_view (const ACMFlat *flat, ACMFlatView *view);
/* Failure-resolved (DFA) rows of states [0, n_rows) of a byte-alphabet machine:
 * out[s * width + cls] = delta(s, alpha_lo + cls) | (nb_outputs(next) ? top bit : 0), class
 * `span` (when width == span + 1) standing for every symbol outside [lo, lo + span).
 * entry_bytes is 2 (n_states <= 32768) or 4. */
int acm_flat_dense_rows (const ACMFlat *flat, uint32_t n_rows, uint32_t entry_bytes, void *out);

/* Serialised form of the flat tables (versioned, little-endian; layout in acm_flat.c).  The
 * reference keeps a machine in memory only and rebuilds it from its keywords at every start; a
 * blob restores the scan tables without the keywords or the trie.  Loading recomputes the failure
 * function and every derived array from the goto function and compares: a blob that loads is what
 * acm_flatten would have produced.  Values (`void *` of acm_insert_end_of_keyword) are process-
 * local and not part of a blob; keyword ids and spellings are (acm_flat_keyword). */
size_t acm_flat_blob_bytes (const ACMFlat *flat);
int acm_flat_to_blob (const ACMFlat *flat, void *out, size_t capacity);
int acm_flat_from_blob (const void *blob, size_t bytes, ACMFlat **out);
int acm_flat_save (const ACMFlat *flat, const char *path);
int acm_flat_load (const char *path, ACMFlat **out);
/* Spelling of keyword `keyword_id` from the tables alone -- what MatchHolder.letters[] spell
 * (aho_corasick.c:472-479): min(length, capacity) symbols of sym_bytes bytes each, front to back. */
int acm_flat_keyword (const ACMFlat *flat, uint32_t keyword_id, void *symbols, uint32_t capacity, uint32_t *length);

/* ------------------------------------------------------------------ device plan */
typedef struct ACMPlan ACMPlan;

typedef struct {
  int device;
  uint32_t kernel;       /* 1 = dense-row byte kernel (automaton in LDS); 2 = CSR walk (any symbol size, any
                            alignment); 2- and 4-byte symbols in 16-byte aligned buffers: 4 = start-parallel kernel
                            (root table by symbol value, every position verified independently), or
                            3 = sparse automaton walk when the environment says ACM_GPU_SPARSE=walk;
                            5 = 4-gram sieve kernel: byte dictionaries whose hot rows outgrow LDS (more than about
                            1,300 keywords over a-z) with some keyword of 4 symbols or more
                            (ACM_GPU_GRAM=0: kernel 1 instead, ACM_GPU_GRAM=2: kernel 5 whenever possible) */
  uint32_t entry_bytes;  /* dense entries: 2 or 4 */
  uint32_t width;        /* dense row width */
  uint32_t dense_rows;   /* rows resident in HBM */
  uint32_t lds_rows;     /* states whose failure-resolved row is staged in LDS by every workgroup */
  uint32_t lds_hotfail;  /* further states for which LDS holds the nearest failure-chain state that has a row (2 B each) */
  uint32_t lds_bytes;    /* dynamic LDS per workgroup */
  uint32_t block_threads;
  uint32_t grid_blocks;
  uint32_t chunk_bytes;  /* text bytes per lane-stream per tile */
  uint32_t streams;      /* independent streams per lane */
  uint64_t table_bytes;  /* device bytes held by the plan */
  uint32_t delta_keywords; /* keywords added since the tables were made, held by the delta plan (acm_gpu_plan_update) */
  uint32_t merges;         /* updates that rebuilt the tables (the delta had outgrown its share) */
  uint32_t records_direct; /* 1: the scan kernel writes the 16-byte records itself (kernel 5 on alphabets of at most 29
                              symbols; kernel 2); 0: it parks 8-byte items / hits that a second kernel turns into records */
  uint32_t variant;        /* kernel 5: 2 = scan_gram2_kernel (lane-local sieve on two bits per 4-gram: alphabets of up to
                              26 symbols + "other"; ACM_GPU_GRAM2=0 selects the older form); | 4: the keywords of 1-3 symbols
                              have a pass of their own behind it (scan_short_kernel); | 8: that pass reads the keyword ids
                              from HBM (more of them than LDS holds: about 29 K) */
} ACMPlanInfo;

/* Flattens `machine` and uploads the tables to `device`.  The plan is a snapshot: keywords
 * inserted later are not seen by it until acm_gpu_plan_update.
 * A plan owns scratch buffers that its scans share (item regions, the running record count, the
 * class-mapped / aligned copy of the text): scans of ONE plan must be issued from one host thread
 * and on one stream at a time (consecutive scans on the same stream queue up as usual; for
 * concurrent streams or threads make one plan each -- the tables are a few MB). */
int acm_gpu_plan_create (ACMachine *machine, int device, ACMPlan **out);
int acm_gpu_plan_create_flat (const ACMFlat *flat, int device, ACMPlan **out);
/* acm_flatten_classes + acm_gpu_plan_create_flat: plan of a machine with a custom comparator over
 * 1- or 2-byte symbols.  Scans of such a plan first map the text to class ids (one more pass over
 * the text into a buffer the plan owns), then run the same kernels. */
int acm_gpu_plan_create_classes (ACMachine *machine, uint32_t sym_bytes, int device, ACMPlan **out);
/* Brings a plan up to date with the machine it was made from after keywords were added to it
 * (the reference inserts while it scans: README.md:352-356, generic_test.c:214-229), without
 * rebuilding its tables and without waiting for the device:
 *   - plans of the start-parallel kernel (2- and 4-byte symbols) are edited in place: only the
 *     words of the new keywords' own states change, and they are written on the stream of the next
 *     scan, in front of it;
 *   - every other plan keeps its tables; the keywords added since it was made are flattened into a
 *     small delta plan of their own (cost independent of the size of the dictionary), scanned
 *     right after the plan into the same record buffer -- the match set of a dictionary is the
 *     union of its keywords' match sets.  Once the delta holds more than an eighth of the
 *     dictionary (at least 256 keywords) the next update builds one plan of everything again
 *     (that one waits for the device).
 * Not while a stream (acm_gpu_stream_*) is open on the plan.  acm_scan() calls this by itself. */
int acm_gpu_plan_update (ACMPlan *plan, ACMachine *machine);
void acm_gpu_plan_destroy (ACMPlan *plan);
void acm_gpu_plan_info (const ACMPlan *plan, ACMPlanInfo *info);

/* Scan d_text[0 .. n_symbols) from the root state.  Matches whose end index is < emit_from are
 * not reported (warm-up region of a shard: pass the lmax - 1 symbols preceding the shard and
 * emit_from = their number).
 *
 *   end_pos of a match ending at buffer index i  =  pos_base + i
 *
 * Records are appended to d_records in no particular order (at most `capacity`); *d_count
 * (device, 8 bytes) receives the TOTAL number of matches, which may exceed capacity -- nothing is
 * dropped silently: compare and re-run with a larger buffer.  Asynchronous on `stream`. */
int acm_gpu_scan_device (ACMPlan *plan, const void *d_text, uint64_t n_symbols, uint64_t emit_from,
                         uint64_t pos_base, ACMRecord *d_records, uint64_t capacity,
                         uint64_t *d_count, void *stream);

/* Sum of acm_match return values over the buffer (reference: generic_test.c:272-273), no records. */
int acm_gpu_count_device (ACMPlan *plan, const void *d_text, uint64_t n_symbols, uint64_t emit_from,
                          uint64_t *d_count, void *stream);

/* Puts n records into canonical order (end_pos ascending, length descending) in place.
 * d_tmp must hold acm_gpu_sort_tmp_bytes(n) bytes.  Asynchronous on `stream`. */
size_t acm_gpu_sort_tmp_bytes (uint64_t n);
int acm_gpu_sort_records_device (ACMPlan *plan, ACMRecord *d_records, uint64_t n, void *d_tmp,
                                 size_t tmp_bytes, void *stream);

/* The same order for records whose end_pos all lie in [pos_lo, pos_lo + span) -- what a scan of
 * `span` symbols with pos_base = pos_lo (and any emit_from) leaves: three passes over the records
 * (position buckets, then sorts of a few thousand records in LDS) instead of the radix sort's
 * eight.  A record outside the range makes acm_gpu_plan_status report ACM_GPU_E_INTERNAL.
 * d_tmp must hold acm_gpu_order_tmp_bytes(plan, n, span) bytes.  Asynchronous on `stream`. */
size_t acm_gpu_order_tmp_bytes (const ACMPlan *plan, uint64_t n, uint64_t span);
int acm_gpu_order_records_device (ACMPlan *plan, ACMRecord *d_records, uint64_t n, uint64_t pos_lo, uint64_t span,
                                  void *d_tmp, size_t tmp_bytes, void *stream);

/* Records on the wire.  The records of a scan of `span` symbols with pos_base = pos_lo hold
 * positions in [pos_lo, pos_lo + span), lengths up to the plan's lmax and ids below its number of
 * keywords: when that fits 64 bits (acm_gpu_wire_bits returns 0 and the three field widths; config
 * 4's shards: 34 + 4 + 17) a record packs to ONE 8-byte word,
 *     (end_pos - pos_lo) | length << pos_bits | keyword_id << (pos_bits + len_bits),
 * half of what a shard sends to the root in a multi-GPU scan (acm_gpu_multi_* and sharded.py pack
 * behind the canonical order and unpack on the root; the order is kept: index i stays index i).
 * Asynchronous on `stream`; d_packed / d_records on the current device of the call. */
int acm_gpu_wire_bits (const ACMPlan *plan, uint64_t span, uint32_t *pos_bits, uint32_t *len_bits, uint32_t *kw_bits);
int acm_gpu_pack_records_device (const ACMRecord *d_records, uint64_t n, uint64_t pos_lo, uint32_t pos_bits, uint32_t len_bits,
                                 uint64_t *d_packed, void *stream);
int acm_gpu_unpack_records_device (const uint64_t *d_packed, uint64_t n, uint64_t pos_lo, uint32_t pos_bits, uint32_t len_bits,
                                   ACMRecord *d_records, void *stream);

/* acm_gpu_scan_device and the canonical order of what it found in ONE call that only queues work on
 * `stream`: the order passes read the number of records from *d_count on the device, so no host
 * round trip separates the scan from them (the caller loop of aho_corasick.h:47,77 yields its
 * matches in this order; this is that loop's output, complete, with one synchronisation at the
 * end).  d_records[0 .. *d_count) is in canonical order afterwards when *d_count <= capacity; a
 * scan that overflowed leaves the total in *d_count and nothing in order (repeat it with room).
 * Plans of big byte dictionaries (the 4-gram kernel, narrow alphabets, keywords of up to 1,024
 * symbols, no pending delta) scan in tiles into d_tmp and put the records in order in ONE pass
 * over them; every other plan scans as acm_gpu_scan_device does and runs
 * acm_gpu_order_records_device's passes behind it.  Same records, same order either way.
 * d_tmp must hold acm_gpu_scan_ordered_tmp_bytes(plan, capacity, n_symbols) bytes (about
 * 16 bytes per record of capacity, plus a few megabytes). */
size_t acm_gpu_scan_ordered_tmp_bytes (const ACMPlan *plan, uint64_t capacity, uint64_t n_symbols);
int acm_gpu_scan_ordered_device (ACMPlan *plan, const void *d_text, uint64_t n_symbols, uint64_t emit_from,
                                 uint64_t pos_base, ACMRecord *d_records, uint64_t capacity, uint64_t *d_count,
                                 void *d_tmp, size_t tmp_bytes, void *stream);

/* Host-buffer convenience: upload, scan, sort, download; blocking.  On ACM_GPU_E_OVERFLOW
 * *n_found holds the capacity needed. */
int acm_gpu_scan_host (ACMPlan *plan, const void *text, uint64_t n_symbols, uint64_t emit_from,
                       uint64_t pos_base, ACMRecord *records, uint64_t capacity, uint64_t *n_found);

/* The bulk call on the machine itself, for EVERY machine the reference's API can make
 * (aho_corasick.h:33-45: any comparator, any symbol): the records of the caller's loop over
 * text[0 .. n_symbols), in the loop's order.  Keeps a plan cached inside the machine and brings it
 * up to date when the dictionary changed since the last call.  Device = $ACM_GPU_DEVICE or 0.
 *   - ACM_CMP_DEFAULT over 1, 2, 4 or 8 byte symbols: the GPU scan (ACM_SCAN_PATH_GPU);
 *   - another comparator: the library cannot know the symbol size (letters are opaque pointers), so
 *     the caller says it once with acm_set_symbol_bytes.  1, 2 or 4 bytes: the GPU scan over the
 *     comparator's symbol classes (acm_gpu_plan_create_classes; ACM_SCAN_PATH_GPU_CLASSES) -- the
 *     reference's own example, wchar_t + alphacmp (examples/aho_corasick_generic_test.c:48-54), runs
 *     this way.  Any other size, or a comparator that is no consistent order over all symbol values
 *     (acm_flatten_classes refuses it): the loop itself, on the host, with this library's own
 *     acm_match / acm_get_match steps (ACM_SCAN_PATH_CPU_LOOP) -- SURVEY.md 8(b): "otherwise it runs
 *     loop a9 on the CPU".  Without acm_set_symbol_bytes such a machine is ACM_GPU_E_INELIGIBLE.
 * A missing device or a failing HIP call is an ERROR for the machines of the first two kinds
 * (ACM_GPU_E_NODEVICE, ACM_GPU_E_HIP): the GPU path never falls back to the host silently.
 * acm_scan_path says which of the three the machine's last acm_scan ran. */
#define ACM_SCAN_PATH_NONE 0
#define ACM_SCAN_PATH_GPU 1
#define ACM_SCAN_PATH_GPU_CLASSES 2
#define ACM_SCAN_PATH_CPU_LOOP 3
int acm_scan (ACMachine *machine, const void *text, uint64_t n_symbols, ACMRecord *records,
              uint64_t capacity, uint64_t *n_found);
int acm_set_symbol_bytes (ACMachine *machine, uint32_t sym_bytes);
int acm_scan_path (const ACMachine *machine);

/* ------------------------------------------------------------------ streaming scan
 * Text that arrives piece by piece from the host (the reference's callers read files symbol by
 * symbol, generic_test.c:191).  The result is the caller loop's output over the concatenation of
 * all pieces fed so far (end_pos = position in the whole stream): the last lmax - 1 symbols are
 * carried over in front of every piece.  Host-to-device copies of a piece overlap with the scan of
 * the previous one (two device slots, two streams).  One open stream per plan at a time; do not
 * mix with acm_gpu_scan_* calls on the same plan while it is open. */
typedef struct ACMStream ACMStream;
int acm_gpu_stream_open (ACMPlan *plan, uint64_t max_piece_symbols, uint64_t record_capacity, ACMStream **out);
/* Enqueues copy + scan of the next n_symbols (split into pieces of at most max_piece_symbols) and
 * returns; `text` (host memory, pinned for a truly asynchronous copy) must stay untouched until
 * the second next feed or acm_gpu_stream_finish. */
int acm_gpu_stream_feed (ACMStream *stream, const void *text, uint64_t n_symbols);
/* Waits, sorts into canonical order, copies the records of the whole stream so far to the host.
 * ACM_GPU_E_OVERFLOW (with *n_found = number needed) if they exceed the stream's or this call's
 * capacity.  The stream stays open. */
int acm_gpu_stream_finish (ACMStream *stream, ACMRecord *records, uint64_t capacity, uint64_t *n_found);
void acm_gpu_stream_close (ACMStream *stream);

/* ------------------------------------------------------------------ several GPUs of one node, one process
 * The reference's model for parallel work is one shared read-only machine and one cursor per worker
 * (/root/reference/README.md:364, aho_corasick.h:70: the caller owns the `const ACState *`).  The
 * cursor after any symbol depends on the last lmax symbols only, so a text splits into R contiguous
 * shards that are scanned independently: shard r owns [r N / R, (r + 1) N / R),
 * starts from the root lmax - 1 symbols earlier (rounded down to a 16-byte boundary of the text)
 * and reports the matches that END inside its range.  An ACMMulti holds one plan per distinct
 * device (tables replicated) and one stream per device; shard r runs on devices[r] -- a device may
 * appear several times (shards of one device run one after the other on its stream), so that
 * `devices = {0,0,0,0,0,0,0,0}` exercises on one GPU everything but the peer copies.
 * The one exchange step: every shard's records are put in canonical order where they were found
 * and copied into their place in one buffer on devices[0] -- hipMemcpyPeerAsync from the other
 * devices (xGMI is point-to-point: these are the direct peer-to-root transfers of SURVEY.md 8e, one
 * per link), a device-to-device copy for shards of devices[0] itself.  Shards own increasing
 * position ranges: their concatenation in shard order IS the canonical order of the whole text. */
typedef struct ACMMulti ACMMulti;
int acm_gpu_multi_create (ACMachine *machine, const int *devices, int n_shards, ACMMulti **out);
void acm_gpu_multi_destroy (ACMMulti *multi);
/* [read_begin, own_end) is what shard `shard` of a text of n_symbols reads, [own_begin, own_end) where its matches end */
int acm_gpu_multi_shard_bounds (const ACMMulti *multi, uint64_t n_symbols, int shard, uint64_t *read_begin,
                                uint64_t *own_begin, uint64_t *own_end);
/* Text in host memory: the shards are uploaded to their devices, scanned, ordered, gathered on
 * devices[0] and copied to `records` (canonical order of the whole text).  Blocking.
 * ACM_GPU_E_OVERFLOW with *n_found = the capacity needed when `capacity` is too small. */
int acm_gpu_multi_scan_host (ACMMulti *multi, const void *text, uint64_t n_symbols, ACMRecord *records,
                             uint64_t capacity, uint64_t *n_found);
/* Shards already resident: d_shard_text[r] is a buffer on devices[r] (16-byte aligned) holding the
 * symbols [read_begin_r, own_end_r) of acm_gpu_multi_shard_bounds.  The records of the whole text
 * arrive in canonical order in d_records, a buffer of `capacity` records on devices[0]; *n_found
 * (host) = their number, which may exceed capacity (ACM_GPU_E_OVERFLOW: nothing dropped silently).
 * Blocking. */
int acm_gpu_multi_scan_device (ACMMulti *multi, const void *const *d_shard_text, uint64_t n_symbols,
                               ACMRecord *d_records, uint64_t capacity, uint64_t *n_found);

/* ------------------------------------------------------------------ one process per GPU: the gather over RCCL
 * The reference's model for parallel work is one shared read-only machine and one cursor per worker
 * (README.md:364, aho_corasick.h:70); across PROCESSES that is one process per GPU, each with its
 * own plan of the same dictionary, its shard of the text (acm_gpu_multi_shard_bounds' rule:
 * contiguous ranges, an lmax - 1 halo) and its records from acm_gpu_scan_ordered_device.  The one
 * exchange step -- BASELINE's "final RCCL gather of match records over xGMI" -- is this call: the
 * counts travel by ncclAllGather, every rank's ordered records by ncclSend to `root`, which
 * receives them (ncclRecv, one group) into their places in ONE buffer: shards own increasing
 * position ranges, so rank order IS the canonical order.  Records travel as 8-byte words when the
 * rank's plan says they fit (acm_gpu_wire_bits; `plan` may be NULL: 16 bytes) and are unpacked on
 * the root.  xGMI is point-to-point: these are the direct peer-to-root transfers of SURVEY.md 8e.
 *
 * librccl.so is loaded when the first of these calls is made (dlopen; $ACM_GPU_COMM_LIB names
 * another library with the same entry points -- the tests' loopback transport), so a single-GPU
 * user of libac75_amd.so never loads it.  `nccl_comm` is an ncclComm_t (rccl.h), passed as void *:
 * the caller's own, or one made here -- acm_gpu_comm_unique_id on one rank, its 128 bytes handed to
 * the others by whatever the processes share (a file, a socket, MPI), acm_gpu_comm_init_rank on
 * every rank with its device current.
 * acm_gpu_comm_gather_records: collective over the communicator, every rank calls it with its own
 * records (d_local[0 .. n_local), positions in [pos_lo, pos_lo + span)); on the root d_all takes
 * `capacity` records.  *n_total (host, every rank) = the records of all ranks; more than the
 * root's capacity: ACM_GPU_E_OVERFLOW on every rank, nothing sent.  The counts are exchanged
 * before the call returns (one stream synchronisation); the transfers and the unpacking are
 * queued on `stream`.  `counts` (host, `world` entries, may be NULL) = records per rank. */
typedef struct ACMComm ACMComm;
int acm_gpu_comm_unique_id (void *id_128_bytes);
int acm_gpu_comm_init_rank (const void *id_128_bytes, int rank, int world, void **nccl_comm);
int acm_gpu_comm_free (void *nccl_comm);
int acm_gpu_comm_create (void *nccl_comm, int rank, int world, int root, ACMComm **out);
void acm_gpu_comm_destroy (ACMComm *comm);
int acm_gpu_comm_gather_records (ACMComm *comm, const ACMPlan *plan, const ACMRecord *d_local, uint64_t n_local,
                                 uint64_t pos_lo, uint64_t span, ACMRecord *d_all, uint64_t capacity,
                                 uint64_t *n_total, uint64_t *counts, void *stream);

/* Waits for the plan's device and reports ACM_GPU_E_INTERNAL if a device-side consistency check
 * ever failed during its scans (never expected), else ACM_GPU_OK. */
int acm_gpu_plan_status (ACMPlan *plan);

/* Kernel timing with HIP events recorded on the launch stream around the scan kernel only.
 * Enable, run scans, then read: total milliseconds and number of launches since enabling.
 * Reading synchronises on the recorded events.  enable = 1: every launch; enable = N > 1: every
 * N-th launch -- the three events of a launch cost about 10 us on the stream (a step of 1 GiB
 * against 1,000 keywords takes 282 us without them and 292 with: tools/exp_timing_overhead.py), so
 * a caller that measures throughput and kernel time in the same run samples. */
int acm_gpu_plan_timing (ACMPlan *plan, int enable);
int acm_gpu_plan_timing_read (ACMPlan *plan, double *total_ms, uint64_t *launches);
/* the same, and beside the scan kernels' time (scan_ms) the time from the start of each scan kernel
 * to the end of what its launch enqueues behind it -- the expansion of parked items / hits into
 * records, or the closing of the holes in a record buffer the scan kernel wrote itself (all_ms) */
int acm_gpu_plan_timing_read_all (ACMPlan *plan, double *scan_ms, double *all_ms, uint64_t *launches);

/* ------------------------------------------------------------------ synthetic workload (bench/test tooling)
 * text[i] = 'a' + sm(i + 42) % 26, one keyword planted per 4096-symbol block.
 * Generates d_text[0 .. n) for global indices [global_begin, global_begin + n); global_begin must
 * be a multiple of 4096.  kw_data/kw_off: the K keywords packed on the DEVICE (symbols of
 * sym_bytes each; kw_off has K + 1 entries).  For sym_bytes == 4 the unplanted symbol is
 * sm(i + 42) % vocab. */
int acm_gpu_synth_text (int device, void *d_text, uint64_t n, uint64_t global_begin, uint32_t sym_bytes,
                        uint32_t vocab, const void *d_kw_data, const uint32_t *d_kw_off, uint32_t n_kw,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif
