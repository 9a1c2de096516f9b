/* dev_all.h -- the device side of libac75_amd.so as ONE unit: the constants and structs the kernels
 * share and the dev_*.h files that hold them, in an anonymous namespace.  acm_gpu.hip includes this
 * and adds the host side; tools/kernel_regs_one.sh compiles it alone with a single kernel
 * instantiated (seconds instead of the minutes the whole library takes). */
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "acm_internal.h"

#ifdef ACM_GRAM_NT_BOTH /* experiment builds (make expd D=...): non-temporal text loads and record stores in the 4-gram kernel */
#define ACM_GRAM_NT_TEXT
#define ACM_GRAM_NT_REC
#endif

/* Diagnostic build only (-DACM_DIAG, libac75_amd_diag.so, used by tools/diag_*.py): eight per-wave
 * counters (cycle stamps, call counts) whose meaning each kernel defines where it writes them.
 * Nothing of this exists in the product build. */
#ifdef ACM_DIAG
__device__ unsigned long long g_acm_diag[8192][8];
#define DIAG(...) __VA_ARGS__
#else
#define DIAG(...)
#endif

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int WAVE = 64;
constexpr int QCAP = 128;           /* per-wave queue of (position, state) items, 8 B each */
constexpr uint32_t GRAM_Q2 = 96;    /* 4-gram kernel: per-wave queue of walk candidates */
/* 4-gram kernel: its first queue (63 items waiting + up to 64 from one position).  Two other forms
 * of the push -- the fill in a vector register with the exec mask narrowed around the LDS write
 * (9.6 % slower), four positions' compares ahead of their scalar counts (19 % slower: 217 scalar
 * and 115 vector registers spilled) -- are tools/experiments/r03_gram_push_variants.patch */
constexpr uint32_t GRAM_Q1 = 128;
constexpr uint32_t GRAM_NO_PEEK = 0xFFFFFFFFu; /* 4-gram kernel, GramK::g5peek: the state's record has to be looked at */
constexpr int DENSE_THREADS = 1024; /* one workgroup per CU, 16 waves */
/* dense kernel geometry: C = 64 bytes per lane-stream per tile, S = 2 streams per lane
 * (-DACM_DENSE_S=n builds another stream count for experiments: 3 and 4 spill and are 2x slower) */
#ifndef ACM_DENSE_S
#define ACM_DENSE_S 2
#endif
/* a region must hold the items of two 16-step blocks beside a queue's worth: region_make_room */
constexpr uint32_t DENSE_MIN_REGION_ITEMS = 2 * (16 * ACM_DENSE_S * WAVE + QCAP);
constexpr uint64_t SEGMENT = 1ull << 31; /* symbols per launch: positions inside a launch are 32-bit */
/* tests shrink it with ACM_GPU_SEGMENT_LOG2 to cross segment seams on small inputs */

/* tables of the CSR kernel, states in breadth-first numbering (ACMFlatView) */
struct CsrTables {
  const uint32_t *row_ptr, *edge_sym, *edge_next, *fail, *nb_outputs;
  uint32_t lmax;
};

/* what turns a queued (position, state) into records; states in breadth-first numbering.
 * oinfo[s] = { nb_outputs(s), next, length, keyword_id } describes the FIRST output of s (the
 * longest keyword ending there: s itself if terminal, else the nearest terminal state down its
 * failure chain) and `next` = the terminal state holding the following one: one 16-byte load per
 * record.  The second half serves the continuations of the dense kernel. */
struct EmitCtx {
  const uint4 *oinfo;
  ACMRecord *records;
  unsigned long long *count; /* running total of records: slots are reserved by atomicAdd on it */
  uint64_t capacity, pos_base;
  const unsigned char *text;  /* segment */
  const uint32_t *wrows;      /* continuation rows of every state: next | out flag << 15 | depth(next) << 16 */
  const uint16_t *cont_dh;    /* per state: depth of the nearest failure-chain state whose row is in LDS */
  uint32_t W, lo, span, n, emit_from;
  uint32_t chunk; /* bytes per lane-stream chunk of the dense kernel (a power of two) */
  uint32_t n_states;
  unsigned int *error; /* set to 1 if an item with an impossible state id is ever met (never expected) */
  /* 4-gram kernels: keyword id of every depth-4 state by its rank among them (NONE: not a keyword).
   * A hit whose state word carries HIT_LEN4 names such a state by that rank: its record is
   * (position, 4, kw4[rank]) -- a 360 KB table that stays in L2 instead of the 16-byte output
   * records of all 508,339 states (config 3: 26 of 27 M hits per GiB are keywords of 4 symbols) */
  const uint32_t *kw4;
  /* dense kernel, continuation mode: per rowless state s (index s - chain_base) what lies below
   * it when that is ONE path to a leaf: { r | depth (hotfail (s)) << 4, leaf state t, the r <= 8
   * symbols of the path }; r = 0: no such record (walk_continuation goes step by step), r = 15:
   * s is a leaf.  NULL: none. */
  const uint4 *chain;
  uint32_t chain_base;
  /* 4-gram kernel, narrow alphabets: records are written by the scan kernel itself into chunks of
   * the caller's buffer (dev_starts.h: WaveRec); slots past `capacity` go to the plan's spill
   * area, from which close_holes_kernel brings them back into the holes below the dense count */
  uint4 *spill;
  uint64_t spill_slots;
  /* tiled scans (dev_tiles.h): `records` is a raw area of whole chunks, and a wave that reserves
   * chunk c writes chunk_prev[c] = the chunk it filled before (NONE: its first) */
  uint32_t *chunk_prev;
  /* slots per chunk of records a wave reserves with one atomic on the record counter: REC_CHUNK in
   * tiled scans (their directory counts in chunks of that size), REC_CHUNK_BIG in other scans of
   * long texts -- every wave of the chip adds to the same counter, 4.8 ns a time: a dictionary of
   * short keywords (164 M records from 2 GiB) spent 0.78 of 1.98 ms on 161 K reservations */
  uint32_t rec_chunk;
};
constexpr uint32_t HIT_LEN4 = 0x80000000u;
/* the hit's word is the keyword itself: id (below 2^28) | length << 28 (1-3; 0: 4 symbols) | HIT_KW -- what
 * the 4-gram kernel's tables hold for the keywords of up to 4 symbols, so that their records need
 * no lookup (one in 13 positions of a text can end such a keyword) */
constexpr uint32_t HIT_KW = 0x40000000u, HIT_KW_ID = 0x0FFFFFFFu;

/* one launch: a segment of the buffer, positions relative to its first symbol */
struct Launch {
  const unsigned char *text; /* first symbol of the segment */
  uint32_t n;                /* symbols in the segment */
  uint32_t emit_from;        /* matches ending before this index are not reported */
  uint32_t range_begin, range_end; /* dense: tile indices; csr: symbol indices */
  /* dense: tiles [range_begin, static_end) are split evenly between the blocks; [static_end,
   * range_end) is a pool in POOL_CLASSES equal parts, handed out tile by tile through pool_ctr */
  uint32_t static_end, pool_class_tiles;
  uint32_t pool_classes; /* min (POOL_CLASSES, blocks): block b draws from part b * pool_classes / blocks */
  unsigned int *pool_ctr, *pool_reset; /* this launch's counters; the previous launch's, to zero */
};
static constexpr uint32_t POOL_CLASSES = 16, POOL_CTR_STRIDE = 64; /* counters 256 B apart */

/* small uniform constants of the dense kernel */
struct DenseK {
  uint32_t W, rowbytes, lo, span;
  uint32_t HD; /* states [0, HD): failure-resolved row in LDS */
  uint32_t aux_off, queue_off, wub, lmax;
  uint32_t stream_stride; /* 64 * C: distance between the chunks of a lane's consecutive streams */
};

/* queue item, second word, when the dense kernel runs in continuation mode (16-bit states) */
constexpr uint32_t IT_STATE = 0x7FFFu;
constexpr uint32_t IT_CONT = 1u << 15;  /* walk on from this (rowless) state: see walk_continuation */
constexpr uint32_t IT_K_SHIFT = 16;     /* 12 bits: run-over step k (symbols past the end of the lane's chunk) */
constexpr uint32_t IT_RUN = 1u << 28;   /* queued during a chunk's run-over */
constexpr uint32_t IT_OUT = 1u << 29;   /* report the outputs of the state itself at pos */

#include "dev_emit.h"
#include "dev_dense.h"
#include "dev_csr.h"
#include "dev_sparse.h"
#include "dev_starts.h"
#include "dev_gram.h"
#include "dev_gram2.h"
#include "dev_short.h"
#include "dev_misc.h"
#include "dev_order.h"
#include "dev_tiles.h"

} // namespace
