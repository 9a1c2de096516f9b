/*
 * acm_gpu.hip -- device side of the bulk scan (include/acm_gpu.h) for MI355X / gfx950.
 *
 * What runs here is the reference's caller loop (examples/test.c:17-23): per input symbol one
 * automaton step (acm_match -> state_goto, aho_corasick.c:434-448,167-192) and, when the new
 * state has outputs, the walk over keyword-terminal states of its failure chain
 * (acm_get_match, aho_corasick.c:459-466), emitting one 16-byte record per match.
 *
 * Kernels
 *   scan_dense_kernel  byte alphabets.  Failure-resolved rows (one lookup per symbol), the
 *                      shallowest rows staged in LDS by every workgroup, colder rows read from
 *                      the HBM/L2-resident copy.  A wave owns tiles of 64*S*C contiguous bytes;
 *                      each lane walks S independent chunks of C bytes held in registers
 *                      (dwordx4 loads), restarting from the root WU >= lmax-1 bytes before its
 *                      chunk and reporting only matches that end inside the chunk.
 *   scan_csr_kernel    any symbol width (1/2/4 bytes): goto/failure walk over the CSR arrays; also
 *                      covers the head and tail of the text that do not fill whole tiles.
 *   expand/flush       states-with-outputs are queued per wave in LDS as (position, state) and
 *                      expanded to records with one global atomic per <= 64 queue entries
 *                      (wave prefix sum), never one atomic per match.
 *   sort               canonical order (end_pos asc, length desc) by a 64-bit radix sort
 *                      (hipCUB, a library op outside the timed scan).
 * No MFMA anywhere: this is byte/integer pointer chasing bound by LDS lookups and HBM reads.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
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

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int WAVE = 64;
constexpr int QCAP = 128;          /* per-wave queue of (pos, state) items, 16 B each */
constexpr int DENSE_THREADS = 1024; /* one workgroup per CU, 16 waves */
constexpr int DENSE_C = 64;         /* bytes per lane-stream per tile */
constexpr int DENSE_S = 2;          /* independent streams per lane */

struct DevTables {
  const uint32_t *row_ptr, *edge_sym, *edge_next, *fail, *depth, *nb_outputs, *term_kw, *out_link;
  const void *dense; /* [dense_rows * width] entries of entry_bytes */
  uint32_t n_states, width, lo, span, dense_rows, lds_rows, lmax, entry_bytes;
};

struct ScanArgs {
  const unsigned char *text;
  uint64_t n;          /* symbols in the buffer */
  uint64_t emit_from;  /* matches ending before this buffer index are not reported */
  uint64_t pos_base;   /* reported end_pos = pos_base + buffer index */
  ACMRecord *records;
  uint64_t capacity;
  unsigned long long *count;
  /* range of this launch */
  uint64_t range_begin, range_end; /* dense: tile indices; csr: symbol indices */
};

/* ------------------------------------------------------------------ wave helpers */
__device__ __forceinline__ uint32_t
lane_id () {
  return __builtin_amdgcn_mbcnt_hi (~0u, __builtin_amdgcn_mbcnt_lo (~0u, 0u));
}

__device__ __forceinline__ uint32_t
wave_incl_scan (uint32_t v) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    uint32_t o = __shfl_up (v, d, WAVE);
    if ((int)lane_id () >= d)
      v += o;
  }
  return v;
}

/* what the (out-of-line) queue expansion needs, passed by value in registers */
struct EmitCtx {
  const uint32_t *nb_outputs, *term_kw, *out_link, *depth;
  ACMRecord *records;
  unsigned long long *count;
  uint64_t capacity, pos_base;
};

__device__ __forceinline__ EmitCtx
make_emit_ctx (const DevTables &T, const ScanArgs &A) {
  EmitCtx c;
  c.nb_outputs = T.nb_outputs;
  c.term_kw = T.term_kw;
  c.out_link = T.out_link;
  c.depth = T.depth;
  c.records = A.records;
  c.count = A.count;
  c.capacity = A.capacity;
  c.pos_base = A.pos_base;
  return c;
}

/* Expand the wave's queue into records: item = (buffer position, state with nb_outputs > 0).
 * Records of one position stay consecutive and in acm_get_match index order (the state itself if
 * terminal, then the chain of out_link: reference aho_corasick.c:459-466).  Must be called by
 * all 64 lanes. */
template <bool COUNT_ONLY>
__device__ __noinline__ void
flush_queue (EmitCtx E, const uint4 *queue, uint32_t n_items) {
  const uint32_t lane = lane_id ();
  for (uint32_t base = 0; base < n_items; base += WAVE) {
    const uint32_t i = base + lane;
    const bool valid = i < n_items;
    uint4 it = valid ? queue[i] : make_uint4 (0, 0, 0, 0);
    const uint32_t st = it.z;
    const uint32_t cnt = valid ? E.nb_outputs[st] : 0;
    const uint32_t incl = wave_incl_scan (cnt);
    const uint32_t total = __shfl (incl, WAVE - 1, WAVE);
    unsigned long long gbase = 0;
    if (lane == 0 && total)
      gbase = atomicAdd (E.count, (unsigned long long)total);
    gbase = ((unsigned long long)__shfl ((uint32_t)(gbase >> 32), 0, WAVE) << 32) | __shfl ((uint32_t)gbase, 0, WAVE);
    if (!COUNT_ONLY && cnt) {
      const uint64_t pos = E.pos_base + (((uint64_t)it.y << 32) | it.x);
      uint64_t o = gbase + (incl - cnt);
      uint32_t t = E.term_kw[st] != NONE ? st : E.out_link[st];
      while (t) {
        if (o < E.capacity) {
          uint4 rec = make_uint4 ((uint32_t)pos, (uint32_t)(pos >> 32), E.depth[t], E.term_kw[t]);
          *reinterpret_cast<uint4 *> (&E.records[o]) = rec;
        }
        o++;
        t = E.out_link[t];
      }
    }
  }
}

/* ------------------------------------------------------------------ dense byte kernel */
template <typename ENTRY> struct EntryTraits;
template <> struct EntryTraits<uint16_t> {
  static constexpr uint32_t FLAG = 0x8000u;
};
template <> struct EntryTraits<uint32_t> {
  static constexpr uint32_t FLAG = 0x80000000u;
};

/* small uniform constants of the dense kernel, kept in SGPRs */
struct DenseK {
  uint32_t W, rowbytes, lo, span, HL;
};

/* per-wave walking state of the dense kernel */
template <int S> struct Walk {
  uint32_t s[S];   /* current state of each stream */
  uint64_t cs[S];  /* buffer index of the first byte of each stream's chunk */
  uint32_t qn;     /* queue fill (wave-uniform) */
  uint32_t sticky; /* per lane: ~0 while one of its streams sits in a state whose row is not in LDS */
};

__device__ __forceinline__ uint32_t
uniform (uint32_t v) { /* tells the compiler the value is the same in every lane */
  return __builtin_amdgcn_readfirstlane (v);
}

/* Slow side of one step.  The whole wave comes here when some lane looked up an entry >= HL
 * (next state has outputs, or its row lives only in HBM) or some lane is currently in such a
 * cold state (its LDS lookup was meaningless and is redone from the HBM copy here). */
template <typename ENTRY, int S, bool COUNT_ONLY>
__device__ __forceinline__ void
dense_step_slow (const DenseK &K, const EmitCtx &E, uint64_t emit_from, uint64_t emit_end, const ENTRY *__restrict__ gdense, uint4 *queue,
                 Walk<S> &w, uint32_t (&e)[S], const uint32_t (&cls)[S], uint32_t j, bool emit, uint32_t lane) {
  constexpr uint32_t FLAG = EntryTraits<ENTRY>::FLAG;
  constexpr uint32_t IDMASK = FLAG - 1;
  bool cold = false;
#pragma unroll
  for (int q = 0; q < S; q++) {
    if (w.s[q] >= K.HL)
      e[q] = gdense[w.s[q] * K.W + cls[q]];
    const uint32_t ns = e[q] & IDMASK;
    const uint64_t pos = w.cs[q] + j;
    const bool hit = emit && (e[q] & FLAG) && pos >= emit_from && pos < emit_end;
    const uint64_t m = __ballot (hit);
    if (m) {
      if (hit)
        queue[w.qn + __popcll (m & ((1ull << lane) - 1))] = make_uint4 ((uint32_t)pos, (uint32_t)(pos >> 32), ns, 0);
      w.qn = uniform (w.qn + (uint32_t)__popcll (m));
      if (w.qn > QCAP - WAVE) {
        flush_queue<COUNT_ONLY> (E, queue, w.qn);
        w.qn = 0;
      }
    }
    w.s[q] = ns;
    cold |= ns >= K.HL;
  }
  w.sticky = cold ? ~0u : 0u;
}

/* One step of all S streams of a lane: byte b[q] for stream q at chunk offset j.
 * Fast side per stream: class = min(byte - lo, span); one ds_read_u16/b32 at
 * row(state) + class; all streams share one compare-and-branch. */
template <typename ENTRY, int S, bool COUNT_ONLY>
__device__ __forceinline__ void
dense_step (const DenseK &K, const EmitCtx &E, uint64_t emit_from, uint64_t emit_end, const unsigned char *lds,
            const ENTRY *__restrict__ gdense, uint4 *queue, Walk<S> &w, const uint32_t (&b)[S], uint32_t j, bool emit,
            uint32_t lane) {
  uint32_t cls[S], e[S];
#pragma unroll
  for (int q = 0; q < S; q++) {
    cls[q] = min (b[q] - K.lo, K.span);
    /* a cold lane reads past the staged rows: LDS returns 0 for out-of-range addresses and the
     * value is replaced on the slow side */
    const uint32_t addr = __umul24 (w.s[q], K.rowbytes) + cls[q] * (uint32_t)sizeof (ENTRY);
    /* the staged rows start at LDS address 0 (no static LDS in this kernel): address the LDS
     * by integer so that no base is added per lookup */
    e[q] = *reinterpret_cast<const __attribute__ ((address_space (3))) ENTRY *> (addr);
  }
  uint32_t emax = w.sticky;
#pragma unroll
  for (int q = 0; q < S; q++)
    emax = max (emax, e[q]);
  if (__builtin_expect (__ballot (emax >= K.HL) != 0, 0))
    dense_step_slow<ENTRY, S, COUNT_ONLY> (K, E, emit_from, emit_end, gdense, queue, w, e, cls, j, emit, lane);
  else {
#pragma unroll
    for (int q = 0; q < S; q++)
      w.s[q] = e[q];
  }
}

/* 16 steps over one 16-byte block per stream */
template <typename ENTRY, int S, bool COUNT_ONLY>
__device__ __forceinline__ void
dense_block (const DenseK &K, const EmitCtx &E, uint64_t emit_from, uint64_t emit_end, const unsigned char *tab,
             const ENTRY *__restrict__ gdense, uint4 *queue, Walk<S> &w, const uint4 (&blk)[S], uint32_t j0, bool emit,
             uint32_t lane) {
#define ACM_BYTE(COMP, SH, J)                                                                      \
  {                                                                                                \
    uint32_t b_[S];                                                                                \
    _Pragma ("unroll") for (int q = 0; q < S; q++) b_[q] = (blk[q].COMP >> (SH)) & 0xffu;          \
    dense_step<ENTRY, S, COUNT_ONLY> (K, E, emit_from, emit_end, tab, gdense, queue, w, b_, j0 + (J), emit, lane); \
  }
#define ACM_WORD(COMP, J)                                                                          \
  ACM_BYTE (COMP, 0, (J) + 0) ACM_BYTE (COMP, 8, (J) + 1) ACM_BYTE (COMP, 16, (J) + 2) ACM_BYTE (COMP, 24, (J) + 3)
  ACM_WORD (x, 0) ACM_WORD (y, 4) ACM_WORD (z, 8) ACM_WORD (w, 12)
#undef ACM_WORD
#undef ACM_BYTE
}

/* compile-time loop over the C/16 blocks of a chunk (keeps the text registers statically indexed) */
template <typename ENTRY, int S, bool COUNT_ONLY, int K0, int KN> struct BlockLoop {
  static __device__ __forceinline__ void
  run (const DenseK &K, const EmitCtx &E, uint64_t emit_from, uint64_t emit_end, const unsigned char *tab,
       const ENTRY *__restrict__ gdense, uint4 *queue, Walk<S> &w, const uint4 (&d)[KN][S], uint32_t lane) {
    dense_block<ENTRY, S, COUNT_ONLY> (K, E, emit_from, emit_end, tab, gdense, queue, w, d[K0], 16 * K0, true, lane);
    BlockLoop<ENTRY, S, COUNT_ONLY, K0 + 1, KN>::run (K, E, emit_from, emit_end, tab, gdense, queue, w, d, lane);
  }
};
template <typename ENTRY, int S, bool COUNT_ONLY, int KN> struct BlockLoop<ENTRY, S, COUNT_ONLY, KN, KN> {
  static __device__ __forceinline__ void
  run (const DenseK &, const EmitCtx &, uint64_t, uint64_t, const unsigned char *, const ENTRY *__restrict__, uint4 *,
       Walk<S> &, const uint4 (&)[KN][S], uint32_t) {}
};

/* Tiles [range_begin, range_end) of 64*S*C bytes cover the whole buffer, the last one possibly
 * ragged.  16-byte loads are clamped to the last block that holds a valid byte (an aligned
 * 16-byte block never straddles a page, so it cannot fault); what a lane walks beyond the end of
 * the buffer is never reported (emit window [emit_from, n)). */
template <typename ENTRY, int C, int S, bool COUNT_ONLY>
__global__ __launch_bounds__ (DENSE_THREADS) void
scan_dense_kernel (DevTables T, ScanArgs A, const ENTRY *__restrict__ gdense, const unsigned char *__restrict__ text,
                   uint32_t queue_off) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  const unsigned char *tab = smem;
  constexpr uint32_t TILE = WAVE * S * C;
  constexpr int NB = C / 16;

  /* stage the hottest rows: a straight 16-byte-per-lane copy (the source is padded to 16 bytes) */
  {
    const uint32_t bytes = T.lds_rows * T.width * (uint32_t)sizeof (ENTRY);
    const uint4 *src = reinterpret_cast<const uint4 *> (gdense);
    uint4 *dst = reinterpret_cast<uint4 *> (smem);
    for (uint32_t i = threadIdx.x; i < (bytes + 15) / 16; i += blockDim.x)
      dst[i] = src[i];
  }
  __syncthreads ();

  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t wib = uniform (threadIdx.x / WAVE);
  uint4 *queue = reinterpret_cast<uint4 *> (smem + queue_off) + wib * QCAP;
  const uint32_t waves_per_block = blockDim.x / WAVE;
  const uint64_t wave = (uint64_t)blockIdx.x * waves_per_block + wib;
  const uint64_t nwaves = (uint64_t)gridDim.x * waves_per_block;
  const uint32_t wub = (T.lmax > 1 ? (T.lmax - 1 + 15) / 16 : 0); /* warm-up blocks of 16 bytes */
  const DenseK K = { T.width, T.width * (uint32_t)sizeof (ENTRY), T.lo, T.span, T.lds_rows };
  const EmitCtx E = make_emit_ctx (T, A);
  const uint64_t emit_from = A.emit_from, emit_end = A.n;
  const uint64_t last_block = (A.n - 1) & ~15ull; /* byte offset of the last 16-byte block with a valid byte */

  Walk<S> w;
  w.qn = 0;
  w.sticky = 0;

  for (uint64_t tile = A.range_begin + wave; tile < A.range_end; tile += nwaves) {
    const uint64_t tbase = tile * TILE;
    uint4 d[NB][S];
#pragma unroll
    for (int q = 0; q < S; q++) {
      w.cs[q] = tbase + (uint64_t)(q * WAVE + lane) * C;
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const uint64_t off = w.cs[q] + 16 * k;
        d[k][q] = *reinterpret_cast<const uint4 *> (text + (off < last_block ? off : last_block));
      }
      w.s[q] = 0;
    }
    w.sticky = 0;
    /* pull the wave's next tile towards L2 while this one is walked: one dword per lane-stream
     * touches every 64-byte half line of it */
    uint32_t touch[S];
    {
      const uint64_t nbase = tbase + nwaves * TILE;
#pragma unroll
      for (int q = 0; q < S; q++) {
        const uint64_t off = nbase + (uint64_t)(q * WAVE + lane) * C;
        touch[q] = *reinterpret_cast<const uint32_t *> (text + (off < last_block ? off : last_block));
      }
    }
    /* warm-up: wub 16-byte blocks before each chunk, walked from the root without reporting
     * (matches ending there belong to the previous chunk's owner).  A chunk closer than that to
     * the start of the buffer starts from the root at its first in-range block instead. */
    for (uint32_t b = wub; b >= 1; b--) {
      uint4 pre[S];
#pragma unroll
      for (int q = 0; q < S; q++) {
        const uint64_t back = 16ull * b;
        const uint64_t off = w.cs[q] >= back ? w.cs[q] - back : 0;
        pre[q] = *reinterpret_cast<const uint4 *> (text + (off < last_block ? off : last_block));
      }
      dense_block<ENTRY, S, COUNT_ONLY> (K, E, emit_from, emit_end, tab, gdense, queue, w, pre, 0, false, lane);
#pragma unroll
      for (int q = 0; q < S; q++)
        if (w.cs[q] < 16ull * b)
          w.s[q] = 0;
    }
    BlockLoop<ENTRY, S, COUNT_ONLY, 0, NB>::run (K, E, emit_from, emit_end, tab, gdense, queue, w, d, lane);
#pragma unroll
    for (int q = 0; q < S; q++)
      asm volatile("" ::"v"(touch[q])); /* keeps the prefetch loads alive; they landed long ago */
  }
  flush_queue<COUNT_ONLY> (E, queue, w.qn);
}

/* ------------------------------------------------------------------ CSR kernel (any width) */
template <typename SYM>
__device__ __forceinline__ uint32_t
csr_step (const DevTables &T, uint32_t s, uint32_t c) {
  for (;;) {
    uint32_t b = T.row_ptr[s], e = T.row_ptr[s + 1];
    if (e - b > 8) { /* rows are sorted by numeric symbol value */
      while (e - b > 1) {
        uint32_t m = (b + e) >> 1;
        if (T.edge_sym[m] <= c)
          b = m;
        else
          e = m;
      }
      if (T.edge_sym[b] == c)
        return T.edge_next[b];
    } else {
      for (; b < e; b++)
        if (T.edge_sym[b] == c)
          return T.edge_next[b];
    }
    if (s == 0)
      return 0;
    s = T.fail[s];
  }
}

/* One lane walks `chunk` symbols of [range_begin, range_end), restarting from the root lmax-1
 * symbols earlier (or at buffer index 0).  blockDim.x == 64: one wave per block, its queue in
 * static LDS. */
template <typename SYM, bool COUNT_ONLY>
__global__ __launch_bounds__ (WAVE) void
scan_csr_kernel (DevTables T, ScanArgs A, uint32_t chunk) {
  __shared__ uint4 queue[QCAP];
  const uint32_t lane = threadIdx.x;
  const SYM *text = reinterpret_cast<const SYM *> (A.text);
  const uint64_t nchunks = (A.range_end - A.range_begin + chunk - 1) / chunk;
  const uint64_t rounds = (nchunks + (uint64_t)gridDim.x * WAVE - 1) / ((uint64_t)gridDim.x * WAVE);
  const uint64_t warm = T.lmax > 1 ? T.lmax - 1 : 0;
  uint32_t qn = 0;
  const EmitCtx E = make_emit_ctx (T, A);
  for (uint64_t r = 0; r < rounds; r++) {
    const uint64_t ck = (r * gridDim.x + blockIdx.x) * WAVE + lane;
    uint64_t begin = A.range_end, end = A.range_end, i = A.range_end;
    if (ck < nchunks) {
      begin = A.range_begin + ck * chunk;
      end = begin + chunk < A.range_end ? begin + chunk : A.range_end;
      i = begin > warm ? begin - warm : 0;
    }
    uint32_t s = 0;
    /* all lanes iterate together so that the queue stays a wave-level structure */
    const uint64_t steps_max = chunk + warm;
    for (uint64_t k = 0; k < steps_max; k++, i++) {
      bool hit = false;
      if (i < end) {
        s = csr_step<SYM> (T, s, (uint32_t)text[i]);
        hit = i >= begin && i >= A.emit_from && T.nb_outputs[s] != 0;
      }
      const uint64_t m = __ballot (hit);
      if (m) {
        if (hit)
          queue[qn + __popcll (m & ((1ull << lane) - 1))] = make_uint4 ((uint32_t)i, (uint32_t)(i >> 32), s, 0);
        qn += (uint32_t)__popcll (m);
        if (qn > QCAP - WAVE) {
          flush_queue<COUNT_ONLY> (E, queue, qn);
          qn = 0;
        }
      }
    }
  }
  flush_queue<COUNT_ONLY> (E, queue, qn);
}

/* ------------------------------------------------------------------ sort keys */
__global__ void
make_keys_kernel (const ACMRecord *rec, uint64_t n, uint64_t pos_min, uint32_t len_bits, uint64_t *keys) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint64_t lmask = (1ull << len_bits) - 1;
    keys[i] = ((rec[i].end_pos - pos_min) << len_bits) | (lmask - (rec[i].length & lmask));
  }
}

struct Rec16 {
  uint64_t a, b;
};

/* ------------------------------------------------------------------ synthetic text */
__device__ __forceinline__ uint64_t
splitmix64 (uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <typename SYM>
__global__ void
synth_text_kernel (SYM *text, uint64_t n, uint64_t gbegin, uint64_t n_total_hint, uint32_t vocab, const SYM *kw, const uint32_t *kw_off,
                   uint32_t n_kw) {
  (void)n_total_hint;
  constexpr uint64_t P = 4096;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; li < n; li += stride) {
    const uint64_t i = gbegin + li;
    const uint64_t p = i & ~(P - 1);
    uint64_t v = sizeof (SYM) == 1 ? (uint64_t)'a' + splitmix64 (i + 42) % 26 : splitmix64 (i + 42) % vocab;
    if (n_kw) {
      const uint64_t off = p + splitmix64 (p) % (P - 16);
      const uint32_t k = (uint32_t)(splitmix64 (p + 99) % n_kw);
      const uint32_t len = kw_off[k + 1] - kw_off[k];
      if (i >= off && i < off + len)
        v = kw[kw_off[k] + (i - off)];
    }
    text[li] = (SYM)v;
  }
}

} // namespace

/* ====================================================================== host side */
struct ACMPlan {
  int device = 0;
  ACMFlatInfo finfo{};
  DevTables T{};
  ACMPlanInfo info{};
  void *blob = nullptr; /* one device allocation holding every table */
  size_t blob_bytes = 0;
  uint32_t queue_off = 0;
  uint64_t generation = 0; /* for the machine-cached plan */
  int cu_count = 0;
  /* timing */
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  double timing_ms = 0;
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

template <typename T>
size_t
blob_reserve (size_t &cursor, size_t count) {
  cursor = (cursor + 255) & ~(size_t)255;
  size_t at = cursor;
  cursor += count * sizeof (T);
  return at;
}

using DenseKernel16 = void (*) (DevTables, ScanArgs, const uint16_t *, const unsigned char *, uint32_t);
using DenseKernel32 = void (*) (DevTables, ScanArgs, const uint32_t *, const unsigned char *, uint32_t);

DenseKernel16
dense_kernel16 (bool count_only) {
  return count_only ? scan_dense_kernel<uint16_t, DENSE_C, DENSE_S, true> : scan_dense_kernel<uint16_t, DENSE_C, DENSE_S, false>;
}
DenseKernel32
dense_kernel32 (bool count_only) {
  return count_only ? scan_dense_kernel<uint32_t, DENSE_C, DENSE_S, true> : scan_dense_kernel<uint32_t, DENSE_C, DENSE_S, false>;
}

void
drop_cached_plan (void *p) {
  acm_gpu_plan_destroy (static_cast<ACMPlan *> (p));
}

} // namespace

extern "C" int
acm_gpu_plan_create_flat (const ACMFlat *flat, int device, ACMPlan **out) {
  if (!flat || !out)
    return ACM_GPU_E_ARG;
  int ndev = 0;
  if (hipGetDeviceCount (&ndev) != hipSuccess || ndev <= 0)
    return ACM_GPU_E_NODEVICE;
  if (device < 0 || device >= ndev)
    return ACM_GPU_E_ARG;
  HIP_TRY (hipSetDevice (device));
  hipDeviceProp_t prop;
  HIP_TRY (hipGetDeviceProperties (&prop, device));

  ACMFlatInfo fi;
  ACMFlatView fv;
  acm_flat_info (flat, &fi);
  acm_flat_view (flat, &fv);
  if (fi.lmax >= (1u << 24))
    return ACM_GPU_E_INELIGIBLE;

  ACMPlan *p = new (std::nothrow) ACMPlan ();
  if (!p)
    return ACM_GPU_E_NOMEM;
  p->device = device;
  p->finfo = fi;
  p->cu_count = prop.multiProcessorCount;

  /* dense rows for byte alphabets whenever the whole DFA fits comfortably in HBM */
  const uint32_t n = fi.n_states;
  uint32_t entry_bytes = n <= 32768 ? 2 : 4;
  bool dense = fi.sym_bytes == 1 && fi.n_edges > 0 && (uint64_t)n * fi.width < (1ull << 31) &&
               (uint64_t)n * fi.width * entry_bytes <= (8ull << 30);
  const uint32_t nwaves_blk = DENSE_THREADS / WAVE;
  const uint32_t queue_bytes = nwaves_blk * QCAP * 16;
  uint32_t lds_rows = 0, table_lds = 0;
  if (dense) {
    const uint32_t lds_total = (uint32_t)prop.maxSharedMemoryPerMultiProcessor >= 160 * 1024 ? 160 * 1024 : 64 * 1024;
    const uint32_t budget = lds_total - queue_bytes - 1024;
    const uint32_t row_bytes = fi.width * entry_bytes;
    lds_rows = budget / row_bytes;
    if (lds_rows > n)
      lds_rows = n;
    /* keep the LDS-resident prefix below the flag bit */
    table_lds = (lds_rows * row_bytes + 15) & ~15u;
  }

  size_t cur = 0;
  const size_t o_row = blob_reserve<uint32_t> (cur, (size_t)n + 1);
  const size_t o_sym = blob_reserve<uint32_t> (cur, fi.n_edges ? fi.n_edges : 1);
  const size_t o_next = blob_reserve<uint32_t> (cur, fi.n_edges ? fi.n_edges : 1);
  const size_t o_fail = blob_reserve<uint32_t> (cur, n);
  const size_t o_depth = blob_reserve<uint32_t> (cur, n);
  const size_t o_nbo = blob_reserve<uint32_t> (cur, n);
  const size_t o_term = blob_reserve<uint32_t> (cur, n);
  const size_t o_link = blob_reserve<uint32_t> (cur, n);
  const size_t dense_bytes = dense ? (size_t)n * fi.width * entry_bytes : 0;
  const size_t o_dense = blob_reserve<unsigned char> (cur, dense_bytes + 16);
  p->blob_bytes = cur;

  std::vector<unsigned char> host (cur, 0);
  memcpy (&host[o_row], fv.row_ptr, ((size_t)n + 1) * 4);
  memcpy (&host[o_sym], fv.edge_sym, (size_t)fi.n_edges * 4);
  memcpy (&host[o_next], fv.edge_next, (size_t)fi.n_edges * 4);
  memcpy (&host[o_fail], fv.fail, (size_t)n * 4);
  memcpy (&host[o_depth], fv.depth, (size_t)n * 4);
  memcpy (&host[o_nbo], fv.nb_outputs, (size_t)n * 4);
  memcpy (&host[o_term], fv.term_kw, (size_t)n * 4);
  memcpy (&host[o_link], fv.out_link, (size_t)n * 4);
  if (dense) {
    int rc = acm_flat_dense_rows (flat, n, entry_bytes, &host[o_dense]);
    if (rc) {
      delete p;
      return rc;
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
  unsigned char *b = static_cast<unsigned char *> (p->blob);
  DevTables &T = p->T;
  T.row_ptr = reinterpret_cast<uint32_t *> (b + o_row);
  T.edge_sym = reinterpret_cast<uint32_t *> (b + o_sym);
  T.edge_next = reinterpret_cast<uint32_t *> (b + o_next);
  T.fail = reinterpret_cast<uint32_t *> (b + o_fail);
  T.depth = reinterpret_cast<uint32_t *> (b + o_depth);
  T.nb_outputs = reinterpret_cast<uint32_t *> (b + o_nbo);
  T.term_kw = reinterpret_cast<uint32_t *> (b + o_term);
  T.out_link = reinterpret_cast<uint32_t *> (b + o_link);
  T.dense = dense ? b + o_dense : nullptr;
  T.n_states = n;
  T.width = fi.width;
  T.lo = fi.alpha_lo;
  T.span = fi.alpha_span;
  T.dense_rows = dense ? n : 0;
  T.lds_rows = lds_rows;
  T.lmax = fi.lmax;
  T.entry_bytes = entry_bytes;
  p->queue_off = table_lds;

  ACMPlanInfo &I = p->info;
  I.device = device;
  I.kernel = dense ? 1 : 2;
  I.entry_bytes = dense ? entry_bytes : 0;
  I.width = fi.width;
  I.dense_rows = T.dense_rows;
  I.lds_rows = lds_rows;
  I.lds_bytes = dense ? table_lds + queue_bytes : QCAP * 16;
  I.block_threads = dense ? DENSE_THREADS : WAVE;
  I.grid_blocks = dense ? (uint32_t)p->cu_count : (uint32_t)p->cu_count * 16;
  I.chunk_bytes = DENSE_C;
  I.streams = DENSE_S;
  I.table_bytes = cur;

  if (dense) {
    for (int co = 0; co < 2; co++) {
      const void *fn = entry_bytes == 2 ? reinterpret_cast<const void *> (dense_kernel16 (co != 0))
                                        : reinterpret_cast<const void *> (dense_kernel32 (co != 0));
      HIP_TRY (hipFuncSetAttribute (fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)I.lds_bytes));
    }
  }
  *out = p;
  return ACM_GPU_OK;
}

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

extern "C" void
acm_gpu_plan_destroy (ACMPlan *plan) {
  if (!plan)
    return;
  (void)hipSetDevice (plan->device);
  for (auto &ev : plan->events) {
    (void)hipEventDestroy (ev.first);
    (void)hipEventDestroy (ev.second);
  }
  if (plan->blob)
    (void)hipFree (plan->blob);
  delete plan;
}

extern "C" void
acm_gpu_plan_info (const ACMPlan *plan, ACMPlanInfo *info) {
  *info = plan->info;
}

extern "C" int
acm_gpu_plan_timing (ACMPlan *plan, int enable) {
  if (!plan)
    return ACM_GPU_E_ARG;
  plan->timing = enable != 0;
  plan->events_used = 0;
  plan->timing_ms = 0;
  plan->timing_launches = 0;
  return ACM_GPU_OK;
}

extern "C" int
acm_gpu_plan_timing_read (ACMPlan *plan, double *total_ms, uint64_t *launches) {
  if (!plan)
    return ACM_GPU_E_ARG;
  HIP_TRY (hipSetDevice (plan->device));
  for (size_t i = 0; i < plan->events_used; i++) {
    HIP_TRY (hipEventSynchronize (plan->events[i].second));
    float ms = 0;
    HIP_TRY (hipEventElapsedTime (&ms, plan->events[i].first, plan->events[i].second));
    plan->timing_ms += ms;
    plan->timing_launches++;
  }
  plan->events_used = 0;
  if (total_ms)
    *total_ms = plan->timing_ms;
  if (launches)
    *launches = plan->timing_launches;
  return ACM_GPU_OK;
}

namespace {

int
timing_begin (ACMPlan *p, hipStream_t st, hipEvent_t *stop) {
  *stop = nullptr;
  if (!p->timing)
    return ACM_GPU_OK;
  if (p->events_used == p->events.size ()) {
    if (p->events.size () >= 4096) { /* fold what is recorded so far */
      int rc = acm_gpu_plan_timing_read (p, nullptr, nullptr);
      if (rc)
        return rc;
    } else {
      hipEvent_t a, b;
      HIP_TRY (hipEventCreate (&a));
      HIP_TRY (hipEventCreate (&b));
      p->events.emplace_back (a, b);
    }
  }
  auto &ev = p->events[p->events_used++];
  HIP_TRY (hipEventRecord (ev.first, st));
  *stop = ev.second;
  return ACM_GPU_OK;
}

template <bool COUNT_ONLY>
int
launch_csr (ACMPlan *p, const ScanArgs &base, uint64_t begin, uint64_t end, hipStream_t st) {
  if (end <= begin)
    return ACM_GPU_OK;
  ScanArgs a = base;
  a.range_begin = begin;
  a.range_end = end;
  const uint64_t len = end - begin;
  /* chunk length: enough chunks to fill the chip, long enough to amortise the warm-up */
  uint64_t target_lanes = (uint64_t)p->cu_count * 16 * WAVE;
  uint64_t chunk = (len + target_lanes - 1) / target_lanes;
  const uint64_t minchunk = 64 > 8ull * p->finfo.lmax ? 64 : 8ull * p->finfo.lmax;
  if (chunk < minchunk)
    chunk = minchunk;
  if (chunk > 4096)
    chunk = 4096;
  const uint64_t nchunks = (len + chunk - 1) / chunk;
  uint64_t blocks = (nchunks + WAVE - 1) / WAVE;
  const uint64_t maxblocks = (uint64_t)p->cu_count * 32;
  if (blocks > maxblocks)
    blocks = maxblocks;
  dim3 g ((uint32_t)blocks), b (WAVE);
  switch (p->finfo.sym_bytes) {
  case 1: hipLaunchKernelGGL ((scan_csr_kernel<uint8_t, COUNT_ONLY>), g, b, 0, st, p->T, a, (uint32_t)chunk); break;
  case 2: hipLaunchKernelGGL ((scan_csr_kernel<uint16_t, COUNT_ONLY>), g, b, 0, st, p->T, a, (uint32_t)chunk); break;
  default: hipLaunchKernelGGL ((scan_csr_kernel<uint32_t, COUNT_ONLY>), g, b, 0, st, p->T, a, (uint32_t)chunk); break;
  }
  HIP_TRY (hipGetLastError ());
  return ACM_GPU_OK;
}

template <bool COUNT_ONLY>
int
scan_impl (ACMPlan *p, const void *d_text, uint64_t n, uint64_t emit_from, uint64_t pos_base, ACMRecord *d_records,
           uint64_t capacity, uint64_t *d_count, hipStream_t st) {
  HIP_TRY (hipSetDevice (p->device));
  HIP_TRY (hipMemsetAsync (d_count, 0, sizeof (uint64_t), st));
  if (n == 0 || p->finfo.n_edges == 0)
    return ACM_GPU_OK;
  ScanArgs a{};
  a.text = static_cast<const unsigned char *> (d_text);
  a.n = n;
  a.emit_from = emit_from;
  a.pos_base = pos_base;
  a.records = d_records;
  a.capacity = COUNT_ONLY ? 0 : capacity;
  a.count = reinterpret_cast<unsigned long long *> (d_count);

  const uint64_t TILE = (uint64_t)WAVE * DENSE_S * DENSE_C;
  if (p->info.kernel == 1 && (reinterpret_cast<uintptr_t> (d_text) & 15) == 0) {
    /* one launch covers the whole buffer, ragged last tile included */
    ScanArgs f = a;
    f.range_begin = 0;
    f.range_end = (n + TILE - 1) / TILE;
    hipEvent_t stop;
    int rc = timing_begin (p, st, &stop);
    if (rc)
      return rc;
    uint32_t grid = p->info.grid_blocks;
    const uint64_t waves_needed = f.range_end;
    const uint64_t blocks_needed = (waves_needed + DENSE_THREADS / WAVE - 1) / (DENSE_THREADS / WAVE);
    if (blocks_needed < grid)
      grid = (uint32_t)blocks_needed;
    if (p->T.entry_bytes == 2)
      hipLaunchKernelGGL (dense_kernel16 (COUNT_ONLY), dim3 (grid), dim3 (DENSE_THREADS), p->info.lds_bytes, st, p->T, f,
                          static_cast<const uint16_t *> (p->T.dense), f.text, p->queue_off);
    else
      hipLaunchKernelGGL (dense_kernel32 (COUNT_ONLY), dim3 (grid), dim3 (DENSE_THREADS), p->info.lds_bytes, st, p->T, f,
                          static_cast<const uint32_t *> (p->T.dense), f.text, p->queue_off);
    HIP_TRY (hipGetLastError ());
    if (stop)
      HIP_TRY (hipEventRecord (stop, st));
    return ACM_GPU_OK;
  }
  hipEvent_t stop;
  int rc = timing_begin (p, st, &stop);
  if (rc)
    return rc;
  rc = launch_csr<COUNT_ONLY> (p, a, 0, n, st);
  if (rc)
    return rc;
  if (stop)
    HIP_TRY (hipEventRecord (stop, st));
  return ACM_GPU_OK;
}

} // namespace

extern "C" int
acm_gpu_scan_device (ACMPlan *plan, const void *d_text, uint64_t n_symbols, uint64_t emit_from, uint64_t pos_base,
                     ACMRecord *d_records, uint64_t capacity, uint64_t *d_count, void *stream) {
  if (!plan || !d_count || (n_symbols && !d_text) || (capacity && !d_records))
    return ACM_GPU_E_ARG;
  return scan_impl<false> (plan, d_text, n_symbols, emit_from, pos_base, d_records, capacity, d_count,
                           static_cast<hipStream_t> (stream));
}

extern "C" int
acm_gpu_count_device (ACMPlan *plan, const void *d_text, uint64_t n_symbols, uint64_t emit_from, uint64_t *d_count,
                      void *stream) {
  if (!plan || !d_count || (n_symbols && !d_text))
    return ACM_GPU_E_ARG;
  return scan_impl<true> (plan, d_text, n_symbols, emit_from, 0, nullptr, 0, d_count, static_cast<hipStream_t> (stream));
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

extern "C" int
acm_gpu_sort_records_device (ACMPlan *plan, ACMRecord *d_records, uint64_t n, void *d_tmp, size_t tmp_bytes, void *stream) {
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
  /* keys are relative to position 0 of the record space; 64 - len_bits bits remain for positions */
  hipLaunchKernelGGL (make_keys_kernel, dim3 ((uint32_t)((n + 255) / 256)), dim3 (256), 0, st, d_records, n, 0ull, len_bits, k0);
  HIP_TRY (hipGetLastError ());
  hipcub::DoubleBuffer<uint64_t> keys (k0, k1);
  hipcub::DoubleBuffer<Rec16> vals (reinterpret_cast<Rec16 *> (d_records), v1);
  HIP_TRY (hipcub::DeviceRadixSort::SortPairs (cub_tmp, cub_bytes, keys, vals, (int)n, 0, 64, st));
  if (vals.Current () != reinterpret_cast<Rec16 *> (d_records))
    HIP_TRY (hipMemcpyAsync (d_records, vals.Current (), n * 16, hipMemcpyDeviceToDevice, st));
  return ACM_GPU_OK;
}

/* ------------------------------------------------------------------ host-buffer convenience */
extern "C" int
acm_gpu_scan_host (ACMPlan *plan, const void *text, uint64_t n_symbols, uint64_t emit_from, uint64_t pos_base,
                   ACMRecord *records, uint64_t capacity, uint64_t *n_found) {
  if (!plan || !n_found || (n_symbols && !text) || (capacity && !records))
    return ACM_GPU_E_ARG;
  HIP_TRY (hipSetDevice (plan->device));
  const size_t tbytes = (size_t)n_symbols * plan->finfo.sym_bytes;
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
    size_t tb = acm_gpu_sort_tmp_bytes (found);
    HOST_TRY (hipMalloc (&d_tmp, tb));
    rc = acm_gpu_sort_records_device (plan, static_cast<ACMRecord *> (d_rec), found, d_tmp, tb, nullptr);
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

extern "C" int
acm_scan (ACMachine *machine, const void *text, uint64_t n_symbols, ACMRecord *records, uint64_t capacity, uint64_t *n_found) {
  if (!machine || !n_found)
    return ACM_GPU_E_ARG;
  acm_internal_plan_dropper = drop_cached_plan;
  void **slot = acm_internal_plan_slot (machine);
  ACMPlan *plan = static_cast<ACMPlan *> (*slot);
  const uint64_t gen = acm_internal_generation (machine);
  if (!plan || plan->generation != gen) {
    if (plan) {
      acm_gpu_plan_destroy (plan);
      *slot = nullptr;
    }
    int device = 0;
    if (const char *e = getenv ("ACM_GPU_DEVICE"))
      device = atoi (e);
    int rc = acm_gpu_plan_create (machine, device, &plan);
    if (rc)
      return rc;
    plan->generation = gen;
    *slot = plan;
  }
  return acm_gpu_scan_host (plan, text, n_symbols, 0, 0, records, capacity, n_found);
}

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
    hipLaunchKernelGGL (synth_text_kernel<uint8_t>, g, b, 0, st, static_cast<uint8_t *> (d_text), n, global_begin, 0ull, vocab,
                        static_cast<const uint8_t *> (d_kw_data), d_kw_off, n_kw);
  else
    hipLaunchKernelGGL (synth_text_kernel<uint32_t>, g, b, 0, st, static_cast<uint32_t *> (d_text), n, global_begin, 0ull, vocab,
                        static_cast<const uint32_t *> (d_kw_data), d_kw_off, n_kw);
  HIP_TRY (hipGetLastError ());
  return ACM_GPU_OK;
}
