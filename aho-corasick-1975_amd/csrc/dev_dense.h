/* dev_dense.h -- scan_dense_kernel: byte alphabets, automaton in LDS (continuation and sticky modes).
 * Device code of libac75_amd.so; included by acm_gpu.hip inside its anonymous namespace (one
 * translation unit: the kernels share the structs and helpers declared there and in dev_emit.h). */

/* ------------------------------------------------------------------ dense byte kernel
 * ENTRY = uint16_t: "continuation mode".  LDS holds the rows of the HD shallowest states and, for
 *   every other state s, hotfail(s).  A lane never sits in a rowless state: stepping into one
 *   queues a continuation item and the lane carries on from hotfail(s) (walk_continuation explains why
 *   nothing is lost).  The per-symbol path is: class, one ds_read_u16, compare, branch.
 * ENTRY = uint32_t: "sticky mode" for dictionaries with more than 32768 states.  LDS holds the
 *   rows of a breadth-first prefix; a lane in a deeper state is walked through the HBM rows on
 *   the slow side until it comes back. */
template <typename ENTRY> struct EntryTraits;
template <> struct EntryTraits<uint16_t> {
  static constexpr uint32_t FLAG = 0x8000u;
  static constexpr bool CONT = true;
};
template <> struct EntryTraits<uint32_t> {
  static constexpr uint32_t FLAG = 0x80000000u;
  static constexpr bool CONT = false;
};

/* per-wave walking state of the dense kernel */
template <int S> struct Walk {
  uint32_t s[S];   /* current state of each stream */
  uint32_t qn;     /* queue fill (wave-uniform) */
  uint32_t sticky; /* sticky mode, per lane: ~0 while one of its streams sits in a rowless state */
  Spill spill;
  DIAG (unsigned long long d_slow_cycles = 0; unsigned long long d_slow_steps = 0;)
};

template <typename ENTRY>
__device__ __forceinline__ uint32_t
lds_row_entry (uint32_t state, uint32_t rowbytes, uint32_t cls) {
  /* the rows start at LDS address 0 (no static LDS in this kernel): address the LDS by integer
   * so that no base is added per lookup */
  const uint32_t addr = __umul24 (state, rowbytes) + cls * (uint32_t)sizeof (ENTRY);
  return *reinterpret_cast<const __attribute__ ((address_space (3))) ENTRY *> (addr);
}

/* where a step is: pos0 = position of stream 0's byte; phase MAIN (inside the chunk), WARM
 * (sticky mode: before the chunk, nothing is reported) or RUN (continuation mode: k bytes past
 * the chunk, live_from = first state id of depth k + 1) */
enum { PH_MAIN = 0, PH_WARM = 1, PH_RUN = 2 };
struct StepAt {
  uint32_t pos0, k, live_from;
  int phase;
};

/* Slow side of one step: the whole wave comes here when some lane looked up an entry >= HD (next
 * state has outputs and/or no row in LDS), or -- sticky mode -- sits in a rowless state. */
template <typename ENTRY, int S, bool COUNT_ONLY>
__device__ __forceinline__ void
dense_step_slow (const DenseK &K, const EmitCtx &E, uint32_t emit_from, uint32_t emit_end,
                 const ENTRY *__restrict__ gdense, uint2 *queue, Walk<S> &w, uint32_t (&e)[S], const uint32_t (&cls)[S],
                 const StepAt at, uint32_t lane) {
  constexpr uint32_t FLAG = EntryTraits<ENTRY>::FLAG;
  constexpr uint32_t IDMASK = FLAG - 1;
  constexpr bool CONT = EntryTraits<ENTRY>::CONT;
  bool rowless = false;
  uint32_t ns[S], hf[S];
  /* first all the loads, so that their latencies overlap with the queue bookkeeping below */
#pragma unroll
  for (int q = 0; q < S; q++) {
    if (!CONT && w.s[q] >= K.HD) /* sticky mode: the LDS lookup was meaningless, redo it from HBM */
      e[q] = gdense[w.s[q] * K.W + cls[q]];
  }
#pragma unroll
  for (int q = 0; q < S; q++) {
    ns[q] = e[q] & IDMASK;
    hf[q] = 0;
    if (CONT && ns[q] >= K.HD) /* the lane carries on from the nearest state that has a row */
      hf[q] = *reinterpret_cast<const __attribute__ ((address_space (3))) uint16_t *> (K.aux_off + (ns[q] - K.HD) * 2u);
  }
#pragma unroll
  for (int q = 0; q < S; q++) {
    const uint32_t pos = at.pos0 + (uint32_t)q * K.stream_stride;
    const bool window = pos >= emit_from && pos < emit_end;
    if (CONT) {
      /* run-over: a lane whose state is no deeper than k has nothing of its own left */
      const bool live = at.phase != PH_RUN || ns[q] >= at.live_from;
      const bool out = live && (e[q] & FLAG) && window;
      const bool deep = live && ns[q] >= K.HD && pos < emit_end;
      uint32_t word = ns[q] | (deep ? IT_CONT : 0u) | (out ? IT_OUT : 0u);
      if (at.phase == PH_RUN)
        word |= IT_RUN | (at.k << IT_K_SHIFT);
      queue_push<CONT, COUNT_ONLY, true> (E, queue, w.qn, out | deep, pos, word, lane, &w.spill);
    } else {
      const bool out = at.phase == PH_MAIN && (e[q] & FLAG) && window;
      queue_push<CONT, COUNT_ONLY, true> (E, queue, w.qn, out, pos, ns[q], lane, &w.spill);
      rowless |= ns[q] >= K.HD;
    }
  }
  /* the next states go into e[] itself, after its last use above: the caller takes them from
   * there on both sides of its branch, so the looked-up entry and the state can share a register */
#pragma unroll
  for (int q = 0; q < S; q++)
    e[q] = (CONT && ns[q] >= K.HD) ? hf[q] : ns[q];
  if (!CONT)
    w.sticky = rowless ? ~0u : 0u;
}

/* One step of all S streams of a lane: byte b[q] for stream q.
 * Fast side per stream: class = min(byte - lo, span); one ds_read at row(state) + class; all
 * streams share one compare-and-branch. */
template <typename ENTRY, int S, bool COUNT_ONLY>
__device__ __forceinline__ void
dense_step (const DenseK &K, const EmitCtx &E, uint32_t emit_from, uint32_t emit_end, const ENTRY *__restrict__ gdense,
            uint2 *queue, Walk<S> &w, const uint32_t (&b)[S], const StepAt at, uint32_t lane) {
  uint32_t cls[S], e[S];
#pragma unroll
  for (int q = 0; q < S; q++) {
    cls[q] = min (b[q] - K.lo, K.span);
    e[q] = lds_row_entry<ENTRY> (w.s[q], K.rowbytes, cls[q]);
  }
  uint32_t emax = EntryTraits<ENTRY>::CONT ? 0u : w.sticky;
#pragma unroll
  for (int q = 0; q < S; q++)
    emax = max (emax, e[q]);
  if (__builtin_expect (__ballot (emax >= K.HD) != 0, 0)) {
    DIAG (const unsigned long long t0_ = __builtin_readcyclecounter ();)
    dense_step_slow<ENTRY, S, COUNT_ONLY> (K, E, emit_from, emit_end, gdense, queue, w, e, cls, at, lane);
    DIAG (w.d_slow_cycles += __builtin_readcyclecounter () - t0_; w.d_slow_steps++;)
  }
  /* (one assignment for both sides: with `w.s = e` on the fast side only, the register allocator
   * kept the looked-up entries and the states apart and paid two v_mov per step for it) */
#pragma unroll
  for (int q = 0; q < S; q++)
    w.s[q] = e[q];
}

/* 16 steps over one 16-byte block per stream; at = where the block's first byte is */
template <typename ENTRY, int S, bool COUNT_ONLY>
__device__ __forceinline__ void
dense_block (const DenseK &K, const EmitCtx &E, uint32_t emit_from, uint32_t emit_end, const ENTRY *__restrict__ gdense,
             uint2 *queue, Walk<S> &w, const uint4 (&blk)[S], const StepAt at, uint32_t lane) {
  region_make_room<EntryTraits<ENTRY>::CONT, COUNT_ONLY, S> (E, &w.spill);
#define ACM_BYTE(COMP, SH, J)                                                                      \
  {                                                                                                \
    uint32_t b_[S];                                                                                \
    _Pragma ("unroll") for (int q = 0; q < S; q++) b_[q] = (blk[q].COMP >> (SH)) & 0xffu;          \
    const StepAt at_ = { at.pos0 + (J), 0, 0, at.phase };                                          \
    dense_step<ENTRY, S, COUNT_ONLY> (K, E, emit_from, emit_end, gdense, queue, w, b_, at_, lane); \
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
  run (const DenseK &K, const EmitCtx &E, uint32_t emit_from, uint32_t emit_end, const ENTRY *__restrict__ gdense,
       uint2 *queue, Walk<S> &w, const uint4 (&d)[KN][S], uint32_t pos0, uint32_t lane) {
    const StepAt at = { pos0 + 16 * K0, 0, 0, PH_MAIN };
    dense_block<ENTRY, S, COUNT_ONLY> (K, E, emit_from, emit_end, gdense, queue, w, d[K0], at, lane);
    BlockLoop<ENTRY, S, COUNT_ONLY, K0 + 1, KN>::run (K, E, emit_from, emit_end, gdense, queue, w, d, pos0, lane);
  }
};
template <typename ENTRY, int S, bool COUNT_ONLY, int KN> struct BlockLoop<ENTRY, S, COUNT_ONLY, KN, KN> {
  static __device__ __forceinline__ void
  run (const DenseK &, const EmitCtx &, uint32_t, uint32_t, const ENTRY *__restrict__, uint2 *, Walk<S> &,
       const uint4 (&)[KN][S], uint32_t, uint32_t) {}
};

/* Tiles [range_begin, range_end) of 64*S*C bytes cover the whole segment, the last one possibly
 * ragged.  16-byte loads are clamped to the last block that holds a valid byte (an aligned
 * 16-byte block never straddles a page, so it cannot fault); what a lane walks beyond the end of
 * the segment is never reported (emit window [emit_from, n)).
 * LDS image: [HD rows][continuation mode: hotfail of every other state, 2 B each][16 queues]. */
template <typename ENTRY, int C, int S, bool COUNT_ONLY>
__global__ __launch_bounds__ (DENSE_THREADS) void
scan_dense_kernel (DenseK K, EmitCtx E, Launch A, const ENTRY *__restrict__ gdense, const uint4 *__restrict__ lds_image,
                   uint32_t lds_image_bytes, const unsigned char *__restrict__ text, uint2 *items, uint32_t region_items,
                   uint32_t *fill, const uint32_t *__restrict__ dstart) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  constexpr uint32_t TILE = WAVE * S * C;
  constexpr int NB = C / 16;
  constexpr bool CONT = EntryTraits<ENTRY>::CONT;

  /* stage rows (+ hotfail): a straight 16-byte-per-lane copy of the prebuilt image */
  {
    uint4 *dst = reinterpret_cast<uint4 *> (smem);
    for (uint32_t i = threadIdx.x; i < lds_image_bytes / 16; i += blockDim.x)
      dst[i] = lds_image[i];
  }
  /* tiles are handed out dynamically inside the workgroup (its waves do not run at the same
   * pace: a static split left the slowest wave of a block 11% behind the block's mean) */
  uint32_t *next_tile = reinterpret_cast<uint32_t *> (smem + K.queue_off + (DENSE_THREADS / WAVE) * QCAP * 8);
  if (threadIdx.x == 0)
    *next_tile = 0;
  __syncthreads ();

  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t wib = uniform (threadIdx.x / WAVE);
#ifdef ACM_DENSE_PRIO /* experiment (MI355X_MICROARCH.md, two waves per SIMD, item 4): static priority for the younger half of the block */
  if (wib >= (DENSE_THREADS / WAVE) / 2)
    __builtin_amdgcn_s_setprio (1);
#endif
  uint2 *queue = reinterpret_cast<uint2 *> (smem + K.queue_off) + wib * QCAP;
  const uint32_t waves_per_block = blockDim.x / WAVE;
  const uint32_t wave = blockIdx.x * waves_per_block + wib;
  const uint32_t emit_from = A.emit_from, emit_end = A.n;
  const uint32_t last_block = (A.n - 1) & ~15u; /* byte offset of the last 16-byte block with a valid byte */
  /* Work split.  Block b takes the tiles b, b + gridDim.x, ... of [range_begin, static_end),
   * handed to its waves through an LDS counter (interleaved rather than contiguous shares: match
   * density is not even along a text, and the regions of the item buffer fill more evenly, which
   * the expand kernel likes: 54 -> 47 us); once those are gone its waves draw single tiles from
   * the pool of their class (16 consecutive blocks = 2 per XCD; one counter per class keeps the
   * atomics per counter far below what one address sustains).  Blocks ran up to 6% apart. */
  const uint32_t static_tiles = A.static_end - A.range_begin;
  const uint32_t blk_tiles = blockIdx.x < static_tiles ? (static_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const uint32_t cls = blockIdx.x * A.pool_classes / gridDim.x;
  const uint32_t cls_begin = A.static_end + cls * A.pool_class_tiles;
  const uint32_t cls_tiles = cls_begin >= A.range_end ? 0
                             : (A.range_end - cls_begin < A.pool_class_tiles ? A.range_end - cls_begin : A.pool_class_tiles);
  unsigned int *const cls_ctr = A.pool_ctr + cls * POOL_CTR_STRIDE;
  if (blockIdx.x == 0 && threadIdx.x < POOL_CLASSES)
    A.pool_reset[threadIdx.x * POOL_CTR_STRIDE] = 0;
  /* next tile of this wave, valid in lane 0 (A.range_end = none left); made uniform only where
   * it is used, one tile later, so that the atomics' latency stays hidden */
  auto grab_tile = [&] () -> uint32_t {
    uint32_t t = A.range_end;
    if (lane == 0) {
      const uint32_t i = atomicAdd (next_tile, 1u);
      if (i < blk_tiles)
        t = A.range_begin + i * gridDim.x + blockIdx.x;
      else if (cls_tiles) {
        const uint32_t g = atomicAdd (cls_ctr, 1u);
        if (g < cls_tiles)
          t = cls_begin + g;
      }
    }
    return t;
  };
  uint32_t cur = uniform (grab_tile ());
  uint32_t nxt_raw = cur < A.range_end ? grab_tile () : A.range_end;

  Walk<S> w;
  w.qn = 0;
  w.sticky = 0;
  w.spill.region = items + (size_t)wave * region_items;
  w.spill.capacity = region_items; /* >= DENSE_MIN_REGION_ITEMS: ensure_item_buffer */
  w.spill.fill = 0;
  DIAG (const unsigned long long d_t0 = __builtin_readcyclecounter (); const unsigned long long d_w0 = wall_clock64 (); unsigned long long d_text = 0, d_tiles = 0;)

  static_assert (NB == 4, "the software pipeline below is written out for 4 blocks per chunk");
  uint4 d[NB][S], post[S];
  auto load_block = [&] (uint32_t p0, int k, int q) -> uint4 {
    const uint32_t off = p0 + q * (WAVE * C) + 16 * k;
#ifdef ACM_DENSE_NT /* experiment: the text as a stream that should not stay in the caches -- 0.269 -> 0.493 ms per GiB:
                     * a lane owns 64 contiguous bytes, so each of a wave's four loads touches every 64-byte
                     * sector of its 4 KiB and three of the four find it in L1 / L2 only if it may stay there */
    typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
    const u32x4 v = __builtin_nontemporal_load (reinterpret_cast<const u32x4 *> (text + (off < last_block ? off : last_block)));
    return make_uint4 (v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4 *> (text + (off < last_block ? off : last_block));
#endif
  };
  {
    const uint32_t p0 = cur * TILE + lane * C;
#pragma unroll
    for (int q = 0; q < S; q++) {
      d[0][q] = load_block (p0, 0, q);
      d[1][q] = load_block (p0, 1, q);
      d[2][q] = load_block (p0, 2, q);
      d[3][q] = load_block (p0, 3, q);
      post[q] = load_block (p0, NB, q);
    }
  }

  while (cur < A.range_end) {
    const uint32_t nxt = uniform (nxt_raw);
    const uint32_t pos0 = cur * TILE + lane * C;  /* first byte of this lane's stream 0 */
    const uint32_t npos0 = nxt * TILE + lane * C; /* the same lane's place in the wave's next tile */
    cur = nxt;
    if (nxt < A.range_end)
      nxt_raw = grab_tile ();
    DIAG (const unsigned long long d_tl = __builtin_readcyclecounter ();)
#pragma unroll
    for (int q = 0; q < S; q++)
      w.s[q] = 0;
    w.sticky = 0;
    DIAG (asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); d_text += __builtin_readcyclecounter () - d_tl; d_tiles++;)
    if (!CONT) {
      /* sticky mode, ownership by END position.  Warm-up: wub 16-byte blocks before each chunk,
       * walked from the root without reporting (matches ending there belong to the previous
       * chunk's owner).  A chunk closer than that to the start of the segment starts from the
       * root at its first in-range block instead. */
      for (uint32_t b = K.wub; b >= 1; b--) {
        uint4 pre[S];
        const uint32_t back = 16u * b;
#pragma unroll
        for (int q = 0; q < S; q++) {
          const uint32_t cs = pos0 + q * (WAVE * C);
          const uint32_t off = cs >= back ? cs - back : 0;
          pre[q] = *reinterpret_cast<const uint4 *> (text + (off < last_block ? off : last_block));
        }
        const StepAt at = { pos0 - back, 0, 0, PH_WARM };
        dense_block<ENTRY, S, COUNT_ONLY> (K, E, emit_from, emit_end, gdense, queue, w, pre, at, lane);
#pragma unroll
        for (int q = 0; q < S; q++)
          if (pos0 + q * (WAVE * C) < back)
            w.s[q] = 0;
      }
    }
    /* walk blocks 0 .. NB-2, refill their registers from the next tile, walk the last block,
     * refill it.  (Refilling half and half re-touched every 128-byte line half a tile later, when
     * part of them had already left L2: 1.37x the text in L2 misses.  Non-temporal loads were
     * 1.55x slower for the same reason.) */
    /* (written out block by block: a loop over k that the compiler declines to unroll would index
     * the text registers dynamically and push them into scratch memory) */
#define ACM_WALK_BLOCK(k)                                                                          \
  {                                                                                                \
    const StepAt at_ = { pos0 + 16 * (k), 0, 0, PH_MAIN };                                         \
    dense_block<ENTRY, S, COUNT_ONLY> (K, E, emit_from, emit_end, gdense, queue, w, d[k], at_, lane); \
  }
#define ACM_REFILL_BLOCK(k)                                                                        \
  _Pragma ("unroll") for (int q = 0; q < S; q++) d[k][q] = load_block (npos0, k, q);
    ACM_WALK_BLOCK (0)
    ACM_WALK_BLOCK (1)
    ACM_WALK_BLOCK (2)
    ACM_REFILL_BLOCK (0)
    ACM_REFILL_BLOCK (1)
    ACM_REFILL_BLOCK (2)
    ACM_WALK_BLOCK (3)
    ACM_REFILL_BLOCK (3)
#undef ACM_WALK_BLOCK
#undef ACM_REFILL_BLOCK
    if (CONT) {
      /* continuation mode, ownership by START position: run over into the following bytes until
       * no lane's state is deeper than the number of bytes past its chunk (at most lmax - 1) */
      bool done = false;
      for (uint32_t b = 0; b < K.wub && !done; b++) {
        if (b > 0) {
#pragma unroll
          for (int q = 0; q < S; q++)
            post[q] = load_block (pos0, NB + b, q);
        }
        region_make_room<CONT, COUNT_ONLY, S> (E, &w.spill);
#define ACM_RUN_BYTE(COMP, SH, J)                                                                  \
  if (!done) {                                                                                     \
    const uint32_t k_ = 16 * b + (J) + 1;                                                          \
    const uint32_t live_from_ = dstart[k_ + 1 <= K.lmax ? k_ + 1 : K.lmax + 1];                    \
    uint32_t b_[S];                                                                                \
    _Pragma ("unroll") for (int q = 0; q < S; q++) b_[q] = (post[q].COMP >> (SH)) & 0xffu;         \
    const StepAt at_ = { pos0 + C + 16 * b + (J), k_, live_from_, PH_RUN };                        \
    dense_step<ENTRY, S, COUNT_ONLY> (K, E, emit_from, emit_end, gdense, queue, w, b_, at_, lane); \
    bool live_ = false;                                                                            \
    _Pragma ("unroll") for (int q = 0; q < S; q++) live_ |= w.s[q] >= live_from_;                  \
    done = k_ + 1 >= K.lmax || __ballot (live_) == 0;                                              \
  }
#define ACM_RUN_WORD(COMP, J)                                                                      \
  ACM_RUN_BYTE (COMP, 0, (J) + 0) ACM_RUN_BYTE (COMP, 8, (J) + 1) ACM_RUN_BYTE (COMP, 16, (J) + 2) ACM_RUN_BYTE (COMP, 24, (J) + 3)
        ACM_RUN_WORD (x, 0) ACM_RUN_WORD (y, 4) ACM_RUN_WORD (z, 8) ACM_RUN_WORD (w, 12)
#undef ACM_RUN_WORD
#undef ACM_RUN_BYTE
      }
#pragma unroll
      for (int q = 0; q < S; q++)
        post[q] = load_block (npos0, NB, q);
    }
  }
  if (w.qn) /* (at most 64 items: the last block's room covers them) */
    queue_drain<CONT, COUNT_ONLY, true> (E, queue, w.qn, &w.spill, lane);
  if (lane == 0 && fill)
    fill[wave] = w.spill.fill;
  DIAG (if (lane == 0 && wave < 8192) {
    unsigned long long *o = g_acm_diag[wave];
    o[0] = __builtin_readcyclecounter () - d_t0;
    o[1] = wall_clock64 ();
    o[7] = d_w0;
    o[2] = w.spill.fill;
    o[3] = w.d_slow_steps;
    o[4] = w.d_slow_cycles;
    o[5] = d_text;
    o[6] = d_tiles;
  })
}
