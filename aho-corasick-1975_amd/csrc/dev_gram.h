/* dev_gram.h -- scan_gram_kernel: 4-gram sieve kernel for byte alphabets with big dictionaries.
 * Device code of libac75_amd.so; included by acm_gpu.hip inside its anonymous namespace (one
 * translation unit: the kernels share the structs and helpers declared there and in dev_emit.h). */

/* ------------------------------------------------------------------ 4-gram sieve kernel (byte alphabets, big dictionaries)
 * Dictionaries whose automaton does not fit the LDS scheme of the dense kernel (more than 32,768
 * states: config 3 has 508,339) make every step of a carried-state walk a dependent gather into
 * tens of megabytes of rows (99 GB/s).  When the alphabet is small (width W = span + 1 <= 30)
 * the start-parallel idea works for bytes too.  Keywords of 4 symbols or more:
 *   1. LDS holds one bit per possible 4-gram over the W classes (W^4 bits: 66 KB for a-z): "some
 *      keyword starts with it".  Every position is tested with one ds_read_b32 on a rolling
 *      4-gram index (config 3: 19.6% pass);
 *   2. the survivors are queued per wave as (position, 4-gram index, class of the 5th symbol)
 *      and leave the queue 64 at a time: two Bloom filters in LDS say which of them can have
 *      anything to report (GramK::bloom5_bits), those get the depth-4 state's {terminal bit | mask
 *      of its children} by rank (GramK::g4prefix) in two pipelined gathers; two batches are in
 *      flight behind the scan (config 3: 3.2% of the positions pass -- all real: a keyword of
 *      length 4 ends there, reported on the spot, or a 5-symbol prefix of a keyword does);
 *   3. the latter (0.75%) go to walk_starts (shared with the start-parallel kernel) at the
 *      depth-4 state; the trie records below depth 4 are laid out depth-first.
 * Keywords of 1-3 symbols (template SHORTS, only if the dictionary has any): a nibble per 3-gram
 * in LDS (bit d-1: the first d symbols are a keyword) looked up on a second rolling index, a third
 * queue, and per 3-gram the states of its three prefixes in HBM for the records.
 * Alphabets of more than 29 symbols (template WIDE): the raw 4-byte
 * window replaces the base-W index -- a multiplicative hash into Bloom bits in LDS for stage 1, an
 * open-addressing table {window, depth-4 state | flags} in HBM for stage 2.
 * Text is read as in the start-parallel kernel (1 KiB groups, 16 bytes per lane, four groups in
 * flight). */
struct GramK {
  const uint2 *g4rec;     /* [W^4] {children mask | terminal << 31, state id of the depth-4 node (0: none)} (sieve kernel only) */
  const uint32_t *g4bits; /* [g4words] one bit per 4-gram, staged in LDS */
  const uint4 *srec;      /* trie records of the states of depth >= 4, depth-first order (see StartsK::remap) */
  const uint2 *sedge;
  const uint32_t *g4gid;  /* [states of depth 4] record index of each depth-4 state */
  uint32_t d4_begin;      /* breadth-first id of the first depth-4 state */
  uint32_t g4words;
  /* dictionaries with keywords of 1-3 symbols: per 3-gram a nibble in LDS (bit d-1: the first d
   * symbols are a keyword) and a record {state of the 1-, 2-, 3-symbol prefix, -} in HBM */
  const uint4 *g3rec;
  uint32_t g3_off, g3_bytes; /* nibble table in LDS, right after the 4-gram bits; 0 bytes: no short keywords */
  /* scan_short_kernel's image (dev_short.h): the nibbles, per 8 3-grams the number of set nibble
   * bits in front of them, the keyword ids in the order of those bits */
  const uint32_t *sh_img;
  uint32_t sh_nib_bytes, sh_base_bytes, sh_ids_bytes;
  /* wide alphabets (template WIDE): g4bits is a Bloom filter of 2^bloom_log2 bits on the hashed
   * 4-byte window, wtab an open-addressing table {window, depth-4 state | has children << 30 |
   * terminal << 31} of 2^wtab_log2 slots (state 0: empty slot) */
  const uint2 *wtab;
  uint32_t bloom_log2, wtab_log2;
  /* wide alphabets with keywords of 1-3 symbols: their windows, keyed (bytes | length << 24), share
   * the Bloom bits and have their own table {key, state}; short_lens bit d-1: some keyword has d symbols */
  const uint2 *stab;
  uint32_t stab_log2, short_lens;
  uint32_t W, lo, span, W4; /* class = min (byte - lo, span); W = span + 1 */
  /* narrow alphabets: two Bloom filters in LDS behind the 4-gram bits, looked at when a batch of
   * first-queue items sends for its records: "this 4-gram is a keyword" (key = the 4-gram index)
   * and "these 5 symbols start a keyword" (key = index * W + class of the 5th symbol).  An item
   * that is in neither needs no record: 4 of 5 on config 3 (19.6 % of the positions pass the
   * 4-gram bits, 3.2 % have something to report or to walk on from), and the record gather --
   * 64 lanes, 64 lines, two thirds of them L2 misses on a 4 MB table -- is what the kernel waits
   * for.  bloom5_bits == 0: no filters. */
  uint32_t bloom_off, bloomT_bits, bloom5_bits;
  /* narrow alphabets: the second stage's records by RANK.  The depth-4 states are numbered in the
   * order of their 4-gram index (breadth-first ids, children in symbol order), so the state of an
   * existing 4-gram is d4_begin + its rank among the set bits of g4bits:
   *   rank = g4prefix[index >> 5] + popcount (bits of its word below it),
   * and g4entry[rank] = children mask | terminal << 31 is all the second stage needs: 66 KB + 4
   * bytes per existing 4-gram (config 3: 360 KB) that stay in L2, where the 8-byte records by
   * index (W^4 of them, 4.25 MB for a-z) did not fit beside the text streaming through: two of
   * three record gathers went to memory for a 128-byte line each, 5.5 x the algorithmic bytes. */
  const uint32_t *g4prefix;
  const uint32_t *g4entry; /* by rank, 3 words: {children mask | terminal << 31, state id of the first child, keyword id} */
  uint32_t kw_inline;     /* wide alphabets: keyword ids fit a hit's word beside HIT_KW (always, short of 2^28 keywords) */
  /* the walks start one level down, at the depth-5 state the 5th symbol leads to (children are
   * numbered consecutively in symbol order: first child + set mask bits below the class), and ask
   * g5peek[that state - d5_begin] = {its record, the symbol of its only edge | GRAM_NO_PEEK}
   * before any record: 708 KB that stay in L2 */
  const uint32_t *g5peek; /* 4 bytes per state when peek_packed (record | symbol << 23 | GRAM_NO_PEEK's bit 31), else 8 */
  uint32_t d5_begin, peek_packed;
  uint32_t d5_rel;        /* fewer than 2^24 depth-5 states: walk items name them by index and carry the 6th symbol's class */
  uint32_t R;             /* groups per tile */
  uint32_t queue_off;     /* LDS: [g4 bits][16 x first queue][16 x second queue][16 x hit buffer][tile counter] */
  /* scan_gram2_kernel (dev_gram2.h): two bits per 4-gram, staged in LDS -- bit 2i: "a keyword starts
   * with 4-gram i" (the same as g4bits), bit 2i + 1: "4-gram i is a keyword, or the tail (symbols
   * 2-5) of some keyword's first five"; g2_off: LDS offset of the waves' areas behind the table.
   * tab2 == NULL: the plan does not use that kernel.
   * rows2 (scan_gram2_kernel): what the second stage needs of an existing
   * 4-gram, ONE 64-byte line per 16 4-grams: slot r < 7 = the entry of the r-th 4-gram of the 16
   * that exists, {children mask | "goes on as well" << 30 | keyword << 31, x} with x = the
   * keyword's id (a keyword that does not go on), the first child's state id (no keyword), or the
   * record of the depth-4 state itself (a keyword of 4 symbols that other keywords go on from: 2 %
   * of config 3's 4-grams -- such a position becomes a walk item AT that record, which reports the
   * keyword and goes on: nothing else has to be fetched for it); slot 7 = the 8th entry, or, when there
   * are nine or more, {1 << 29 | index into over2 of the 8th entry, -} (one line in 600 of config
   * 3's: the dependent gather it costs is exposed, so it must be rare per BATCH).  One gather per survivor
   * instead of two dependent ones (prefix count, then entry by rank: scan_gram_kernel keeps those --
   * with its Bloom filters in front, 5.5 % of the positions gather instead of 7.4 %, and the rows
   * measured SLOWER there: count-only 1.87 -> 2.33 ms per 2 GiB of config 3; the 2.1 MB of rows miss
   * L2 more often than the 66 KB + 1 MB of the tables by rank). */
  const uint32_t *tab2;
  const uint2 *rows2, *over2;
  uint32_t tab2_words, g2_off;
};

constexpr uint32_t WIDE_H1 = 0x9E3779B1u, WIDE_H2 = 0x85EBCA6Bu; /* multiplicative hashes: Bloom bits, table slots */
/* The two bit positions of a key in a Bloom filter of m bits (m a multiple of 32, below 2^21;
 * host and device alike), in full-rate 24-bit multiplies only (v_mul_lo_u32 / v_mul_hi_u32 hold
 * the SIMD four times as long, and a batch of the kernel's second stage asks for four positions):
 * h = key x odd constant (24 x 24 bits, low word); its low 24 bits scaled to the filter's words
 * pick the word, its low 5 bits the bit. */
constexpr uint32_t BLOOM_C1 = 0x9E3779u, BLOOM_C2 = 0x85EBCBu, BLOOM_D1 = 0xC2B2AFu, BLOOM_D2 = 0x27D4EBu;
/* key = a 4-gram index; with a 5th symbol's class c5 (5-gram filter) the same product plus c5 x constant */
__host__ __device__ __forceinline__ uint32_t
gram_bloom_hash (uint32_t idx, int which) {
  return (idx & 0xFFFFFFu) * (which ? BLOOM_C2 : BLOOM_C1);
}
__host__ __device__ __forceinline__ uint32_t
gram_bloom_hash5 (uint32_t idx, uint32_t c5, int which) {
  return gram_bloom_hash (idx, which) + c5 * (which ? BLOOM_D2 : BLOOM_D1);
}
/* Both bits of a key lie in ONE word (a blocked filter: one LDS read per key instead of two, and
 * half the address arithmetic; the false-positive rate of a filter this full -- 5 to 11 of a word's
 * 32 bits set -- is the same within a tenth): the word from the hash's high bits scaled to the
 * filter's words, the bits from its bits 0-4 and 5-9 (`second`). */
__host__ __forceinline__ uint32_t
gram_bloom_slot (uint32_t h, uint32_t m, int second) {
  const uint32_t word = (uint32_t)(((uint64_t)(h & 0xFFFFFFu) * ((m / 32) << 8)) >> 32);
  return word * 32 + ((second ? h >> 5 : h) & 31u);
}
__device__ __forceinline__ uint32_t
mul_hi_u24 (uint32_t a, uint32_t b) { /* bits 32..47 of the product of the operands' low 24 bits */
  uint32_t r;
  asm ("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
  return r;
}
__device__ __forceinline__ uint32_t
mad_u24 (uint32_t a, uint32_t b, uint32_t c) { /* (the compiler turns the C expression into a 64-bit mad when it likes) */
  uint32_t r;
  asm ("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}
constexpr uint32_t WT_TERM = 0x80000000u, WT_KIDS = 0x40000000u;
/* first-queue item of the narrow kernel, second word: 4-gram index (20 bits) | class of the 5th
 * symbol << 20 | class of the 6th << 25, GRAM_CLS_UNKNOWN where it is not at hand (a lane holds the
 * classes of its 16 symbols and of the next 4: the 6th symbol of its last position is beyond them;
 * classes are below 30) */
constexpr uint32_t GRAM_CLS_UNKNOWN = 31;
__device__ __forceinline__ uint32_t
gram_item_word (const uint32_t (&c)[20], int j, uint32_t idx) {
  const uint32_t y = lshl_or (c[j + 4], 20u, idx);
  return lshl_or (j + 5 < 20 ? c[j + 5] : GRAM_CLS_UNKNOWN, 25u, y);
}

template <bool COUNT_ONLY, bool SHORTS, bool WIDE, bool TILED>
__global__ __launch_bounds__ (SPARSE_THREADS) void
scan_gram_kernel (GramK K, EmitCtx E, Launch A, const unsigned char *__restrict__ text, uint2 *items, uint32_t region_items,
                  uint32_t *fill, RecHole *holes, uint32_t resume, TileEntry *dir, uint32_t dir_base) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  constexpr uint32_t WAVES = SPARSE_THREADS / WAVE;
  constexpr uint32_t GROUP = WAVE * 16;
  {
    uint4 *dst = reinterpret_cast<uint4 *> (smem);
    const uint4 *src = reinterpret_cast<const uint4 *> (K.g4bits);
    const uint32_t image = K.bloom5_bits ? K.bloom_off + (K.bloomT_bits + K.bloom5_bits + 7) / 8 : K.g3_off + K.g3_bytes;
    for (uint32_t i = threadIdx.x; i < (image + 15) / 16; i += blockDim.x)
      dst[i] = src[i]; /* 4-gram bits, then (g3_off) the nibbles of the short keywords, then (bloom_off) the Bloom filters */
  }
  constexpr uint32_t NQ = SHORTS ? 3 : 2; /* queues per wave */
  /* LDS queues per wave, 8-byte items: first queue (QCAP: up to 63 waiting + 64 from one position),
   * walk queue (GRAM_Q2: up to 63 waiting + what one batch sends on, with room made first when
   * that is more than 33), short-keyword queue (QCAP, SHORTS only), hit buffer (wide alphabets) */
  constexpr bool DIRECT = !WIDE; /* narrow alphabets: 16-byte records straight into the caller's buffer (dev_starts.h: WaveRec), no hits */
  constexpr uint32_t HB = DIRECT ? 0u : HITS_STRIDE;
  uint32_t *next_tile = reinterpret_cast<uint32_t *> (smem + K.queue_off + WAVES * (GRAM_Q1 + GRAM_Q2 + (NQ - 2) * QCAP + HB) * 8);
  StartsK *Ks = reinterpret_cast<StartsK *> (next_tile + 4); /* see scan_starts_kernel */
  EmitCtx *Es = reinterpret_cast<EmitCtx *> (reinterpret_cast<unsigned char *> (Ks) + WALK_CTX_K);
  WaveRec *Ws = reinterpret_cast<WaveRec *> (reinterpret_cast<unsigned char *> (Ks) + WALK_CTX_K + WALK_CTX_E); /* one per wave */
  if (threadIdx.x == 0) {
    *next_tile = 0;
    StartsK Kc{};
    Kc.srec = K.srec;
    Kc.sedge = K.sedge;
    Kc.remap = K.g4gid;
    Kc.remap_base = WIDE ? K.d4_begin : K.d5_begin;
    Kc.peek = K.g5peek;
    Kc.peek_packed = K.peek_packed;
    Kc.peek_rel = K.d5_rel;
    Kc.lo = K.lo;
    Kc.span = K.span;
    *Ks = Kc;
    *Es = E;
  }
  __syncthreads ();

  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t wib = uniform (threadIdx.x / WAVE);
  uint2 *q1 = reinterpret_cast<uint2 *> (smem + K.queue_off) + wib * GRAM_Q1;
  uint2 *q2 = reinterpret_cast<uint2 *> (smem + K.queue_off) + WAVES * GRAM_Q1 + wib * GRAM_Q2;
  uint2 *q3 = reinterpret_cast<uint2 *> (smem + K.queue_off) + WAVES * (GRAM_Q1 + GRAM_Q2) + wib * QCAP; /* SHORTS only */
  const uint32_t wave_id = blockIdx.x * WAVES + wib;
  uint2 *hits = DIRECT ? reinterpret_cast<uint2 *> (Ws + wib)
                       : reinterpret_cast<uint2 *> (smem + K.queue_off) + WAVES * (GRAM_Q1 + GRAM_Q2 + (NQ - 2) * QCAP) + wib * HITS_STRIDE + 2;
  if (!DIRECT)
    hits_init (hits, (!COUNT_ONLY && items) ? items + (size_t)wave_id * region_items : nullptr, region_items, lane);
  else if (lane == 0) {
    /* no chunk yet (limit 0 sends the first batch through emit_records_slow) -- or, in a later
     * segment of the same scan (`resume`), the chunk this wave was filling when the segment before
     * ended: what it had left of it is in its hole descriptor.  The holes of a scan are closed once,
     * behind its last segment (seven fewer close_holes_kernel launches per 16 GiB). */
    WaveRec w0{};
    if (resume && !COUNT_ONLY && holes) {
      const RecHole h = holes[wave_id];
      if (h.len) {
        const unsigned long long at = ((unsigned long long)h.start_hi << 32) | h.start_lo; /* first free slot */
        const unsigned long long base = at + h.len - E.rec_chunk;
        const bool below = base + E.rec_chunk <= E.capacity;
        const bool above = base >= E.capacity && base - E.capacity + E.rec_chunk <= E.spill_slots;
        const uint64_t dst = below ? reinterpret_cast<uint64_t> (&E.records[base]) : (above ? reinterpret_cast<uint64_t> (E.spill + (base - E.capacity)) : 0ull);
        w0.dst_lo = (uint32_t)dst;
        w0.dst_hi = (uint32_t)(dst >> 32);
        w0.base_lo = (uint32_t)base;
        w0.base_hi = (uint32_t)(base >> 32);
        w0.limit = (below || above) ? E.rec_chunk : 0u;
        w0.have = 1;
        w0.pad[0] = E.rec_chunk - h.len; /* slots used */
        w0.pad[1] = 1;
        w0.prev1 = w0.prev2 = NONE;
      }
    }
    if (!w0.have)
      w0.prev1 = w0.prev2 = NONE;
    Ws[wib] = w0;
  }
  const uint4 *text16 = reinterpret_cast<const uint4 *> (text);
  /* block b takes the tiles b, b + gridDim.x, ... (match density is rarely even along a text:
   * contiguous shares left two blocks of config 3 working 1 ms after all others had finished),
   * handed to its waves through the LDS counter */
  const TileShare share (A);
  const uint32_t last_blk = (A.n - 1) / 16;
  uint32_t qn1 = 0, qn2 = 0, qn3 = 0;
  unsigned long long counted = 0;
  RecState rs = { 0ull, 0u }; /* DIRECT: the wave's chunk of records (none yet, or the one a resumed scan left) */
  if (DIRECT && !COUNT_ONLY && resume) {
    rs = rec_state_load (hits);
    counted = uniform (reinterpret_cast<const WaveRec *> (hits)->pad[0]);
  }
  /* batches of the first queue whose records are in flight: a pipeline of GRAM_DEPTH batches
   * (the gather of a batch has the time it takes the scan to fill that many more before it is
   * looked at: one batch ahead left the L2 / MALL latency exposed).  Narrow alphabets: the record
   * comes in two dependent lookups, one per pipeline step -- slot 2 holds {prefix count asked for,
   * rank inside the word | NEED}, slots 1 and 0 {entry, state}.  A value is only ever copied one
   * step AFTER its load was issued: a copy in the same step (the load landing in a temporary and
   * moved to its slot at once) is an s_waitcnt vmcnt(0) per batch, latency of the gather and of
   * the prefetched text included. */
  constexpr int GRAM_DEPTH = 3;
  DIAG (const unsigned long long d_t0 = __builtin_readcyclecounter (); unsigned long long d_walk = 0, d_calls = 0, d_items = 0, d_b1 = 0, d_cons = 0, d_tiles = 0;)
  uint2 pend_item[GRAM_DEPTH];
  uint32_t pend_rx[GRAM_DEPTH], pend_ry[GRAM_DEPTH], pend_rz[1] = { 0 }, pend_rw[1] = { 0 };
  uint3 pend_e1 = make_uint3 (0, 0, 0);
  uint2 pend_w2 = make_uint2 (0, 0); /* wide alphabets: the newest batch's table slots */ /* the record's words, apart: a 64-bit register pair half in flight pins both (rz: narrow alphabets, the first child) */
  uint32_t pend_n[GRAM_DEPTH]; /* wave-uniform: items in each pending batch, [0] the oldest */
  constexpr uint32_t PEND_NEED = 0x80000000u;
#pragma unroll
  for (int d = 0; d < GRAM_DEPTH; d++) {
    pend_item[d] = make_uint2 (0, 0);
    pend_rx[d] = pend_ry[d] = 0;
    pend_n[d] = 0;
  }

  auto load_group = [&] (uint32_t g) -> uint4 {
    const uint32_t blk = g * WAVE + lane;
#ifdef ACM_GRAM_NT_TEXT /* experiment: the text as a stream that should not push the second stage's tables out of L2 */
    typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
    const u32x4 v = __builtin_nontemporal_load (reinterpret_cast<const u32x4 *> (text16) + (blk < last_blk ? blk : last_blk));
    return make_uint4 (v.x, v.y, v.z, v.w);
#else
    return text16[blk < last_blk ? blk : last_blk];
#endif
  };
  auto walk_batch = [&] (uint32_t n_items) {
    DIAG (const unsigned long long d_c0 = __builtin_readcyclecounter ();)
    const unsigned long long r = walk_starts<uint8_t, COUNT_ONLY, WIDE ? 1 : 2> (Ks, Es, text, q2, qn2, n_items, hits, counted);
    qn2 = uniform ((uint32_t)(r >> 32));
    counted = (uint32_t)r;
    if (DIRECT && !COUNT_ONLY)
      rs = rec_state_load (hits); /* (the walk may have gone on to the next chunk) */
    DIAG (d_walk += __builtin_readcyclecounter () - d_c0; d_calls++; d_items += n_items;)
  };
  /* Second sieve on the oldest pending batch, in three parts so that a pipeline step can put the
   * LDS round trips of the NEW batch (its items, then its filter bits) behind them: a keyword of
   * 4 symbols ends here; the 5th symbol is an edge of the depth-4 state; the pipeline moves up. */
  bool st_term = false;
  uint32_t st_pos = 0, st_what = 0;
  auto consume_terminal = [&] () {
#if defined(ACM_GRAM_ABLATE) && ACM_GRAM_ABLATE == 3 /* experiment: the records are gathered and only looked at */
    st_term = false;
    if (pend_n[0])
      asm volatile ("" :: "v"(pend_rx[0]), "v"(pend_rw[0]));
    return;
#endif
    if (pend_n[0]) {
      if (WIDE) {
        /* the slot that came back is the first of the probe sequence: a hit, an empty slot (the
         * Bloom bit was a false positive), or somebody else's window -- then probe on (rare) */
        uint2 e = make_uint2 (pend_rx[0], pend_ry[0]);
        uint32_t slot = (pend_item[0].y * WIDE_H2) >> (32 - K.wtab_log2);
        while (lane < pend_n[0] && (e.y & ST_STATE) != 0 && e.x != pend_item[0].y) {
          slot = (slot + 1) & ((1u << K.wtab_log2) - 1);
          e = K.wtab[slot];
        }
        /* same shape as the exact table's record: terminal bit 31, "goes on" as bit 0 of the mask */
        pend_rx[0] = (e.y & ST_STATE) != 0 ? (e.y & WT_TERM) | ((e.y & WT_KIDS) ? 1u : 0u) : 0u;
        pend_ry[0] = e.y & ST_STATE;
      }
      /* a keyword of length 4 ends here: reported at once, its record needs nothing but the
       * state id (the trie records of the 508,339 states are 16 MB of HBM: not worth a visit) */
      const bool term = lane < pend_n[0] && (pend_rx[0] >> 31) && pend_item[0].x + 3 >= E.emit_from;
      /* (narrow alphabets: the entry brought the keyword's id along, and the hit carries it --
       * expand_hits_kernel spent 0.2 of its 0.52 ms per 2 GiB of config 3 on the gather by rank) */
      st_term = term;
      st_pos = pend_item[0].x + 3;
      st_what = DIRECT ? pend_rw[0] : (pend_ry[0] - K.d4_begin) | HIT_LEN4; /* narrow: the keyword's id itself (the entry brought it) */
    } else
      st_term = false;
  };
  /* (reported apart from the test: a pipeline step of the narrow kernel puts the store behind
   * its gathers, so that the step's own wait for the older gathers does not wait for it) */
  auto emit_stashed = [&] () {
    emit_terminals<COUNT_ONLY, DIRECT> (E, st_term, st_pos, st_what, 4u, lane, hits, counted, Es, DIRECT ? &rs : nullptr);
    if (!COUNT_ONLY)
      counted = uniform ((uint32_t)counted);
    st_term = false;
  };
  auto consume_pass = [&] () {
#if defined(ACM_GRAM_ABLATE) && ACM_GRAM_ABLATE == 3
    return;
#endif
    if (pend_n[0]) {
      const uint32_t c4 = WIDE ? 0u : (pend_item[0].y >> 20) & 31u;
      const bool pass = lane < pend_n[0] && ((pend_rx[0] >> (WIDE ? 0u : c4)) & 1u);
      const uint64_t m = __ballot (pass);
      if (m) {
        /* (a walk call handles one level of the newest items and never leaves more than it took,
         * and every item ends within lmax levels) */
        while (qn2 + (uint32_t)__popcll (m) > GRAM_Q2)
          walk_batch (qn2 < WAVE ? qn2 : WAVE);
        if (pass) {
          if (WIDE)
            q2[qn2 + rank_below (m)] = make_uint2 (pend_item[0].x + 3, pend_ry[0] | WI_REPORTED);
          else { /* the depth-5 state, at the 5th symbol */
            const uint32_t st5 = pend_rz[0] + __popc (pend_rx[0] & ((1u << c4) - 1u));
            /* (with the class of the 6th symbol when the first queue's item brought it: the walk's
             * first look at a candidate -- its peek entry against the next symbol, where 25 of 26
             * end -- then needs no text: re-reading a byte of text that has long left L2 was a
             * 128-byte line from the Infinity Cache per candidate, 2 to 3 GB per 2 GiB launch) */
            q2[qn2 + rank_below (m)] = make_uint2 (pend_item[0].x + 4, K.d5_rel ? (st5 - K.d5_begin) | ((pend_item[0].y >> 25) & 31u) << 24 : st5);
          }
        }
        qn2 = uniform (qn2 + (uint32_t)__popcll (m));
#if defined(ACM_GRAM_ABLATE) && ACM_GRAM_ABLATE == 4 /* experiment: the walk candidates are queued and dropped */
        if (qn2 >= WAVE)
          qn2 -= WAVE;
#endif
        while (qn2 >= WAVE)
          walk_batch (WAVE);
      }
    }
  };
  auto pipeline_shift = [&] () {
    pend_item[0] = pend_item[1];
    pend_ry[0] = pend_ry[1];
    pend_n[0] = pend_n[1];
    pend_item[1] = pend_item[2];
    pend_n[1] = pend_n[2];
    pend_n[2] = 0;
    if (WIDE) {
      /* (the newest batch's table slots came in one 8-byte gather: a register pair that stays
       * where it landed until its words are copied out here, one step later) */
      pend_rx[0] = pend_rx[1];
      pend_rx[1] = pend_w2.x;
      pend_ry[1] = pend_w2.y;
      asm volatile ("" : "+v"(pend_rx[0]), "+v"(pend_ry[0]), "+v"(pend_rx[1]), "+v"(pend_ry[1]), "+v"(pend_item[0].x), "+v"(pend_item[0].y),
                    "+v"(pend_item[1].x), "+v"(pend_item[1].y));
      __builtin_amdgcn_sched_barrier (0);
    } else {
      /* the middle batch's entries came in one 16-byte gather (a register tuple: it stays where
       * it landed until its words are copied out here, one step later) */
      pend_rx[0] = pend_e1.x;
      pend_rz[0] = pend_e1.y;
      pend_rw[0] = pend_e1.z;
      /* the batch that has just left the newest slot: its prefix counts are here, now the entries
       * (issued after the copies above, straight into the tuple they will be read from) */
      asm volatile ("" : "+v"(pend_rx[0]), "+v"(pend_ry[0]), "+v"(pend_rz[0]), "+v"(pend_rw[0]), "+v"(pend_item[0].x), "+v"(pend_item[0].y),
                    "+v"(pend_item[1].x), "+v"(pend_item[1].y));
      __builtin_amdgcn_sched_barrier (0);
      const uint32_t rank = pend_rx[2] + (pend_ry[2] & ~PEND_NEED);
      uint3 ent = make_uint3 (0, 0, 0);
      if (pend_n[1] && (pend_ry[2] & PEND_NEED))
        ent = *reinterpret_cast<const uint3 *> (K.g4entry + 3 * (size_t)rank);
      pend_e1 = ent;
      pend_ry[1] = K.d4_begin + rank;
    }
  };
  auto consume_oldest = [&] () {
    consume_terminal ();
    emit_stashed ();
    consume_pass ();
    pipeline_shift ();
  };
  /* One pipeline step: the oldest batch is consumed, the newest n items of the first queue take
   * the slot that frees up and send for their records.  Narrow alphabets: the new batch's two LDS
   * round trips (items; then the words of the two Bloom filters and of the 4-gram bits) are put
   * behind the two halves of the consumption -- four waves per SIMD do not hide them. */
  auto batch_step = [&] (uint32_t n_items) {
    qn1 -= n_items;
    const uint2 it = lane < n_items ? q1[qn1 + lane] : make_uint2 (0, 0);
    if constexpr (WIDE) {
      consume_oldest ();
      pend_item[GRAM_DEPTH - 1] = it;
      pend_w2 = K.wtab[(it.y * WIDE_H2) >> (32 - K.wtab_log2)];
    } else {
#if defined(ACM_GRAM_ABLATE) && ACM_GRAM_ABLATE == 1 /* experiment: the items are taken off the queue and dropped */
      asm volatile ("" :: "v"(it.x), "v"(it.y));
      return;
#endif
      consume_terminal ();
      const uint32_t idx = it.y & 0xFFFFFu, c5 = (it.y >> 20) & 31u;
      /* the words of the two filters (word scale and byte offset: wave-uniform, in SGPRs; without
       * filters scale 0 reads some word and the answer is ignored -- no branch here: a value
       * loaded under one must be looked at before its block ends) and of the 4-gram bits */
      const uint32_t scaleT = (K.bloomT_bits / 32) << 8, scale5 = (K.bloom5_bits / 32) << 8;
      const uint32_t offT = K.bloom_off, off5 = K.bloom_off + K.bloomT_bits / 8;
      auto lds_word = [&] (uint32_t byte_off) -> uint32_t {
        return *reinterpret_cast<const __attribute__ ((address_space (3))) uint32_t *> (byte_off);
      };
      /* one word per filter: both bits of a key lie in it (gram_bloom_slot) */
      const uint32_t h1 = __umul24 (idx, BLOOM_C1);
      const uint32_t h3 = mad_u24 (c5, BLOOM_D1, h1);
      const uint32_t w1 = lds_word (offT + mul_hi_u24 (h1, scaleT) * 4u);
      const uint32_t w3 = lds_word (off5 + mul_hi_u24 (h3, scale5) * 4u);
      /* the 4-gram bits' word: the plain kernel asks for it BEHIND the walks consume_pass may run
       * (asked for beside the filters' words it was one register too many across them -- spilled,
       * with a wait for all three LDS reads in front of the spill: 1.5 % of config 3); the tiled
       * instantiation, whose registers fall differently, is 13 % slower that way and keeps it here */
      uint32_t word = 0;
      if constexpr (TILED)
        word = lds_word ((idx >> 5) * 4u);
      consume_pass ();
      pipeline_shift ();
      if constexpr (!TILED)
        word = lds_word ((idx >> 5) * 4u);
      pend_item[GRAM_DEPTH - 1] = it;
      const uint32_t tf = ((w1 >> (h1 & 31u)) & (w1 >> ((h1 >> 5) & 31u))) | ((w3 >> (h3 & 31u)) & (w3 >> ((h3 >> 5) & 31u))) | (K.bloom5_bits ? 0u : 1u);
      /* (a lane that needs nothing asks for nothing: a gather costs by the line) */
#if defined(ACM_GRAM_ABLATE) && ACM_GRAM_ABLATE == 2 /* experiment: the filters are asked, nothing is gathered */
      const bool need = lane < n_items && (tf & 1u) != 0 && w1 == 0x12345u;
#else
      const bool need = lane < n_items && (tf & 1u) != 0;
#endif
      uint32_t pre = 0;
      if (need)
        pre = K.g4prefix[idx >> 5];
      pend_rx[GRAM_DEPTH - 1] = pre;
      pend_ry[GRAM_DEPTH - 1] = __popc (word & ((1u << (idx & 31u)) - 1u)) | (need ? PEND_NEED : 0u);
      emit_stashed ();
    }
    pend_n[GRAM_DEPTH - 1] = n_items;
  };

  /* the newest n items of the short-keyword queue: item = (position, 3-gram index | nibble << 20);
   * the keywords of 1, 2, 3 symbols that start there end at position, +1, +2 */
  auto short_batch = [&] (uint32_t n_items) {
    qn3 -= n_items;
    const uint2 it = lane < n_items ? q3[qn3 + lane] : make_uint2 (0, 0);
    if constexpr (WIDE) {
      /* item = (position, bytes | length << 24): look the key up (the Bloom bit may have lied) */
      bool valid = lane < n_items;
      uint32_t slot = (it.y * WIDE_H2) >> (32 - K.stab_log2);
      uint2 e = K.stab[slot];
      while (valid && e.y != 0 && e.x != it.y) {
        slot = (slot + 1) & ((1u << K.stab_log2) - 1);
        e = K.stab[slot];
      }
      const uint32_t end = it.x + (it.y >> 24) - 1;
      emit_terminals<COUNT_ONLY, DIRECT> (E, valid && e.y != 0 && end >= E.emit_from, end, e.y, 0u, lane, hits, counted, Es, DIRECT ? &rs : nullptr);
      if (!COUNT_ONLY)
        counted = uniform ((uint32_t)counted);
      return;
    }
    const uint32_t nib = it.y >> 20;
    uint4 rec = make_uint4 (0, 0, 0, 0);
    if (!COUNT_ONLY)
      rec = K.g3rec[it.y & 0xFFFFFu];
#pragma unroll
    for (uint32_t d = 0; d < 3; d++) {
      const bool hit = ((nib >> d) & 1u) && it.x + d >= E.emit_from;
      emit_terminals<COUNT_ONLY, DIRECT> (E, hit, it.x + d, d == 0 ? rec.x : (d == 1 ? rec.y : rec.z), d + 1, lane, hits, counted, Es, DIRECT ? &rs : nullptr);
      if (!COUNT_ONLY)
        counted = uniform ((uint32_t)counted);
    }
  };

  /* one group: cur = this lane's 16 bytes, next_x = the first 4 bytes of every lane of the next group */
  auto walk_group = [&] (const uint4 cur, const uint32_t next_x, const uint32_t g, uint4 &prefetched) {
    uint32_t after = __shfl_down (cur.x, 1, WAVE);
    const uint32_t after_group = uniform (next_x);
    if (lane == WAVE - 1)
      after = after_group;
    /* the group four ahead is asked for HERE, after the first look at this one: the compiler
     * cannot count the loads in flight across the pipeline steps' gathers and waits for all of
     * them (vmcnt(0)) wherever it waits -- asked for before that first look, the prefetch was
     * waited for on the spot, one HBM latency per group and wave */
    asm volatile ("" : "+v"(after));
    __builtin_amdgcn_sched_barrier (0);
    prefetched = load_group (g + 4);
    const uint32_t pos0 = g * GROUP + lane * 16;
    const uint32_t w[5] = { cur.x, cur.y, cur.z, cur.w, after };
    if constexpr (WIDE) {
      /* the 4-byte window at every position (one v_alignbit each), hashed into the Bloom bits;
       * windows that reach past the end of the segment are no starts (last groups only) */
      const bool tail = pos0 + 20 > A.n;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        uint32_t win[8], word[8], hb[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int jj = 8 * h + j;
          win[j] = jj % 4 ? __builtin_amdgcn_alignbit (w[jj / 4 + 1], w[jj / 4], 8 * (jj % 4)) : w[jj / 4];
          hb[j] = (win[j] * WIDE_H1) >> (32 - K.bloom_log2);
          word[j] = *reinterpret_cast<const __attribute__ ((address_space (3))) uint32_t *> ((hb[j] >> 5) * 4u);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint32_t p = pos0 + 8 * h + j;
          const bool push = ((word[j] >> (hb[j] & 31u)) & 1u) && (!tail || p + 3 < A.n);
          const uint64_t m = __ballot (push);
          if (m) {
            if (push)
              q1[qn1 + rank_below (m)] = make_uint2 (p, win[j]);
            qn1 = uniform (qn1 + (uint32_t)__popcll (m));
            if (__builtin_expect (qn1 >= WAVE, 0))
              batch_step (WAVE);
          }
        }
        if (SHORTS) {
          /* keywords of d = 1, 2, 3 symbols (only the lengths the dictionary has): the first d
           * bytes of the window, tagged with d, through the same Bloom bits */
#pragma unroll
          for (uint32_t d = 1; d <= 3; d++) {
            if (!((K.short_lens >> (d - 1)) & 1u))
              continue;
            uint32_t key[8], sw[8], sh[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
              key[j] = (win[j] & ((1u << (8 * d)) - 1u)) | (d << 24);
              sh[j] = (key[j] * WIDE_H1) >> (32 - K.bloom_log2);
              sw[j] = *reinterpret_cast<const __attribute__ ((address_space (3))) uint32_t *> ((sh[j] >> 5) * 4u);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const uint32_t p = pos0 + 8 * h + j;
              const bool push = ((sw[j] >> (sh[j] & 31u)) & 1u) && (!tail || p + d - 1 < A.n);
              const uint64_t m = __ballot (push);
              if (m) {
                if (push)
                  q3[qn3 + rank_below (m)] = make_uint2 (p, key[j]);
                qn3 = uniform (qn3 + (uint32_t)__popcll (m));
                if (qn3 >= WAVE)
                  short_batch (WAVE);
              }
            }
          }
        }
      }
      return;
    }
#if defined(ACM_GRAM_ABLATE) && ACM_GRAM_ABLATE == 7 /* experiment: the text is streamed and not looked at */
    asm volatile ("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]));
    return;
#endif
    uint32_t c[20];
#pragma unroll
    for (int j = 0; j < 20; j++) {
      const uint32_t b = (w[j / 4] >> (8 * (j % 4))) & 0xFFu;
      c[j] = min (b - K.lo, K.span);
    }
    /* symbols past the end of the segment count as outside the alphabet (only the last groups) */
    if (pos0 + 20 > A.n) {
#pragma unroll
      for (int j = 0; j < 20; j++)
        if (pos0 + j >= A.n)
          c[j] = K.span;
    }
    /* (24-bit multiplies: everything here is below 2^24, and v_mul_u32_u24 / v_mad_u32_u24 run at
     * full rate where the 32-bit multiply and the 64-bit mad the compiler picks otherwise take
     * four times as long) */
    /* pair[j] = index of the symbols j, j + 1; a 4-gram index is two pairs, a 3-gram index a pair
     * and a symbol: two multiply-adds per position instead of the three of a rolling update */
    uint32_t pair[19];
#pragma unroll
    for (int j = 0; j < 19; j++)
      pair[j] = __umul24 (c[j], K.W) + c[j + 1];
    const uint32_t W2 = K.W * K.W;
    /* eight positions at a time: all their table words are asked for before any is looked at
     * (slot by slot, every ds_read waited behind the queue's ds_write of the slot before it,
     * which the compiler must assume to alias: 16 LDS round trips in a row per group) */
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t ix[8], word[8], ix3[SHORTS ? 8 : 1], nibs[SHORTS ? 8 : 1];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t idx = __umul24 (pair[8 * h + j], W2) + pair[8 * h + j + 2];
        ix[j] = idx;
#if defined(ACM_GRAM_ABLATE) && ACM_GRAM_ABLATE == 6 /* experiment: classes and indices, no table look-up, no push */
        word[j] = idx;
#else
        word[j] = *reinterpret_cast<const __attribute__ ((address_space (3))) uint32_t *> ((idx >> 5) * 4u);
#endif
        if (SHORTS) {
          const uint32_t idx3 = __umul24 (pair[8 * h + j], K.W) + c[8 * h + j + 2];
          ix3[j] = idx3;
          nibs[j] = *reinterpret_cast<const __attribute__ ((address_space (3))) unsigned char *> (K.g3_off + (idx3 >> 1));
        }
      }
      if (SHORTS) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint32_t nib = (nibs[j] >> ((ix3[j] & 1u) * 4u)) & 7u;
          const uint64_t m = __ballot (nib != 0);
          if (m) {
            if (nib)
              q3[qn3 + rank_below (m)] = make_uint2 (pos0 + 8 * h + j, ix3[j] | (nib << 20));
            qn3 = uniform (qn3 + (uint32_t)__popcll (m));
            if (qn3 >= WAVE)
              short_batch (WAVE);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool push = __builtin_amdgcn_ubfe (word[j], ix[j], 1u) != 0; /* v_bfe_u32 (it takes the low 5 bits of the offset itself) */
        const uint64_t m = __ballot (push);
#if defined(ACM_GRAM_ABLATE) && (ACM_GRAM_ABLATE == 5 || ACM_GRAM_ABLATE == 6) /* experiment: the 4-gram bits are looked up, nothing is queued */
        asm volatile ("" :: "s"(m));
        continue;
#endif
        if (m) {
          if (push)
            q1[qn1 + rank_below (m)] = make_uint2 (pos0 + 8 * h + j, gram_item_word (c, 8 * h + j, ix[j]));
          qn1 = uniform (qn1 + (uint32_t)__popcll (m));
          if (__builtin_expect (qn1 >= WAVE, 0)) {
            DIAG (const unsigned long long d_c1 = __builtin_readcyclecounter ();)
            batch_step (WAVE);
            DIAG (d_cons += __builtin_readcyclecounter () - d_c1; d_b1++;)
          }
        }
      }
    }
  };

  /* tiled scan: the wave's stream index = records it has written so far (up to a constant) */
  static_assert (!TILED || (!WIDE && !COUNT_ONLY), "tiled scans write their records themselves");
  constexpr bool tiled = TILED; /* (an instantiation of its own: the loop below with the drain inside it costs the plain scan registers) */
  auto stream_index = [&] (const WaveRec &w) -> uint32_t { return w.have ? (w.pad[1] - 1u) * REC_CHUNK + (uint32_t)counted : 0u; };
  if (tiled && lane == 0)
    Ws[wib].tile = NONE;
  /* the queues and the pipeline are emptied: when the text is through, and in a tiled scan behind
   * every tile (all its records are then written, side by side) */
  auto drain = [&] () {
      if (qn1)
        batch_step (qn1);
#pragma unroll
      for (int d = 0; d < GRAM_DEPTH; d++)
        consume_oldest ();
      while (qn2)
        walk_batch (qn2 < WAVE ? qn2 : WAVE);
      if (SHORTS && qn3)
        short_batch (qn3);
  };
  for (;;) {
    const uint32_t tile = share.next (next_tile, lane);
    if (tiled) {
      drain ();
      if (lane == 0) {
        const WaveRec w = Ws[wib];
        if (w.tile != NONE) {
          const uint32_t s_end = stream_index (w), cur_tile = w.tile;
          const uint32_t tile_begin = cur_tile * K.R * GROUP, tile_len = K.R * GROUP;
          const uint32_t tile_end = A.n - tile_begin < tile_len ? A.n : tile_begin + tile_len;
          TileEntry e;
          e.end_slot = w.have ? (((unsigned long long)w.base_hi << 32) | w.base_lo) + (uint32_t)counted : 0ull;
          e.lo = E.pos_base + (tile_begin > E.emit_from ? tile_begin : E.emit_from);
          e.hi = E.pos_base + tile_end;
          e.n = s_end - w.s_begin;
          e.n_late = s_end - w.s_late;
          e.c1 = w.prev1;
          e.c2 = w.prev1 != NONE ? w.prev2 : NONE;
          e.pad[0] = e.pad[1] = 0;
          dir[dir_base + (cur_tile - A.range_begin)] = e;
        }
      }
    }
    if (tile == NONE)
      break;
    DIAG (d_tiles++;)
    if (tiled && lane == 0) {
      WaveRec *W = Ws + wib;
      W->tile = tile;
      W->s_begin = W->s_late = stream_index (*W);
    }
    const uint32_t g0 = tile * K.R;
    uint4 c0 = load_group (g0), c1 = load_group (g0 + 1), c2 = load_group (g0 + 2), c3 = load_group (g0 + 3);
    for (uint32_t k = 0; k < K.R; k++) {
      uint4 n3;
      if (tiled && k + 1 == K.R && lane == 0) /* (keywords are no longer than a group here: what ends beyond the tile starts in its last group) */
        Ws[wib].s_late = stream_index (Ws[wib]);
      walk_group (c0, c1.x, g0 + k, n3);
      c0 = c1;
      c1 = c2;
      c2 = c3;
      c3 = n3;
    }
  }
  if (!tiled)
    drain ();
  if (COUNT_ONLY) {
    const uint32_t incl = wave_incl_scan ((uint32_t)counted);
    const uint32_t total = __shfl (incl, WAVE - 1, WAVE);
    if (lane == 0 && total)
      atomicAdd (E.count, (unsigned long long)total);
  } else {
    if (DIRECT) {
      /* what is left of the wave's last chunk is a hole for close_holes_kernel */
      if (lane == 0 && holes) {
        const WaveRec w = Ws[wib];
        const unsigned long long at = (((unsigned long long)w.base_hi << 32) | w.base_lo) + (uint32_t)counted;
        RecHole h = { (uint32_t)at, (uint32_t)(at >> 32), w.have ? E.rec_chunk - (uint32_t)counted : 0u, 0u };
        holes[wave_id] = h;
      }
    } else {
      if (counted)
        flush_hits (E, hits, (uint32_t)counted, lane);
      if (lane == 0 && fill)
        fill[wave_id] = hits[-1].y;
    }
  }
  DIAG (if (lane == 0 && wave_id < 8192) {
    unsigned long long *o = g_acm_diag[wave_id];
    o[0] = __builtin_readcyclecounter () - d_t0;
    o[1] = d_walk;
    o[2] = d_calls;
    o[3] = d_items;
    o[4] = d_b1;
    o[5] = d_cons;
    o[6] = wall_clock64 ();
    o[7] = d_tiles;
  })
}
