/* dev_gram2.h -- scan_gram2_kernel: the 4-gram kernel with a lane-local sieve (narrow alphabets whose
 * 2-bit table fits LDS; keywords of 1-3 symbols, if the dictionary has any, are scan_short_kernel's).
 * Device code of libac75_amd.so; included by dev_all.h inside its anonymous namespace.
 *
 * What it replaces (reference: the caller's loop acm_match -> acm_get_match, aho_corasick.c:434-482):
 * scan_gram_kernel (dev_gram.h) asks one LDS bit per position -- "a keyword starts with this
 * 4-gram", 19.6 % of config 3's positions -- and pushes the survivors into a per-wave queue position
 * by position: ballot -> scalar count -> branch -> LDS write, 21 instructions in one dependent
 * chain per position and wave, and empties the queue in batches with Bloom filters in front of
 * two dependent gathers by rank.  Here:
 *   1. two bits per 4-gram in LDS, T = "a keyword starts with it" and H = "it IS a keyword of 4
 *      symbols, or it is the TAIL (symbols 2-5) of some keyword's first five".  A position can
 *      only have something to report if T (p) and (H (p) or H (p + 1)) -- its 4-gram is a keyword,
 *      or its 5-gram can be a prefix: 8.7 % of config 3's positions (3.2 % real) instead of 19.6 %,
 *      for the same one LDS word per position (the word of position p + 1 is read for p + 1
 *      anyway).  No Bloom filters: LDS has nothing left for them;
 *   2. a lane keeps the answers of its 16 positions as bits of ONE register (one v_bfe + one
 *      v_lshl_or per position, no ballot, no scalar instruction, no branch); per group of 1,024
 *      positions one wave-wide prefix sum (six DPP adds) says where every lane's survivors go, and
 *      the lanes write their POSITIONS (2 bytes each) side by side into a list in LDS;
 *   3. the group's text is staged in LDS (one ds_write_b128 per lane), so the lane that takes
 *      survivor number s off the list rebuilds its item -- 4-gram index, classes of the 5th and
 *      6th symbol -- from six bytes at a run-time LDS address (not a run-time register index);
 *      batches of 64 are put together in registers across groups;
 *   4. ONE gather per survivor: its entry {children mask | flags, keyword id or first child} from
 *      the 64-byte line of its 16 4-grams (GramK::rows2), asked for in one step and looked at in
 *      the next; a keyword of 4 symbols is reported on the spot, a 5th symbol that goes on becomes
 *      a walk candidate whose first look (the peek entry of its depth-5 state) is asked for in the
 *      same way, a step ahead -- only what passes it (1 in 26) reaches walk_starts.
 * LDS: 2 x W^4 bits (a-z + "other": 132,864 B) + per wave 1,040 B of staged text, 256 B of list and
 * a walk queue of 64 items; alphabets of up to 26 symbols fit.  Everything else (tiles, records
 * written by the wave itself into chunks of the caller's buffer, tiled scans with a directory,
 * resumed segments) is scan_gram_kernel's.
 *
 * Measured on config 3, 2 GiB per launch, both kernels in one process (tools/exp_gram2.py): records
 * 2.34 -> 1.98 ms, count only 1.85 -> 1.60, tiled 2.43 -> 2.15.  What the cycle stamps of a
 * diagnostic build (tools/diag_gram2.py) say of a wave's 7,300 cycles per group: sieve 1,800, list
 * and rebuild 2,700, entries looked at 1,000, walks 500 to 1,200, waits for the gathers ~100 --
 * four waves per SIMD (one block per CU: the table fills LDS) and a chain of LDS round trips per
 * group; the SIMD issues ~600 instructions per group and wave.
 * Tried on the way, measured, not kept:
 *   - the entry by RANK as scan_gram_kernel has it (prefix count of the word, then entry: two
 *     dependent gathers, 406 M requests to L2 per launch): same time within 2 % once nothing sat
 *     in scratch memory -- a uint2 assigned under a condition did, two scratch stores and two
 *     flat loads per step, +0.5 ms, found as SQ_INSTS_VMEM_RD going UP when a gather was removed;
 *   - the rows in scan_gram_kernel (its Bloom filters leave 5.5 % of the positions to gather):
 *     count only 1.87 -> 2.33 ms -- the 2.1 MB of rows miss L2 more often than its 66 KB + 1 MB;
 *   - the text in four fixed sets of registers loaded by inline assembly with a counted wait
 *     (s_waitcnt vmcnt(6) at the top of a group instead of the compiler's vmcnt(0), which also
 *     waits for the gathers the last step has just issued): the registers have to be kept from the
 *     compiler -- a value with a load in flight must never be copied -- and amdgpu_num_vgpr does
 *     not do that: once the kernel grew, the record-mode instantiation used four of them for
 *     values of its own, the text was garbage and a gather went out of bounds.  The wait it
 *     saved was 130 to 220 of 7,300 cycles;
 *   - complete items written by the lane that owns the position, in straight-line code over its
 *     16 positions (no staged text, no rebuild, one LDS round trip per batch instead of two and
 *     no loop over the set bits): 16 exec-mask changes and 32 more vector instructions per group;
 *     count only 1.60 -> 1.79 ms, records 2.04 -> 2.09. */
constexpr uint32_t G2_STAGE = WAVE * 16 + 16; /* a group of text + the 8 bytes behind it (padded) */
constexpr uint32_t G2_LIST = 128;             /* survivor positions listed per round, 2 bytes each */
constexpr uint32_t G2_Q2 = 64;                /* walk queue, 8-byte items */
constexpr uint32_t G2_WAVE_BYTES = G2_STAGE + G2_LIST * 2 + G2_Q2 * 8;
static_assert (G2_WAVE_BYTES % 16 == 0, "the staged text is written 16 bytes per lane");
constexpr uint32_t G2_LDS_FIXED = (SPARSE_THREADS / WAVE) * G2_WAVE_BYTES + WALK_CTX_BYTES; /* + the table */

/* inclusive prefix sum over the wave in six DPP adds (row shifts, then the two row broadcasts);
 * every lane must be active */
__device__ __forceinline__ uint32_t
wave_incl_scan_dpp (uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x111, 0xf, 0xf, false); /* row_shr:1 */
  v += (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x112, 0xf, 0xf, false); /* row_shr:2 */
  v += (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x114, 0xf, 0xf, false); /* row_shr:4 */
  v += (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x118, 0xf, 0xf, false); /* row_shr:8 */
  v += (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x142, 0xa, 0xf, false); /* row_bcast:15 -> rows 1, 3 */
  v += (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x143, 0xc, 0xf, false); /* row_bcast:31 -> rows 2, 3 */
  return v;
}

template <bool COUNT_ONLY, bool TILED>
__global__ __launch_bounds__ (SPARSE_THREADS) void
scan_gram2_kernel (GramK K, EmitCtx E, Launch A, const unsigned char *__restrict__ text, uint2 *items, uint32_t region_items,
                   uint32_t *fill, RecHole *holes, uint32_t resume, TileEntry *dir, uint32_t dir_base) {
  (void)items;
  (void)region_items;
  (void)fill;
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  constexpr uint32_t WAVES = SPARSE_THREADS / WAVE;
  constexpr uint32_t GROUP = WAVE * 16;
  {
    uint4 *dst = reinterpret_cast<uint4 *> (smem);
    const uint4 *src = reinterpret_cast<const uint4 *> (K.tab2);
    for (uint32_t i = threadIdx.x; i < (K.tab2_words + 3) / 4; i += blockDim.x)
      dst[i] = src[i];
  }
  uint32_t *next_tile = reinterpret_cast<uint32_t *> (smem + K.g2_off + WAVES * G2_WAVE_BYTES);
  StartsK *Ks = reinterpret_cast<StartsK *> (next_tile + 4); /* see scan_starts_kernel */
  EmitCtx *Es = reinterpret_cast<EmitCtx *> (reinterpret_cast<unsigned char *> (Ks) + WALK_CTX_K);
  WaveRec *Ws = reinterpret_cast<WaveRec *> (reinterpret_cast<unsigned char *> (Ks) + WALK_CTX_K + WALK_CTX_E); /* one per wave */
  if (threadIdx.x == 0) {
    *next_tile = 0;
    StartsK Kc{};
    Kc.srec = K.srec;
    Kc.sedge = K.sedge;
    Kc.remap = K.g4gid;
    Kc.remap_base = K.d5_begin;
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
  /* the wave's part of LDS, by byte offset (address space 3) */
  const uint32_t stage_off = K.g2_off + wib * G2_WAVE_BYTES;
  const uint32_t list_off = stage_off + G2_STAGE;
  uint2 *q2 = reinterpret_cast<uint2 *> (smem + stage_off + G2_STAGE + G2_LIST * 2);
  const uint32_t wave_id = blockIdx.x * WAVES + wib;
  uint2 *hits = reinterpret_cast<uint2 *> (Ws + wib); /* records straight into the caller's buffer (dev_starts.h: WaveRec) */
  if (lane == 0) {
    /* (as in scan_gram_kernel: no chunk yet, or the chunk a resumed segment carries over) */
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
  const TileShare share (A);
  const uint32_t last_blk = (A.n - 1) / 16;
  uint32_t qn2 = 0;
  unsigned long long counted = 0;
  RecState rs = { 0ull, 0u };
  if (!COUNT_ONLY && resume) {
    rs = rec_state_load (hits);
    counted = uniform (reinterpret_cast<const WaveRec *> (hits)->pad[0]);
  }
  /* the batch that is being put together: lanes [0, pk) hold an item (position, 4-gram index |
   * class of the 5th symbol << 20 | class of the 6th << 25) */
  uint32_t it_x = 0, it_y = 0, pk = 0;
  /* the batch whose gather is in flight: its items, the entry every lane asked for (landing), the
   * number of existing 4-grams in front of the lane's in its word | NEED, the number of items */
  uint2 pend_item = make_uint2 (0, 0), pend_e = make_uint2 (0, 0);
  uint32_t pend_r = 0, pend_n = 0;
  constexpr uint32_t PEND_NEED = 0x80000000u;
  auto lds_word = [&] (uint32_t byte_off) -> uint32_t {
    return *reinterpret_cast<const __attribute__ ((address_space (3))) uint32_t *> (byte_off);
  };
  const uint4 *text16 = reinterpret_cast<const uint4 *> (text);
  auto load_group = [&] (uint32_t g) -> uint4 {
    const uint32_t blk = g * WAVE + lane;
    /* (non-temporal: the text is read once -- same time, and the rows' lines stay in L2 more often:
     * FETCH_SIZE per launch 2.22 -> 1.84 M KB, TCC_MISS 42 -> 36 M) */
    typedef uint32_t nt_u32x4 __attribute__ ((ext_vector_type (4)));
    const nt_u32x4 v = __builtin_nontemporal_load (reinterpret_cast<const nt_u32x4 *> (text16) + (blk < last_blk ? blk : last_blk));
    return make_uint4 (v.x, v.y, v.z, v.w);
  };
  /* diagnostic build (make diag, tools/diag_gram2.py): cycles per wave by part of the group loop */
  DIAG (const unsigned long long d_t0 = __builtin_readcyclecounter (); unsigned long long d_top = 0, d_sieve = 0, d_take = 0, d_cwait = 0, d_cons = 0, d_walk = 0, d_steps = 0;)
  auto walk_batch = [&] (uint32_t n_items) {
    DIAG (const unsigned long long d_wk = __builtin_readcyclecounter ();)
    const unsigned long long r = walk_starts<uint8_t, COUNT_ONLY, 2> (Ks, Es, text, q2, qn2, n_items, hits, counted);
    qn2 = uniform ((uint32_t)(r >> 32));
    counted = (uint32_t)r;
    if (!COUNT_ONLY)
      rs = rec_state_load (hits); /* (the walk may have gone on to the next chunk) */
    DIAG (d_walk += __builtin_readcyclecounter () - d_wk;)
  };
  /* Walk candidates between the entry and the walk queue: a candidate's first look is its depth-5
   * state's peek entry -- {its record, the symbol of its only edge} -- against the 6th symbol, whose
   * class came along: 25 of 26 end there.  Left to walk_starts (as scan_gram_kernel does) that look
   * is a call with a gather and a wait of its own, 1,200 to 2,200 of the 7,300 to 9,100 cycles a wave
   * spent per group (cycle stamps of a diagnostic build); here the gather is issued when the entry
   * is looked at and looked at a step later, and only what passes goes to the walk queue, as the
   * record item walk_starts would have made of it.  (Plans with 4-byte peek entries and relative
   * state ids: automata of fewer than 2^23 states; others queue the candidates as before.) */
  uint32_t peek_pos = 0, peek_cls = 0, peek_val = 0, peek_n = 0; /* per lane: position, class of the next symbol | valid << 31, the entry (landing); wave: any in flight */
  /* (record scans: 2.04 -> 1.98 ms per 2 GiB of config 3; a count-only scan has no records to write in
   * the walk and measured 4 % SLOWER with the extra stage: it keeps the plain queue) */
  const bool peek_inline = !COUNT_ONLY && K.peek_packed && K.d5_rel;
  auto push_walk = [&] (bool go, uint32_t ix, uint32_t iy) {
    const uint64_t m = __ballot (go);
    if (m) {
      /* (a walk call handles one level of the newest items and never leaves more than it took,
       * and every item ends within lmax levels) */
      while (qn2 + (uint32_t)__popcll (m) > G2_Q2)
        walk_batch (qn2 < WAVE ? qn2 : WAVE);
      if (go)
        q2[qn2 + rank_below (m)] = make_uint2 (ix, iy);
      qn2 = uniform (qn2 + (uint32_t)__popcll (m));
#if defined(ACM_GRAM2_ABLATE) && ACM_GRAM2_ABLATE == 2 /* experiment: the walk candidates are queued and dropped */
      qn2 = 0;
#endif
      while (qn2 >= WAVE)
        walk_batch (WAVE);
    }
  };
  auto resolve_peeks = [&] () {
    if (peek_n == 0)
      return;
    const uint32_t ev = peek_val, cls = peek_cls & 31u;
    const uint32_t sym = (ev >> 31) ? GRAM_NO_PEEK : (ev >> 23) & 0xFFu; /* (GramK::g5peek, packed: record | symbol << 23 | "look at the record" << 31) */
    const uint32_t c1 = cls < K.span ? K.lo + cls : 0x100u;              /* (no symbol of the alphabet, or beyond the text: matches no edge) */
    const bool on = (peek_cls >> 31) != 0 && (sym == GRAM_NO_PEEK || sym == c1);
    peek_n = 0;
    push_walk (on, peek_pos, (ev & 0x7FFFFFu) | WI_RECORD | cls << 24);
  };
  /* The batch in flight is looked at: a keyword of 4 symbols ends here (reported at once: the
   * entry brought its id along); the 5th symbol is an edge of the depth-4 state (a walk candidate
   * at the depth-5 state it leads to, the class of the 6th symbol riding along: the walk's first
   * look needs no text). */
  auto consume_pending = [&] () {
    if (pend_n == 0)
      return;
    DIAG (const unsigned long long d_c0 = __builtin_readcyclecounter (); asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long d_c1 = __builtin_readcyclecounter (); d_cwait += d_c1 - d_c0; d_steps++; const unsigned long long d_w0 = d_walk;)
    /* (words, not a struct: a uint2 assigned under a condition was put in scratch memory) */
    struct { uint32_t x, y; } e = { pend_e.x, pend_e.y };
    const bool valid = lane < pend_n && (pend_r & PEND_NEED) != 0;
    const uint32_t r = pend_r & 0xFFu;
#if defined(ACM_GRAM2_ABLATE) && (ACM_GRAM2_ABLATE == 3 || ACM_GRAM2_ABLATE == 8) /* experiment: the entries are gathered and only looked at */
    asm volatile ("" :: "v"(e.x), "v"(e.y));
    pend_n = 0;
    return;
#endif
    /* (rare: the word has nine entries or more -- slot 7 says where the 8th and later ones are) */
    const bool far = valid && r >= 7 && (e.x & 0x20000000u) != 0;
    if (__ballot (far)) {
      const uint2 o = K.over2[far ? (e.x & 0x1FFFFFFFu) + (r - 7) : 0u];
      e.x = far ? o.x : e.x;
      e.y = far ? o.y : e.y;
    }
    if (!valid)
      e.x = 0;
    /* a keyword of 4 symbols that other keywords go on from is left to the walk: an item at the
     * depth-4 state's own record, which reports it and looks at the 5th symbol itself */
    const bool at_record = (e.x & 0x40000000u) != 0;
#if !(defined(ACM_GRAM2_ABLATE) && ACM_GRAM2_ABLATE == 1)
    emit_terminals<COUNT_ONLY, true> (E, (e.x >> 31) != 0 && !at_record && pend_item.x + 3 >= E.emit_from, pend_item.x + 3, e.y, 4u, lane, hits, counted, Es, &rs);
    if (!COUNT_ONLY)
      counted = uniform ((uint32_t)counted);
#endif
    const uint32_t c4 = (pend_item.y >> 20) & 31u, c5 = (pend_item.y >> 25) & 31u;
    const bool child = ((e.x >> c4) & 1u) != 0 && !at_record; /* (classes are below 30: never one of the flag bits) */
    const uint32_t st5 = e.y + __popc (e.x & ((1u << c4) - 1u));
    if (peek_inline) {
      /* the 5th symbol leads to depth-5 state st5: its peek entry is asked for, to be looked at a step later */
      resolve_peeks ();
      uint32_t pv = 0;
      if (child)
        pv = K.g5peek[st5 - K.d5_begin];
      peek_val = pv;
      peek_pos = pend_item.x + 4;
      peek_cls = c5 | (child ? 0x80000000u : 0u);
      peek_n = __ballot (child) != 0 ? 1u : 0u;
      push_walk (at_record, pend_item.x + 3, e.y | WI_RECORD | c4 << 24);
    } else
      push_walk (child || at_record, at_record ? pend_item.x + 3 : pend_item.x + 4,
                 at_record ? e.y | WI_RECORD | (K.d5_rel ? c4 << 24 : 0u) : (K.d5_rel ? (st5 - K.d5_begin) | c5 << 24 : st5));
    pend_n = 0;
    DIAG (d_cons += (__builtin_readcyclecounter () - d_c1) - (d_walk - d_w0);)
  };
  /* One step: the batch in flight is looked at, the n items in lanes [0, n) of (it_x, it_y) send
   * for their entries -- the word of the table says which slot of the word's line is theirs. */
  auto batch_step = [&] (uint32_t n_items) {
    const uint32_t idx = it_y & 0xFFFFFu;
    const uint32_t word = lds_word ((idx >> 4) * 4u);
    consume_pending ();
    const uint32_t sh = (idx & 15u) * 2u;
    /* (a lane that needs nothing asks for nothing: a gather costs by the line) */
    const bool need = lane < n_items && ((word >> sh) & 1u) != 0;
    const uint32_t r = __popc (word & 0x55555555u & ((1u << sh) - 1u));
    uint2 e = make_uint2 (0, 0);
#if defined(ACM_GRAM2_ABLATE) && ACM_GRAM2_ABLATE == 8 /* experiment: every lane asks for the same line */
    if (need)
      e = K.rows2[r < 7 ? r : 7];
#else
    if (need)
      e = K.rows2[(size_t)(idx >> 4) * 8 + (r < 7 ? r : 7)];
#endif
    pend_e = e;
    pend_item = make_uint2 (it_x, it_y);
    pend_r = r | (need ? PEND_NEED : 0u);
    pend_n = n_items;
  };

  /* one group, first half: the sieve.  S holds the group's text (waited for here) and is asked for
   * the group four ahead as soon as its bytes have become classes; leaves the lane's survivors as
   * the even bits of `pass`, the number of its first one among the group's in `my`, returns their total */
  auto sieve_group = [&] (const uint4 cur, const uint32_t next_x, const uint32_t next_y, const uint32_t g, uint4 &prefetched, uint32_t &pass_out, uint32_t &my_out) -> uint32_t {
    typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
    typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
    uint32_t t4 = (uint32_t)__builtin_amdgcn_update_dpp ((int)uniform (next_x), (int)cur.x, 0x130, 0xf, 0xf, false);
    const uint32_t t5 = (uint32_t)__builtin_amdgcn_update_dpp ((int)uniform (next_y), (int)cur.y, 0x130, 0xf, 0xf, false);
    asm volatile ("" : "+v"(t4));
    __builtin_amdgcn_sched_barrier (0);
    prefetched = load_group (g + 4);
    const uint32_t t0 = cur.x, t1 = cur.y, t2 = cur.z, t3 = cur.w;
    /* the group's text where the lanes that take its survivors off the list find it */
    *reinterpret_cast<__attribute__ ((address_space (3))) u32x4 *> (stage_off + lane * 16u) = u32x4{ t0, t1, t2, t3 };
    if (lane == WAVE - 1)
      *reinterpret_cast<__attribute__ ((address_space (3))) u32x2 *> (stage_off + GROUP) = u32x2{ t4, t5 };
    const uint32_t pos0 = g * GROUP + lane * 16;
    const uint32_t w[5] = { t0, t1, t2, t3, t4 };
#if defined(ACM_GRAM2_ABLATE) && ACM_GRAM2_ABLATE == 7 /* experiment: the text is streamed and staged, not looked at */
    asm volatile ("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]));
    pass_out = 0;
    return 0;
#endif
    uint32_t c[20];
#pragma unroll
    for (int j = 0; j < 20; j++) {
      const uint32_t b = (w[j / 4] >> (8 * (j % 4))) & 0xFFu;
      c[j] = min (b - K.lo, K.span);
    }
    /* symbols past the end of the segment count as outside the alphabet (only the last groups: a
     * condition on the GROUP, in a scalar register -- on the lane's own position the compiler turned
     * the branch into 20 compares and 20 selects for every group, a sixth of the sieve's instructions) */
    if (__builtin_expect (uniform (g * GROUP + GROUP + 20) > A.n, 0)) {
#pragma unroll
      for (int j = 0; j < 20; j++)
        if (pos0 + j >= A.n)
          c[j] = K.span;
    }
    /* (the set has been copied out: its registers take the group four ahead) */
    uint32_t pair[19];
#pragma unroll
    for (int j = 0; j < 19; j++)
      pair[j] = __umul24 (c[j], K.W) + c[j + 1];
    const uint32_t W2 = K.W * K.W;
    /* bits 2j, 2j + 1 of acc: T and H of this lane's position j */
    uint32_t acc = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t ix[8], word[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        ix[j] = __umul24 (pair[8 * h + j], W2) + pair[8 * h + j + 2];
        word[j] = lds_word ((ix[j] >> 4) * 4u);
      }
#pragma unroll
      for (int j = 0; j < 8; j++)
        acc |= __builtin_amdgcn_ubfe (word[j], ix[j] << 1, 2u) << (2 * (8 * h + j)); /* (v_bfe_u32 takes the low 5 bits of the offset itself; v_lshl_or_b32) */
    }
    /* H of the position behind the lane's last (the next lane's first) */
    const uint32_t ix16 = __umul24 (pair[16], W2) + pair[18];
    const uint32_t h16 = __builtin_amdgcn_ubfe (lds_word ((ix16 >> 4) * 4u), (ix16 << 1) | 1u, 1u);
    /* T (j) and (H (j) or H (j + 1)), at the even bits */
    const uint32_t pass = acc & ((acc >> 1) | (acc >> 3) | (h16 << 30)) & 0x55555555u;
    pass_out = pass;
#if defined(ACM_GRAM2_ABLATE) && ACM_GRAM2_ABLATE == 6 /* experiment: the sieve alone */
    asm volatile ("" :: "v"(pass));
    return 0;
#endif
    const uint32_t cnt = __popc (pass);
    const uint32_t incl = wave_incl_scan_dpp (cnt);
    my_out = incl - cnt;
    return (uint32_t)__builtin_amdgcn_readlane ((int)incl, WAVE - 1);
  };
  /* one group, second half: the survivors listed, their items rebuilt from the staged text, full
   * batches sent through the pipeline */
  auto take_group = [&] (const uint32_t g, const uint32_t total, uint32_t pass, uint32_t my) {
    const uint32_t W2 = K.W * K.W;
    const bool tail = g * GROUP + GROUP + 8 > A.n; /* (the last groups of the segment) */
    for (uint32_t base = 0; base < total; base += G2_LIST) {
      const uint32_t n_here = total - base < G2_LIST ? total - base : G2_LIST;
      /* the positions (within the group) of the survivors base .. base + n_here - 1, side by side */
      {
        const uint32_t lim = base + n_here;
        while (pass != 0 && my < lim) {
          const uint32_t b = (uint32_t)__builtin_ctz (pass);
          *reinterpret_cast<__attribute__ ((address_space (3))) uint16_t *> (list_off + (my - base) * 2u) = (uint16_t)(lane * 16u + (b >> 1));
          pass &= pass - 1u;
          my++;
        }
      }
#if defined(ACM_GRAM2_ABLATE) && ACM_GRAM2_ABLATE == 5 /* experiment: the survivors are listed and dropped */
      continue;
#endif
      for (uint32_t off = 0; off < n_here;) {
        const uint32_t take = n_here - off < WAVE - pk ? n_here - off : WAVE - pk;
        const bool mine = lane >= pk && lane < pk + take;
        /* survivor off + (lane - pk): its position q in the group, the six symbols from there */
        const uint32_t li = mine ? off + lane - pk : 0u;
        const uint32_t q = *reinterpret_cast<const __attribute__ ((address_space (3))) uint16_t *> (list_off + li * 2u);
        const uint32_t a = stage_off + (q & ~3u);
        const uint32_t d0 = lds_word (a), d1 = lds_word (a + 4u), d2 = lds_word (a + 8u);
        const uint32_t lo4 = __builtin_amdgcn_alignbyte (d1, d0, q), hi4 = __builtin_amdgcn_alignbyte (d2, d1, q);
        uint32_t r[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const uint32_t b = ((k < 4 ? lo4 : hi4) >> (8 * (k % 4))) & 0xFFu;
          r[k] = min (b - K.lo, K.span);
        }
        const uint32_t p = g * GROUP + q;
        if (tail) {
#pragma unroll
          for (int k = 0; k < 6; k++)
            if (p + k >= A.n)
              r[k] = K.span;
        }
        const uint32_t idx = __umul24 (__umul24 (r[0], K.W) + r[1], W2) + __umul24 (r[2], K.W) + r[3];
        const uint32_t y = idx | (r[4] << 20) | (r[5] << 25);
        it_x = mine ? p : it_x;
        it_y = mine ? y : it_y;
        pk += take;
        off += take;
#if defined(ACM_GRAM2_ABLATE) && ACM_GRAM2_ABLATE == 4 /* experiment: the items are rebuilt and dropped */
        if (pk == WAVE) {
          asm volatile ("" :: "v"(it_x), "v"(it_y));
          pk = 0;
        }
        continue;
#endif
        if (pk == WAVE) {
          batch_step (WAVE);
          pk = 0;
        }
      }
    }
  };

  /* tiled scan: the wave's stream index = records it has written so far (up to a constant) */
  static_assert (!TILED || !COUNT_ONLY, "tiled scans write their records themselves");
  constexpr bool tiled = TILED;
  auto stream_index = [&] (const WaveRec &w) -> uint32_t { return w.have ? (w.pad[1] - 1u) * REC_CHUNK + (uint32_t)counted : 0u; };
  if (tiled && lane == 0)
    Ws[wib].tile = NONE;
  /* the batch in the making, the pipeline and the walk queue are emptied: when the text is
   * through, and in a tiled scan behind every tile (all its records are then written, side by side) */
  auto drain = [&] () {
    if (pk) {
      batch_step (pk);
      pk = 0;
    }
    consume_pending ();
    resolve_peeks ();
    while (qn2)
      walk_batch (qn2 < WAVE ? qn2 : WAVE);
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
    if (tiled && lane == 0) {
      WaveRec *W = Ws + wib;
      W->tile = tile;
      W->s_begin = W->s_late = stream_index (*W);
    }
    const uint32_t g0 = tile * K.R;
    uint4 c0 = load_group (g0), c1 = load_group (g0 + 1), c2 = load_group (g0 + 2), c3 = load_group (g0 + 3);
    for (uint32_t k = 0; k < K.R; k++) {
      if (tiled && k + 1 == K.R && lane == 0)
        Ws[wib].s_late = stream_index (Ws[wib]);
      uint4 n3;
      uint32_t pass = 0, my = 0;
      /* (diagnostic build: the wait for the group's text apart from the sieve) */
      DIAG (const unsigned long long d_g0 = __builtin_readcyclecounter (); asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long d_g1 = __builtin_readcyclecounter (); d_top += d_g1 - d_g0;)
      const uint32_t total = sieve_group (c0, c1.x, c1.y, g0 + k, n3, pass, my);
      DIAG (const unsigned long long d_g2 = __builtin_readcyclecounter (); d_sieve += d_g2 - d_g1; const unsigned long long d_in = d_cwait + d_cons + d_walk;)
      if (total)
        take_group (g0 + k, total, pass, my);
      DIAG (d_take += (__builtin_readcyclecounter () - d_g2) - (d_cwait + d_cons + d_walk - d_in);)
      c0 = c1;
      c1 = c2;
      c2 = c3;
      c3 = n3;
    }
  }
  if (!tiled)
    drain ();
  DIAG (if (lane == 0 && wave_id < 8192) {
    unsigned long long *o = g_acm_diag[wave_id];
    o[0] = __builtin_readcyclecounter () - d_t0;
    o[1] = d_top;
    o[2] = d_sieve;
    o[3] = d_take;
    o[4] = d_cwait;
    o[5] = d_cons;
    o[6] = d_walk;
    o[7] = d_steps;
  })
  if (COUNT_ONLY) {
    const uint32_t incl = wave_incl_scan ((uint32_t)counted);
    const uint32_t total = __shfl (incl, WAVE - 1, WAVE);
    if (lane == 0 && total)
      atomicAdd (E.count, (unsigned long long)total);
  } else {
    /* what is left of the wave's last chunk is a hole for close_holes_kernel */
    if (lane == 0 && holes) {
      const WaveRec w = Ws[wib];
      const unsigned long long at = (((unsigned long long)w.base_hi << 32) | w.base_lo) + (uint32_t)counted;
      RecHole h = { (uint32_t)at, (uint32_t)(at >> 32), w.have ? E.rec_chunk - (uint32_t)counted : 0u, 0u };
      holes[wave_id] = h;
    }
  }
}
