/* dev_short.h -- scan_short_kernel: the keywords of 1-3 symbols of a big byte dictionary, a pass of
 * their own behind the 4-gram kernel (narrow alphabets).
 * Device code of libac75_amd.so; included by dev_all.h inside its anonymous namespace.
 *
 * A dictionary that the 4-gram kernels take may also hold keywords of 1-3 symbols -- the
 * reference's own dictionaries do ("he", "u", "hi": examples/test.c:6,
 * aho_corasick_generic_test.c:73-99).  scan_gram_kernel looked for them beside the 4-grams: a
 * nibble per 3-gram in LDS on a second rolling index, a third queue, a third kind of batch -- the
 * record-mode instantiation spilled 319 vector registers and ran at 0.2 to 0.6 TB/s.  They are now
 * a pass of their own over the same text, behind the 4-gram pass (which no longer knows of them),
 * into the same record buffer: the match set of a dictionary is the union of its keywords' match
 * sets, and the caller loop's order (acm_match -> acm_get_match, aho_corasick.c:434-482) is
 * restored by the order passes either way.
 *   - LDS: one nibble per 3-gram over the W classes (bit d - 1: "the first d symbols are a keyword";
 *     W = 27: 9.8 KB), the waves' staged text and lists;
 *   - a lane looks its 16 positions' nibbles up and keeps "something ends here" as bits of one
 *     register; a wave-wide prefix sum per group, the survivors' positions listed in LDS, batches of
 *     64 put together in registers across groups (the sieve of scan_gram2_kernel, dev_gram2.h);
 *   - a batch asks for the keyword ids of its 3-grams' three prefixes (GramK::g3rec: 16 bytes from
 *     W^3 entries, 315 KB for a-z) and writes the records a step later, straight into the wave's
 *     chunk of the caller's buffer (dev_starts.h: WaveRec, emit_terminals); the holes its waves
 *     leave are closed behind its last segment (close_holes_kernel, a second set of descriptors). */
constexpr uint32_t SH_STAGE = WAVE * 16 + 16, SH_LIST = 256; /* staged text of a group + 8 bytes behind it; survivor positions per round */
constexpr uint32_t SH_WAVE_BYTES = SH_STAGE + SH_LIST * 2;
static_assert (SH_WAVE_BYTES % 16 == 0, "the staged text is written 16 bytes per lane");
constexpr uint32_t SH_CTX_BYTES = 16 + WALK_CTX_E + (SPARSE_THREADS / WAVE) * sizeof (WaveRec);
constexpr uint32_t SH_LDS_FIXED = (SPARSE_THREADS / WAVE) * SH_WAVE_BYTES + SH_CTX_BYTES; /* + the nibbles */

template <bool COUNT_ONLY>
__global__ __launch_bounds__ (SPARSE_THREADS) void
scan_short_kernel (GramK K, EmitCtx E, Launch A, const unsigned char *__restrict__ text, RecHole *holes, uint32_t resume) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  constexpr uint32_t WAVES = SPARSE_THREADS / WAVE;
  constexpr uint32_t GROUP = WAVE * 16;
  const uint32_t nib_bytes = (K.g3_bytes + 15) & ~15u;
  {
    uint4 *dst = reinterpret_cast<uint4 *> (smem);
    const uint4 *src = reinterpret_cast<const uint4 *> (reinterpret_cast<const unsigned char *> (K.g4bits) + K.g3_off);
    for (uint32_t i = threadIdx.x; i < nib_bytes / 16; i += blockDim.x)
      dst[i] = src[i];
  }
  uint32_t *next_tile = reinterpret_cast<uint32_t *> (smem + nib_bytes + WAVES * SH_WAVE_BYTES);
  EmitCtx *Es = reinterpret_cast<EmitCtx *> (next_tile + 4);
  WaveRec *Ws = reinterpret_cast<WaveRec *> (reinterpret_cast<unsigned char *> (Es) + WALK_CTX_E);
  if (threadIdx.x == 0) {
    *next_tile = 0;
    *Es = E;
  }
  __syncthreads ();
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t wib = uniform (threadIdx.x / WAVE);
  const uint32_t stage_off = nib_bytes + wib * SH_WAVE_BYTES, list_off = stage_off + SH_STAGE;
  const uint32_t wave_id = blockIdx.x * WAVES + wib;
  uint2 *hits = reinterpret_cast<uint2 *> (Ws + wib);
  if (lane == 0) {
    /* (as in scan_gram_kernel: no chunk yet, or the chunk a resumed segment carries over) */
    WaveRec w0{};
    if (resume && !COUNT_ONLY && holes) {
      const RecHole h = holes[wave_id];
      if (h.len) {
        const unsigned long long at = ((unsigned long long)h.start_hi << 32) | h.start_lo;
        const unsigned long long base = at + h.len - REC_CHUNK;
        const bool below = base + REC_CHUNK <= E.capacity;
        const bool above = base >= E.capacity && base - E.capacity + REC_CHUNK <= E.spill_slots;
        const uint64_t dst = below ? reinterpret_cast<uint64_t> (&E.records[base]) : (above ? reinterpret_cast<uint64_t> (E.spill + (base - E.capacity)) : 0ull);
        w0.dst_lo = (uint32_t)dst;
        w0.dst_hi = (uint32_t)(dst >> 32);
        w0.base_lo = (uint32_t)base;
        w0.base_hi = (uint32_t)(base >> 32);
        w0.limit = (below || above) ? REC_CHUNK : 0u;
        w0.have = 1;
        w0.pad[0] = REC_CHUNK - h.len;
        w0.pad[1] = 1;
        w0.prev1 = w0.prev2 = NONE;
      }
    }
    if (!w0.have)
      w0.prev1 = w0.prev2 = NONE;
    Ws[wib] = w0;
  }
  const uint4 *text16 = reinterpret_cast<const uint4 *> (text);
  const TileShare share (A);
  const uint32_t last_blk = (A.n - 1) / 16;
  unsigned long long counted = 0;
  RecState rs = { 0ull, 0u };
  if (!COUNT_ONLY && resume) {
    rs = rec_state_load (hits);
    counted = uniform (reinterpret_cast<const WaveRec *> (hits)->pad[0]);
  }
  typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
  typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
  auto lds_word = [&] (uint32_t byte_off) -> uint32_t {
    return *reinterpret_cast<const __attribute__ ((address_space (3))) uint32_t *> (byte_off);
  };
  auto nibble = [&] (uint32_t idx3) -> uint32_t { /* bit d - 1: the first d symbols of the 3-gram are a keyword */
    const uint32_t b = *reinterpret_cast<const __attribute__ ((address_space (3))) unsigned char *> (idx3 >> 1);
    return (b >> ((idx3 & 1u) * 4u)) & 7u;
  };
  auto load_group = [&] (uint32_t g) -> uint4 {
    const uint32_t blk = g * WAVE + lane;
    const u32x4 v = __builtin_nontemporal_load (reinterpret_cast<const u32x4 *> (text16) + (blk < last_blk ? blk : last_blk));
    return make_uint4 (v.x, v.y, v.z, v.w);
  };
  /* the batch that is being put together: lanes [0, pk) hold (position, 3-gram index | nibble << 20);
   * the batch whose ids are in flight */
  uint32_t it_x = 0, it_y = 0, pk = 0;
  uint2 pend_item = make_uint2 (0, 0);
  uint32_t pend_k1 = 0, pend_k2 = 0, pend_k3 = 0, pend_n = 0; /* (words, not a uint4 temporary) */
  auto consume_pending = [&] () {
    if (pend_n == 0)
      return;
    uint32_t nib = lane < pend_n ? pend_item.y >> 20 : 0u;
    /* (a keyword of d symbols that starts at position p ends at p + d - 1: not before emit_from) */
    if (pend_item.x < E.emit_from)
      nib &= pend_item.x + 1 >= E.emit_from ? 6u : (pend_item.x + 2 >= E.emit_from ? 4u : 0u);
    const uint32_t mine = __popc (nib);
    if (COUNT_ONLY) {
      counted += mine;
      pend_n = 0;
      return;
    }
    /* the batch's records in ONE reservation: every lane's up to three side by side (a prefix sum
     * over the lanes' counts; a ballot and a rank per length were three of each per batch) */
    const uint32_t incl = wave_incl_scan_dpp (mine);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane ((int)incl, WAVE - 1);
    const uint32_t used = (uint32_t)counted;
    if (total && used + total <= rs.limit) {
      typedef uint32_t g_u32x4 __attribute__ ((ext_vector_type (4)));
      uint64_t at = rs.dst + ((uint64_t)(used + incl - mine) << 4);
#pragma unroll
      for (uint32_t d = 0; d < 3; d++) {
        if ((nib >> d) & 1u) {
          const uint64_t gp = E.pos_base + pend_item.x + d;
          *reinterpret_cast<__attribute__ ((address_space (1))) g_u32x4 *> (at) =
            g_u32x4{ (uint32_t)gp, (uint32_t)(gp >> 32), d + 1, d == 0 ? pend_k1 : (d == 1 ? pend_k2 : pend_k3) };
          at += 16;
        }
      }
      counted = used + total;
    } else if (total) {
      /* (the chunk ends inside the batch, or there is none yet: length by length through the path that reserves the next) */
#pragma unroll
      for (uint32_t d = 0; d < 3; d++) {
        emit_terminals<COUNT_ONLY, true> (E, ((nib >> d) & 1u) != 0, pend_item.x + d, d == 0 ? pend_k1 : (d == 1 ? pend_k2 : pend_k3), d + 1, lane, hits, counted, Es, &rs);
        counted = uniform ((uint32_t)counted);
      }
    }
    pend_n = 0;
  };
  auto batch_step = [&] (uint32_t n_items) {
    consume_pending ();
    uint4 ids = make_uint4 (0, 0, 0, 0);
    if (!COUNT_ONLY && lane < n_items)
      ids = K.g3rec[it_y & 0xFFFFFu];
    pend_k1 = ids.x;
    pend_k2 = ids.y;
    pend_k3 = ids.z;
    pend_item = make_uint2 (it_x, it_y);
    pend_n = n_items;
  };
  auto drain = [&] () {
    if (pk) {
      batch_step (pk);
      pk = 0;
    }
    consume_pending ();
  };
  /* one group: cur = this lane's 16 bytes, (next_x, next_y) = the first 8 bytes of every lane of the next group */
  auto walk_group = [&] (const uint4 cur, const uint32_t next_x, const uint32_t next_y, const uint32_t g, uint4 &prefetched) {
    uint32_t t4 = (uint32_t)__builtin_amdgcn_update_dpp ((int)uniform (next_x), (int)cur.x, 0x130, 0xf, 0xf, false); /* word 0 of the next lane */
    const uint32_t t5 = (uint32_t)__builtin_amdgcn_update_dpp ((int)uniform (next_y), (int)cur.y, 0x130, 0xf, 0xf, false);
    asm volatile ("" : "+v"(t4));
    __builtin_amdgcn_sched_barrier (0);
    prefetched = load_group (g + 4);
    *reinterpret_cast<__attribute__ ((address_space (3))) u32x4 *> (stage_off + lane * 16u) = u32x4{ cur.x, cur.y, cur.z, cur.w };
    if (lane == WAVE - 1)
      *reinterpret_cast<__attribute__ ((address_space (3))) u32x2 *> (stage_off + GROUP) = u32x2{ t4, t5 };
    const uint32_t pos0 = g * GROUP + lane * 16;
    const uint32_t w[5] = { cur.x, cur.y, cur.z, cur.w, t4 };
    uint32_t c[18];
#pragma unroll
    for (int j = 0; j < 18; j++) {
      const uint32_t b = (w[j / 4] >> (8 * (j % 4))) & 0xFFu;
      c[j] = min (b - K.lo, K.span);
    }
    /* symbols past the end of the segment count as outside the alphabet (only the last groups) */
    if (pos0 + 18 > A.n) {
#pragma unroll
      for (int j = 0; j < 18; j++)
        if (pos0 + j >= A.n)
          c[j] = K.span;
    }
    uint32_t pass = 0, ix[16], nb[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      ix[j] = __umul24 (__umul24 (c[j], K.W) + c[j + 1], K.W) + c[j + 2];
      nb[j] = *reinterpret_cast<const __attribute__ ((address_space (3))) unsigned char *> (ix[j] >> 1);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t nib = (nb[j] >> ((ix[j] & 1u) * 4u)) & 7u;
      pass |= min (nib, 1u) << j;
    }
    if (pos0 + 16 > A.n) /* (a position beyond the segment is no position) */
      pass &= pos0 < A.n ? (1u << (A.n - pos0)) - 1u : 0u;
    const uint32_t cnt = __popc (pass);
    const uint32_t incl = wave_incl_scan_dpp (cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane ((int)incl, WAVE - 1);
    if (total == 0)
      return;
    uint32_t my = incl - cnt;
    const bool tail = g * GROUP + GROUP + 8 > A.n;
    for (uint32_t base = 0; base < total; base += SH_LIST) {
      const uint32_t n_here = total - base < SH_LIST ? total - base : SH_LIST;
      {
        const uint32_t lim = base + n_here;
        while (pass != 0 && my < lim) {
          const uint32_t b = (uint32_t)__builtin_ctz (pass);
          *reinterpret_cast<__attribute__ ((address_space (3))) uint16_t *> (list_off + (my - base) * 2u) = (uint16_t)(lane * 16u + b);
          pass &= pass - 1u;
          my++;
        }
      }
      for (uint32_t off = 0; off < n_here;) {
        const uint32_t take = n_here - off < WAVE - pk ? n_here - off : WAVE - pk;
        const bool mine = lane >= pk && lane < pk + take;
        const uint32_t li = mine ? off + lane - pk : 0u;
        const uint32_t q = *reinterpret_cast<const __attribute__ ((address_space (3))) uint16_t *> (list_off + li * 2u);
        const uint32_t a = stage_off + (q & ~3u);
        const uint32_t d0 = lds_word (a), d1 = lds_word (a + 4u);
        const uint32_t lo4 = __builtin_amdgcn_alignbyte (d1, d0, q);
        uint32_t r[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const uint32_t b = (lo4 >> (8 * k)) & 0xFFu;
          r[k] = min (b - K.lo, K.span);
        }
        const uint32_t p = g * GROUP + q;
        if (tail) {
#pragma unroll
          for (int k = 0; k < 3; k++)
            if (p + k >= A.n)
              r[k] = K.span;
        }
        const uint32_t idx3 = __umul24 (__umul24 (r[0], K.W) + r[1], K.W) + r[2];
        const uint32_t y = idx3 | nibble (idx3) << 20;
        it_x = mine ? p : it_x;
        it_y = mine ? y : it_y;
        pk += take;
        off += take;
        if (pk == WAVE) {
          batch_step (WAVE);
          pk = 0;
        }
      }
    }
  };
  for (;;) {
    const uint32_t tile = share.next (next_tile, lane);
    if (tile == NONE)
      break;
    const uint32_t g0 = tile * K.R;
    uint4 c0 = load_group (g0), c1 = load_group (g0 + 1), c2 = load_group (g0 + 2), c3 = load_group (g0 + 3);
    for (uint32_t k = 0; k < K.R; k++) {
      uint4 n3;
      walk_group (c0, c1.x, c1.y, g0 + k, n3);
      c0 = c1;
      c1 = c2;
      c2 = c3;
      c3 = n3;
    }
  }
  drain ();
  if (COUNT_ONLY) {
    const uint32_t incl = wave_incl_scan ((uint32_t)counted);
    const uint32_t total = __shfl (incl, WAVE - 1, WAVE);
    if (lane == 0 && total)
      atomicAdd (E.count, (unsigned long long)total);
  } else if (lane == 0 && holes) {
    /* what is left of the wave's last chunk is a hole for close_holes_kernel */
    const WaveRec w = Ws[wib];
    const unsigned long long at = (((unsigned long long)w.base_hi << 32) | w.base_lo) + (uint32_t)counted;
    RecHole h = { (uint32_t)at, (uint32_t)(at >> 32), w.have ? REC_CHUNK - (uint32_t)counted : 0u, 0u };
    holes[wave_id] = h;
  }
}
