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
 *   - the keyword ids by RANK, from LDS: the number of set nibble bits in front of every 8 3-grams
 *     + the set bits of the word in front of the 3-gram's = where its ids are in a table that has
 *     one entry per set bit (a few KB; a keyword of d symbols is the id of W^(3-d) 3-grams).  No
 *     load from memory but the text: the first version asked HBM for 16 bytes per surviving
 *     position (W^3 entries) a batch ahead, and every wait for those was a wait for the text
 *     prefetched behind them too (vmcnt counts in order) -- 2.0 ms per 2 GiB against 1.07 ms
 *     count-only, and neither the lines asked for nor the stores were what it cost
 *     (tools/ablate_short.sh).  Dictionaries whose ids do not fit LDS read them from the image in
 *     HBM (IDS_LDS = false);
 *   - records: one reservation per batch (a prefix sum over the lanes' 0-3 records), written
 *     straight into the wave's chunk of the caller's buffer (dev_starts.h: WaveRec); the holes the
 *     waves leave are closed behind the last segment (close_holes_kernel, a second set of descriptors);
 *   - the text: four register sets that swap roles over four groups (a tile is a multiple of four
 *     groups), no copies between them -- a copy of the newest set at the top of every group had
 *     made the top wait for the load issued one group before. */
constexpr uint32_t SH_STAGE = WAVE * 16 + 16, SH_LIST = 256; /* staged text of a group + 8 bytes behind it; survivor positions per round */
constexpr uint32_t SH_WAVE_BYTES = SH_STAGE + SH_LIST * 2;
static_assert (SH_WAVE_BYTES % 16 == 0, "the staged text is written 16 bytes per lane");
constexpr uint32_t SH_CTX_BYTES = 16 + WALK_CTX_E + (SPARSE_THREADS / WAVE) * sizeof (WaveRec);
constexpr uint32_t SH_LDS_FIXED = (SPARSE_THREADS / WAVE) * SH_WAVE_BYTES + SH_CTX_BYTES; /* + the image */

/* count-only: 61 registers, two blocks a CU; with records one block (a spilled register's reload
 * sat right behind the prefetch and waited for it) */
#ifndef SH_OCCUPANCY
#define SH_OCCUPANCY __attribute__ ((amdgpu_waves_per_eu (COUNT_ONLY ? 8 : 4, 8)))
#endif
template <bool COUNT_ONLY, bool IDS_LDS>
__global__ __launch_bounds__ (SPARSE_THREADS) SH_OCCUPANCY void
scan_short_kernel (GramK K, EmitCtx E, Launch A, const unsigned char *__restrict__ text, RecHole *holes, uint32_t resume) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  constexpr uint32_t WAVES = SPARSE_THREADS / WAVE;
  constexpr uint32_t GROUP = WAVE * 16;
  /* LDS: nibbles | (records) bases | (records, IDS_LDS) ids | the waves' areas | context */
  const uint32_t base_off = K.sh_nib_bytes, ids_off = base_off + K.sh_base_bytes;
  const uint32_t image = COUNT_ONLY ? K.sh_nib_bytes : ids_off + (IDS_LDS ? K.sh_ids_bytes : 0u);
  {
    uint4 *dst = reinterpret_cast<uint4 *> (smem);
    const uint4 *src = reinterpret_cast<const uint4 *> (K.sh_img);
    for (uint32_t i = threadIdx.x; i < image / 16; i += blockDim.x)
      dst[i] = src[i];
  }
  uint32_t *next_tile = reinterpret_cast<uint32_t *> (smem + image + WAVES * SH_WAVE_BYTES);
  EmitCtx *Es = reinterpret_cast<EmitCtx *> (next_tile + 4);
  WaveRec *Ws = reinterpret_cast<WaveRec *> (reinterpret_cast<unsigned char *> (Es) + WALK_CTX_E);
  if (threadIdx.x == 0) {
    *next_tile = 0;
    *Es = E;
  }
  __syncthreads ();
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t wib = uniform (threadIdx.x / WAVE);
  const uint32_t stage_off = image + wib * SH_WAVE_BYTES, list_off = stage_off + SH_STAGE;
  const uint32_t wave_id = blockIdx.x * WAVES + wib;
  uint2 *hits = reinterpret_cast<uint2 *> (Ws + wib);
  if (lane == 0) {
    /* (as in scan_gram_kernel: no chunk yet, or the chunk a resumed segment carries over) */
    WaveRec w0{};
    if (resume && !COUNT_ONLY && holes) {
      const RecHole h = holes[wave_id];
      if (h.len) {
        const unsigned long long at = ((unsigned long long)h.start_hi << 32) | h.start_lo;
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
        w0.pad[0] = E.rec_chunk - h.len;
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
  auto load_group = [&] (uint32_t g) -> uint4 {
    const uint32_t blk = g * WAVE + lane;
    const u32x4 v = __builtin_nontemporal_load (reinterpret_cast<const u32x4 *> (text16) + (blk < last_blk ? blk : last_blk));
    return make_uint4 (v.x, v.y, v.z, v.w);
  };
  auto id_at = [&] (uint32_t rank) -> uint32_t {
    if (IDS_LDS)
      return lds_word (ids_off + rank * 4u);
    return K.sh_img[(ids_off >> 2) + rank];
  };
  /* the batch that is being put together: lanes [0, pk) hold (position, rank of the 3-gram's first id | nibble << 20) */
  uint32_t it_x = 0, it_y = 0, pk = 0;
  /* a batch's records: bit d - 1 of a lane's nibble = "the keyword of d symbols that starts here",
   * its id the (set bits below it)th behind the lane's rank */
  auto emit_batch = [&] (uint32_t n_items) {
    const uint32_t nib0 = lane < n_items ? it_y >> 20 : 0u;
    uint32_t nib = nib0;
    /* (a keyword of d symbols that starts at position p ends at p + d - 1: not before emit_from) */
    if (it_x < E.emit_from)
      nib &= it_x + 1 >= E.emit_from ? 6u : (it_x + 2 >= E.emit_from ? 4u : 0u);
    const uint32_t mine = __popc (nib);
    if (COUNT_ONLY) {
      counted += mine;
      return;
    }
    const uint32_t incl = wave_incl_scan_dpp (mine);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane ((int)incl, WAVE - 1);
    if (total == 0)
      return;
#if defined(ACM_SHORT_ABLATE) && ACM_SHORT_ABLATE == 3 /* experiment: the batch's records are counted across the wave, nothing else */
    asm volatile ("" :: "v"(incl), "v"(it_y));
    return;
#endif
    const uint32_t rank = it_y & 0xFFFFFu;
    const uint32_t r1 = rank + (nib0 & 1u), r2 = r1 + ((nib0 >> 1) & 1u);
    uint32_t k1 = 0, k2 = 0, k3 = 0;
#if defined(ACM_SHORT_ABLATE) && ACM_SHORT_ABLATE == 4 /* experiment: the ranks instead of the ids */
    k1 = rank, k2 = r1, k3 = r2;
#else
    if (nib & 1u)
      k1 = id_at (rank);
    if (nib & 2u)
      k2 = id_at (r1);
    if (nib & 4u)
      k3 = id_at (r2);
#endif
    const uint32_t used = (uint32_t)counted;
    if (used + total <= rs.limit) {
      /* ONE reservation: every lane's up to three records side by side */
      typedef uint32_t g_u32x4 __attribute__ ((ext_vector_type (4)));
      uint64_t at = rs.dst + ((uint64_t)(used + incl - mine) << 4);
#pragma unroll
      for (uint32_t d = 0; d < 3; d++) {
        if ((nib >> d) & 1u) {
          const uint64_t gp = E.pos_base + it_x + d;
#if defined(ACM_SHORT_ABLATE) && ACM_SHORT_ABLATE == 2 /* experiment: the records are put together, not written */
          asm volatile ("" :: "v"(gp), "v"(k1), "v"(k2), "v"(k3), "v"(at));
          continue;
#endif
          *reinterpret_cast<__attribute__ ((address_space (1))) g_u32x4 *> (at) =
            g_u32x4{ (uint32_t)gp, (uint32_t)(gp >> 32), d + 1, d == 0 ? k1 : (d == 1 ? k2 : k3) };
          at += 16;
        }
      }
      counted = used + total;
    } else {
      /* (the chunk ends inside the batch, or there is none yet: length by length through the path that reserves the next) */
#pragma unroll
      for (uint32_t d = 0; d < 3; d++) {
        emit_terminals<COUNT_ONLY, true> (E, ((nib >> d) & 1u) != 0, it_x + d, d == 0 ? k1 : (d == 1 ? k2 : k3), d + 1, lane, hits, counted, Es, &rs);
        counted = uniform ((uint32_t)counted);
      }
    }
  };
  auto drain = [&] () {
    if (pk) {
      emit_batch (pk);
      pk = 0;
    }
  };
  /* one group: cur = this lane's 16 bytes, (next_x, next_y) = the first 8 bytes of every lane of the
   * next group; `prefetched` takes the group three ahead (a register set of its own) */
  auto walk_group = [&] (const uint4 cur, const uint32_t next_x, const uint32_t next_y, const uint32_t g, uint4 &prefetched) {
    uint32_t t4 = (uint32_t)__builtin_amdgcn_update_dpp ((int)uniform (next_x), (int)cur.x, 0x130, 0xf, 0xf, false); /* word 0 of the next lane */
    const uint32_t t5 = (uint32_t)__builtin_amdgcn_update_dpp ((int)uniform (next_y), (int)cur.y, 0x130, 0xf, 0xf, false);
    asm volatile ("" : "+v"(t4));
    __builtin_amdgcn_sched_barrier (0);
    prefetched = load_group (g + 3);
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
    /* symbols past the end of the segment count as outside the alphabet (only the last groups: asked of
     * the group, in a scalar register -- asked of the lane it became 18 compares and selects per group) */
    if (__builtin_expect (uniform (g * GROUP + GROUP + 18) > A.n, 0)) {
#pragma unroll
      for (int j = 0; j < 18; j++)
        if (pos0 + j >= A.n)
          c[j] = K.span;
    }
    uint32_t pass = 0;
#pragma unroll
    for (int h = 0; h < 16; h += 8) { /* (two rounds of eight reads: the indices and bytes of sixteen were 32 registers) */
      uint32_t ix[8], nb[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        ix[j] = __umul24 (__umul24 (c[h + j], K.W) + c[h + j + 1], K.W) + c[h + j + 2];
        nb[j] = *reinterpret_cast<const __attribute__ ((address_space (3))) unsigned char *> (ix[j] >> 1);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t nib = (nb[j] >> ((ix[j] & 1u) * 4u)) & 7u;
        pass |= min (nib, 1u) << (h + j);
      }
      if (h == 0)
        __builtin_amdgcn_sched_barrier (0);
    }
    if (__builtin_expect (uniform (g * GROUP + GROUP) > A.n, 0)) /* (a position beyond the segment is no position) */
      pass &= pos0 + 16 <= A.n ? ~0u : (pos0 < A.n ? (1u << (A.n - pos0)) - 1u : 0u);
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
        /* the nibbles of the 8 3-grams around it (one word): its own, and how many set bits lie in front of it */
        const uint32_t word = lds_word ((idx3 >> 3) * 4u) & 0x77777777u;
        const uint32_t sh = (idx3 & 7u) * 4u;
        uint32_t y = ((word >> sh) & 7u) << 20;
        if (!COUNT_ONLY)
          y |= lds_word (base_off + (idx3 >> 3) * 4u) + __popc (word & ((1u << sh) - 1u));
        it_x = mine ? p : it_x;
        it_y = mine ? y : it_y;
        pk += take;
        off += take;
        if (pk == WAVE) {
          emit_batch (WAVE);
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
    /* (asked for in this order: the top of the loop waits for "all but the newest", whichever way it is entered) */
    uint4 c0 = load_group (g0), c3;
    __builtin_amdgcn_sched_barrier (0);
    uint4 c1 = load_group (g0 + 1);
    __builtin_amdgcn_sched_barrier (0);
    uint4 c2 = load_group (g0 + 2);
    __builtin_amdgcn_sched_barrier (0);
    for (uint32_t k = 0; k < K.R; k += 4) { /* (K.R is a multiple of 4) */
      walk_group (c0, c1.x, c1.y, g0 + k, c3);
      walk_group (c1, c2.x, c2.y, g0 + k + 1, c0);
      walk_group (c2, c3.x, c3.y, g0 + k + 2, c1);
      walk_group (c3, c0.x, c0.y, g0 + k + 3, c2);
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
    RecHole h = { (uint32_t)at, (uint32_t)(at >> 32), w.have ? E.rec_chunk - (uint32_t)counted : 0u, 0u };
    holes[wave_id] = h;
  }
}
