/* dev_sieve.h -- scan_sieve_kernel: byte alphabets, small dictionaries of keywords of 4 symbols or
 * more (config 2: the headline kernel).
 * Device code of libac75_amd.so; included by acm_gpu.hip inside its anonymous namespace (one
 * translation unit: the kernels share the structs and helpers declared there and in dev_emit.h). */

/* ------------------------------------------------------------------ trigram sieve kernel
 * What bounds scan_dense_kernel is not HBM but its one random ds_read_u16 per symbol: 64 lanes
 * hit the 32 banks of a lane group at random, ~3.3 lanes per busy bank, 6.5 LDS cycles per wave
 * instruction (PMC: SQ_LDS_BANK_CONFLICT 68 % of SQ_LDS_IDX_ACTIVE), and more streams per lane do
 * not help (S = 3 without spills: 0.289 against 0.275 ms per GiB): it is the LDS array's rate.
 * A conflict-free lookup needs every lane on a bank of its own whatever it asks for, i.e. the
 * table once PER BANK -- affordable only for a tiny table.  For a small dictionary the trigram set
 * is such a table: "some keyword starts with these 3 symbols" is W^2 words of W bits (a-z: 729
 * words = 2.9 KB, 93 KB when every word is stored 32 times, at byte address word * 128 +
 * (lane & 31) * 4), and it is selective: 979 of 19,683 trigrams on config 2 (5 %).
 *
 *   0. every position: one conflict-free ds_read_b32 on the index of its symbol pair, the bit of
 *      the third symbol; no state is carried, nothing depends on the previous symbol;
 *   1. the positions that pass are compacted into the wave's queue (a ballot per round, one
 *      candidate per lane and round) and leave it 64 at a time: the five symbols at the position
 *      are fetched again (L2: the wave has just read them), the 4-gram index formed, and from
 *      there on the candidate is a first-queue item of the 4-gram kernel: its 8-byte record
 *      {terminal | children mask, depth-4 state} (dev_gram.h, stage 2), a keyword of 4 symbols is
 *      reported on the spot, a fifth symbol that continues goes to walk_starts.
 * Text is read as in the 4-gram kernel (1 KiB groups, 16 bytes per lane, four groups in flight);
 * hits are parked per wave and become records in expand_hits_kernel.
 * LDS: [W^2 words x 32 copies][16 x first queue][16 x second queue][16 x hit buffer][tile counter, walk context]. */
struct SieveK {
  GramK G;              /* second and third stage: the 4-gram kernel's tables */
  const uint32_t *tri;  /* [W * W] bit c2 of word c0 * W + c1: some keyword starts with c0 c1 c2 */
  uint32_t tri_words;   /* W * W */
};

template <bool COUNT_ONLY>
__global__ __launch_bounds__ (SPARSE_THREADS) void
scan_sieve_kernel (SieveK S, EmitCtx E, Launch A, const unsigned char *__restrict__ text, uint2 *items, uint32_t region_items,
                   uint32_t *fill) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  constexpr uint32_t WAVES = SPARSE_THREADS / WAVE;
  constexpr uint32_t GROUP = WAVE * 16;
  const GramK &K = S.G;
  {
    /* word w, copy r at dword w * 32 + r */
    uint32_t *dst = reinterpret_cast<uint32_t *> (smem);
    for (uint32_t i = threadIdx.x; i < S.tri_words * 32u; i += blockDim.x)
      dst[i] = S.tri[i >> 5];
  }
  uint32_t *next_tile = reinterpret_cast<uint32_t *> (smem + K.queue_off + WAVES * (2 * QCAP + HITS_STRIDE) * 8);
  StartsK *Ks = reinterpret_cast<StartsK *> (next_tile + 4); /* see scan_starts_kernel */
  EmitCtx *Es = reinterpret_cast<EmitCtx *> (reinterpret_cast<unsigned char *> (Ks) + WALK_CTX_K);
  if (threadIdx.x == 0) {
    *next_tile = 0;
    StartsK Kc{};
    Kc.srec = K.srec;
    Kc.sedge = K.sedge;
    Kc.remap = K.g4gid;
    Kc.remap_base = K.d4_begin;
    *Ks = Kc;
    *Es = E;
  }
  __syncthreads ();

  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t wib = uniform (threadIdx.x / WAVE);
  uint2 *q1 = reinterpret_cast<uint2 *> (smem + K.queue_off) + wib * QCAP;
  uint2 *q2 = reinterpret_cast<uint2 *> (smem + K.queue_off) + WAVES * QCAP + wib * QCAP;
  uint2 *hits = reinterpret_cast<uint2 *> (smem + K.queue_off) + 2 * WAVES * QCAP + wib * HITS_STRIDE + 2;
  const uint32_t wave_id = blockIdx.x * WAVES + wib;
  hits_init (hits, (!COUNT_ONLY && items) ? items + (size_t)wave_id * region_items : nullptr, region_items, lane);
  const uint4 *text16 = reinterpret_cast<const uint4 *> (text);
  const uint32_t *text32 = reinterpret_cast<const uint32_t *> (text);
  const TileShare share (A);
  const uint32_t last_blk = (A.n - 1) / 16;
  const uint32_t last_word = (A.n - 1) / 4;
  const uint32_t lane_off = (lane & 31u) * 4u; /* this lane's copy of the trigram words: a bank of its own */
  uint32_t qn1 = 0, qn2 = 0;
  unsigned long long counted = 0;

  /* candidates on their way: [3] text words asked for, [2] waiting, [1] 4-gram record asked for,
   * [0] looked at next.  A batch moves one slot every time the first queue has 64 more. */
  constexpr int DEPTH = 4;
  uint32_t pend_pos[DEPTH], pend_c4[DEPTH], pend_n[DEPTH];
  uint2 pend_a[DEPTH]; /* text words, then the 4-gram record */
#pragma unroll
  for (int d = 0; d < DEPTH; d++) {
    pend_pos[d] = 0;
    pend_c4[d] = 0;
    pend_n[d] = 0;
    pend_a[d] = make_uint2 (0, 0);
  }

  auto load_group = [&] (uint32_t g) -> uint4 {
    const uint32_t blk = g * WAVE + lane;
    return text16[blk < last_blk ? blk : last_blk];
  };
  auto cls = [&] (uint32_t byte) -> uint32_t { return min (byte - K.lo, K.span); };
  auto walk_batch = [&] (uint32_t n_items) {
    const unsigned long long r = walk_starts<uint8_t, COUNT_ONLY, 1> (Ks, Es, text, q2, qn2, n_items, hits, counted);
    qn2 = uniform ((uint32_t)(r >> 32));
    counted = (uint32_t)r;
  };
  /* the oldest batch against its 4-gram records (scan_gram_kernel's second stage), then every
   * batch moves up; the one that reaches slot 1 sends for its records */
  auto advance = [&] () {
    if (pend_n[0]) {
      const bool valid = lane < pend_n[0] && pend_a[0].y != 0;
      const bool term = valid && (pend_a[0].x >> 31) && pend_pos[0] + 3 >= E.emit_from;
      emit_terminals<COUNT_ONLY> (E, term, pend_pos[0] + 3, (pend_a[0].y - K.d4_begin) | HIT_LEN4, lane, hits, counted);
      if (!COUNT_ONLY)
        counted = uniform ((uint32_t)counted);
      const bool pass = valid && ((pend_a[0].x >> pend_c4[0]) & 1u);
      const uint64_t m = __ballot (pass);
      if (m) {
        if (pass)
          q2[qn2 + rank_below (m)] = make_uint2 (pend_pos[0] + 3, pend_a[0].y | WI_REPORTED);
        qn2 = uniform (qn2 + (uint32_t)__popcll (m));
        while (qn2 >= WAVE)
          walk_batch (WAVE);
      }
    }
#pragma unroll
    for (int d = 0; d + 1 < DEPTH; d++) {
      pend_pos[d] = pend_pos[d + 1];
      pend_c4[d] = pend_c4[d + 1];
      pend_n[d] = pend_n[d + 1];
      pend_a[d] = pend_a[d + 1];
    }
    pend_n[DEPTH - 1] = 0;
    if (pend_n[1]) {
      /* the five symbols at the position: bytes p .. p + 4 of the two words */
      const uint32_t sh = (pend_pos[1] & 3u) * 8u;
      const uint64_t two = (((uint64_t)pend_a[1].y << 32) | pend_a[1].x) >> sh;
      const uint32_t w4 = (uint32_t)two;
      const uint32_t c0 = cls (w4 & 0xFFu), c1 = cls ((w4 >> 8) & 0xFFu), c2 = cls ((w4 >> 16) & 0xFFu), c3 = cls (w4 >> 24);
      const uint32_t idx = __umul24 (__umul24 (__umul24 (c0, K.W) + c1, K.W) + c2, K.W) + c3;
      /* a fifth symbol past the end of the segment is no symbol of the alphabet */
      pend_c4[1] = pend_pos[1] + 4 < A.n ? cls ((uint32_t)(two >> 32) & 0xFFu) : K.span;
      pend_a[1] = K.g4rec[lane < pend_n[1] ? idx : 0u];
    }
  };
  auto issue_batch = [&] (uint32_t n_items) {
    qn1 -= n_items;
    const uint32_t p = lane < n_items ? q1[qn1 + lane].x : 0u;
    const uint32_t w = p >> 2;
    pend_pos[DEPTH - 1] = p;
    pend_a[DEPTH - 1] = make_uint2 (text32[w < last_word ? w : last_word], text32[w + 1 < last_word ? w + 1 : last_word]);
    pend_n[DEPTH - 1] = n_items;
  };

  /* one group: cur = this lane's 16 bytes, next_x = the first 4 bytes of every lane of the next group */
  auto walk_group = [&] (const uint4 cur, const uint32_t next_x, const uint32_t g) {
    uint32_t after = __shfl_down (cur.x, 1, WAVE);
    const uint32_t after_group = uniform (next_x);
    if (lane == WAVE - 1)
      after = after_group;
    const uint32_t pos0 = g * GROUP + lane * 16;
    const uint32_t w[5] = { cur.x, cur.y, cur.z, cur.w, after };
    uint32_t c[18];
#pragma unroll
    for (int j = 0; j < 18; j++)
      c[j] = cls ((w[j / 4] >> (8 * (j % 4))) & 0xFFu);
    /* symbols past the end of the segment count as outside the alphabet (only the last groups) */
    if (pos0 + 18 > A.n) {
#pragma unroll
      for (int j = 0; j < 18; j++)
        if (pos0 + j >= A.n)
          c[j] = K.span;
    }
    /* all 16 words are asked for before any is looked at; bit j of mask: position j passes */
    uint32_t word[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t pair = __umul24 (c[j], K.W) + c[j + 1];
      word[j] = *reinterpret_cast<const __attribute__ ((address_space (3))) uint32_t *> ((pair << 7) + lane_off);
    }
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < 16; j++)
      mask |= __builtin_amdgcn_ubfe (word[j], c[j + 2], 1u) << j;
#ifdef ACM_SIEVE_STAGE0_ONLY /* experiment: what the first stage alone costs (the candidates are only counted) */
    counted += __popc (mask);
    return;
#endif
    /* compaction: one candidate per lane and round */
    for (;;) {
      const uint64_t m = __ballot (mask != 0);
      if (!m)
        break;
      if (mask) {
        const uint32_t b = (uint32_t)__builtin_ctz (mask);
        mask &= mask - 1u;
        q1[qn1 + rank_below (m)] = make_uint2 (pos0 + b, 0u);
      }
      qn1 = uniform (qn1 + (uint32_t)__popcll (m));
      if (qn1 >= WAVE) {
        advance ();
        issue_batch (WAVE);
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
      const uint4 n3 = load_group (g0 + k + 4);
      walk_group (c0, c1.x, g0 + k);
      c0 = c1;
      c1 = c2;
      c2 = c3;
      c3 = n3;
    }
  }
  if (qn1) {
    advance ();
    issue_batch (qn1);
  }
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
    advance ();
  while (qn2)
    walk_batch (qn2 < WAVE ? qn2 : WAVE);
  if (COUNT_ONLY) {
    const uint32_t incl = wave_incl_scan ((uint32_t)counted);
    const uint32_t total = __shfl (incl, WAVE - 1, WAVE);
    if (lane == 0 && total)
      atomicAdd (E.count, (unsigned long long)total);
  } else {
    if (counted)
      flush_hits (E, hits, (uint32_t)counted, lane);
    if (lane == 0 && fill)
      fill[wave_id] = hits[-1].y;
  }
}
