/* dev_starts.h -- scan_starts_kernel: start-parallel kernel for 2- and 4-byte symbols; walk_starts, hit parking, expand_hits_kernel.
 * Device code of libac75_amd.so; included by acm_gpu.hip inside its anonymous namespace (one
 * translation unit: the kernels share the structs and helpers declared there and in dev_emit.h). */

/* ------------------------------------------------------------------ start-parallel kernel (2- and 4-byte symbols)
 * The same match set, computed without a state carried from symbol to symbol: a keyword occurs
 * at [i, i + L) iff the goto function alone (the trie, no failure transitions) leads from the root
 * through text[i .. i + L) to its terminal state.  With a large alphabet nearly every start dies
 * at once, so instead of one dependent table lookup per symbol (the sparse walk above waits
 * ~2.7 us of memory latency per step) every position is tested independently, in three sieves:
 *   1. one LDS lookup per symbol in the root table: child state | ALWAYS << 31 | SECOND << 30
 *      (SECOND: the symbol is the second symbol of some keyword).  A start survives if its
 *      symbol has a child and the next symbol has SECOND (config 5: 7% of the positions);
 *   2. the two smallest edge symbols of the child (8 bytes, a 69 KB table on config 5) against
 *      the next symbol -- the load is issued at once and looked at one group (1 KiB of text)
 *      later, so nobody waits for it.  Children that are keywords themselves or have more than
 *      two edges carry ALWAYS and pass both sieves unseen;
 *   3. what is left (config 5: 3 starts per 10,000 symbols) is queued per wave and walked down
 *      the trie 64 at a time (walk_starts); terminal states give records (end position, depth,
 *      keyword id).
 * The text is read the coalesced way: a wave takes 1 KiB groups, lane l the 16 bytes at 16 l,
 * four groups walked while the next four are in flight.  Every record belongs to the start
 * position that finds it: nothing is warmed up, nothing is found twice.  Worst case (every start
 * walks lmax symbols) is lmax dependent loads per symbol; ACM_GPU_SPARSE=walk selects the sparse
 * automaton walk instead. */
struct StartsK {
  const uint4 *srec;   /* 2 x uint4 per state: {-, n_edges, edge_begin, terminal} {sym0, next0, sym1, next1} */
  const uint2 *sedge;  /* per goto edge, rows in ascending symbol order: {symbol, next} */
  const uint2 *pairs;  /* states 0 .. root fan-out: the symbols of the first two edges (repeated / 0 when fewer) */
  const uint32_t *lut; /* by symbol value: child | SECOND << 30 | ALWAYS << 31 */
  uint32_t lut_size;
  uint32_t R;          /* groups per tile, a multiple of 4 */
  uint32_t queue_off;  /* LDS: [lut if staged][16 queues of 128][16 hit buffers of 64][tile counter] */
  /* 4-gram kernel only: its records are laid out depth-first below depth 4 (a keyword's tail in
   * consecutive records); an item names a depth-4 state by its breadth-first id, remap[id -
   * remap_base] is its record, and word 0 of a record is the state's breadth-first id */
  const uint32_t *remap;
  uint32_t remap_base;
  /* 4-gram kernel, narrow alphabets (walk_starts<.., 2>): an item without WI_RECORD names a
   * depth-5 state; peek[id - remap_base] = {its record, the symbol of its only edge or
   * GRAM_NO_PEEK} */
  const uint32_t *peek;
  uint32_t peek_packed; /* 4 bytes per state: record | symbol << 23 | "look at the record" << 31; else {record, symbol or GRAM_NO_PEEK} */
  /* walk_starts<.., 2>: a fresh item's second word is (depth-5 state - remap_base) | class of the
   * symbol after its position << 24 (31: not known), and [lo, lo + span) is the alphabet */
  uint32_t peek_rel, lo, span;
};
constexpr uint32_t ST_ALWAYS = 0x80000000u, ST_SECOND = 0x40000000u, ST_STATE = 0x3FFFFFFFu;

/* Matches found by a wave collect in its LDS hit buffer as (end position, terminal state) and
 * leave up to 64 at a time for the wave's private region of the plan's item buffer in HBM (plain
 * coalesced stores); expand_hits_kernel turns the regions into records afterwards with one atomic
 * per block.  (One atomic on the record counter per find was 0.7 ms of serialised atomics on
 * config 5, one per 64 finds still 6 ms on config 3's 27 M matches: a single address sustains
 * ~90 atomics per microsecond.)  A full region falls back to records straight from here.
 * LDS per wave: [region pointer][capacity, fill][64 hits]. */
constexpr uint32_t HITS_STRIDE = WAVE + 2; /* in 8-byte units */

__device__ __forceinline__ void
hits_init (uint2 *hits, uint2 *region, uint32_t capacity, uint32_t lane) {
  if (lane == 0) {
    const uint64_t a = reinterpret_cast<uint64_t> (region);
    hits[-2] = make_uint2 ((uint32_t)a, (uint32_t)(a >> 32));
    hits[-1] = make_uint2 (region ? capacity : 0u, 0u);
  }
}

/* the record of a hit: (length, keyword id) of the terminal state it names -- its first output is
 * its own keyword -- or, for a keyword of 4 symbols found by a 4-gram kernel, the keyword the hit
 * carries (HIT_KW) or (HIT_LEN4: the depth-4 state by rank) from the
 * small table of the depth-4 states */
__device__ __forceinline__ void
write_hit_record (const EmitCtx &E, uint2 h, unsigned long long slot) {
  uint32_t length, kw;
  if (h.y & HIT_KW) {
    length = (h.y >> 28) & 3u ? (h.y >> 28) & 3u : 4u;
    kw = h.y & HIT_KW_ID;
  } else if (h.y & HIT_LEN4) {
    length = 4;
    kw = E.kw4[h.y & ~HIT_LEN4];
  } else {
    const uint4 oi = E.oinfo[h.y];
    length = oi.z;
    kw = oi.w;
  }
  const uint64_t gp = E.pos_base + h.x;
  *reinterpret_cast<uint4 *> (&E.records[slot]) = make_uint4 ((uint32_t)gp, (uint32_t)(gp >> 32), length, kw);
}

__device__ __forceinline__ void
flush_hits (const EmitCtx &E, uint2 *hits, uint32_t n, uint32_t lane) {
  const uint2 rp = hits[-2], cf = hits[-1];
  if (cf.y + n <= cf.x) {
    /* (a global store, not a flat one through a generic pointer: see emit_terminals) */
    typedef uint32_t g_u32x2 __attribute__ ((ext_vector_type (2)));
    const uint64_t region = ((uint64_t)rp.y << 32) | rp.x;
    if (lane < n) {
      const uint2 h = hits[lane];
      *reinterpret_cast<__attribute__ ((address_space (1))) g_u32x2 *> (region + (uint64_t)(cf.y + lane) * 8u) = g_u32x2{ h.x, h.y };
    }
    if (lane == 0)
      hits[-1] = make_uint2 (cf.x, cf.y + n);
    return;
  }
  unsigned long long base = 0;
  if (lane == 0)
    base = atomicAdd (E.count, (unsigned long long)n);
  base = ((unsigned long long)__shfl ((uint32_t)(base >> 32), 0, WAVE) << 32) | __shfl ((uint32_t)base, 0, WAVE);
  if (lane < n && base + lane < E.capacity)
    write_hit_record (E, hits[lane], base + lane);
}

/* one record per parked hit; a block takes REGIONS consecutive regions and reserves their
 * records with one atomic; it zeroes the fill counters it consumed */
template <int THREADS, int REGIONS>
__global__ __launch_bounds__ (THREADS) void
expand_hits_kernel (EmitCtx E, const uint2 *items, uint32_t region_items, uint32_t *fill, uint32_t n_regions) {
  __shared__ uint32_t s_off[REGIONS + 1];
  __shared__ unsigned long long s_base;
  const uint32_t tid = threadIdx.x;
  const uint32_t r0 = blockIdx.x * REGIONS;
  if (tid < REGIONS) {
    const uint32_t r = r0 + tid;
    s_off[tid + 1] = r < n_regions ? fill[r] : 0;
    if (r < n_regions)
      fill[r] = 0;
  }
  __syncthreads ();
  if (tid == 0) {
    uint32_t acc = 0;
    for (int r = 0; r < REGIONS; r++) {
      const uint32_t v = s_off[r + 1];
      s_off[r] = acc;
      acc += v;
    }
    s_off[REGIONS] = acc;
    s_base = acc ? atomicAdd (E.count, (unsigned long long)acc) : 0ull;
  }
  __syncthreads ();
  const uint32_t total = s_off[REGIONS];
  const unsigned long long base = s_base;
  /* four hits per thread and round: their loads are independent, so four item loads and then four
   * table lookups are in flight per thread instead of one (two blocks of 1,024 threads per CU with
   * one hit each left the kernel waiting on memory latency: 0.68 ms for the 54 M hits of a 2 GiB
   * segment of config 3, which is 1.3 GB of traffic) */
  constexpr int UNROLL = 4;
  for (uint32_t i0 = tid; i0 < total; i0 += THREADS * UNROLL) {
    uint2 h[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint32_t i = i0 + u * THREADS;
      h[u] = make_uint2 (0, 0);
      if (i < total) {
        uint32_t r = 0;
#pragma unroll
        for (int k = 1; k < REGIONS; k++)
          r += s_off[k] <= i ? 1u : 0u;
        h[u] = items[(size_t)(r0 + r) * region_items + (i - s_off[r])];
      }
    }
    uint32_t length[UNROLL], kw[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint32_t i = i0 + u * THREADS;
      length[u] = 4;
      kw[u] = 0;
      if (i < total && base + i < E.capacity) {
        if (h[u].y & HIT_KW) {
          length[u] = (h[u].y >> 28) & 3u ? (h[u].y >> 28) & 3u : 4u;
          kw[u] = h[u].y & HIT_KW_ID;
        } else if (h[u].y & HIT_LEN4)
          kw[u] = E.kw4[h[u].y & ~HIT_LEN4];
        else {
          const uint4 oi = E.oinfo[h[u].y]; /* terminal state: its first output is its own keyword */
          length[u] = oi.z;
          kw[u] = oi.w;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint32_t i = i0 + u * THREADS;
      if (i < total && base + i < E.capacity) {
        const uint64_t gp = E.pos_base + h[u].x;
        *reinterpret_cast<uint4 *> (&E.records[base + i]) = make_uint4 ((uint32_t)gp, (uint32_t)(gp >> 32), length[u], kw[u]);
      }
    }
  }
}

/* ---- records straight from the 4-gram kernel (narrow alphabets; DIRECT below)
 * Round 2 parked 8-byte hits in a region per wave and had expand_hits_kernel turn them into
 * 16-byte records afterwards: 0.30 ms beside 2.1 ms of scanning per 2 GiB of config 3, 8 B written
 * + 8 B read back per record.  Now a wave writes the records themselves, into chunks of
 * REC_CHUNK slots of the caller's buffer that it reserves with one atomicAdd on the record counter
 * (13 per wave and launch on config 3; a single address takes ~90 per microsecond).  What a wave
 * leaves unused of its last chunk is a hole: it reports {first slot, length} in RecHole and
 * close_holes_kernel moves the records that lie beyond the dense count into the holes and rewinds
 * the counter (at most one chunk per wave).  The caller's buffer may be exactly as long as the
 * number of matches: slots past its capacity are kept in the plan's spill area (one chunk per
 * wave), which the closing pass reads from -- so nothing is dropped because of holes.
 * Per wave in LDS (WaveRec): where slot 0 of the current chunk is (in the caller's buffer or in the
 * spill area), the chunk's first slot, and how many slots the fast path may use. */
#ifndef ACM_REC_CHUNK
#define ACM_REC_CHUNK 1024
#endif
constexpr uint32_t REC_CHUNK = ACM_REC_CHUNK, REC_CHUNK_BIG = 4096;
constexpr uint32_t HOLE_BITS = 13; /* close_holes_kernel's sort keys: first slot << HOLE_BITS | length */
static_assert (REC_CHUNK <= REC_CHUNK_BIG && REC_CHUNK_BIG < (1u << HOLE_BITS) && REC_CHUNK % 1024 == 0, "a hole's length fits its key");
struct WaveRec {
  uint32_t dst_lo, dst_hi;   /* address of slot 0 of the current chunk (meaningless while limit == 0) */
  uint32_t base_lo, base_hi; /* index of that slot */
  uint32_t limit;            /* REC_CHUNK for a chunk that lies entirely on one side of the capacity, else 0: slow path only */
  uint32_t have;             /* a chunk has been reserved */
  uint32_t pad[2];           /* [0] resumed scans: slots used of the chunk that is carried over; [1] chunks reserved so far */
  /* tiled scans: the tile at hand, the wave's stream index (records written so far, up to a
   * constant) when it began and when its last group began -- kept here, not in registers: the
   * scan kernel has none to spare */
  uint32_t tile, s_begin, s_late;
  uint32_t prev1, prev2;     /* the two chunks filled before the current one (NONE: not known) */
  uint32_t pad2;
};
/* A tiled scan (the 4-gram kernel with a directory: acm_gpu_scan_ordered_device) empties its
 * queues at the end of every tile, so that the records of a tile lie side by side in the wave's
 * stream of chunks, and says where: the slot behind the tile's last record, how many records, how
 * many of them were written after the tile's last group began (those that END beyond the tile --
 * they belong to the next tile's stretch of the output -- are among these), and the range of end
 * positions the tile owns. */
struct TileEntry {
  uint64_t end_slot;
  uint64_t lo, hi;
  uint32_t n, n_late;
  uint32_t c1, c2;           /* the chunks in front of end_slot's (NONE: follow chunk_prev) -- most tiles end within three */
  uint32_t pad[2];
};
struct RecHole {
  uint32_t start_lo, start_hi, len, pad;
};

/* where record `slot` goes: the caller's buffer below the capacity, the spill area for the next
 * spill_slots, nowhere beyond (a true overflow: the count alone tells) */
__device__ __forceinline__ uint4 *
record_address (const EmitCtx &E, unsigned long long slot) {
  if (slot < E.capacity)
    return reinterpret_cast<uint4 *> (&E.records[slot]);
  if (slot - E.capacity < E.spill_slots)
    return E.spill + (slot - E.capacity);
  return nullptr;
}

/* the batch does not fit what is left of the chunk (or the chunk straddles the capacity): out of
 * line.  Fills the chunk, reserves the next one, returns the slots used in the current chunk. */
__device__ __noinline__ uint32_t
emit_records_slow (const EmitCtx *Ep, WaveRec *W, uint32_t used, bool hit, uint32_t p, uint32_t length, uint32_t kw) {
  const EmitCtx &E = *Ep;
  const uint64_t m = __ballot (hit);
  const uint32_t total = (uint32_t)__popcll (m), rank = rank_below (m);
  const uint32_t lane = lane_id ();
  unsigned long long base = ((unsigned long long)W->base_hi << 32) | W->base_lo;
  const uint32_t chunk = E.rec_chunk;
  const uint32_t room = W->have ? chunk - used : 0u;
  const uint64_t gp = E.pos_base + p;
  const uint4 rec = make_uint4 ((uint32_t)gp, (uint32_t)(gp >> 32), length, kw);
  if (hit && rank < room) {
    uint4 *a = record_address (E, base + used + rank);
    if (a)
      *a = rec;
  }
  if (total <= room)
    return used + total;
  /* next chunk: one atomic for the wave */
  unsigned long long nb = 0;
  if (lane == 0)
    nb = atomicAdd (E.count, (unsigned long long)chunk);
  nb = ((unsigned long long)__shfl ((uint32_t)(nb >> 32), 0, WAVE) << 32) | __shfl ((uint32_t)nb, 0, WAVE);
  if (hit && rank >= room) {
    uint4 *a = record_address (E, nb + (rank - room));
    if (a)
      *a = rec;
  }
  if (lane == 0) {
    const bool below = nb + chunk <= E.capacity;
    const bool above = nb >= E.capacity && nb - E.capacity + chunk <= E.spill_slots;
    const uint64_t dst = below ? reinterpret_cast<uint64_t> (&E.records[nb]) : (above ? reinterpret_cast<uint64_t> (E.spill + (nb - E.capacity)) : 0ull);
    W->dst_lo = (uint32_t)dst;
    W->dst_hi = (uint32_t)(dst >> 32);
    W->base_lo = (uint32_t)nb;
    W->base_hi = (uint32_t)(nb >> 32);
    W->limit = (below || above) ? chunk : 0u;
    if (E.chunk_prev) { /* (tiled scans: chunks of REC_CHUNK) */
      const uint32_t was = W->have ? (uint32_t)(base / REC_CHUNK) : NONE;
      if (below)
        E.chunk_prev[nb / REC_CHUNK] = was;
      W->prev2 = W->prev1;
      W->prev1 = was;
    }
    W->have = 1;
    W->pad[1]++; /* chunks reserved so far (a tiled scan's stream index: TileEntry) */
  }
  return total - room;
}

/* Closes the holes the waves of a scan kernel left in their last chunks (RecHole, one per wave):
 * with T = the counter (slots reserved so far) and H = the holes' total length, the records are
 * the filled slots of [0, T) and there are C = T - H of them; those at C or beyond (in the
 * caller's buffer or, past its capacity, in the spill area) move into the holes below C, the k-th
 * hole slot taking the k-th such record, and the counter is rewound to C.
 * Every block sorts the holes by first slot in LDS (keys of start << HOLE_BITS | length, counted into place),
 * takes their prefix sums P, and handles the holes b, b + gridDim.x, ...: target = start_i + t is
 * hole slot number k = P[i] + t; its source is the filled slot of rank (C - M) + k (M = hole slots
 * below C), found by bisection on G[i] = start_i - P[i] = filled slots in front of hole i. */
constexpr int CLOSE_THREADS = 1024;
__global__ __launch_bounds__ (CLOSE_THREADS) void
close_holes_kernel (EmitCtx E, const RecHole *holes, uint32_t n_waves, uint32_t npow, unsigned int *ticket, uint32_t network_only) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  unsigned long long *key = reinterpret_cast<unsigned long long *> (smem); /* [npow] */
  uint32_t *P = reinterpret_cast<uint32_t *> (smem + (size_t)npow * 8);    /* [npow + 1] exclusive prefix of the lengths, then spare */
  __shared__ unsigned long long s_T;
  __shared__ uint32_t s_part[CLOSE_THREADS / WAVE];
  __shared__ uint32_t s_M;
  const uint32_t tid = threadIdx.x;
  constexpr unsigned long long PAD = ~0ull << HOLE_BITS;
  constexpr uint32_t HOLE_MASK = (1u << HOLE_BITS) - 1;
  /* The sort.  A bitonic network over 4,096 keys is 78 passes with a barrier each: 67 of the kernel's
   * 107 us (an ablation build that sorted and moved nothing).  The keys are distinct and spread
   * over a known range, so they are COUNTED into place instead: npow buckets by first slot (min
   * and max by a reduction; a monotone map), an LDS atomic per key for its bucket's count and its
   * arrival number, one prefix sum, every key written at its bucket's base + arrival number, and
   * the few keys that share a bucket ranked among themselves -- five barriers.  The waves' last
   * chunks cluster at the end of the range (a straggler's chunk far in front stretches it), which
   * crowds buckets: past 64 keys in one the network below sorts instead. */
  constexpr uint32_t PER_MAX = 8;
  __shared__ unsigned long long s_lo[CLOSE_THREADS / WAVE], s_hi[CLOSE_THREADS / WAVE];
  __shared__ uint32_t s_crowd;
  unsigned long long kreg[PER_MAX];
  unsigned long long klo = ~0ull, khi = 0;
#pragma unroll
  for (uint32_t q = 0; q < PER_MAX; q++) {
    const uint32_t i = tid + q * CLOSE_THREADS;
    unsigned long long k = PAD;
    if (i < n_waves && i < npow) {
      const RecHole h = holes[i];
      if (h.len)
        k = (((unsigned long long)h.start_hi << 32 | h.start_lo) << HOLE_BITS) | h.len;
    }
    kreg[q] = k;
    if (i < npow)
      key[i] = k;
    if (k != PAD) {
      klo = (k >> HOLE_BITS) < klo ? (k >> HOLE_BITS) : klo;
      khi = (k >> HOLE_BITS) > khi ? (k >> HOLE_BITS) : khi;
    }
  }
  for (uint32_t i = tid + PER_MAX * CLOSE_THREADS; i < npow; i += CLOSE_THREADS) { /* (more holes than the counting takes: the network) */
    unsigned long long k = PAD;
    if (i < n_waves) {
      const RecHole h = holes[i];
      if (h.len)
        k = (((unsigned long long)h.start_hi << 32 | h.start_lo) << HOLE_BITS) | h.len;
    }
    key[i] = k;
  }
  uint32_t *hist = P; /* [npow + 1] while the keys are being sorted (P is made afterwards) */
  for (uint32_t i = tid; i <= npow; i += CLOSE_THREADS)
    hist[i] = 0;
  for (int d = 1; d < WAVE; d <<= 1) {
    const unsigned long long ol = __shfl_xor (klo, d, WAVE), oh = __shfl_xor (khi, d, WAVE);
    klo = ol < klo ? ol : klo;
    khi = oh > khi ? oh : khi;
  }
  if ((tid & (WAVE - 1)) == 0) {
    s_lo[tid / WAVE] = klo;
    s_hi[tid / WAVE] = khi;
  }
  if (tid == 0) {
    s_T = __hip_atomic_load (E.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_crowd = 0;
  }
  __syncthreads ();
  bool counted = npow <= CLOSE_THREADS * PER_MAX && !network_only; /* (ACM_GPU_CLOSE_SORT=network: tests of the fallback) */
  if (counted) {
    for (int w = 0; w < CLOSE_THREADS / WAVE; w++) {
      klo = s_lo[w] < klo ? s_lo[w] : klo;
      khi = s_hi[w] > khi ? s_hi[w] : khi;
    }
    const double scale = khi >= klo ? (double)npow / (double)(khi - klo + 1) : 0.0;
    uint32_t bkt[PER_MAX], slot[PER_MAX];
#pragma unroll
    for (uint32_t q = 0; q < PER_MAX; q++) {
      bkt[q] = slot[q] = 0;
      if (kreg[q] != PAD) {
        const uint32_t b = (uint32_t)((double)((kreg[q] >> HOLE_BITS) - klo) * scale);
        bkt[q] = b < npow - 1 ? b : npow - 1;
        slot[q] = atomicAdd (&hist[bkt[q]], 1u);
      }
    }
    __syncthreads ();
    /* exclusive prefix of the counts in place (hist[npow] = the number of holes), the fullest bucket */
    const uint32_t per = npow >= CLOSE_THREADS ? npow / CLOSE_THREADS : 1;
    uint32_t run = 0, most = 0;
    for (uint32_t q = 0; q < per; q++) {
      const uint32_t i = tid * per + q;
      if (i < npow) {
        run += hist[i];
        most = hist[i] > most ? hist[i] : most;
      }
    }
    const uint32_t incl = wave_incl_scan (run);
    if ((tid & (WAVE - 1)) == WAVE - 1)
      s_part[tid / WAVE] = incl;
    if (most > 64)
      s_crowd = 1; /* (benign race: every writer writes 1) */
    __syncthreads ();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < CLOSE_THREADS / WAVE; w++) {
      before += w < tid / WAVE ? s_part[w] : 0u;
      all += s_part[w];
    }
    counted = s_crowd == 0;
    __syncthreads (); /* (s_part is used again below; hist is rewritten) */
    if (counted) {
      uint32_t acc = before + incl - run;
      for (uint32_t q = 0; q < per; q++) {
        const uint32_t i = tid * per + q;
        if (i < npow) {
          const uint32_t c = hist[i];
          hist[i] = acc;
          acc += c;
        }
      }
      if (tid == 0)
        hist[npow] = all;
      __syncthreads ();
#pragma unroll
      for (uint32_t q = 0; q < PER_MAX; q++)
        if (kreg[q] != PAD)
          key[hist[bkt[q]] + slot[q]] = kreg[q];
      for (uint32_t i = all + tid; i < npow; i += CLOSE_THREADS)
        key[i] = PAD;
      __syncthreads ();
      uint32_t place[PER_MAX];
#pragma unroll
      for (uint32_t q = 0; q < PER_MAX; q++) {
        place[q] = 0;
        if (kreg[q] != PAD) {
          const uint32_t base = hist[bkt[q]], end = hist[bkt[q] + 1];
          uint32_t r = 0;
          for (uint32_t x = base; x < end; x++)
            r += key[x] < kreg[q] ? 1u : 0u;
          place[q] = base + r;
        }
      }
      __syncthreads ();
#pragma unroll
      for (uint32_t q = 0; q < PER_MAX; q++)
        if (kreg[q] != PAD)
          key[place[q]] = kreg[q];
      __syncthreads ();
    }
  }
  for (uint32_t k = 2; !counted && k <= npow; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < npow; i += CLOSE_THREADS) {
        const uint32_t o = i ^ j;
        if (o > i) {
          const unsigned long long a = key[i], b = key[o];
          if ((a > b) == ((i & k) == 0)) {
            key[i] = b;
            key[o] = a;
          }
        }
      }
      __syncthreads ();
    }
  /* exclusive prefix of the lengths: every thread sums a run of npow / THREADS (>= 1) keys,
   * the runs' sums are scanned by wave and block */
  const uint32_t per = npow >= CLOSE_THREADS ? npow / CLOSE_THREADS : 1;
  uint32_t run = 0;
  for (uint32_t q = 0; q < per; q++) {
    const uint32_t i = tid * per + q;
    if (i < npow)
      run += (uint32_t)(key[i] & HOLE_MASK);
  }
  const uint32_t incl = wave_incl_scan (run);
  if ((tid & (WAVE - 1)) == WAVE - 1)
    s_part[tid / WAVE] = incl;
  __syncthreads ();
  if (tid == 0) {
    uint32_t acc = 0;
    for (int w = 0; w < CLOSE_THREADS / WAVE; w++) {
      const uint32_t v = s_part[w];
      s_part[w] = acc;
      acc += v;
    }
    P[npow] = acc;
  }
  __syncthreads ();
  {
    uint32_t acc = s_part[tid / WAVE] + incl - run;
    for (uint32_t q = 0; q < per; q++) {
      const uint32_t i = tid * per + q;
      if (i < npow) {
        P[i] = acc;
        acc += (uint32_t)(key[i] & HOLE_MASK);
      }
    }
  }
  __syncthreads ();
  const unsigned long long T = s_T, H = P[npow], C = T - H;
  /* M: hole slots below C */
  uint32_t mine = 0;
  for (uint32_t i = tid; i < npow; i += CLOSE_THREADS) {
    const unsigned long long st = key[i] >> HOLE_BITS;
    const uint32_t len = (uint32_t)(key[i] & HOLE_MASK);
    if (len && st < C)
      mine += C - st < len ? (uint32_t)(C - st) : len;
  }
  const uint32_t mi = wave_incl_scan (mine);
  __syncthreads ();
  if ((tid & (WAVE - 1)) == WAVE - 1)
    s_part[tid / WAVE] = mi;
  __syncthreads ();
  if (tid == 0) {
    uint32_t acc = 0;
    for (int w = 0; w < CLOSE_THREADS / WAVE; w++)
      acc += s_part[w];
    s_M = acc;
  }
  __syncthreads ();
  const unsigned long long first_rank = C - s_M; /* rank (among the filled slots) of the first record at C or beyond */
  /* The move: hole slot number k (k-th slot of the holes below C, in slot order) takes the filled
   * slot of rank first_rank + k.  The M slots are dealt out in equal stretches, one per wave of the
   * grid (by hole -- a block per hole -- the blocks whose holes lie beyond C had nothing to do
   * and the others 16 holes of up to 4,095 slots one after the other, a bisection per slot):
   * a wave finds the hole of its first slot and the holes in front of its first source by
   * bisection ONCE, then every lane walks on from where its last slot was (k and the rank only
   * grow), four slots in flight per lane. */
  const uint32_t M = s_M;
#if defined(ACM_CLOSE_ABLATE) && ACM_CLOSE_ABLATE == 1 /* experiment: the holes sorted and summed, nothing moved */
  if (M)
    goto moved;
#endif
  {
    const uint32_t lane = tid & (WAVE - 1);
    const unsigned long long gw = (unsigned long long)blockIdx.x * (CLOSE_THREADS / WAVE) + tid / WAVE, GW = (unsigned long long)gridDim.x * (CLOSE_THREADS / WAVE);
    const uint32_t k0 = (uint32_t)((unsigned long long)M * gw / GW), k1 = (uint32_t)((unsigned long long)M * (gw + 1) / GW);
    auto real = [&] (uint32_t x) -> bool { return x < npow && (key[x] & HOLE_MASK) != 0; };
    auto G = [&] (uint32_t x) -> unsigned long long { return (key[x] >> HOLE_BITS) - P[x]; };
    if (k0 < k1) {
      /* hh: the hole that holds slot k0 (the last one with P <= k0 among the holes that have slots) */
      uint32_t lo = 0, hi = npow;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (real (mid) && P[mid] <= k0)
          lo = mid + 1;
        else
          hi = mid;
      }
      uint32_t hh = lo - 1; /* (k0 < M: hole 0 starts at P = 0 <= k0, so lo >= 1) */
      /* gg: the holes in front of the filled slot of rank first_rank + k0: those with G = start - P <= rank (G ascends) */
      const unsigned long long r0 = first_rank + k0;
      lo = 0, hi = npow;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (real (mid) && G (mid) <= r0) /* (the padding sorts last and is in front of nothing) */
          lo = mid + 1;
        else
          hi = mid;
      }
      uint32_t gg = lo;
      /* where slot k goes and where its record comes from (a lane without a slot reads something
       * harmless: a load under a condition, or an array of the four in flight, went to scratch memory) */
      auto prep = [&] (uint32_t k, unsigned long long &to, bool &ok) -> const uint4 * {
        const uint4 *from = reinterpret_cast<const uint4 *> (holes);
        ok = false;
        to = 0;
        if (k < k1) {
          while (real (hh + 1) && P[hh + 1] <= k)
            hh++;
          to = (key[hh] >> HOLE_BITS) + (k - P[hh]);
          const unsigned long long r = first_rank + k;
          while (real (gg) && G (gg) <= r)
            gg++;
          const uint4 *at = record_address (E, r + P[gg]);
          ok = at != nullptr && to < E.capacity;
          from = ok ? at : from;
        }
        return from;
      };
      for (uint32_t base = k0 + lane; base < k1; base += WAVE * 4) {
        unsigned long long t0, t1, t2, t3;
        bool o0, o1, o2, o3;
        const uint4 *f0 = prep (base, t0, o0), *f1 = prep (base + WAVE, t1, o1), *f2 = prep (base + 2 * WAVE, t2, o2), *f3 = prep (base + 3 * WAVE, t3, o3);
        const uint4 v0 = *f0, v1 = *f1, v2 = *f2, v3 = *f3;
        if (o0)
          *reinterpret_cast<uint4 *> (&E.records[t0]) = v0;
        if (o1)
          *reinterpret_cast<uint4 *> (&E.records[t1]) = v1;
        if (o2)
          *reinterpret_cast<uint4 *> (&E.records[t2]) = v2;
        if (o3)
          *reinterpret_cast<uint4 *> (&E.records[t3]) = v3;
      }
    }
  }
#if defined(ACM_CLOSE_ABLATE) && ACM_CLOSE_ABLATE == 1
moved:
#endif
  /* the block that finishes last rewinds the counter: every block has read T by then */
  __syncthreads ();
  if (tid == 0) {
    if (atomicAdd (ticket, 1u) == gridDim.x - 1) {
      *ticket = 0;
      __hip_atomic_store (E.count, C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

/* tally: this lane's finds (count-only mode; summed over the wave at the end of the kernel), the
 * fill of the LDS hit buffer (record mode, wave-uniform), or -- DIRECT (4-gram kernel, narrow
 * alphabets) -- the slots used of the wave's current chunk of records: `hits` is then the wave's
 * WaveRec in LDS, `kw` / `length` the record's second half (other modes: `kw` is the hit's word --
 * a terminal state or a HIT_KW / HIT_LEN4 word -- and `length` is not looked at); Ep = the LDS copy
 * of E for the out-of-line path. */
/* the wave's chunk state as the scan kernel keeps it in scalar registers between batches (a
 * pipeline step that asked LDS for it every time paid an LDS round trip per batch: the scan kernel
 * 2.12 -> 2.36 ms per 2 GiB of config 3); reloaded from the WaveRec after the out-of-line paths */
struct RecState {
  unsigned long long dst; /* address of slot 0 of the current chunk */
  uint32_t limit;
};
__device__ __forceinline__ RecState
rec_state_load (const uint2 *hits) {
  const WaveRec *W = reinterpret_cast<const WaveRec *> (hits);
  RecState r;
  r.dst = ((unsigned long long)uniform (W->dst_hi) << 32) | uniform (W->dst_lo);
  r.limit = uniform (W->limit);
  return r;
}

template <bool COUNT_ONLY, bool DIRECT = false>
__device__ __forceinline__ void
emit_terminals (const EmitCtx &E, bool hit, uint32_t p, uint32_t kw, uint32_t length, uint32_t lane, uint2 *hits, unsigned long long &tally,
                const EmitCtx *Ep = nullptr, RecState *rs = nullptr) {
  if (COUNT_ONLY) {
    tally += hit ? 1u : 0u;
    return;
  }
  const uint64_t m = __ballot (hit);
  if (m) {
    const uint32_t total = (uint32_t)__popcll (m);
    uint32_t hn = (uint32_t)tally;
    if (DIRECT) {
      WaveRec *W = reinterpret_cast<WaveRec *> (hits);
      RecState here = rs ? *rs : rec_state_load (hits);
      if (hn + total <= here.limit) {
        /* the chunk's slot 0 + (used + rank) * 16: a wave-uniform base and a 32-bit offset */
        if (hit) {
          const uint64_t gp = E.pos_base + p;
          const uint32_t off = (hn + rank_below (m)) << 4;
#ifdef ACM_GRAM_NT_REC /* experiment: the records as a stream that should not occupy L2 */
          typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
          u32x4 v = { (uint32_t)gp, (uint32_t)(gp >> 32), length, kw };
          __builtin_nontemporal_store (v, reinterpret_cast<u32x4 *> (reinterpret_cast<unsigned char *> (here.dst) + off));
#else
          /* (a GLOBAL store: through a generic pointer -- the chunk's address comes out of LDS as two
           * integers -- the compiler emits flat_store, which counts on lgkmcnt as well, and every LDS
           * read behind it then waits until the store has been acknowledged by L2) */
          typedef uint32_t g_u32x4 __attribute__ ((ext_vector_type (4)));
          *reinterpret_cast<__attribute__ ((address_space (1))) g_u32x4 *> (here.dst + off) = g_u32x4{ (uint32_t)gp, (uint32_t)(gp >> 32), length, kw };
#endif
        }
        tally = hn + total;
      } else {
        tally = emit_records_slow (Ep, W, hn, hit, p, length, kw);
        if (rs)
          *rs = rec_state_load (hits);
      }
      return;
    }
    if (hn + total > WAVE) {
      flush_hits (E, hits, hn, lane);
      hn = 0;
    }
    if (hit)
      hits[hn + rank_below (m)] = make_uint2 (p, kw);
    tally = hn + total;
  }
}

/* third sieve: the newest n_items (<= 64) of a wave's queue, item = (position p of the last
 * symbol read, state reached | flags).  ONE level per call: report the state if it is terminal,
 * look its goto edge on text[p + 1] up, and put the starts that go on back into the queue -- the
 * few long walks (a planted keyword of 12 symbols) then travel in full batches with everybody
 * else's instead of holding 63 idle lanes for 8 rounds of memory latency each.
 * Returns (new queue fill << 32) | tally.
 * (The structs come by pointer to copies the caller makes on the spot: taking the address of the
 * kernel's own K and E would move them from scalar registers to scratch memory for the whole
 * kernel -- measured 2x on the main loop.) */
constexpr uint32_t WALK_CTX_K = 128, WALK_CTX_E = 256; /* LDS after the tile counter: StartsK, EmitCtx, one WaveRec per wave */
constexpr uint32_t WALK_CTX_BYTES = 16 + WALK_CTX_K + WALK_CTX_E + (SPARSE_THREADS / WAVE) * sizeof (WaveRec);
static_assert (sizeof (StartsK) <= WALK_CTX_K && sizeof (EmitCtx) <= WALK_CTX_E, "walk context does not fit its LDS slot");
constexpr uint32_t WI_REPORTED = 0x80000000u; /* what ends in this state has been reported by the caller */
constexpr uint32_t WI_RECORD = 0x40000000u;   /* 4-gram kernel: the index is a record index already (StartsK::remap) */
template <typename SYM, bool COUNT_ONLY, int GRAM = 0>
__device__ __noinline__ unsigned long long
walk_starts (const StartsK *Kp, const EmitCtx *Ep, const SYM *text, uint2 *queue, uint32_t qn, uint32_t n_items, uint2 *hits,
             unsigned long long counted) {
  const StartsK &K = *Kp;
  const EmitCtx &E = *Ep;
  const uint32_t lane = lane_id ();
  const uint32_t base = qn - n_items;
  const bool alive = lane < n_items;
  const uint2 it = alive ? queue[base + lane] : make_uint2 (0, 0);
  const uint32_t p = it.x;
  uint32_t st = it.y & ST_STATE;
  /* GRAM == 2 with peek_rel: bits 24-28 of an item's second word hold the class of the symbol
   * after its position, or 31 for "not known" -- a fresh item (bits 0-23: depth-5 state -
   * remap_base) got it from the first queue, the record item it turns into at the same position
   * (bits 0-22: the record) keeps it: the two looks most candidates get need no text */
  const bool with_cls = GRAM == 2 && K.peek_rel && alive;
  const uint32_t next_cls = with_cls ? (it.y >> 24) & 31u : 31u;
  if (with_cls)
    st = (it.y & WI_RECORD) ? it.y & 0x7FFFFFu : (it.y & 0xFFFFFFu) + K.remap_base;
  /* GRAM == 2: a fresh item (a depth-5 state by its id) asks the peek table first and comes back
   * as a record item at the same position if the next symbol can go on (or the state's record has
   * to be seen anyway); only record items touch the records */
  const bool fresh = GRAM == 2 && alive && !(it.y & WI_RECORD);
  uint2 pk = make_uint2 (0, 0);
  if (fresh) {
    if (K.peek_packed) {
      const uint32_t e = K.peek[st - K.remap_base];
      pk = make_uint2 (e & 0x7FFFFFu, (e >> 31) ? GRAM_NO_PEEK : (e >> 23) & 0xFFu);
    } else
      pk = *reinterpret_cast<const uint2 *> (K.peek + 2 * (size_t)(st - K.remap_base));
  }
  if (GRAM == 1 && !(it.y & WI_RECORD))
    st = alive ? K.remap[st - K.remap_base] : 0u;
  const bool regular = alive && !fresh;
  uint4 ra = make_uint4 (0, 0, 0, 0), rb = make_uint4 (0, 0, 0, 0);
  if (GRAM != 2 || regular) {
    ra = K.srec[2 * st];
    rb = K.srec[2 * st + 1];
  }
  const bool more = alive && p + 1 < E.n;
  uint32_t c1 = 0;
  if (next_cls != 31u) /* the symbol itself, or no symbol of the alphabet: no text is read */
    c1 = next_cls < K.span ? K.lo + next_cls : 0x100u;
  else if (more)
    c1 = (uint32_t)text[p + 1];
  /* (4-gram kernels: word 1 of a record is n_edges | depth << 16, word 3 the keyword id + 1 of a
   * terminal state -- what a record written on the spot needs; fill_gram_tables) */
  emit_terminals<COUNT_ONLY, GRAM == 2> (E, regular && !(it.y & WI_REPORTED) && ra.w != 0 && p >= E.emit_from, p,
                                         GRAM == 2 ? ra.w - 1u : (GRAM ? ra.x : st), ra.y >> 16, lane, hits, counted, Ep);
  uint32_t nx = NONE;
  if (more && regular) {
    const uint32_t ne = GRAM ? ra.y & 0xFFFFu : ra.y;
    if (ne >= 1 && rb.x == c1)
      nx = rb.y;
    else if (ne >= 2 && rb.z == c1)
      nx = rb.w;
    else if (ne > 2) {
      uint32_t lo = ra.z, hi = ra.z + ne;
      while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (K.sedge[mid].x < c1)
          lo = mid + 1;
        else
          hi = mid;
      }
      if (lo < ra.z + ne) {
        const uint2 e = K.sedge[lo];
        if (e.x == c1)
          nx = e.y;
      }
    }
  }
  const bool on = fresh && (pk.y == GRAM_NO_PEEK || (more && pk.y == c1));
  const bool go = nx != NONE || on;
  const uint64_t m = __ballot (go);
  if (go)
    queue[base + rank_below (m)] = on ? make_uint2 (p, pk.x | WI_RECORD | (GRAM == 2 && K.peek_rel ? next_cls << 24 : 0u))
                                      : make_uint2 (p + 1, nx | (GRAM ? WI_RECORD : 0u) | (GRAM == 2 && K.peek_rel ? 31u << 24 : 0u));
  const uint32_t fill = base + (uint32_t)__popcll (m);
  if (COUNT_ONLY)
    return ((unsigned long long)fill << 32) | (uint32_t)counted;
  /* record mode: the tally is the fill of the hit buffer, one value for the wave */
  return ((unsigned long long)fill << 32) | uniform ((uint32_t)counted);
}

template <bool LUT_LDS>
__device__ __forceinline__ uint32_t
starts_root (const StartsK &K, uint32_t c) {
  if (c < K.lut_size) {
    if (LUT_LDS)
      return *reinterpret_cast<const __attribute__ ((address_space (3))) uint32_t *> (c * 4u);
    return K.lut[c];
  }
  /* beyond the table: bisect the root row; no sieve can be applied, let everything pass */
  const uint4 a = K.srec[0];
  uint32_t lo = a.z, hi = a.z + a.y;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (K.sedge[mid].x < c)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo < a.z + a.y) {
    const uint2 e = K.sedge[lo];
    if (e.x == c)
      return e.y | ST_SECOND | ST_ALWAYS;
  }
  return ST_SECOND;
}

/* Which tile a wave of the start-parallel / 4-gram kernels takes next.  Tiles [range_begin,
 * static_end) go to the blocks interleaved (block b: b, b + gridDim.x, ...: match density is
 * rarely even along a text, contiguous shares left two blocks of config 3 working 1 ms after all
 * others), handed to the block's waves through an LDS counter; the last sixteenth, [static_end,
 * range_end), is a pool in POOL_CLASSES parts drawn tile by tile through one global counter per
 * part (Launch::pool_ctr, as in the dense kernel): the XCDs do not run at the same pace. */
struct TileShare {
  uint32_t begin, static_tiles, blk_tiles, cls_begin, cls_tiles;
  unsigned int *cls_ctr;
  __device__ __forceinline__
  TileShare (const Launch &A) {
    begin = A.range_begin;
    static_tiles = A.static_end - A.range_begin;
    blk_tiles = blockIdx.x < static_tiles ? (static_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t cls = blockIdx.x * A.pool_classes / gridDim.x;
    cls_begin = A.static_end + cls * A.pool_class_tiles;
    cls_tiles = cls_begin >= A.range_end ? 0 : (A.range_end - cls_begin < A.pool_class_tiles ? A.range_end - cls_begin : A.pool_class_tiles);
    cls_ctr = A.pool_ctr + cls * POOL_CTR_STRIDE;
    if (blockIdx.x == 0 && threadIdx.x < POOL_CLASSES)
      A.pool_reset[threadIdx.x * POOL_CTR_STRIDE] = 0; /* the previous launch's counters */
  }
  /* wave-uniform tile index, NONE when there is nothing left */
  __device__ __forceinline__ uint32_t
  next (uint32_t *lds_counter, uint32_t lane) const {
    uint32_t t = NONE;
    if (lane == 0) {
      const uint32_t i = atomicAdd (lds_counter, 1u);
      if (i < blk_tiles)
        t = begin + i * gridDim.x + blockIdx.x;
      else if (cls_tiles) {
        const uint32_t g = atomicAdd (cls_ctr, 1u);
        if (g < cls_tiles)
          t = cls_begin + g;
      }
    }
    return uniform (t);
  }
};

/* a start between the first and the second sieve */
struct PendingStart {
  uint2 pair;     /* the child's first two edge symbols (load in flight) */
  uint32_t ntok;  /* the symbol after the start */
  uint32_t child; /* root-table entry of the start's symbol; 0 = none pending */
  uint32_t pos;
};

template <typename SYM, bool LUT_LDS, bool COUNT_ONLY>
__global__ __launch_bounds__ (SPARSE_THREADS) void
scan_starts_kernel (StartsK K, EmitCtx E, Launch A, const unsigned char *__restrict__ text8, uint2 *items, uint32_t region_items,
                    uint32_t *fill) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  constexpr uint32_t PER = 16 / sizeof (SYM);  /* symbols per lane per group */
  constexpr uint32_t PERW = 4 / sizeof (SYM);  /* symbols per 32-bit word */
  constexpr uint32_t GROUP = WAVE * PER;
  constexpr uint32_t SYM_MASK = sizeof (SYM) == 4 ? 0xFFFFFFFFu : (1u << (8 * (sizeof (SYM) & 3))) - 1u;
  if (LUT_LDS) {
    uint4 *dst = reinterpret_cast<uint4 *> (smem);
    const uint4 *src = reinterpret_cast<const uint4 *> (K.lut);
    for (uint32_t i = threadIdx.x; i < (K.lut_size + 3) / 4; i += blockDim.x)
      dst[i] = src[i];
  }
  constexpr uint32_t WAVES = SPARSE_THREADS / WAVE;
  uint32_t *next_tile = reinterpret_cast<uint32_t *> (smem + K.queue_off + WAVES * (QCAP + HITS_STRIDE) * 8);
  /* what walk_starts needs of K and E, once per block in LDS: handing it the kernel's own
   * structs by address would move them from scalar registers to scratch for the whole kernel,
   * and a copy per call is 13 KB of scratch traffic per wave and call */
  StartsK *Ks = reinterpret_cast<StartsK *> (next_tile + 4);
  EmitCtx *Es = reinterpret_cast<EmitCtx *> (reinterpret_cast<unsigned char *> (Ks) + WALK_CTX_K);
  if (threadIdx.x == 0) {
    *next_tile = 0;
    *Ks = K;
    *Es = E;
  }
  __syncthreads ();

  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t wib = uniform (threadIdx.x / WAVE);
  uint2 *queue = reinterpret_cast<uint2 *> (smem + K.queue_off) + wib * QCAP;
  uint2 *hits = reinterpret_cast<uint2 *> (smem + K.queue_off) + WAVES * QCAP + wib * HITS_STRIDE + 2;
  const uint32_t wave_id = blockIdx.x * WAVES + wib;
  hits_init (hits, (!COUNT_ONLY && items) ? items + (size_t)wave_id * region_items : nullptr, region_items, lane);
  const SYM *text = reinterpret_cast<const SYM *> (text8);
  const uint4 *text16 = reinterpret_cast<const uint4 *> (text8);
  /* block b takes the tiles b, b + gridDim.x, ... (match density is rarely even along a text:
   * contiguous shares left two blocks of config 3 working 1 ms after all others had finished),
   * handed to its waves through the LDS counter */
  const TileShare share (A);
  const uint32_t last_blk = (uint32_t)(((uint64_t)A.n * sizeof (SYM) - 1) / 16);
  uint32_t qn = 0;
  unsigned long long counted = 0;
  DIAG (const unsigned long long d_t0 = __builtin_readcyclecounter (); unsigned long long d_walk = 0, d_calls = 0, d_cands = 0, d_deep = 0, d_tiles = 0;)
  PendingStart pend[4][PERW];
#pragma unroll
  for (int w = 0; w < 4; w++)
#pragma unroll
    for (uint32_t i = 0; i < PERW; i++)
      pend[w][i].child = 0;

  auto load_group = [&] (uint32_t g) -> uint4 {
    const uint32_t blk = g * WAVE + lane;
    return text16[blk < last_blk ? blk : last_blk];
  };
  /* second sieve on a start whose pair has arrived; survivors go to the wave's queue */
  auto resolve = [&] (PendingStart &P) {
    const bool deep = P.child != 0 && ((P.child & ST_ALWAYS) || P.pair.x == P.ntok || P.pair.y == P.ntok);
    const uint64_t m = __ballot (deep);
    if (m) {
      if (deep)
        queue[qn + rank_below (m)] = make_uint2 (P.pos, P.child & ST_STATE);
      qn = uniform (qn + (uint32_t)__popcll (m));
      while (qn >= WAVE) {
        DIAG (const unsigned long long d_c0 = __builtin_readcyclecounter ();)
        {
          const unsigned long long r = walk_starts<SYM, COUNT_ONLY> (Ks, Es, text, queue, qn, WAVE, hits, counted);
          qn = uniform ((uint32_t)(r >> 32));
          counted = (uint32_t)r;
        }
        DIAG (d_walk += __builtin_readcyclecounter () - d_c0; d_calls++;)
      }
    }
    DIAG (d_cands += __popcll (__ballot (P.child != 0)); d_deep += __popcll (m);)
    P.child = 0;
  };
  /* first sieve on the start at position p: e0 = entry of its symbol, e1 = entry of the next
   * symbol (0 when there is none), ntok = the next symbol */
  auto sieve = [&] (PendingStart &P, uint32_t e0, uint32_t e1, uint32_t ntok, uint32_t p) {
    resolve (P); /* the start that used this slot one group ago */
    /* (straight-line: every lane loads, the ones without a start pairs[0]; a load under a branch
     * would make the compiler wait for everything in flight, prefetched text included) */
    const bool cand = (e0 & ST_STATE) != 0 && ((e0 & ST_ALWAYS) || (e1 & ST_SECOND));
    P.child = cand ? e0 : 0u;
    P.pair = K.pairs[P.child & ST_STATE];
    P.ntok = ntok;
    P.pos = p;
  };
  /* one group: `cur` = this lane's 16 bytes, next_x = word 0 of every lane of the following group */
  auto walk_group = [&] (const uint4 cur, const uint32_t next_x, const uint32_t g) {
    const uint32_t pos0 = g * GROUP + lane * PER; /* position of this lane's first symbol */
    /* word 0 of the next lane (of the next group for lane 63): holds the successor of this lane's last symbol */
    uint32_t after = __shfl_down (cur.x, 1, WAVE);
    const uint32_t after_group = uniform (next_x);
    if (lane == WAVE - 1)
      after = after_group;
    const uint32_t words[5] = { cur.x, cur.y, cur.z, cur.w, after };
    uint32_t e_first = pos0 < A.n ? starts_root<LUT_LDS> (K, cur.x & SYM_MASK) : 0u;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t p = pos0 + w * PERW;
      const uint32_t ntok_word = words[w + 1] & SYM_MASK;
      const uint32_t e_next = p + PERW < A.n ? starts_root<LUT_LDS> (K, ntok_word) : 0u;
      if (PERW == 2) {
        const uint32_t mid = words[w] >> 16;
        const uint32_t e_mid = p + 1 < A.n ? starts_root<LUT_LDS> (K, mid) : 0u;
        sieve (pend[w][0], e_first, e_mid, mid, p);
        sieve (pend[w][PERW - 1], e_mid, e_next, ntok_word, p + 1);
      } else
        sieve (pend[w][0], e_first, e_next, ntok_word, p);
      e_first = e_next;
    }
  };

  for (;;) {
    const uint32_t tile = share.next (next_tile, lane);
    if (tile == NONE)
      break;
    DIAG (d_tiles++;)
    const uint32_t g0 = tile * K.R;
#ifndef ACM_STARTS16_DEEP /* (experiment: four and four for 2-byte symbols too) */
#define ACM_STARTS16_DEEP 0
#endif
    if (sizeof (SYM) == 2 && !COUNT_ONLY && !ACM_STARTS16_DEEP) {
      /* 2-byte symbols, with records: two pending starts per word, 40 registers of them -- two groups
       * walked while two are in flight.  With four and four the kernel spilled 16 registers, the
       * text it had just asked for among them: 1.49 -> 1.10 ms per Gi tokens (tools/exp_c5_16.py;
       * count-only has registers to spare and keeps four: 1.00 against 1.05) */
      uint4 c0 = load_group (g0), c1 = load_group (g0 + 1);
      for (uint32_t k = 0; k < K.R; k += 2) {
        const uint32_t g = g0 + k;
        const uint4 n0 = load_group (g + 2), n1 = load_group (g + 3);
        walk_group (c0, c1.x, g);
        walk_group (c1, n0.x, g + 1);
        c0 = n0;
        c1 = n1;
      }
      continue;
    }
    uint4 c0 = load_group (g0), c1 = load_group (g0 + 1), c2 = load_group (g0 + 2), c3 = load_group (g0 + 3);
    for (uint32_t k = 0; k < K.R; k += 4) {
      const uint32_t g = g0 + k;
      const uint4 n0 = load_group (g + 4), n1 = load_group (g + 5), n2 = load_group (g + 6), n3 = load_group (g + 7);
      walk_group (c0, c1.x, g);
      walk_group (c1, c2.x, g + 1);
      walk_group (c2, c3.x, g + 2);
      walk_group (c3, n0.x, g + 3);
      c0 = n0;
      c1 = n1;
      c2 = n2;
      c3 = n3;
    }
  }
#pragma unroll
  for (int w = 0; w < 4; w++)
#pragma unroll
    for (uint32_t i = 0; i < PERW; i++)
      resolve (pend[w][i]);
  while (qn) {
    const unsigned long long r = walk_starts<SYM, COUNT_ONLY> (Ks, Es, text, queue, qn, qn < WAVE ? qn : WAVE, hits, counted);
    qn = uniform ((uint32_t)(r >> 32));
    counted = (uint32_t)r;
  }
  if (COUNT_ONLY) {
    const uint32_t incl = wave_incl_scan ((uint32_t)counted); /* a lane finds far fewer than 2^32 / 64 */
    const uint32_t total = __shfl (incl, WAVE - 1, WAVE);
    if (lane == 0 && total)
      atomicAdd (E.count, (unsigned long long)total);
  } else {
    if (counted)
      flush_hits (E, hits, (uint32_t)counted, lane);
    if (lane == 0 && fill)
      fill[wave_id] = hits[-1].y;
  }
  DIAG (if (lane == 0) {
    const uint32_t wave = blockIdx.x * (SPARSE_THREADS / WAVE) + wib;
    if (wave < 8192) {
      unsigned long long *o = g_acm_diag[wave];
      o[0] = __builtin_readcyclecounter () - d_t0;
      o[1] = d_walk;
      o[2] = d_calls;
      o[3] = d_cands;
      o[4] = d_deep;
      o[5] = d_tiles;
    }
  })
}
