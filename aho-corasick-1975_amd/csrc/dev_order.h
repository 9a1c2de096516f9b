/* dev_order.h -- canonical order of a record set without a global radix sort.
 * Device code of libac75_amd.so; included by acm_gpu.hip inside its anonymous namespace.
 *
 * Canonical order = (end_pos ascending, length descending): what the reference's caller loop emits
 * (acm_get_match index order, aho_corasick.c:459-466).  The scan leaves records unordered but far
 * from random: a wave writes the records of one stretch of text side by side.  Round 2 ran a 64-bit
 * radix sort over everything (eight passes, 3.95 ms each on config 3's 430 M records: 2.9 x the
 * scan).  Here instead, for records whose positions lie in a known range [pos_lo, pos_lo + span):
 *   A. bucket = (end_pos - pos_lo) / 4,096.  Histogram: a wave takes 512 consecutive records and
 *      adds the count of every distinct bucket among them to the global histogram with ONE atomic
 *      (records that lie side by side fall into few buckets);
 *   B. exclusive prefix sum of the histogram (hipCUB), then the same walk again: every (piece,
 *      bucket) pair reserves its run in the bucket with one atomic and the records are copied
 *      there -- bucket by bucket the records are now in position order, unordered inside;
 *   C. every bucket is put in order and written to its final place: up to 256 records by one wave
 *      (rank = number of smaller keys), more by a counting sort over the bucket's positions in LDS
 *      (a bitonic sort of 4,096-record windows in LDS took 94 us per window: 19 ms on config 3).
 * Three passes over the records (one of them reads the positions only) instead of eight.
 * acm_gpu_scan_ordered_device queues them behind a scan with the record count read on the device
 * (OrderK::n_dev); 4-gram plans take dev_tiles.h's single pass instead. */
constexpr int ORDER_THREADS = 256, ORDER_PER = 8, ORDER_PIECE = ORDER_PER * WAVE; /* records a wave takes at a time */
constexpr int ORDER_ROUNDS = 12; /* distinct buckets of a piece taken one at a time before the rest goes lane by lane */

struct OrderK {
  const ACMRecord *in;   /* the unordered records */
  uint64_t n;
  uint64_t pos_lo;
  uint32_t wlog;         /* bucket width = 1 << wlog symbols */
  uint32_t n_buckets;
  uint32_t len_bits;     /* bits of a length */
  unsigned int *error;   /* set when a record lies outside [pos_lo, pos_lo + span) */
  /* a scan's records put in order behind it on the same stream, no host in between: the number of
   * records is the scan's count in device memory (n is then the buffer's capacity: a count beyond
   * it means the scan overflowed and is repeated -- nothing is put in order), and whether the set
   * is sparse is decided here too (mode 2; 0: dense, 1: sparse) */
  const unsigned long long *n_dev;
  uint64_t span;
  uint32_t mode;
};
/* (counting the buckets in the dense kernel's put_outputs instead of pass A -- one atomic per record
 * where the record is written -- cost the scan kernel 277 -> 396 us on config 2: measured, not kept) */

__device__ __forceinline__ uint64_t
order_n (const OrderK &K) {
  if (!K.n_dev)
    return K.n;
  const unsigned long long c = *K.n_dev;
  return c > K.n ? 0 : c;
}
/* fewer than 8 records per 4,096 positions: pass C goes by windows of buckets, not by bucket */
__host__ __device__ __forceinline__ bool
order_is_sparse (uint64_t n, uint64_t span) {
  return n < (span >> 9);
}
__device__ __forceinline__ bool
order_sparse (const OrderK &K, uint64_t n) {
  return K.mode == 2 ? order_is_sparse (n, K.span) : K.mode == 1;
}

/* passes A (SCATTER = false: histogram into `hist`) and B (SCATTER = true: `hist` holds the buckets'
 * cursors, the records go to `out`).  A wave takes ORDER_PIECE = 512 consecutive records, eight per
 * lane, and goes through the distinct buckets among them: every lane whose record lies in the
 * bucket at hand is counted by ballot, ONE atomic reserves the run (pass B) or adds the count (pass
 * A).  Records that lie side by side come from one stretch of text (a wave of the scan kernel wrote
 * them), so there are a handful of buckets per piece -- and where two stretches meet (the scan
 * kernel's chunks of 1,024 records: neighbours in the buffer, megabytes apart in the text) it is
 * two handfuls, nothing worse.  No LDS, no barrier.  (Two earlier forms counted a block's 2,048
 * records in an LDS array indexed by bucket - lowest bucket: pieces that straddle two chunks fell
 * out of the array into one global atomic per record -- 27 ms per pass on config 3 -- and an LDS
 * atomic per record on some ten counters serialised.) */
template <bool SCATTER>
__global__ __launch_bounds__ (ORDER_THREADS) void
order_bucket_kernel (OrderK K, uint32_t *hist, ACMRecord *out) {
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE, waves = (uint64_t)gridDim.x * blockDim.x / WAVE;
  const uint64_t n = order_n (K);
  if (SCATTER && n && hist[K.n_buckets] != n) {
    /* the buckets' counts do not add up to the records (never expected): nothing is moved */
    if (K.error && threadIdx.x == 0)
      *K.error = 1;
    return;
  }
  const uint64_t pieces = (n + ORDER_PIECE - 1) / ORDER_PIECE;
  for (uint64_t piece = wave; piece < pieces; piece += waves) {
    const uint64_t first = piece * ORDER_PIECE;
    uint32_t b[ORDER_PER], rk[ORDER_PER];
    uint4 rec[ORDER_PER];
    uint64_t todo[ORDER_PER];
#pragma unroll
    for (int q = 0; q < ORDER_PER; q++) {
      const uint64_t i = first + (uint64_t)q * WAVE + lane;
      b[q] = 0xFFFFFFFFu;
      rk[q] = 0;
      if (i < n) {
        uint64_t pos;
        if (SCATTER) {
          rec[q] = *reinterpret_cast<const uint4 *> (&K.in[i]);
          pos = ((uint64_t)rec[q].y << 32) | rec[q].x;
        } else
          pos = K.in[i].end_pos;
        uint64_t bb = (pos - K.pos_lo) >> K.wlog;
        if (pos < K.pos_lo || bb >= K.n_buckets) { /* outside the range the caller named: kept (in the last bucket), reported */
          if (K.error)
            *K.error = 1;
          bb = K.n_buckets - 1;
        }
        b[q] = (uint32_t)bb;
      }
      todo[q] = __ballot (b[q] != 0xFFFFFFFFu);
    }
    /* up to ORDER_ROUNDS distinct buckets are taken one at a time: all their records in the piece
     * are counted by ballot and the bucket gets ONE atomic for them.  Nothing here waits for an
     * atomic: round r leaves its bucket and its count in lane r, every record remembers its round
     * and its rank among that round's records, and when the rounds are over the lanes issue their
     * atomics together (one memory round trip for the piece instead of one per round: 12 in a row
     * made pass B 8.5 ms on config 3 and 33 us on the half a million records of config 2).  Four
     * rounds that find two records each say the piece is spread thin (sparse matches: config 2 has
     * two records per bucket) -- the rest goes lane by lane, every lane its own atomic.  (One
     * thin round says nothing: a dense piece has stragglers in the bucket next door, and the 200
     * records behind them must not go to ONE counter one by one -- that was 27 ms.) */
    uint32_t rnd[ORDER_PER], rel[ORDER_PER];
#pragma unroll
    for (int q = 0; q < ORDER_PER; q++)
      rnd[q] = 0xFFFFFFFFu, rel[q] = 0;
    uint32_t my_bucket = 0, my_total = 0, rounds = 0, placed = 0;
    for (int round = 0; round < ORDER_ROUNDS; round++) {
      /* the bucket of the first record not yet placed */
      uint32_t b0 = 0xFFFFFFFFu;
#pragma unroll
      for (int q = ORDER_PER - 1; q >= 0; q--)
        if (todo[q])
          b0 = __builtin_amdgcn_readlane (b[q], (uint32_t)__builtin_ctzll (todo[q]));
      if (b0 == 0xFFFFFFFFu)
        break;
      uint32_t total = 0;
#pragma unroll
      for (int q = 0; q < ORDER_PER; q++) {
        const uint64_t m = __ballot (b[q] == b0);
        if (b[q] == b0) {
          rnd[q] = (uint32_t)round;
          rel[q] = total + rank_below (m);
        }
        total += (uint32_t)__popcll (m);
        todo[q] &= ~m;
      }
      if (lane == (uint32_t)round) {
        my_bucket = b0;
        my_total = total;
      }
      rounds = (uint32_t)round + 1;
      placed += total;
      if (round == 3 && placed <= 8)
        break;
    }
    uint32_t at = 0;
#if defined(ACM_ORDER_ABLATE) && ACM_ORDER_ABLATE == 1 /* experiment: pass A without its atomics */
    if (!SCATTER) {
      asm volatile ("" :: "v"(my_bucket), "v"(my_total), "v"(b[0]), "v"(b[7]));
      continue;
    }
#endif
    if (lane < rounds) {
      if (SCATTER)
        at = atomicAdd (&hist[my_bucket], my_total); /* pass B: where this piece's run in the bucket begins */
      else
        atomicAdd (&hist[my_bucket], my_total);
    }
#pragma unroll
    for (int q = 0; q < ORDER_PER; q++)
      if ((todo[q] >> lane) & 1ull) {
        if (SCATTER)
          rk[q] = atomicAdd (&hist[b[q]], 1u);
        else
          atomicAdd (&hist[b[q]], 1u);
      }
    if (SCATTER) {
#pragma unroll
      for (int q = 0; q < ORDER_PER; q++) {
        const uint32_t begins = __shfl (at, (int)(rnd[q] & (WAVE - 1)), WAVE);
        if (rnd[q] != 0xFFFFFFFFu)
          rk[q] = begins + rel[q];
      }
    }
    if (SCATTER) {
#pragma unroll
      for (int q = 0; q < ORDER_PER; q++)
        if (b[q] != 0xFFFFFFFFu)
          *reinterpret_cast<uint4 *> (&out[rk[q]]) = rec[q];
    }
  }
}

/* pass C.  A bucket covers 1 << wlog <= ORDER_POSITIONS positions; its records lie side by side in
 * `bucketed`, in no order.
 * order_small_role: buckets of at most 256 records, one wave each, four records per lane -- a
 * record's place is the number of records of the bucket with a smaller key (position, then longer
 * before shorter): one v_readlane and four compares per other record, no barrier, no LDS (a block
 * per bucket with a counting sort in LDS took 24 us per bucket of 200 records: five barriers with
 * a memory round trip between each).
 * order_count_role: the other buckets, one block each, whatever they hold (the records are
 * streamed, LDS holds a counter per position): a counting sort by position -- count, exclusive
 * prefix sums, place -- and then, position by position, the few records that end at the same
 * symbol put longest first. */
constexpr uint32_t ORDER_POSITIONS = 4096;
constexpr int ORDER_COUNT_THREADS = 256;

__device__ __forceinline__ unsigned long long
order_key (const OrderK &K, uint64_t lo, uint64_t pos, uint32_t length) {
  const uint64_t lmask = (1ull << K.len_bits) - 1;
  return ((pos - lo) << K.len_bits) | (lmask - (length & lmask));
}

constexpr uint32_t ORDER_SMALL = 256; /* records a wave orders by itself: 4 per lane */

/* the wave puts the cnt <= 256 records bucketed[base ...] in order into out[base ...]: a record's
 * place is the number of records with a smaller key (position from `lo` on, then longer before
 * shorter; keys differ).  KEY = uint32_t when the positions span at most a bucket. */
template <typename KEY, int R>
__device__ __forceinline__ void
order_wave_sort_r (const OrderK &K, uint64_t lo, uint32_t base, uint32_t cnt, const ACMRecord *__restrict__ bucketed, ACMRecord *__restrict__ out,
                   uint32_t lane) {
  uint4 rec[R];
  KEY key[R];
  uint32_t rank[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    rec[r] = make_uint4 (0, 0, 0, 0);
    key[r] = ~(KEY)0;
    rank[r] = 0;
    const uint32_t i = r * WAVE + lane;
    if (i < cnt) {
      rec[r] = *reinterpret_cast<const uint4 *> (&bucketed[base + i]);
      key[r] = (KEY)order_key (K, lo, ((uint64_t)rec[r].y << 32) | rec[r].x, rec[r].z);
    }
  }
#pragma unroll
  for (int s = 0; s < R; s++) {
    if ((uint32_t)s * WAVE >= cnt)
      break;
    const uint32_t upto = cnt - s * WAVE < WAVE ? cnt - s * WAVE : WAVE;
    for (uint32_t j = 0; j < upto; j++) {
      KEY kj;
      if (sizeof (KEY) == 8)
        kj = (KEY)(((unsigned long long)__builtin_amdgcn_readlane ((uint32_t)((unsigned long long)key[s] >> 32), j) << 32) |
                   __builtin_amdgcn_readlane ((uint32_t)key[s], j));
      else
        kj = (KEY)__builtin_amdgcn_readlane ((uint32_t)key[s], j);
#pragma unroll
      for (int r = 0; r < R; r++)
        rank[r] += kj < key[r] ? 1u : 0u;
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++)
    if ((uint32_t)r * WAVE + lane < cnt)
      *reinterpret_cast<uint4 *> (&out[base + rank[r]]) = rec[r];
}

/* (R = records per lane: a bucket of 100 records costs half the compares of one of 256) */
template <typename KEY>
__device__ __forceinline__ void
order_wave_sort (const OrderK &K, uint64_t lo, uint32_t base, uint32_t cnt, const ACMRecord *__restrict__ bucketed, ACMRecord *__restrict__ out,
                 uint32_t lane) {
  if (cnt <= WAVE)
    order_wave_sort_r<KEY, 1> (K, lo, base, cnt, bucketed, out, lane);
  else if (cnt <= 2 * WAVE)
    order_wave_sort_r<KEY, 2> (K, lo, base, cnt, bucketed, out, lane);
  else
    order_wave_sort_r<KEY, 4> (K, lo, base, cnt, bucketed, out, lane);
}

/* dense record sets: a wave per bucket of up to 256 records */
__device__ __forceinline__ void
order_small_role (const OrderK &K, const uint32_t *__restrict__ P, const ACMRecord *__restrict__ bucketed, ACMRecord *__restrict__ out, uint32_t blk,
                  uint32_t nblk) {
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t wave = (blk * blockDim.x + threadIdx.x) / WAVE, waves = nblk * blockDim.x / WAVE;
  if (order_sparse (K, order_n (K)))
    return; /* order_window_role's */
  for (uint32_t b = wave; b < K.n_buckets; b += waves) {
    const uint32_t base = uniform (P[b]), cnt = uniform (P[b + 1]) - base;
    if (cnt == 0 || cnt > ORDER_SMALL)
      continue;
    order_wave_sort<uint32_t> (K, K.pos_lo + ((uint64_t)b << K.wlog), base, cnt, bucketed, out, lane);
  }
}

/* sparse record sets (most buckets hold a record or two): a wave per WINDOW of whole buckets --
 * from the first bucket that begins at or after record 128 k to the first that begins at or after
 * record 128 (k + 1) -- ordered in one go when it holds at most 256 records; else (a crowded
 * bucket among sparse ones: the start of config 2's text has one of 934) its buckets one by one,
 * those of more than 256 left to order_count_role. */
constexpr uint32_t ORDER_WINDOW = 128;
/* the first bucket that begins at or after record `want` (P is ascending): the wave probes 64
 * places of the range at a time -- three dependent loads for config 2's 262,144 buckets where a
 * bisection took eighteen, and that was most of this kernel's 28 us */
__device__ __forceinline__ uint32_t
order_first_bucket_at (const uint32_t *__restrict__ P, uint32_t n_buckets, uint64_t want, uint32_t lane) {
  uint32_t lo = 0, hi = n_buckets;
  while (lo < hi) {
    const uint32_t step = (hi - lo + WAVE - 1) / WAVE;
    const uint64_t idx = (uint64_t)lo + (uint64_t)lane * step;
    const bool less = idx < hi && P[idx] < want;
    const uint32_t c = (uint32_t)__popcll (__ballot (less)); /* (the probes that are less come first) */
    if (c == 0)
      hi = lo;
    else {
      const uint64_t first_not = (uint64_t)lo + (uint64_t)c * step;
      lo = lo + (c - 1) * step + 1;
      if (first_not < hi)
        hi = (uint32_t)first_not;
    }
  }
  return lo;
}
/* both ends of a window in one go: the two searches' loads travel together (three round trips, not six) */
__device__ __forceinline__ void
order_window_bounds (const uint32_t *__restrict__ P, uint32_t n_buckets, uint64_t want0, uint64_t want1, uint32_t lane, uint32_t &b0, uint32_t &b1) {
  uint32_t lo0 = 0, hi0 = n_buckets, lo1 = 0, hi1 = n_buckets;
  while (lo0 < hi0 || lo1 < hi1) {
    const uint32_t step0 = (hi0 - lo0 + WAVE - 1) / WAVE, step1 = (hi1 - lo1 + WAVE - 1) / WAVE;
    const uint64_t idx0 = (uint64_t)lo0 + (uint64_t)lane * step0, idx1 = (uint64_t)lo1 + (uint64_t)lane * step1;
    const uint32_t p0 = idx0 < hi0 ? P[idx0] : 0xFFFFFFFFu, p1 = idx1 < hi1 ? P[idx1] : 0xFFFFFFFFu;
    const uint32_t c0 = (uint32_t)__popcll (__ballot (idx0 < hi0 && p0 < want0)), c1 = (uint32_t)__popcll (__ballot (idx1 < hi1 && p1 < want1));
    if (lo0 < hi0) {
      if (c0 == 0)
        hi0 = lo0;
      else {
        const uint64_t first_not = (uint64_t)lo0 + (uint64_t)c0 * step0;
        lo0 = lo0 + (c0 - 1) * step0 + 1;
        if (first_not < hi0)
          hi0 = (uint32_t)first_not;
      }
    }
    if (lo1 < hi1) {
      if (c1 == 0)
        hi1 = lo1;
      else {
        const uint64_t first_not = (uint64_t)lo1 + (uint64_t)c1 * step1;
        lo1 = lo1 + (c1 - 1) * step1 + 1;
        if (first_not < hi1)
          hi1 = (uint32_t)first_not;
      }
    }
  }
  b0 = lo0;
  b1 = lo1;
}

__device__ __forceinline__ void
order_window_role (const OrderK &K, const uint32_t *__restrict__ P, const ACMRecord *__restrict__ bucketed, ACMRecord *__restrict__ out, uint32_t blk,
                   uint32_t nblk) {
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint64_t wave = ((uint64_t)blk * blockDim.x + threadIdx.x) / WAVE, waves = (uint64_t)nblk * blockDim.x / WAVE;
  const uint64_t n = order_n (K);
  if (!order_sparse (K, n))
    return; /* order_small_role's */
  const uint64_t windows = (n + ORDER_WINDOW - 1) / ORDER_WINDOW;
  for (uint64_t k = wave; k < windows; k += waves) {
    uint32_t b0, b1;
    order_window_bounds (P, K.n_buckets, k * ORDER_WINDOW, (k + 1) * ORDER_WINDOW, lane, b0, b1);
    if (b1 <= b0)
      continue;
    const uint32_t base = uniform (P[b0]), cnt = uniform (P[b1]) - base;
    if (cnt == 0)
      continue;
    if (cnt <= ORDER_SMALL) {
      order_wave_sort<unsigned long long> (K, K.pos_lo + ((uint64_t)b0 << K.wlog), base, cnt, bucketed, out, lane);
      continue;
    }
    for (uint32_t b = b0; b < b1; b++) {
      const uint32_t bb = uniform (P[b]), bc = uniform (P[b + 1]) - bb;
      if (bc && bc <= ORDER_SMALL)
        order_wave_sort<uint32_t> (K, K.pos_lo + ((uint64_t)b << K.wlog), bb, bc, bucketed, out, lane);
    }
  }
}

__device__ __forceinline__ void
order_count_role (const OrderK &K, const uint32_t *__restrict__ P, const ACMRecord *__restrict__ bucketed, ACMRecord *out, uint32_t blk, uint32_t nblk) {
  __shared__ uint32_t ctr[ORDER_POSITIONS];
  __shared__ uint32_t s_part[ORDER_COUNT_THREADS / WAVE];
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  const uint32_t width = 1u << K.wlog; /* <= ORDER_POSITIONS */
  constexpr uint32_t PER = ORDER_POSITIONS / ORDER_COUNT_THREADS;
  /* a block looks at ORDER_COUNT_THREADS buckets at a time (a thread each) and then takes the
   * crowded ones among them one by one: where there are none -- most of the time -- that is all */
  __shared__ uint32_t s_list[ORDER_COUNT_THREADS];
  __shared__ uint32_t s_nlist;
  for (uint32_t first = blk * ORDER_COUNT_THREADS; first < K.n_buckets; first += nblk * ORDER_COUNT_THREADS) {
    if (tid == 0)
      s_nlist = 0;
    __syncthreads ();
    if (first + tid < K.n_buckets && P[first + tid + 1] - P[first + tid] > ORDER_SMALL)
      s_list[atomicAdd (&s_nlist, 1u)] = first + tid;
    __syncthreads ();
    const uint32_t nlist = s_nlist;
  for (uint32_t li = 0; li < nlist; li++) {
    const uint32_t b = s_list[li];
    const uint32_t base = P[b], cnt = P[b + 1] - base;
    const uint64_t lo = K.pos_lo + ((uint64_t)b << K.wlog);
    for (uint32_t i = tid; i < ORDER_POSITIONS; i += ORDER_COUNT_THREADS)
      ctr[i] = 0;
    __syncthreads ();
    for (uint32_t i = tid; i < cnt; i += ORDER_COUNT_THREADS)
      atomicAdd (&ctr[(uint32_t)(bucketed[base + i].end_pos - lo) & (ORDER_POSITIONS - 1)], 1u);
    __syncthreads ();
    /* exclusive prefix sums in place: a run of PER counters per thread */
    uint32_t run = 0;
#pragma unroll
    for (uint32_t q = 0; q < PER; q++)
      run += ctr[tid * PER + q];
    const uint32_t incl = wave_incl_scan (run);
    if (lane == WAVE - 1)
      s_part[wid] = incl;
    __syncthreads ();
    uint32_t acc = incl - run;
    for (uint32_t w = 0; w < wid; w++)
      acc += s_part[w];
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
      const uint32_t c = ctr[tid * PER + q];
      ctr[tid * PER + q] = acc;
      acc += c;
    }
    __syncthreads ();
    /* place: the counter of a position runs from where its records begin to where they end */
    for (uint32_t i = tid; i < cnt; i += ORDER_COUNT_THREADS) {
      const uint4 rec = *reinterpret_cast<const uint4 *> (&bucketed[base + i]);
      const uint64_t pos = ((uint64_t)rec.y << 32) | rec.x;
      const uint32_t slot = atomicAdd (&ctr[(uint32_t)(pos - lo) & (ORDER_POSITIONS - 1)], 1u);
      *reinterpret_cast<uint4 *> (&out[base + slot]) = rec;
    }
    /* the block's own stores, then its loads of them (one CU: the L1 all its waves share) */
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
    __syncthreads ();
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
    /* records of one position: longest first (their lengths differ; there are few of them) */
    for (uint32_t i = tid; i < width; i += ORDER_COUNT_THREADS) {
      const uint32_t end = ctr[i], begin = i ? ctr[i - 1] : 0u;
      for (uint32_t a = begin + 1; a < end; a++) { /* insertion sort */
        const uint4 r = *reinterpret_cast<const uint4 *> (&out[base + a]);
        uint32_t at = a;
        while (at > begin) {
          const uint4 prev = *reinterpret_cast<const uint4 *> (&out[base + at - 1]);
          if (prev.z >= r.z)
            break;
          *reinterpret_cast<uint4 *> (&out[base + at]) = prev;
          at--;
        }
        if (at != a)
          *reinterpret_cast<uint4 *> (&out[base + at]) = r;
      }
    }
    __syncthreads ();
  }
    __syncthreads ();
  }
}

/* pass C in one launch: blocks [0, wgrid) take the windows of a sparse set, [wgrid, wgrid + sgrid)
 * the buckets of a dense one (with the record count on the device both are there and the kind of
 * set decides which of them works), the rest the crowded buckets */
static_assert (ORDER_COUNT_THREADS == 256, "the three roles share a launch");
__global__ __launch_bounds__ (256) void
order_finish_kernel (OrderK K, const uint32_t *__restrict__ P, const ACMRecord *__restrict__ bucketed, ACMRecord *out, uint32_t wgrid, uint32_t sgrid) {
  if (blockIdx.x < wgrid)
    order_window_role (K, P, bucketed, out, blockIdx.x, wgrid);
  else if (blockIdx.x < wgrid + sgrid)
    order_small_role (K, P, bucketed, out, blockIdx.x - wgrid, sgrid);
  else
    order_count_role (K, P, bucketed, out, blockIdx.x - wgrid - sgrid, gridDim.x - wgrid - sgrid);
}
