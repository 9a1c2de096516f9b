/* dev_tiles.h -- canonical order of a tiled scan's records in ONE pass over them.
 * Device code of libac75_amd.so; included by acm_gpu.hip inside its anonymous namespace.
 *
 * acm_gpu_order_records_device (dev_order.h) takes any record set: a histogram pass, a scatter pass
 * and a sort pass, 14 ms for config 3's 430 M records where the scan takes 17.6.  A scan that is
 * asked for ordered records up front (acm_gpu_scan_ordered_device) can leave them so that the first
 * two passes have nothing to do.  The 4-gram kernel hands the text out in tiles (R groups of 1,024
 * symbols: 32,768 symbols on config 3, ~820 records); in a tiled scan a wave empties its queues at
 * the end of every tile, so a tile's records lie side by side in the wave's stream of chunks, and
 * writes a TileEntry (dev_starts.h) saying where.  The chunks of a wave are linked backwards
 * (EmitCtx::chunk_prev), so `the n records in front of slot s` can be followed from any entry.
 * Every record with an end position in [lo, hi) of tile d is then either one of tile d's own
 * records or one of the few that tile d - 1 wrote after its last group began (a keyword that starts
 * in d - 1 and ends in d; keywords are at most a group long here).  So:
 *   1. tile_size_kernel: per tile, how many of its late records end beyond it -> the size of every
 *      tile's stretch of the output; exclusive prefix sums (hipCUB) give where it begins;
 *   2. tile_gather_kernel: a block per tile.  Every thread reads up to eight of the tile's records
 *      (and of the late ones of the tile before) and keeps them in registers; a 4-byte key per kept
 *      record goes to LDS by bucket of 2,048 positions, a wave ranks every bucket of up to 192 keys
 *      (four keys per broadcast LDS read), the ranks come back through LDS and every thread writes
 *      its records to their final places.  Crowded tiles (more source records than the registers
 *      hold, or a bucket of more than 192): see the second half of the kernel.
 * One read and one write of every record; no atomics outside LDS; holes in the chunks are never
 * looked at (close_holes_kernel does not run).  Config 3, 430 M records: 4.2 ms. */
constexpr uint32_t TILE_THREADS = 256, TILE_PER = 8, TILE_SRC_MAX = TILE_THREADS * TILE_PER; /* records a thread keeps in registers */
constexpr uint32_t TILE_BUCKET_LOG2 = 11;                   /* buckets of 2,048 positions (config 3: ~50 records: ranks cost the square) */
constexpr uint32_t TILE_SUBCAP = 192, TILE_NSUB = 33;       /* records a wave ranks by itself; buckets of a tile of 64 groups + 1 */
constexpr uint32_t TILE_LIST = 4, TILE_CLIST = 1024;
constexpr uint32_t TILE_CROWDED_POSITIONS = 2048; /* the crowded path's counters: a bucket at a time */
/* dynamic LDS of tile_gather_kernel for tiles of `nsub` buckets:
 * TILE_SUBCAP records per bucket (keys 4 B, ranks 2 B); the crowded path's counters and chunk list lie over them */
__host__ __device__ constexpr uint32_t
tile_lds_bytes (uint32_t nsub) {
  return nsub * TILE_SUBCAP * 6 > TILE_CROWDED_POSITIONS * 4 + TILE_CLIST * 4 ? nsub * TILE_SUBCAP * 6 : TILE_CROWDED_POSITIONS * 4 + TILE_CLIST * 4;
  /* (normal path: 4-byte keys, 2-byte ranks) */
}

struct TileK {
  const ACMRecord *raw;        /* the scan's chunks */
  const uint32_t *chunk_prev;
  const TileEntry *dir;
  uint32_t n_tiles;
  uint32_t *size;              /* [n_tiles + 1] records of every tile's stretch (0 behind the last) */
  const uint32_t *begin;       /* [n_tiles + 1] exclusive prefix sums of `size` */
  ACMRecord *out;
  uint64_t capacity;
  unsigned long long *d_count; /* the caller's count: the total, written by the gather kernel */
  unsigned long long *reserved; /* the plan's running total (slots reserved by the scan): zeroed */
  uint32_t len_bits;
  uint32_t nsub;               /* buckets of 2,048 positions a tile spans at most (its groups / 2 + 1) */
  uint32_t *crowded;           /* [1 + n_tiles] how many tiles tile_gather_kernel left to tile_crowded_kernel, then which */
  /* A scan that reserved more slots than the raw area has (`*reserved` > raw_slots: far more
   * records than the caller's capacity) dropped the chunks beyond it -- their records, and their
   * chunk_prev links.  Nothing of the raw area is then looked at: tile_size_kernel adds the tiles'
   * own counts up in 64 bits (*over_total, zero otherwise: the sizes and their prefix sums are 32
   * bits wide) and the later kernels report that total and return. */
  uint64_t raw_slots;
  unsigned long long *over_total;
  unsigned int *error;
};
/* the scan's total when it is known not to fit (see TileK::over_total), else what the prefix sums say */
__device__ __forceinline__ unsigned long long
tile_total (const TileK &K) {
  const unsigned long long over = *K.over_total;
  return over ? over : (unsigned long long)K.begin[K.n_tiles];
}

/* the slot of the record j places in front of `end_slot` in its wave's stream (j = 0: the last one) */
__device__ __forceinline__ unsigned long long
tile_slot_walk (const uint32_t *__restrict__ chunk_prev, unsigned long long end_slot, uint32_t j) {
  uint32_t c = (uint32_t)((end_slot - 1) / REC_CHUNK);
  uint32_t top = (uint32_t)(end_slot - (unsigned long long)c * REC_CHUNK);
  while (j >= top) {
    j -= top;
    c = chunk_prev[c];
    top = REC_CHUNK;
  }
  return (unsigned long long)c * REC_CHUNK + (top - 1 - j);
}

/* the same with the stream's last chunks listed (list[0] = the chunk of end_slot - 1): no loads
 * while j lies within the list */
__device__ __forceinline__ unsigned long long
tile_slot_listed (const uint32_t *__restrict__ chunk_prev, const uint32_t *list, uint32_t nlist, unsigned long long end_slot, uint32_t j) {
  const uint32_t top0 = (uint32_t)(end_slot - (unsigned long long)list[0] * REC_CHUNK);
  if (j < top0)
    return (unsigned long long)list[0] * REC_CHUNK + (top0 - 1 - j);
  const uint32_t j1 = j - top0, k = 1 + j1 / REC_CHUNK;
  uint32_t c;
  if (k < nlist)
    c = list[k];
  else {
    c = list[nlist - 1];
    for (uint32_t q = nlist - 1; q < k; q++)
      c = chunk_prev[c];
  }
  return (unsigned long long)c * REC_CHUNK + (REC_CHUNK - 1 - j1 % REC_CHUNK);
}

/* the last TILE_LIST chunks of a stream in registers (every thread of the block follows the links
 * itself: the addresses are the same for all, the loads are scalar, nobody waits for a thread 0) */
struct TileRun {
  unsigned long long end_slot;
  uint32_t top0;            /* records of the stream's last chunk that lie in front of end_slot */
  uint32_t c[TILE_LIST];
};
__device__ __forceinline__ TileRun
tile_run (const uint32_t *__restrict__ chunk_prev, const TileEntry &e, uint32_t need) {
  const unsigned long long end_slot = e.end_slot;
  TileRun r;
  r.end_slot = end_slot;
  r.top0 = 0;
#pragma unroll
  for (uint32_t k = 0; k < TILE_LIST; k++)
    r.c[k] = 0;
  if (need == 0)
    return r;
  uint32_t c = (uint32_t)((end_slot - 1) / REC_CHUNK);
  r.top0 = (uint32_t)(end_slot - (unsigned long long)c * REC_CHUNK);
  r.c[0] = c;
  uint32_t have = r.top0;
#pragma unroll
  for (uint32_t k = 1; k < TILE_LIST; k++) {
    if (have < need) {
      /* (the entry names the two chunks before: no load for the tiles that end within three) */
      c = (k == 1 && e.c1 != NONE) ? e.c1 : ((k == 2 && e.c2 != NONE) ? e.c2 : chunk_prev[c]);
      r.c[k] = c;
      have += REC_CHUNK;
    }
  }
  return r;
}
/* (j below TILE_SRC_MAX: within the list) */
__device__ __forceinline__ unsigned long long
tile_run_slot (const TileRun &r, uint32_t j) {
  if (j < r.top0)
    return (unsigned long long)r.c[0] * REC_CHUNK + (r.top0 - 1 - j);
  const uint32_t j1 = j - r.top0, k = 1 + j1 / REC_CHUNK;
  uint32_t c = r.c[1];
#pragma unroll
  for (uint32_t q = 2; q < TILE_LIST; q++)
    c = k == q ? r.c[q] : c;
  return (unsigned long long)c * REC_CHUNK + (REC_CHUNK - 1 - j1 % REC_CHUNK);
}

__device__ __forceinline__ uint32_t
tile_build_list (const uint32_t *__restrict__ chunk_prev, unsigned long long end_slot, uint32_t need, uint32_t *list, uint32_t cap) {
  if (need == 0) {
    list[0] = 0;
    return 1;
  }
  uint32_t c = (uint32_t)((end_slot - 1) / REC_CHUNK);
  uint32_t have = (uint32_t)(end_slot - (unsigned long long)c * REC_CHUNK), k = 1;
  list[0] = c;
  while (have < need && k < cap) {
    c = chunk_prev[c];
    list[k++] = c;
    have += REC_CHUNK;
  }
  return k;
}

/* how many of the tile's late records end at or beyond its range (wave-uniform result) */
__device__ __forceinline__ uint32_t
tile_count_over (const TileK &K, const TileEntry &e, uint32_t lane) {
  const uint32_t cnt = e.n_late < e.n ? e.n_late : e.n;
  uint32_t c = 0;
  for (uint32_t j0 = 0; j0 < cnt; j0 += WAVE) {
    const uint32_t j = j0 + lane;
    if (j < cnt && K.raw[tile_slot_walk (K.chunk_prev, e.end_slot, j)].end_pos >= e.hi)
      c++;
  }
  const uint32_t incl = wave_incl_scan (c);
  return __shfl (incl, WAVE - 1, WAVE);
}

__global__ __launch_bounds__ (256) void
tile_size_kernel (TileK K) {
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, waves = gridDim.x * blockDim.x / WAVE;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    K.crowded[0] = 0;
  const bool over = __hip_atomic_load (K.reserved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > K.raw_slots;
  unsigned long long sum = 0;
  for (uint32_t d = wave; d <= K.n_tiles; d += waves) {
    uint32_t sz = 0;
    if (d < K.n_tiles) {
      const TileEntry e = K.dir[d];
      if (over)
        sum += e.n; /* (every record is one of its tile's own; the late ones only move between neighbours) */
      else {
        sz = e.n - tile_count_over (K, e, lane);
        if (d)
          sz += tile_count_over (K, K.dir[d - 1], lane);
      }
    }
    if (lane == 0)
      K.size[d] = sz;
  }
  if (over && lane == 0 && sum)
    atomicAdd (K.over_total, sum);
}

/* a wave ranks the cnt <= TILE_SUBCAP keys key[0 .. cnt): rank[k] = number of smaller keys (they
 * differ).  Four keys per LDS read (one address for the wave: a broadcast); the list is padded to
 * a multiple of four with keys that are smaller than none. */
template <int R>
__device__ __forceinline__ void
tile_wave_rank (uint32_t *key, uint16_t *rank, uint32_t cnt, uint32_t lane) {
  const uint32_t padded = (cnt + 3) & ~3u;
  if (lane < 4 && cnt + lane < padded)
    key[cnt + lane] = 0xFFFFFFFFu;
  uint32_t kk[R], rk[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t i = r * WAVE + lane;
    kk[r] = i < cnt ? key[i] : 0u; /* (0: nothing is smaller, and nobody asks) */
    rk[r] = 0;
  }
  for (uint32_t j = 0; j < padded; j += 4) {
    const uint4 k4 = *reinterpret_cast<const uint4 *> (key + j);
#pragma unroll
    for (int r = 0; r < R; r++)
      rk[r] += (k4.x < kk[r] ? 1u : 0u) + (k4.y < kk[r] ? 1u : 0u) + (k4.z < kk[r] ? 1u : 0u) + (k4.w < kk[r] ? 1u : 0u);
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t i = r * WAVE + lane;
    if (i < cnt)
      rank[i] = (uint16_t)rk[r];
  }
}

__global__ __launch_bounds__ (TILE_THREADS) void
tile_gather_kernel (TileK K) {
  /* normal path: keys and source indices by bucket; crowded path: a counter per position of a
   * bucket (over s_key) and the long chunk list */
  extern __shared__ __attribute__ ((aligned (16))) unsigned char tile_smem[]; /* tile_lds_bytes (K.nsub) */
  uint32_t *s_key = reinterpret_cast<uint32_t *> (tile_smem);
  uint16_t *s_rank = reinterpret_cast<uint16_t *> (tile_smem + (size_t)K.nsub * TILE_SUBCAP * 4);
  __shared__ uint32_t s_cnt[TILE_NSUB], s_off[TILE_NSUB];
  static_assert (TILE_SRC_MAX <= (TILE_LIST - 1) * REC_CHUNK, "a tile of the normal path lies within the chunks a TileRun lists");
  static_assert (TILE_NSUB <= WAVE && TILE_SUBCAP <= 256, "one lane per bucket; a record's place in its bucket is a byte");
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  const unsigned long long total = tile_total (K);
  if (blockIdx.x == 0 && tid == 0) {
    *K.d_count = total;
    *K.reserved = 0;
  }
  if (total > K.capacity)
    return; /* the scan overflowed: the caller repeats it with room, nothing is put in order */
  const uint32_t lmask = (1u << K.len_bits) - 1;
  for (uint32_t d = blockIdx.x; d < K.n_tiles; d += gridDim.x) {
    /* (all four asked for together: one round trip) */
    const uint32_t out_begin = K.begin[d], size = K.begin[d + 1] - out_begin;
    const TileEntry e = K.dir[d];
    TileEntry ep = K.dir[d ? d - 1 : 0];
    if (d == 0)
      ep.n = ep.n_late = 0;
    if (size == 0)
      continue;
    const uint32_t n_own = e.n, n_prev = ep.n_late < ep.n ? ep.n_late : ep.n, n_src = n_own + n_prev;
    bool crowded = n_src > TILE_SRC_MAX;
    __syncthreads (); /* (the tile before is done with the shared arrays) */
    if (tid < TILE_NSUB)
      s_cnt[tid] = 0;
    if (!crowded) {
      /* record i of the tile's source: its own records first (the last one written first), then the
       * late ones of the tile before; kept if it ends in [lo, hi).  A thread keeps its records in
       * registers until their places are known: every record is read once and written once. */
      const TileRun own = tile_run (K.chunk_prev, e, n_own), prev = tile_run (K.chunk_prev, ep, n_prev);
      uint4 rec[TILE_PER];
      uint32_t where[TILE_PER]; /* bucket << 8 | place in it; NONE: not this tile's */
      /* (the rounds a tile does not reach are jumped over, not executed under an empty mask:
       * config 3's tiles hold ~900 records, four rounds of the eight) */
#pragma unroll
      for (uint32_t q = 0; q < TILE_PER; q++) {
        where[q] = NONE;
        rec[q] = make_uint4 (0, 0, 0, 0);
      }
#pragma unroll
      for (uint32_t q = 0; q < TILE_PER; q++) {
        if (q * TILE_THREADS >= n_src)
          break;
        const uint32_t i = q * TILE_THREADS + tid;
        if (i < n_src) {
          const unsigned long long slot = i < n_own ? tile_run_slot (own, i) : tile_run_slot (prev, i - n_own);
          rec[q] = *reinterpret_cast<const uint4 *> (&K.raw[slot]);
        }
      }
      __syncthreads ();
#pragma unroll
      for (uint32_t q = 0; q < TILE_PER; q++) {
        if (q * TILE_THREADS >= n_src)
          break;
        const uint32_t i = q * TILE_THREADS + tid;
        const unsigned long long pos = ((unsigned long long)rec[q].y << 32) | rec[q].x;
        if (i < n_src && (i < n_own ? pos < e.hi : pos >= e.lo)) {
          uint32_t sub = (uint32_t)((pos - e.lo) >> TILE_BUCKET_LOG2);
          if (pos < e.lo || sub >= K.nsub) { /* (never expected) */
            if (K.error)
              *K.error = 1;
            sub = K.nsub - 1;
          }
          const uint32_t k = atomicAdd (&s_cnt[sub], 1u);
          if (k < TILE_SUBCAP) {
            const uint32_t off = (uint32_t)(pos - e.lo) & ((1u << TILE_BUCKET_LOG2) - 1);
            s_key[sub * TILE_SUBCAP + k] = (off << K.len_bits) | (lmask - (rec[q].z & lmask));
            where[q] = sub << 8 | k;
          }
        }
      }
      __syncthreads ();
      /* every wave sums the buckets' counts for itself (one lane per bucket) */
      const uint32_t mine = lane < K.nsub ? s_cnt[lane] : 0u;
      const uint32_t incl = wave_incl_scan (mine);
      crowded = __ballot (mine > TILE_SUBCAP) != 0;
      const uint32_t kept_all = __shfl (incl, WAVE - 1, WAVE);
      if (tid == 0 && kept_all != size && K.error)
        *K.error = 1;
      if (!crowded) {
        if (wid == 0 && lane < K.nsub)
          s_off[lane] = incl - mine;
        for (uint32_t sb = wid; sb < K.nsub; sb += TILE_THREADS / WAVE) {
          const uint32_t cnt = __shfl (mine, (int)sb, WAVE);
          if (cnt == 0)
            continue;
          uint32_t *key = s_key + sb * TILE_SUBCAP;
          uint16_t *rank = s_rank + sb * TILE_SUBCAP;
          if (cnt <= WAVE)
            tile_wave_rank<1> (key, rank, cnt, lane);
          else if (cnt <= 2 * WAVE)
            tile_wave_rank<2> (key, rank, cnt, lane);
          else
            tile_wave_rank<3> (key, rank, cnt, lane);
        }
        __syncthreads ();
#pragma unroll
        for (uint32_t q = 0; q < TILE_PER; q++) {
          if (q * TILE_THREADS >= n_src)
            break;
          if (where[q] != NONE) {
            const uint32_t sb = where[q] >> 8, k = where[q] & 255u;
            *reinterpret_cast<uint4 *> (&K.out[out_begin + s_off[sb] + s_rank[sb * TILE_SUBCAP + k]]) = rec[q];
          }
        }
        continue;
      }
    }
    /* crowded: left to tile_crowded_kernel (a kernel of its own: its registers -- a bucket's records
     * held between loads and stores -- would halve this one's occupancy) */
    if (tid == 0)
      K.crowded[1 + atomicAdd (&K.crowded[0], 1u)] = d;
  }
}

/* Crowded tiles (more than TILE_SRC_MAX source records, or a bucket of more than TILE_SUBCAP), a
 * block each. */
__global__ __launch_bounds__ (TILE_THREADS) void
tile_crowded_kernel (TileK K) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char tile_smem[]; /* tile_lds_bytes (K.nsub) */
  uint32_t *ctr = reinterpret_cast<uint32_t *> (tile_smem);
  uint32_t *s_clist = reinterpret_cast<uint32_t *> (tile_smem + TILE_CROWDED_POSITIONS * 4);
  __shared__ uint32_t s_plist[TILE_LIST];
  __shared__ uint32_t s_cnt[TILE_NSUB], s_off[TILE_NSUB], s_cur[TILE_NSUB];
  __shared__ uint32_t s_nown, s_flag, s_part[TILE_THREADS / WAVE];
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  if (tile_total (K) > K.capacity)
    return;
  const uint32_t n_list = K.crowded[0];
  for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
    const uint32_t d = K.crowded[1 + li];
    const uint32_t out_begin = K.begin[d], size = K.begin[d + 1] - out_begin;
    const TileEntry e = K.dir[d];
    TileEntry ep = K.dir[d ? d - 1 : 0];
    if (d == 0)
      ep.n = ep.n_late = 0;
    const uint32_t n_own = e.n, n_prev = ep.n_late < ep.n ? ep.n_late : ep.n, n_src = n_own + n_prev;
    /* crowded tile: the chunk lists in LDS (they may be long), then ... */
    __syncthreads ();
    if (tid == 0) {
      s_plist[0] = 0;
      if (n_prev)
        s_flag = tile_build_list (K.chunk_prev, ep.end_slot, n_prev, s_plist, TILE_LIST);
      else
        s_flag = 1;
      s_nown = tile_build_list (K.chunk_prev, e.end_slot, n_own, s_clist, TILE_CLIST);
    }
    __syncthreads ();
    const uint32_t nl_own = s_nown, nl_prev = s_flag;
    auto source = [&] (uint32_t i) -> uint4 {
      const unsigned long long slot = i < n_own ? tile_slot_listed (K.chunk_prev, s_clist, nl_own, e.end_slot, i)
                                                : tile_slot_listed (K.chunk_prev, s_plist, nl_prev, ep.end_slot, i - n_own);
      return *reinterpret_cast<const uint4 *> (&K.raw[slot]);
    };
    auto kept = [&] (uint32_t i, unsigned long long pos) -> bool { return i < n_own ? pos < e.hi : pos >= e.lo; };
    /* Crowded tile (more than TILE_SRC_MAX source records, or a bucket of more than TILE_SUBCAP):
     * the source is read twice -- the buckets' counts, then the records written to the tile's
     * stretch of the output GROUPED by bucket -- and every bucket is then put in order where it
     * lies: up to TILE_SRC_MAX records by a counting sort over its 2,048 positions with the
     * records held in registers between the loads and the stores (order_count_role does the same
     * out of place for a bucket of the general path); a bucket with more than that -- more than a
     * record per position -- the same way but streaming the tile's source twice more.  (First
     * form: every bucket streamed the whole tile twice; 2 GiB with 0.18 records per symbol took
     * 42 ms where the three general passes take 18.) */
    constexpr uint32_t POSITIONS = TILE_CROWDED_POSITIONS, PER = POSITIONS / TILE_THREADS;
    static_assert (POSITIONS == (1u << TILE_BUCKET_LOG2), "the crowded path sorts a bucket at a time");
    auto bucket_of = [&] (unsigned long long pos) -> uint32_t {
      uint32_t sub = (uint32_t)((pos - e.lo) >> TILE_BUCKET_LOG2);
      if (pos < e.lo || sub >= K.nsub) { /* (never expected) */
        if (K.error)
          *K.error = 1;
        sub = K.nsub - 1;
      }
      return sub;
    };
    __syncthreads ();
    if (tid < TILE_NSUB)
      s_cnt[tid] = 0;
    __syncthreads ();
    for (uint32_t i = tid; i < n_src; i += TILE_THREADS) {
      const uint4 rec = source (i);
      const unsigned long long pos = ((unsigned long long)rec.y << 32) | rec.x;
      if (kept (i, pos))
        atomicAdd (&s_cnt[bucket_of (pos)], 1u);
    }
    __syncthreads ();
    {
      const uint32_t mine = lane < K.nsub ? s_cnt[lane] : 0u;
      const uint32_t incl = wave_incl_scan (mine);
      if (wid == 0 && lane < K.nsub)
        s_off[lane] = s_cur[lane] = incl - mine;
      const uint32_t kept_all = __shfl (incl, WAVE - 1, WAVE);
      if (tid == 0 && kept_all != size && K.error)
        *K.error = 1;
    }
    __syncthreads ();
    for (uint32_t i = tid; i < n_src; i += TILE_THREADS) {
      const uint4 rec = source (i);
      const unsigned long long pos = ((unsigned long long)rec.y << 32) | rec.x;
      if (kept (i, pos)) {
        const uint32_t slot = atomicAdd (&s_cur[bucket_of (pos)], 1u);
        if (slot < size)
          *reinterpret_cast<uint4 *> (&K.out[out_begin + slot]) = rec;
      }
    }
    /* the block's own stores, then its loads of them (one CU: the L1 all its waves share) */
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
    __syncthreads ();
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
    uint32_t done = 0;
    for (uint32_t sb = 0; sb < K.nsub; sb++) {
      const uint32_t in_bucket = s_cnt[sb];
      if (in_bucket == 0)
        continue;
      ACMRecord *dst = K.out + out_begin + s_off[sb];
      const bool held = in_bucket <= TILE_SRC_MAX; /* the bucket's records fit the block's registers */
      const unsigned long long b_lo = e.lo + (unsigned long long)sb * POSITIONS, b_hi = b_lo + POSITIONS;
      __syncthreads ();
      for (uint32_t i = tid; i < POSITIONS; i += TILE_THREADS)
        ctr[i] = 0;
      uint4 rec[TILE_PER];
      if (held) {
#pragma unroll
        for (uint32_t q = 0; q < TILE_PER; q++) {
          const uint32_t i = q * TILE_THREADS + tid;
          rec[q] = make_uint4 (0, 0, 0, 0);
          if (i < in_bucket)
            rec[q] = *reinterpret_cast<const uint4 *> (&dst[i]);
        }
      }
      __syncthreads ();
      if (held) {
#pragma unroll
        for (uint32_t q = 0; q < TILE_PER; q++)
          if (q * TILE_THREADS + tid < in_bucket)
            atomicAdd (&ctr[(uint32_t)((((unsigned long long)rec[q].y << 32) | rec[q].x) - b_lo) & (POSITIONS - 1)], 1u);
      } else {
        for (uint32_t i = tid; i < n_src; i += TILE_THREADS) {
          const uint4 r = source (i);
          const unsigned long long pos = ((unsigned long long)r.y << 32) | r.x;
          if (kept (i, pos) && pos >= b_lo && pos < b_hi)
            atomicAdd (&ctr[(uint32_t)(pos - b_lo)], 1u);
        }
      }
      __syncthreads ();
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
      __syncthreads (); /* (every load of the bucket's records lies in front of this: the stores may begin) */
      if (held) {
#pragma unroll
        for (uint32_t q = 0; q < TILE_PER; q++)
          if (q * TILE_THREADS + tid < in_bucket) {
            const uint32_t slot = atomicAdd (&ctr[(uint32_t)((((unsigned long long)rec[q].y << 32) | rec[q].x) - b_lo) & (POSITIONS - 1)], 1u);
            *reinterpret_cast<uint4 *> (&dst[slot]) = rec[q];
          }
      } else {
        for (uint32_t i = tid; i < n_src; i += TILE_THREADS) {
          const uint4 r = source (i);
          const unsigned long long pos = ((unsigned long long)r.y << 32) | r.x;
          if (kept (i, pos) && pos >= b_lo && pos < b_hi) {
            const uint32_t slot = atomicAdd (&ctr[(uint32_t)(pos - b_lo)], 1u);
            *reinterpret_cast<uint4 *> (&dst[slot]) = r;
          }
        }
      }
      __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
      __syncthreads ();
      __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
      /* records of one position: longest first (their lengths differ; there are few of them) */
      for (uint32_t i = tid; i < POSITIONS; i += TILE_THREADS) {
        const uint32_t end = ctr[i], begin = i ? ctr[i - 1] : 0u;
        for (uint32_t a = begin + 1; a < end; a++) { /* insertion sort */
          const uint4 r = *reinterpret_cast<const uint4 *> (&dst[a]);
          uint32_t at = a;
          while (at > begin) {
            const uint4 prev = *reinterpret_cast<const uint4 *> (&dst[at - 1]);
            if (prev.z >= r.z)
              break;
            *reinterpret_cast<uint4 *> (&dst[at]) = prev;
            at--;
          }
          if (at != a)
            *reinterpret_cast<uint4 *> (&dst[at]) = r;
        }
      }
      done += in_bucket;
    }
    if (tid == 0 && done != size && K.error)
      *K.error = 1;
  }
}
