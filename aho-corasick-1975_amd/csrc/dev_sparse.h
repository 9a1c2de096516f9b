/* dev_sparse.h -- scan_sparse_kernel: automaton walk for 2- and 4-byte symbols.
 * Device code of libac75_amd.so; included by acm_gpu.hip inside its anonymous namespace (one
 * translation unit: the kernels share the structs and helpers declared there and in dev_emit.h). */

/* ------------------------------------------------------------------ sparse kernel (2- and 4-byte symbols)
 * Large alphabets (token ids, UTF-16 units) make automata whose root has thousands of children
 * and whose other states have a handful: a random symbol usually leads from anywhere back to the
 * root or one of its children.  So:
 *   - the root row is a direct-indexed table by symbol value (in LDS when it fits: 128 KB);
 *   - every other state is a 32-byte record {fail, n_edges, edge_begin | sym0, next0, sym1, next1}:
 *     two 16-byte loads issued together resolve a state with at most two children whose
 *     failure state is the root -- anything else takes the general side (sparse_resolve);
 *   - a lane walks SPARSE_S independent chunks; its text comes 128 bytes (one cache line) at a
 *     time into registers, so that every line is fetched once, in one burst.
 * Ownership by END position: a chunk is warmed up over the lmax - 1 symbols before it. */
constexpr int SPARSE_S = 2;
constexpr int SPARSE_THREADS = 1024;
struct SparseK {
  const uint4 *srec;   /* 2 x uint4 per state */
  const uint2 *sedge;  /* per goto edge, rows in ascending symbol order: {symbol, next | out flag << 31} */
  const uint32_t *lut; /* root transitions by symbol value: next | out flag << 31 (0: stay at the root) */
  uint32_t lut_size;   /* symbols >= lut_size bisect the root row instead */
  uint32_t R;          /* 128-byte sub-chunks per lane-stream chunk */
  uint32_t warm_subs;  /* sub-chunks walked before a chunk (>= lmax - 1 symbols) */
  uint32_t warm_skip;  /* symbols at the start of the warm-up that need not be walked */
  uint32_t queue_off;  /* LDS: [lut if staged][16 queues][tile counter] */
};

__device__ __forceinline__ uint32_t
sparse_find (const SparseK &K, uint32_t begin, uint32_t ne, uint32_t c) {
  uint32_t lo = begin, hi = begin + ne;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (K.sedge[mid].x < c)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo < begin + ne) {
    const uint2 e = K.sedge[lo];
    if (e.x == c)
      return e.y;
  }
  return NONE;
}

/* delta(t, c) in full: goto edge of t, else of f(t), ... else stay at the root (reference :167-192) */
__device__ __noinline__ uint32_t
sparse_resolve (const SparseK &K, uint32_t t, uint32_t c) {
  for (;;) {
    if (t == 0 && c < K.lut_size)
      return K.lut[c];
    const uint4 a = K.srec[2 * t];
    const uint32_t ent = sparse_find (K, a.z, a.y, c);
    if (ent != NONE)
      return ent;
    if (t == 0)
      return 0;
    t = a.x;
  }
}

template <int I>
__device__ __forceinline__ uint32_t
word_of (const uint4 &v) {
  if constexpr (I == 0)
    return v.x;
  else if constexpr (I == 1)
    return v.y;
  else if constexpr (I == 2)
    return v.z;
  else
    return v.w;
}

template <typename SYM, bool LUT_LDS, bool COUNT_ONLY>
__device__ __forceinline__ void
sparse_step (const SparseK &K, const EmitCtx &E, uint2 *queue, uint32_t &qn, uint32_t (&s)[SPARSE_S],
             const uint32_t (&tok)[SPARSE_S], const bool (&act)[SPARSE_S], const bool (&emit)[SPARSE_S],
             const uint32_t (&pos)[SPARSE_S], uint32_t lane) {
  constexpr int S = SPARSE_S;
  uint32_t rootent[S], ent[S];
  uint4 ra[S], rb[S];
  bool slow[S], any_slow = false;
#pragma unroll
  for (int q = 0; q < S; q++) {
    ra[q] = K.srec[2 * s[q]];
    rb[q] = K.srec[2 * s[q] + 1];
  }
#pragma unroll
  for (int q = 0; q < S; q++) {
    const uint32_t c = tok[q];
    rootent[q] = NONE;
    if (c < K.lut_size) {
      if (LUT_LDS) /* the table starts at LDS address 0 */
        rootent[q] = *reinterpret_cast<const __attribute__ ((address_space (3))) uint32_t *> (c * 4u);
      else
        rootent[q] = K.lut[c];
    }
  }
#pragma unroll
  for (int q = 0; q < S; q++) {
    const uint32_t c = tok[q];
    uint32_t e = rootent[q];
    bool need = false;
    if (s[q] != 0) {
      const uint32_t ne = ra[q].y;
      if (ne >= 1 && rb[q].x == c)
        e = rb[q].y;
      else if (ne >= 2 && rb[q].z == c)
        e = rb[q].w;
      else if (ne > 2 || ra[q].x != 0)
        need = true; /* more edges to look at, or a failure state other than the root */
    }
    if (e == NONE)
      need = true; /* symbol beyond the root table */
    ent[q] = e;
    slow[q] = act[q] && need;
    any_slow |= slow[q];
  }
  if (__ballot (any_slow)) {
#pragma unroll
    for (int q = 0; q < S; q++)
      if (slow[q])
        ent[q] = sparse_resolve (K, s[q], tok[q]);
  }
  bool hit[S], any_hit = false;
#pragma unroll
  for (int q = 0; q < S; q++) {
    if (act[q])
      s[q] = ent[q] & 0x7FFFFFFFu;
    hit[q] = act[q] && emit[q] && (ent[q] >> 31);
    any_hit |= hit[q];
  }
  if (__ballot (any_hit)) {
#pragma unroll
    for (int q = 0; q < S; q++)
      queue_push<false, COUNT_ONLY> (E, queue, qn, hit[q], pos[q], s[q], lane);
  }
}

/* compile-time loop over the symbols of a 128-byte sub-chunk (text registers statically indexed) */
template <typename SYM, bool LUT_LDS, bool COUNT_ONLY, int J, int JN> struct SubLoop {
  static __device__ __forceinline__ void
  run (const SparseK &K, const EmitCtx &E, uint2 *queue, uint32_t &qn, uint32_t (&s)[SPARSE_S],
       const uint4 (&d)[SPARSE_S][8], const uint32_t (&base)[SPARSE_S], const uint32_t (&lim)[SPARSE_S], uint32_t skip,
       bool own, uint32_t lane) {
    constexpr int PER = 16 / (int)sizeof (SYM);     /* symbols per 16-byte register group */
    constexpr int PERW = 4 / (int)sizeof (SYM);     /* symbols per 32-bit word */
    if ((uint32_t)J >= skip) {
      uint32_t tok[SPARSE_S], pos[SPARSE_S];
      bool act[SPARSE_S], emit[SPARSE_S];
#pragma unroll
      for (int q = 0; q < SPARSE_S; q++) {
        const uint32_t w = word_of<(J % PER) / PERW> (d[q][J / PER]);
        tok[q] = sizeof (SYM) == 4 ? w : (w >> (8 * sizeof (SYM) * (J % PERW))) & ((1u << (8 * (sizeof (SYM) & 3))) - 1u);
        act[q] = (uint32_t)J < lim[q];
        pos[q] = base[q] + J;
        emit[q] = own && pos[q] >= E.emit_from;
      }
      sparse_step<SYM, LUT_LDS, COUNT_ONLY> (K, E, queue, qn, s, tok, act, emit, pos, lane);
    }
    SubLoop<SYM, LUT_LDS, COUNT_ONLY, J + 1, JN>::run (K, E, queue, qn, s, d, base, lim, skip, own, lane);
  }
};
template <typename SYM, bool LUT_LDS, bool COUNT_ONLY, int JN> struct SubLoop<SYM, LUT_LDS, COUNT_ONLY, JN, JN> {
  static __device__ __forceinline__ void
  run (const SparseK &, const EmitCtx &, uint2 *, uint32_t &, uint32_t (&)[SPARSE_S], const uint4 (&)[SPARSE_S][8],
       const uint32_t (&)[SPARSE_S], const uint32_t (&)[SPARSE_S], uint32_t, bool, uint32_t) {}
};

/* A.range_end = number of lane-stream chunks (K.R sub-chunks each) that cover the segment; tile t
 * = chunks [t * 64 * S, (t + 1) * 64 * S); blocks own contiguous tiles, handed to their waves
 * through an LDS counter. */
template <typename SYM, bool LUT_LDS, bool COUNT_ONLY>
__global__ __launch_bounds__ (SPARSE_THREADS) void
scan_sparse_kernel (SparseK K, EmitCtx E, Launch A, const unsigned char *__restrict__ text) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  constexpr int S = SPARSE_S;
  constexpr uint32_t TPS = 128 / sizeof (SYM);
  if (LUT_LDS) {
    uint4 *dst = reinterpret_cast<uint4 *> (smem);
    const uint4 *src = reinterpret_cast<const uint4 *> (K.lut);
    for (uint32_t i = threadIdx.x; i < (K.lut_size + 3) / 4; i += blockDim.x)
      dst[i] = src[i];
  }
  uint32_t *next_tile = reinterpret_cast<uint32_t *> (smem + K.queue_off + (SPARSE_THREADS / WAVE) * QCAP * 8);
  if (threadIdx.x == 0)
    *next_tile = 0;
  __syncthreads ();

  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t wib = uniform (threadIdx.x / WAVE);
  uint2 *queue = reinterpret_cast<uint2 *> (smem + K.queue_off) + wib * QCAP;
  const uint32_t nchunks = A.range_end;
  const uint32_t ntiles = (nchunks + WAVE * S - 1) / (WAVE * S);
  const uint32_t tiles_per_block = (ntiles + gridDim.x - 1) / gridDim.x;
  const uint32_t blk_begin = blockIdx.x * tiles_per_block;
  const uint32_t blk_tiles = blk_begin >= ntiles ? 0 : (ntiles - blk_begin < tiles_per_block ? ntiles - blk_begin : tiles_per_block);
  const uint32_t last_blk = (uint32_t)(((uint64_t)A.n * sizeof (SYM) - 1) / 16); /* last 16-byte block with a valid symbol */
  const uint4 *text16 = reinterpret_cast<const uint4 *> (text);
  uint32_t qn = 0;

  for (;;) {
    uint32_t t = 0;
    if (lane == 0)
      t = atomicAdd (next_tile, 1u);
    t = uniform (t);
    if (t >= blk_tiles)
      break;
    const uint32_t tile = blk_begin + t;
    uint32_t s[S];
    int32_t sub_first[S];
    bool ok[S];
#pragma unroll
    for (int q = 0; q < S; q++) {
      const uint32_t c = tile * (WAVE * S) + q * WAVE + lane;
      ok[q] = c < nchunks;
      sub_first[q] = (int32_t)(c * K.R) - (int32_t)K.warm_subs;
      s[q] = 0;
    }
    const uint32_t rounds = K.R + K.warm_subs;
    for (uint32_t k = 0; k < rounds; k++) {
      uint4 d[S][8];
      uint32_t base[S], lim[S];
#pragma unroll
      for (int q = 0; q < S; q++) {
        const int32_t u = sub_first[q] + (int32_t)k;
        const bool valid = ok[q] && u >= 0;
        const uint32_t ublk = valid ? (uint32_t)u * 8u : 0u;
#pragma unroll
        for (int v = 0; v < 8; v++)
          d[q][v] = text16[ublk + v < last_blk ? ublk + v : last_blk];
        base[q] = valid ? (uint32_t)u * TPS : 0u;
        lim[q] = (valid && base[q] < A.n) ? (A.n - base[q] < TPS ? A.n - base[q] : TPS) : 0u;
      }
      const uint32_t skipped = k * TPS;
      const uint32_t skip = k < K.warm_subs ? (K.warm_skip > skipped ? (K.warm_skip - skipped < TPS ? K.warm_skip - skipped : TPS) : 0u) : 0u;
      const bool own = k >= K.warm_subs;
      SubLoop<SYM, LUT_LDS, COUNT_ONLY, 0, (int)TPS>::run (K, E, queue, qn, s, d, base, lim, skip, own, lane);
    }
  }
  flush_queue<false, COUNT_ONLY> (E, queue, qn);
}
