/* dev_emit.h -- wave helpers, queue items -> records (put_outputs, walk_continuation, flush_queue, expand_items_kernel).
 * Device code of libac75_amd.so; included by acm_gpu.hip inside its anonymous namespace (one
 * translation unit: the kernels share the structs and helpers declared there and in dev_emit.h). */

/* ------------------------------------------------------------------ wave helpers */
__device__ __forceinline__ uint32_t
lane_id () {
  return __builtin_amdgcn_mbcnt_hi (~0u, __builtin_amdgcn_mbcnt_lo (~0u, 0u));
}

/* number of set bits of a ballot below this lane: two v_mbcnt (the and-and-popcount the compiler
 * makes of `popcount (m & ((1 << lane) - 1))` is four instructions, at every queue push) */
__device__ __forceinline__ uint32_t
rank_below (uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi ((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo ((uint32_t)m, 0u));
}

/* (a << s) | b in one instruction */
__device__ __forceinline__ uint32_t
lshl_or (uint32_t a, uint32_t s, uint32_t b) {
  uint32_t r;
  asm ("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(s), "v"(b));
  return r;
}

__device__ __forceinline__ uint32_t
uniform (uint32_t v) { /* tells the compiler the value is the same in every lane */
  return __builtin_amdgcn_readfirstlane (v);
}

__device__ __forceinline__ uint32_t
wave_incl_scan (uint32_t v) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    uint32_t o = __shfl_up (v, d, WAVE);
    if ((int)lane_id () >= d)
      v += o;
  }
  return v;
}

/* outputs of state st longer than `bound`, in acm_get_match index order (longest first:
 * reference aho_corasick.c:459-466): counted, and written from offset o when WRITE.  `oi` is
 * oinfo[st], already loaded by the caller. */
template <bool WRITE>
__device__ __forceinline__ uint32_t
put_outputs (const EmitCtx &E, uint4 oi, uint32_t pos, uint32_t bound, uint64_t o) {
  uint32_t cnt = 0;
  for (uint32_t left = oi.x; left; left--) {
    if (oi.z <= bound)
      break;
    if (WRITE && o + cnt < E.capacity) {
      const uint64_t gp = E.pos_base + pos;
      *reinterpret_cast<uint4 *> (&E.records[o + cnt]) = make_uint4 ((uint32_t)gp, (uint32_t)(gp >> 32), oi.z, oi.w);
    }
    cnt++;
    if (left > 1)
      oi = E.oinfo[oi.y];
  }
  return cnt;
}

/* what the walk of a continuation item found (it rarely finds more than one match position) */
struct ContResult {
  uint32_t cnt;      /* records in all */
  uint32_t events;   /* positions with records */
  uint32_t ev_pos, ev_bound; /* the first of them ... */
  uint4 ev_oi;               /* ... and the oinfo of the state reached there */
};

/* Ownership in the continuation-mode dense kernel: a match belongs to the chunk in which it
 * STARTS.  A lane starts from the root at the first byte of its chunk [cs, ce), walks the chunk
 * and then runs over into the following bytes until its state is no deeper than the number of
 * bytes past ce (no match that started before ce can still be open); during that run-over only
 * outputs longer than the bytes past ce are its own.
 *
 * Continuation item: at `pos` the lane stepped into a state s that has no row in LDS and carried
 * on from hotfail(s), the longest suffix state that has one (depth dh).  From there it still finds
 * every match that starts within the last dh symbols or later; the ones it can no longer see
 * started earlier and end after pos.  They are recovered here by walking on from s itself
 * through the HBM rows: j symbols later every output longer than j + dh -- and, past ce, longer
 * than the bytes past ce -- is such a match, and once the state is no deeper than that bound
 * nothing more can be missing.
 * One dependent load per symbol: the row entry carries the next state, its output flag and its
 * depth; the text byte of the following step is fetched alongside. */
__device__ __forceinline__ uint32_t
item_chunk_end (const EmitCtx &E, uint2 it) {
  return (it.y & IT_RUN) ? it.x + 1 - ((it.y >> IT_K_SHIFT) & 0xFFFu) : (it.x | (E.chunk - 1)) + 1;
}

template <bool WRITE>
__device__ __forceinline__ ContResult
walk_continuation (const EmitCtx &E, uint2 it, uint64_t o) {
  ContResult r = { 0, 0, 0, 0, make_uint4 (0, 0, 0, 0) };
  const uint32_t pos = it.x, st = it.y & IT_STATE;
  const uint32_t ce = item_chunk_end (E, it);
  if (E.chain && st >= E.chain_base && pos + 9 <= E.n) {
    /* all that lies below st is one path of len symbols to a leaf t (chain record, acm_gpu.hip):
     * what the walk below would find is t's keyword len symbols on, or nothing */
    const uint4 c = E.chain[st - E.chain_base];
    const uint32_t len = c.x & 15u;
    if (len) {
      uint64_t have; /* the next 8 text bytes (any alignment) */
      __builtin_memcpy (&have, E.text + pos + 1, 8);
      const uint64_t want = ((uint64_t)c.w << 32) | c.z;
      const uint32_t p = pos + len;
      if (len != 15u && ((want ^ have) << (64u - 8u * len)) == 0 && p >= E.emit_from) {
        const uint32_t past = p + 1 > ce ? p + 1 - ce : 0, jd = len + (c.x >> 4);
        const uint32_t bound = jd > past ? jd : past;
        const uint4 oi = E.oinfo[c.y];
        const uint32_t cnt = put_outputs<WRITE> (E, oi, p, bound, o);
        if (cnt) {
          r.ev_pos = p;
          r.ev_oi = oi;
          r.ev_bound = bound;
          r.events = 1;
          r.cnt = cnt;
        }
      }
      return r;
    }
  }
  const uint32_t dh = E.cont_dh[st];
  uint32_t s2 = st;
  uint32_t byte = pos + 1 < E.n ? E.text[pos + 1] : 0;
  for (uint32_t j = 1; pos + j < E.n; j++) {
    const uint32_t p = pos + j;
    const uint32_t ent = E.wrows[s2 * E.W + min (byte - E.lo, E.span)];
    byte = p + 1 < E.n ? E.text[p + 1] : 0;
    s2 = ent & IT_STATE;
    const uint32_t past = p + 1 > ce ? p + 1 - ce : 0;
    const uint32_t bound = j + dh > past ? j + dh : past;
    if ((ent >> 16) <= bound)
      break;
    if ((ent & 0x8000u) && p >= E.emit_from) {
      const uint4 oi = E.oinfo[s2];
      const uint32_t c = put_outputs<WRITE> (E, oi, p, bound, o + r.cnt);
      if (c) {
        if (!r.events) {
          r.ev_pos = p;
          r.ev_oi = oi;
          r.ev_bound = bound;
        }
        r.events++;
        r.cnt += c;
      }
    }
  }
  return r;
}

/* number of records of one queue item (walks its continuation, remembering what it found) */
template <bool CONT>
__device__ __forceinline__ uint32_t
item_count (const EmitCtx &E, bool valid, uint2 it, uint32_t &own_cnt, uint4 &own_oi, ContResult &r) {
  const uint32_t st = CONT ? (it.y & IT_STATE) : it.y;
  if (valid && st >= E.n_states) { /* cannot happen; never index the tables with it */
    if (E.error)
      *E.error = 1;
    valid = false;
  }
  const bool own = valid && (!CONT || (it.y & IT_OUT));
  own_cnt = 0;
  own_oi = make_uint4 (0, 0, 0, 0);
  if (own) {
    own_oi = E.oinfo[st];
    if (CONT && (it.y & IT_RUN)) /* run-over: only outputs longer than the bytes past the chunk */
      own_cnt = put_outputs<false> (E, own_oi, it.x, (it.y >> IT_K_SHIFT) & 0xFFFu, 0);
    else
      own_cnt = own_oi.x;
  }
  r = ContResult{ 0, 0, 0, 0, make_uint4 (0, 0, 0, 0) };
  if (CONT && valid && (it.y & IT_CONT))
    r = walk_continuation<false> (E, it, 0);
  return own_cnt + r.cnt;
}

/* writes them from offset o: the state's own outputs first, then the continuation's */
template <bool CONT>
__device__ __forceinline__ void
item_write (const EmitCtx &E, uint2 it, uint32_t own_cnt, uint4 own_oi, const ContResult &r, uint64_t o) {
  if (own_cnt)
    (void)put_outputs<true> (E, own_oi, it.x, (CONT && (it.y & IT_RUN)) ? ((it.y >> IT_K_SHIFT) & 0xFFFu) : 0u, o);
  if (r.events == 1)
    (void)put_outputs<true> (E, r.ev_oi, r.ev_pos, r.ev_bound, o + own_cnt);
  else if (r.events > 1)
    (void)walk_continuation<true> (E, it, o + own_cnt);
}

/* Expand a wave's queue into records: one global atomic per <= 64 items (wave prefix sum of the
 * per-item counts), records of one item contiguous.  Must be called by all 64 lanes. */
template <bool CONT, bool COUNT_ONLY>
__device__ __noinline__ void
flush_queue (EmitCtx E, const uint2 *queue, uint32_t n_items) {
  const uint32_t lane = lane_id ();
  for (uint32_t base = 0; base < n_items; base += WAVE) {
    const uint32_t i = base + lane;
    const bool valid = i < n_items;
    const uint2 it = valid ? queue[i] : make_uint2 (0, 0);
    uint32_t own_cnt;
    uint4 own_oi;
    ContResult r;
    const uint32_t cnt = item_count<CONT> (E, valid, it, own_cnt, own_oi, r);
    const uint32_t incl = wave_incl_scan (cnt);
    const uint32_t total = __shfl (incl, WAVE - 1, WAVE);
    unsigned long long gbase = 0;
    if (lane == 0 && total)
      gbase = atomicAdd (E.count, (unsigned long long)total);
    gbase = ((unsigned long long)__shfl ((uint32_t)(gbase >> 32), 0, WAVE) << 32) | __shfl ((uint32_t)gbase, 0, WAVE);
    if (!COUNT_ONLY && cnt)
      item_write<CONT> (E, it, own_cnt, own_oi, r, gbase + (incl - cnt));
  }
}

/* Where a wave of the dense kernel parks its queue when it fills up: a private region of the
 * plan's item buffer in HBM (plain coalesced stores, nothing to wait for).  expand_items_kernel
 * turns the parked items into records afterwards with the whole chip's parallelism; a wave whose
 * region is full expands in place instead (flush_queue), so nothing is ever dropped. */
struct Spill {
  uint2 *region;     /* this wave's region */
  uint32_t capacity; /* items per region */
  uint32_t fill;     /* items parked so far (wave-uniform) */
};

template <bool CONT, bool COUNT_ONLY, bool PARK_ONLY = false>
__device__ __forceinline__ void
queue_drain (const EmitCtx &E, const uint2 *queue, uint32_t qn, Spill *sp, uint32_t lane) {
  /* PARK_ONLY (dense kernel): the caller has made room in the region (region_make_room), so there
   * is no call to flush_queue on this path -- it runs at every slow step of 128 unrolled ones */
  if (PARK_ONLY || (sp && sp->fill + qn <= sp->capacity)) {
    /* (the lane index through an opaque move: &queue[lane] is the same at every one of the dense
     * kernel's 128 inlined slow steps, so the compiler hoisted it to the kernel's top, found no
     * register for it over the whole kernel and kept it in scratch -- the record-mode kernel's one
     * spilled VGPR; one v_lshl_add at the step that needs it instead) */
    uint32_t l = lane;
    asm volatile ("" : "+v"(l));
    for (uint32_t i = l; i < qn; i += WAVE)
      sp->region[sp->fill + i] = queue[i];
    sp->fill = uniform (sp->fill + qn);
  } else
    flush_queue<CONT, COUNT_ONLY> (E, queue, qn);
}

/* append one item per lane with `hit`; wave-uniform bookkeeping in qn */
template <bool CONT, bool COUNT_ONLY, bool PARK_ONLY = false>
__device__ __forceinline__ void
queue_push (const EmitCtx &E, uint2 *queue, uint32_t &qn, bool hit, uint32_t pos, uint32_t word, uint32_t lane,
            Spill *sp = nullptr) {
  const uint64_t m = __ballot (hit);
  if (m) {
    if (hit)
      queue[qn + rank_below (m)] = make_uint2 (pos, word);
    qn = uniform (qn + (uint32_t)__popcll (m));
    if (qn > QCAP - WAVE) {
      queue_drain<CONT, COUNT_ONLY, PARK_ONLY> (E, queue, qn, sp, lane);
      qn = 0;
    }
  }
}

/* Dense kernel, before every 16-step block: the block parks at most 16 * S * 64 items (one per
 * lane, stream and step) beside what the queue already holds; a region that cannot take that much
 * any more is expanded in place (records straight from here, one atomic per 64 items) and starts
 * over.  Regions are sized so that this happens only on texts with a match every few symbols. */
template <bool CONT, bool COUNT_ONLY, int S>
__device__ __forceinline__ void
region_make_room (const EmitCtx &E, Spill *sp) {
  if (sp->capacity - sp->fill < 16u * S * WAVE + QCAP) {
    /* flush_queue reads back, lane by lane, items that OTHER lanes of this wave parked with plain
     * stores (queue_drain), possibly at addresses an earlier flush has read before (the region
     * starts over below).  One wave's vector memory operations reach the CU's L1 in program order
     * and the L1 is shared by the whole CU, so the loads see the stores -- but nothing in the
     * program said so: the ordering is spelled out here (release at workgroup scope + the wave's
     * own stores drained) rather than left to that.  The path runs once per ~2,000 items of a
     * wave; tests/test_gpu_parity.py::test_dense_region_overflow_many_times_per_wave drives it
     * hundreds of times per wave on a one-block grid (ACM_GPU_GRID_BLOCKS=1). */
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    flush_queue<CONT, COUNT_ONLY> (E, sp->region, sp->fill);
    /* ... and the region is written again from its start only after those loads have returned */
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    sp->fill = 0;
  }
}

/* Expands the items parked by REGIONS consecutive waves of the dense kernel: THREADS threads take
 * THREADS items per round, a block-wide prefix sum of the per-item record counts gives every item
 * its slot, and ONE global atomic per round reserves the records (a single counter sustains only
 * ~90 atomics per microsecond, so they are kept to a few hundred per launch).
 * The launch leaves its own bookkeeping clean: each block zeroes the fill counters it consumed,
 * and the block that finishes last hands the total to the caller's counter (when this is the
 * last segment of a scan) and resets the running total and the ticket. */
struct ExpandTail {
  unsigned long long *user_count; /* where the caller wants the total */
  unsigned int *ticket;
  int last_segment;
};

template <bool CONT, bool COUNT_ONLY, int THREADS, int REGIONS>
__global__ __launch_bounds__ (THREADS) void
expand_items_kernel (EmitCtx E, const uint2 *items, uint32_t region_items, uint32_t *fill, ExpandTail tail) {
  __shared__ uint32_t s_off[REGIONS + 1];
  __shared__ uint32_t s_wave[THREADS / WAVE];
  __shared__ unsigned long long s_base;
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  if (tid < REGIONS) { /* fill counters of this block's regions: read and zero them in parallel */
    s_off[tid + 1] = fill[blockIdx.x * REGIONS + tid];
    fill[blockIdx.x * REGIONS + tid] = 0;
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
  }
  __syncthreads ();
  const uint32_t total = s_off[REGIONS];
  for (uint32_t base = 0; base < total; base += THREADS) {
    const uint32_t i = base + tid;
    const bool valid = i < total;
    uint2 it = make_uint2 (0, 0);
    if (valid) {
      uint32_t r = 0;
#pragma unroll
      for (int k = 1; k < REGIONS; k++)
        r += s_off[k] <= i ? 1u : 0u;
      it = items[(size_t)(blockIdx.x * REGIONS + r) * region_items + (i - s_off[r])];
    }
    uint32_t own_cnt;
    uint4 own_oi;
    ContResult res;
    const uint32_t cnt = item_count<CONT> (E, valid, it, own_cnt, own_oi, res);
    const uint32_t incl = wave_incl_scan (cnt);
    if (lane == WAVE - 1)
      s_wave[wid] = incl;
    __syncthreads ();
    if (tid == 0) {
      uint32_t acc = 0;
      for (int k = 0; k < THREADS / WAVE; k++) {
        const uint32_t v = s_wave[k];
        s_wave[k] = acc;
        acc += v;
      }
      s_base = acc ? atomicAdd (E.count, (unsigned long long)acc) : 0ull;
    }
    __syncthreads ();
    if (!COUNT_ONLY && cnt)
      item_write<CONT> (E, it, own_cnt, own_oi, res, s_base + s_wave[wid] + (incl - cnt));
    __syncthreads ();
  }
  if (tid == 0) {
    /* this block's adds to the running total have returned (their values were used), so the block
     * that draws the last ticket sees every add */
    if (atomicAdd (tail.ticket, 1u) == gridDim.x - 1) {
      *tail.ticket = 0;
      if (tail.last_segment) {
        *tail.user_count = atomicAdd (E.count, 0ull);
        *E.count = 0;
      }
    }
  }
}

/* Second form of the expansion: every thread keeps the items of up to ROUNDS rounds (and what
 * counting them found) in registers, so that the rounds' loads overlap instead of queueing behind
 * one another's atomics, and the block reserves the records of all of them with ONE atomicAdd
 * (256 per launch on config 2 instead of ~770: a single counter sustains ~90 per microsecond, and
 * each of them was a round trip between two barriers). */
template <bool CONT, bool COUNT_ONLY, int THREADS, int REGIONS, int ROUNDS>
__global__ __launch_bounds__ (THREADS) void
expand_items_once_kernel (EmitCtx E, const uint2 *items, uint32_t region_items, uint32_t *fill, ExpandTail tail) {
  constexpr int WAVES = THREADS / WAVE;
  __shared__ uint32_t s_off[REGIONS + 1];
  __shared__ uint32_t s_wave[ROUNDS][WAVES];
  __shared__ unsigned long long s_base;
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  if (tid < REGIONS) {
    s_off[tid + 1] = fill[blockIdx.x * REGIONS + tid];
    fill[blockIdx.x * REGIONS + tid] = 0;
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
  }
  __syncthreads ();
  const uint32_t total = s_off[REGIONS];
  for (uint32_t base = 0; base < total; base += THREADS * ROUNDS) {
    uint2 it[ROUNDS];
    uint32_t cnt[ROUNDS], own_cnt[ROUNDS], incl[ROUNDS];
    uint4 own_oi[ROUNDS];
    ContResult res[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; k++) {
      const uint32_t i = base + k * THREADS + tid;
      const bool valid = i < total;
      it[k] = make_uint2 (0, 0);
      if (valid) {
        uint32_t r = 0;
#pragma unroll
        for (int j = 1; j < REGIONS; j++)
          r += s_off[j] <= i ? 1u : 0u;
        it[k] = items[(size_t)(blockIdx.x * REGIONS + r) * region_items + (i - s_off[r])];
      }
    }
#pragma unroll
    for (int k = 0; k < ROUNDS; k++) {
      const bool valid = base + k * THREADS + tid < total;
      cnt[k] = item_count<CONT> (E, valid, it[k], own_cnt[k], own_oi[k], res[k]);
      incl[k] = wave_incl_scan (cnt[k]);
      if (lane == WAVE - 1)
        s_wave[k][wid] = incl[k];
    }
    __syncthreads ();
    if (tid == 0) {
      uint32_t acc = 0;
      for (int k = 0; k < ROUNDS; k++)
        for (int w = 0; w < WAVES; w++) {
          const uint32_t v = s_wave[k][w];
          s_wave[k][w] = acc;
          acc += v;
        }
      s_base = acc ? atomicAdd (E.count, (unsigned long long)acc) : 0ull;
    }
    __syncthreads ();
    if (!COUNT_ONLY) {
#pragma unroll
      for (int k = 0; k < ROUNDS; k++)
        if (cnt[k])
          item_write<CONT> (E, it[k], own_cnt[k], own_oi[k], res[k], s_base + s_wave[k][wid] + (incl[k] - cnt[k]));
    }
    __syncthreads ();
  }
  if (tid == 0) {
    if (atomicAdd (tail.ticket, 1u) == gridDim.x - 1) {
      *tail.ticket = 0;
      if (tail.last_segment) {
        *tail.user_count = atomicAdd (E.count, 0ull);
        *E.count = 0;
      }
    }
  }
}
