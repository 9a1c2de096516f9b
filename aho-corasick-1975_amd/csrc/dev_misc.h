/* dev_misc.h -- classmap, patch, sort-key and synthetic-text kernels.
 * Device code of libac75_amd.so; included by acm_gpu.hip inside its anonymous namespace (one
 * translation unit: the kernels share the structs and helpers declared there and in dev_emit.h). */

/* ------------------------------------------------------------------ comparator classes
 * Plans of machines flattened over comparator classes (acm_flatten_classes) map the text to
 * class ids first: one table of 65,536 16-bit entries in LDS serves both symbol sizes -- 2-byte
 * symbols index it directly, bytes go through it two at a time ((class(hi) << 8) | class(lo)), so
 * either way it is one ds_read_u16 per two bytes of text. */
__global__ __launch_bounds__ (1024) void
classmap_kernel (const uint4 *__restrict__ in, uint4 *__restrict__ out, uint64_t n_blocks16, const uint16_t *__restrict__ lut) {
  extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
  {
    uint4 *dst = reinterpret_cast<uint4 *> (smem);
    const uint4 *src = reinterpret_cast<const uint4 *> (lut);
    for (uint32_t i = threadIdx.x; i < 65536 * 2 / 16; i += blockDim.x)
      dst[i] = src[i];
  }
  __syncthreads ();
  const uint16_t *l = reinterpret_cast<const uint16_t *> (smem);
  auto map2 = [&] (uint32_t w) -> uint32_t { return (uint32_t)l[w & 0xFFFFu] | ((uint32_t)l[w >> 16] << 16); };
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_blocks16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = in[i];
    out[i] = make_uint4 (map2 (v.x), map2 (v.y), map2 (v.z), map2 (v.w));
  }
}

/* element-wise form for the last bytes of a buffer and for buffers that are not 16-byte aligned */
template <typename SYM>
__global__ void
classmap_tail_kernel (const SYM *__restrict__ in, SYM *__restrict__ out, uint64_t begin, uint64_t n, const uint16_t *__restrict__ lut) {
  for (uint64_t i = begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = (SYM)lut[in[i]]; /* a byte v indexes entry (0 << 8) | v: class(v) in its low byte */
}

/* ------------------------------------------------------------------ 8-byte symbols
 * The dictionary's distinct symbols get the ids 1 .. K (ACMFlatView::keys64), every other symbol
 * of a text the id 0; the text is mapped to 4-byte ids through an open-addressing hash table
 * {key, id} (at most half full, linear probing) and walked by the kernels for 4-byte symbols. */
__device__ __forceinline__ uint64_t
intern_hash (uint64_t x) { /* splitmix64 finaliser */
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void
intern_kernel (const uint64_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n, const uint4 *__restrict__ table, uint32_t mask) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t k = in[i];
    uint32_t h = (uint32_t)intern_hash (k) & mask, id = 0;
    for (;;) {
      const uint4 slot = table[h]; /* {key lo, key hi, id (0: empty slot), -} */
      if (slot.z == 0)
        break;
      if (slot.x == (uint32_t)k && slot.y == (uint32_t)(k >> 32)) {
        id = slot.z;
        break;
      }
      h = (h + 1) & mask;
    }
    out[i] = id;
  }
}

/* ------------------------------------------------------------------ comparator classes of 4-byte symbols
 * The text is mapped to class ids (what the tables were flattened over) through an open-addressing
 * table {symbol, class + 1} that holds every symbol classified so far.  A symbol met for the first
 * time claims a slot with the value CLS32_PENDING and goes on the list the host classifies with
 * the machine's comparator; the pass is then repeated (its output was provisional). */
constexpr uint32_t CLS32_PENDING = 0xFFFFFFFFu;
__global__ void
classify32_kernel (const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n, unsigned long long *table, uint32_t mask,
                   uint32_t *unknown, uint32_t *unknown_count, uint32_t unknown_cap) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t k = in[i];
    uint32_t h = (uint32_t)intern_hash (k) & mask, cls = 0;
    for (uint32_t probes = 0; probes <= mask; probes++) {
      unsigned long long slot = table[h]; /* {symbol, class + 1}; 0 = empty */
      if ((uint32_t)(slot >> 32) == 0) {
        /* the list is full: claim nothing more (the table has room for the symbols the host knows
         * plus one list's worth: claims beyond that would fill it and turn every later miss into a
         * walk over all of it).  The pass is repeated anyway once the host has classified the list. */
        if (__hip_atomic_load (unknown_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= unknown_cap)
          break;
        const unsigned long long mine = ((unsigned long long)CLS32_PENDING << 32) | k;
        const unsigned long long old = atomicCAS (&table[h], 0ull, mine);
        if (old == 0) {
          const uint32_t at = atomicAdd (unknown_count, 1u);
          if (at < unknown_cap)
            unknown[at] = k;
          break;
        }
        slot = old;
      }
      if ((uint32_t)slot == k) {
        const uint32_t v = (uint32_t)(slot >> 32);
        cls = v == CLS32_PENDING ? 0u : v - 1u;
        break;
      }
      h = (h + 1) & mask;
    }
    out[i] = cls;
  }
}

/* ------------------------------------------------------------------ incremental updates
 * word patches for the tables of the start-parallel kernel: {table, index, value, -} */
struct PatchTables {
  uint32_t *t[5];
};
__global__ void
patch_kernel (PatchTables T, const uint4 *__restrict__ patches, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint4 q = patches[i];
    T.t[q.x][q.y] = q.z;
  }
}

/* ------------------------------------------------------------------ sort keys */
__global__ void
make_keys_kernel (const ACMRecord *rec, uint64_t n, uint32_t len_bits, uint64_t pos_lo, uint64_t *keys) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint64_t lmask = (1ull << len_bits) - 1;
    keys[i] = ((rec[i].end_pos - pos_lo) << len_bits) | (lmask - (rec[i].length & lmask));
  }
}

struct Rec16 {
  uint64_t a, b;
};

/* ------------------------------------------------------------------ records on the wire (include/acm_gpu.h: acm_gpu_pack_records_device)
 * A shard's ordered records hold far less than their 16 bytes: end positions within the shard's
 * range, lengths up to lmax, ids below the number of keywords.  Packed to 8 bytes --
 * (end_pos - pos_lo) | length << pos_bits | keyword_id << (pos_bits + len_bits) -- they are what a
 * shard sends to the root over its one xGMI link; the root unpacks them where they belong.  Two
 * records per thread and load / store instruction on the 16-byte side. */
__global__ void
pack_records_kernel (const ACMRecord *__restrict__ rec, uint64_t n, uint64_t pos_lo, uint32_t pos_bits, uint32_t len_bits, uint64_t *__restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint4 r = *reinterpret_cast<const uint4 *> (rec + i);
    const uint64_t pos = (((uint64_t)r.y << 32) | r.x) - pos_lo;
    out[i] = pos | ((uint64_t)r.z << pos_bits) | ((uint64_t)r.w << (pos_bits + len_bits));
  }
}
__global__ void
unpack_records_kernel (const uint64_t *__restrict__ in, uint64_t n, uint64_t pos_lo, uint32_t pos_bits, uint32_t len_bits, ACMRecord *__restrict__ rec) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t pmask = (1ull << pos_bits) - 1, lmask = (1ull << len_bits) - 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t v = in[i];
    const uint64_t pos = (v & pmask) + pos_lo;
    *reinterpret_cast<uint4 *> (rec + i) = make_uint4 ((uint32_t)pos, (uint32_t)(pos >> 32), (uint32_t)((v >> pos_bits) & lmask), (uint32_t)(v >> (pos_bits + len_bits)));
  }
}

/* ------------------------------------------------------------------ synthetic text */
__device__ __forceinline__ uint64_t
splitmix64 (uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <typename SYM>
__global__ void
synth_text_kernel (SYM *text, uint64_t n, uint64_t gbegin, uint32_t vocab, const SYM *kw, const uint32_t *kw_off, uint32_t n_kw) {
  constexpr uint64_t P = 4096;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; li < n; li += stride) {
    const uint64_t i = gbegin + li;
    const uint64_t p = i & ~(P - 1);
    uint64_t v = sizeof (SYM) == 1 ? (uint64_t)'a' + splitmix64 (i + 42) % 26 : splitmix64 (i + 42) % vocab;
    if (n_kw) {
      const uint64_t off = p + splitmix64 (p) % (P - 16);
      const uint32_t k = (uint32_t)(splitmix64 (p + 99) % n_kw);
      const uint32_t len = kw_off[k + 1] - kw_off[k];
      if (i >= off && i < off + len)
        v = kw[kw_off[k] + (i - off)];
    }
    text[li] = (SYM)v;
  }
}

/* ------------------------------------------------------------------ plans with a delta (acm_gpu_plan_update)
 * the scans of a plan and of its delta add to one running total; this hands it to the caller's
 * counter and leaves the total zero for the next scan */
__global__ void
finish_count_kernel (unsigned long long *total, unsigned long long *user_count) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *user_count = *total;
    *total = 0;
  }
}
