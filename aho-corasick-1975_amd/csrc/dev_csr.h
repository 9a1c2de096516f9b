/* dev_csr.h -- scan_csr_kernel: goto/failure walk over CSR rows, any symbol width and alignment.
 * Device code of libac75_amd.so; included by acm_gpu.hip inside its anonymous namespace (one
 * translation unit: the kernels share the structs and helpers declared there and in dev_emit.h). */

/* ------------------------------------------------------------------ CSR kernel (any width) */
__device__ __forceinline__ uint32_t
csr_step (const CsrTables &T, uint32_t s, uint32_t c) {
  for (;;) {
    uint32_t b = T.row_ptr[s], e = T.row_ptr[s + 1];
    if (e - b > 8) { /* rows are sorted by numeric symbol value */
      while (e - b > 1) {
        uint32_t m = (b + e) >> 1;
        if (T.edge_sym[m] <= c)
          b = m;
        else
          e = m;
      }
      if (T.edge_sym[b] == c)
        return T.edge_next[b];
    } else {
      for (; b < e; b++)
        if (T.edge_sym[b] == c)
          return T.edge_next[b];
    }
    if (s == 0)
      return 0;
    s = T.fail[s];
  }
}

/* One lane walks `chunk` symbols of [range_begin, range_end), restarting from the root lmax-1
 * symbols earlier (or at index 0 of the segment).  blockDim.x == 64: one wave per block. */
template <typename SYM, bool COUNT_ONLY>
__global__ __launch_bounds__ (WAVE) void
scan_csr_kernel (CsrTables T, EmitCtx E, Launch A, uint32_t chunk) {
  __shared__ uint2 queue[QCAP];
  const uint32_t lane = threadIdx.x;
  const SYM *text = reinterpret_cast<const SYM *> (A.text);
  const uint32_t nchunks = (A.range_end - A.range_begin + chunk - 1) / chunk;
  const uint32_t per_round = gridDim.x * WAVE;
  const uint32_t rounds = (nchunks + per_round - 1) / per_round;
  const uint32_t warm = T.lmax > 1 ? T.lmax - 1 : 0;
  uint32_t qn = 0;
  for (uint32_t r = 0; r < rounds; r++) {
    const uint32_t ck = (r * gridDim.x + blockIdx.x) * WAVE + lane;
    uint32_t begin = A.range_end, end = A.range_end, i = A.range_end;
    if (ck < nchunks) {
      begin = A.range_begin + ck * chunk;
      end = A.range_end - begin > chunk ? begin + chunk : A.range_end;
      i = begin > warm ? begin - warm : 0;
    }
    uint32_t s = 0;
    /* all lanes iterate together so that the queue stays a wave-level structure */
    const uint32_t steps_max = chunk + warm;
    for (uint32_t k = 0; k < steps_max; k++, i++) {
      bool hit = false;
      if (i < end) {
        s = csr_step (T, s, (uint32_t)text[i]);
        hit = i >= begin && i >= A.emit_from && T.nb_outputs[s] != 0;
      }
      queue_push<false, COUNT_ONLY> (E, queue, qn, hit, i, s, lane);
    }
  }
  flush_queue<false, COUNT_ONLY> (E, queue, qn);
}
