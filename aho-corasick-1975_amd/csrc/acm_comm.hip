/* acm_comm.hip -- acm_gpu_comm_*: the gather of the ranks' ordered records on a root over RCCL, for
 * callers that run one process per GPU (include/acm_gpu.h has the contract).  Part of
 * libac75_amd.so; no device code of its own (the 8-byte wire form is acm_gpu_pack_records_device /
 * acm_gpu_unpack_records_device).
 *
 * Reference model: one shared read-only machine, one cursor per worker
 * (/root/reference/README.md:364, aho_corasick.h:70); BASELINE's "final RCCL gather of match
 * records over xGMI".  RCCL is not linked: its entry points are looked up in librccl.so when the
 * first of these calls is made, so that single-GPU users never load it. */
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <vector>

#include "acm_gpu.h"

namespace {

/* the part of rccl.h this file uses (same ABI: ncclUniqueId is 128 opaque bytes passed by value,
 * ncclUint64 = 5, ncclSuccess = 0; /opt/rocm/include/rccl/rccl.h:43,52,464) */
struct NcclId {
  char internal[128];
};
constexpr int NCCL_UINT64 = 5;
struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId) (NcclId *) = nullptr;
  int (*CommInitRank) (void **, int, NcclId, int) = nullptr;
  int (*CommDestroy) (void *) = nullptr;
  int (*AllGather) (const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*Send) (const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv) (void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart) () = nullptr;
  int (*GroupEnd) () = nullptr;
  const char *(*GetErrorString) (int) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl *
rccl () {
  std::call_once (g_rccl_once, [] {
    const char *named = getenv ("ACM_GPU_COMM_LIB");
    const char *names[] = { named, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char *n : names) {
      if (!n || !*n)
        continue;
      g_rccl.lib = dlopen (n, RTLD_NOW | RTLD_LOCAL);
      if (g_rccl.lib || n == named) /* (a library named by the caller is that one or none) */
        break;
    }
    if (!g_rccl.lib) {
      fprintf (stderr, "acm_gpu_comm: cannot load librccl.so (%s)\n", dlerror ());
      return;
    }
    auto sym = [] (const char *s) -> void * {
      void *p = dlsym (g_rccl.lib, s);
      if (!p)
        fprintf (stderr, "acm_gpu_comm: %s not found in the communication library\n", s);
      return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype (g_rccl.GetUniqueId)> (sym ("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype (g_rccl.CommInitRank)> (sym ("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype (g_rccl.CommDestroy)> (sym ("ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<decltype (g_rccl.AllGather)> (sym ("ncclAllGather"));
    g_rccl.Send = reinterpret_cast<decltype (g_rccl.Send)> (sym ("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype (g_rccl.Recv)> (sym ("ncclRecv"));
    g_rccl.GroupStart = reinterpret_cast<decltype (g_rccl.GroupStart)> (sym ("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype (g_rccl.GroupEnd)> (sym ("ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<decltype (g_rccl.GetErrorString)> (sym ("ncclGetErrorString"));
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.Send && g_rccl.Recv &&
                g_rccl.GroupStart && g_rccl.GroupEnd;
  });
  return g_rccl.ok ? &g_rccl : nullptr;
}

#define COMM_HIP(call)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      fprintf (stderr, "acm_gpu_comm: %s -> %s\n", #call, hipGetErrorString (e_));            \
      return ACM_GPU_E_HIP;                                                                   \
    }                                                                                         \
  } while (0)
#define COMM_NCCL(R, call)                                                                    \
  do {                                                                                        \
    int e_ = (call);                                                                          \
    if (e_ != 0) {                                                                            \
      fprintf (stderr, "acm_gpu_comm: %s -> %s\n", #call, (R)->GetErrorString ? (R)->GetErrorString (e_) : "error"); \
      return ACM_GPU_E_COMM;                                                                  \
    }                                                                                         \
  } while (0)

constexpr int META = 4; /* words a rank tells the others: records, first position, wire form, capacity (root) */

} // namespace

struct ACMComm {
  void *comm = nullptr;
  int rank = 0, world = 1, root = 0, device = 0;
  uint64_t *d_meta = nullptr; /* [world + 1][META]: everybody's, then this rank's own */
  uint64_t *h_meta = nullptr; /* pinned, same shape */
  /* 8-byte words: this rank's packed records (a sender) or everybody's as they arrive (the root); grow-only */
  uint64_t *stage = nullptr;
  uint64_t stage_cap = 0;
};

extern "C" int
acm_gpu_comm_unique_id (void *id_128_bytes) {
  const Rccl *R = rccl ();
  if (!R)
    return ACM_GPU_E_COMM;
  if (!id_128_bytes)
    return ACM_GPU_E_ARG;
  NcclId id;
  COMM_NCCL (R, R->GetUniqueId (&id));
  memcpy (id_128_bytes, id.internal, sizeof id.internal);
  return ACM_GPU_OK;
}

extern "C" int
acm_gpu_comm_init_rank (const void *id_128_bytes, int rank, int world, void **nccl_comm) {
  const Rccl *R = rccl ();
  if (!R)
    return ACM_GPU_E_COMM;
  if (!id_128_bytes || !nccl_comm || world < 1 || rank < 0 || rank >= world)
    return ACM_GPU_E_ARG;
  NcclId id;
  memcpy (id.internal, id_128_bytes, sizeof id.internal);
  *nccl_comm = nullptr;
  COMM_NCCL (R, R->CommInitRank (nccl_comm, world, id, rank));
  return ACM_GPU_OK;
}

extern "C" int
acm_gpu_comm_free (void *nccl_comm) {
  const Rccl *R = rccl ();
  if (!R)
    return ACM_GPU_E_COMM;
  if (nccl_comm)
    COMM_NCCL (R, R->CommDestroy (nccl_comm));
  return ACM_GPU_OK;
}

extern "C" void
acm_gpu_comm_destroy (ACMComm *c) {
  if (!c)
    return;
  (void)hipSetDevice (c->device);
  if (c->d_meta)
    (void)hipFree (c->d_meta);
  if (c->h_meta)
    (void)hipHostFree (c->h_meta);
  if (c->stage)
    (void)hipFree (c->stage);
  delete c;
}

extern "C" int
acm_gpu_comm_create (void *nccl_comm, int rank, int world, int root, ACMComm **out) {
  if (!out)
    return ACM_GPU_E_ARG;
  *out = nullptr;
  if (!nccl_comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
    return ACM_GPU_E_ARG;
  if (!rccl ())
    return ACM_GPU_E_COMM;
  ACMComm *c = new ACMComm;
  c->comm = nccl_comm;
  c->rank = rank;
  c->world = world;
  c->root = root;
  if (hipGetDevice (&c->device) != hipSuccess) {
    delete c;
    return ACM_GPU_E_NODEVICE;
  }
  const size_t bytes = (size_t)(world + 1) * META * sizeof (uint64_t);
  if (hipMalloc (reinterpret_cast<void **> (&c->d_meta), bytes) != hipSuccess ||
      hipHostMalloc (reinterpret_cast<void **> (&c->h_meta), bytes, hipHostMallocDefault) != hipSuccess) {
    acm_gpu_comm_destroy (c);
    return ACM_GPU_E_NOMEM;
  }
  *out = c;
  return ACM_GPU_OK;
}

extern "C" int
acm_gpu_comm_gather_records (ACMComm *c, const ACMPlan *plan, const ACMRecord *d_local, uint64_t n_local, uint64_t pos_lo, uint64_t span,
                             ACMRecord *d_all, uint64_t capacity, uint64_t *n_total, uint64_t *counts, void *stream) {
  const Rccl *R = rccl ();
  if (!R)
    return ACM_GPU_E_COMM;
  if (!c || !n_total || (n_local && !d_local) || (c->rank == c->root && capacity && !d_all))
    return ACM_GPU_E_ARG;
  hipStream_t st = static_cast<hipStream_t> (stream);
  COMM_HIP (hipSetDevice (c->device));
  const int W = c->world;
  /* what this rank has, and in which form it will send it */
  uint32_t pb = 0, lb = 0, kb = 0;
  const char *wire_env = getenv ("ACM_GPU_WIRE"); /* 0: 16-byte records on the wire */
  const bool wire = plan && !(wire_env && atoi (wire_env) == 0) && acm_gpu_wire_bits (plan, span, &pb, &lb, &kb) == ACM_GPU_OK;
  uint64_t *mine = c->h_meta + (size_t)W * META;
  mine[0] = n_local;
  mine[1] = pos_lo;
  mine[2] = wire ? (1ull | (uint64_t)pb << 8 | (uint64_t)lb << 16) : 0ull;
  mine[3] = c->rank == c->root ? capacity : 0ull;
  COMM_HIP (hipMemcpyAsync (c->d_meta + (size_t)W * META, mine, META * sizeof (uint64_t), hipMemcpyHostToDevice, st));
  COMM_NCCL (R, R->AllGather (c->d_meta + (size_t)W * META, c->d_meta, META, NCCL_UINT64, c->comm, st));
  COMM_HIP (hipMemcpyAsync (c->h_meta, c->d_meta, (size_t)W * META * sizeof (uint64_t), hipMemcpyDeviceToHost, st));
  COMM_HIP (hipStreamSynchronize (st));
  uint64_t total = 0, packed_words = 0;
  for (int r = 0; r < W; r++) {
    const uint64_t *m = c->h_meta + (size_t)r * META;
    if (counts)
      counts[r] = m[0];
    total += m[0];
    if (r != c->root && (m[2] & 1u))
      packed_words += m[0];
  }
  *n_total = total;
  if (total > c->h_meta[(size_t)c->root * META + 3])
    return ACM_GPU_E_OVERFLOW; /* (every rank sees the root's capacity: all of them stop here) */
  const bool is_root = c->rank == c->root;
  const uint64_t need = is_root ? packed_words : (wire ? n_local : 0);
  if (need > c->stage_cap) {
    if (c->stage)
      COMM_HIP (hipFree (c->stage));
    c->stage = nullptr;
    c->stage_cap = 0;
    if (hipMalloc (reinterpret_cast<void **> (&c->stage), need * sizeof (uint64_t)) != hipSuccess)
      return ACM_GPU_E_NOMEM;
    c->stage_cap = need;
  }
  if (!is_root) {
    if (n_local == 0)
      return ACM_GPU_OK;
    if (wire) {
      int rc = acm_gpu_pack_records_device (d_local, n_local, pos_lo, pb, lb, c->stage, st);
      if (rc)
        return rc;
      COMM_NCCL (R, R->Send (c->stage, n_local, NCCL_UINT64, c->root, c->comm, st));
    } else
      COMM_NCCL (R, R->Send (d_local, n_local * 2, NCCL_UINT64, c->root, c->comm, st));
    return ACM_GPU_OK;
  }
  /* the root: its own records by a copy on the device, everybody else's in one group of receives */
  uint64_t off = 0, soff = 0;
  COMM_NCCL (R, R->GroupStart ());
  for (int r = 0; r < W; r++) {
    const uint64_t *m = c->h_meta + (size_t)r * META;
    if (r != c->root && m[0]) {
      if (m[2] & 1u) {
        COMM_NCCL (R, R->Recv (c->stage + soff, m[0], NCCL_UINT64, r, c->comm, st));
        soff += m[0];
      } else
        COMM_NCCL (R, R->Recv (d_all + off, m[0] * 2, NCCL_UINT64, r, c->comm, st));
    }
    off += m[0];
  }
  COMM_NCCL (R, R->GroupEnd ());
  off = soff = 0;
  for (int r = 0; r < W; r++) {
    const uint64_t *m = c->h_meta + (size_t)r * META;
    if (r == c->root) {
      if (m[0])
        COMM_HIP (hipMemcpyAsync (d_all + off, d_local, m[0] * sizeof (ACMRecord), hipMemcpyDeviceToDevice, st));
    } else if (m[0] && (m[2] & 1u)) {
      int rc = acm_gpu_unpack_records_device (c->stage + soff, m[0], m[1], (uint32_t)(m[2] >> 8) & 0xFFu, (uint32_t)(m[2] >> 16) & 0xFFu, d_all + off, st);
      if (rc)
        return rc;
      soff += m[0];
    }
    off += m[0];
  }
  return ACM_GPU_OK;
}
