/* loopback_comm.hip -- a stand-in for librccl.so inside ONE process (test infrastructure only).
 * The entry points acm_comm.hip looks up (ncclGetUniqueId, ncclCommInitRank, ncclAllGather,
 * ncclSend, ncclRecv, ncclGroupStart / End, ...) for "ranks" that are threads of one process on one
 * GPU: collectives meet at a barrier, a send leaves its buffer in a mailbox, the matching receive
 * copies it device to device.  It lets the product's gather (who sends what to whom, offsets, the
 * 8-byte wire form, the too-small root buffer) run at world sizes of 2 and more on a one-GPU box;
 * it says nothing about RCCL itself (that is what the world-size-1 test and an 8-GPU node are for).
 * ACM_GPU_COMM_LIB=tests/helpers/libloopback_comm.so selects it. */
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct Id {
  char internal[128];
};
struct Group {                        /* one per unique id */
  int world = 0, joined = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0, generation = 0;    /* barrier */
  std::vector<const void *> gather_src;
  struct Mail {
    const void *buf = nullptr;
    size_t bytes = 0;
    bool full = false;
  };
  std::vector<Mail> mail;             /* [src * world + dst] */
};
struct Comm {
  Group *g;
  int rank;
};
std::mutex g_mu;
std::map<uint64_t, Group *> g_groups;
uint64_t g_next = 1;
struct Op {
  bool send;
  void *buf;
  size_t bytes;
  int peer;
  Comm *c;
  hipStream_t st;
};
thread_local std::vector<Op> t_ops;
thread_local int t_depth = 0;

void
barrier (Group *g) {
  std::unique_lock<std::mutex> lk (g->mu);
  const int gen = g->generation;
  if (++g->arrived == g->world) {
    g->arrived = 0;
    g->generation++;
    g->cv.notify_all ();
  } else
    g->cv.wait (lk, [&] { return g->generation != gen; });
}
size_t
width (int dtype) { return dtype == 5 || dtype == 4 || dtype == 8 ? 8 : (dtype <= 1 ? 1 : 4); }

int
run (std::vector<Op> &ops) {
  /* sends first (they only leave a note), then the receives, then every send waits until it was taken */
  for (Op &o : ops)
    if (o.send) {
      if (hipStreamSynchronize (o.st) != hipSuccess) /* (what the sender queued in front of it: the packing) */
        return 1;
      Group *g = o.c->g;
      std::unique_lock<std::mutex> lk (g->mu);
      Group::Mail &m = g->mail[(size_t)o.c->rank * g->world + o.peer];
      g->cv.wait (lk, [&] { return !m.full; });
      m.buf = o.buf;
      m.bytes = o.bytes;
      m.full = true;
      g->cv.notify_all ();
    }
  for (Op &o : ops)
    if (!o.send) {
      Group *g = o.c->g;
      const void *src;
      {
        std::unique_lock<std::mutex> lk (g->mu);
        Group::Mail &m = g->mail[(size_t)o.peer * g->world + o.c->rank];
        g->cv.wait (lk, [&] { return m.full; });
        if (m.bytes != o.bytes)
          return 2; /* (a gather that disagrees about a count would hang real RCCL; here it is an error) */
        src = m.buf;
      }
      if (hipMemcpyAsync (o.buf, src, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess || hipStreamSynchronize (o.st) != hipSuccess)
        return 1;
      std::unique_lock<std::mutex> lk (g->mu);
      g->mail[(size_t)o.peer * g->world + o.c->rank].full = false;
      g->cv.notify_all ();
    }
  for (Op &o : ops)
    if (o.send) {
      Group *g = o.c->g;
      std::unique_lock<std::mutex> lk (g->mu);
      Group::Mail &m = g->mail[(size_t)o.c->rank * g->world + o.peer];
      g->cv.wait (lk, [&] { return !m.full; });
    }
  return 0;
}
} // namespace

extern "C" {
int
ncclGetUniqueId (Id *id) {
  std::lock_guard<std::mutex> lk (g_mu);
  memset (id, 0, sizeof *id);
  const uint64_t k = g_next++;
  memcpy (id->internal, &k, sizeof k);
  g_groups[k] = new Group;
  return 0;
}
int
ncclCommInitRank (void **comm, int world, Id id, int rank) {
  uint64_t k;
  memcpy (&k, id.internal, sizeof k);
  Group *g;
  {
    std::lock_guard<std::mutex> lk (g_mu);
    auto it = g_groups.find (k);
    if (it == g_groups.end ())
      return 3;
    g = it->second;
  }
  {
    std::lock_guard<std::mutex> lk (g->mu);
    if (g->world == 0) {
      g->world = world;
      g->gather_src.assign (world, nullptr);
      g->mail.assign ((size_t)world * world, Group::Mail ());
    } else if (g->world != world)
      return 3;
    g->joined++;
  }
  *comm = new Comm{ g, rank };
  return 0;
}
int
ncclCommDestroy (void *comm) {
  delete static_cast<Comm *> (comm);
  return 0;
}
int
ncclAllGather (const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t st) {
  Comm *c = static_cast<Comm *> (comm);
  Group *g = c->g;
  const size_t bytes = count * width (dtype);
  if (hipStreamSynchronize (st) != hipSuccess)
    return 1;
  {
    std::lock_guard<std::mutex> lk (g->mu);
    g->gather_src[c->rank] = send;
  }
  barrier (g);
  for (int r = 0; r < g->world; r++)
    if (hipMemcpyAsync (static_cast<char *> (recv) + (size_t)r * bytes, g->gather_src[r], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return 1;
  if (hipStreamSynchronize (st) != hipSuccess)
    return 1;
  barrier (g);
  return 0;
}
int
ncclGroupStart () {
  t_depth++;
  return 0;
}
int
ncclGroupEnd () {
  if (--t_depth > 0)
    return 0;
  std::vector<Op> ops;
  ops.swap (t_ops);
  return run (ops);
}
int
ncclSend (const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) {
  t_ops.push_back (Op{ true, const_cast<void *> (buf), count * width (dtype), peer, static_cast<Comm *> (comm), st });
  if (t_depth == 0) {
    std::vector<Op> ops;
    ops.swap (t_ops);
    return run (ops);
  }
  return 0;
}
int
ncclRecv (void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) {
  t_ops.push_back (Op{ false, buf, count * width (dtype), peer, static_cast<Comm *> (comm), st });
  if (t_depth == 0) {
    std::vector<Op> ops;
    ops.swap (t_ops);
    return run (ops);
  }
  return 0;
}
const char *
ncclGetErrorString (int e) { return e == 2 ? "loopback: a receive and its send disagree about the size" : (e == 3 ? "loopback: bad communicator id" : "loopback: HIP error"); }
}
