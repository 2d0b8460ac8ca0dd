// rccl_comm.hip — the collectives of the multi-GPU path as plain C entry points on RCCL (xGMI), no torch in between.
//
// The reference is single-device (pipelines/mod.rs:214-217); SURVEY §8(e) adds: weights broadcast from rank 0, decoded images
// gathered to rank 0 — and §8(f)-4 the two all-to-alls per block of single-image sequence parallelism.  A host in any language
// creates one fmi_comm per process (rank 0 makes the 128-byte id with fmi_comm_unique_id and ships it to the others by whatever
// channel it has: torch.distributed's store here, an env var / TCP in a Rust host) and then
//   * passes fmi_comm_all_to_all + the comm as the (callback, user) pair of fmi_flux_set_sequence_parallel — the exchange is then
//     one ncclAllToAll enqueued on the library's own launch stream: no Python, no host synchronisation per exchange;
//   * calls fmi_comm_broadcast on the weight arenas (fmi_flux_state_buffer) in place, and fmi_comm_gather on the u8 images.
// librccl is opened at run time (dlopen): the library keeps no link-time dependency on it, and in a process that has torch loaded
// the SONAME librccl.so.1 resolves to the copy torch already mapped, so there is one RCCL runtime per process.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "common.h"

using namespace fmi;

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  std::string err;
};
Rccl g_rccl;
std::once_flag g_once;

template <class F>
bool sym(void* h, const char* name, F& f) {
  f = reinterpret_cast<F>(dlsym(h, name));
  return f != nullptr;
}

void open_rccl() {
  Rccl& r = g_rccl;
  const char* cands[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* c : cands) {
    r.h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
    if (r.h) break;
  }
  if (!r.h) {
    r.err = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "?");
    return;
  }
  bool ok = sym(r.h, "ncclGetUniqueId", r.GetUniqueId) && sym(r.h, "ncclCommInitRank", r.CommInitRank) && sym(r.h, "ncclCommDestroy", r.CommDestroy) &&
            sym(r.h, "ncclGetErrorString", r.GetErrorString) && sym(r.h, "ncclAllToAll", r.AllToAll) && sym(r.h, "ncclBroadcast", r.Broadcast) &&
            sym(r.h, "ncclSend", r.Send) && sym(r.h, "ncclRecv", r.Recv) && sym(r.h, "ncclGroupStart", r.GroupStart) && sym(r.h, "ncclGroupEnd", r.GroupEnd);
  if (!ok) {
    r.err = "librccl.so.1 lacks a required symbol";
    r.h = nullptr;
  }
}
int need_rccl() {
  std::call_once(g_once, open_rccl);
  if (!g_rccl.h) return fail(FMI_ERR_UNSUPPORTED, "rccl: " + g_rccl.err);
  return FMI_OK;
}
int nccl_fail(const char* what, ncclResult_t rc) {
  return fail(FMI_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error"));
}
#define FMI_NCCL_TRY(what, expr)                    \
  do {                                              \
    ncclResult_t rc_ = (expr);                      \
    if (rc_ != ncclSuccess) return nccl_fail(what, rc_); \
  } while (0)
}  // namespace

struct fmi_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  unsigned long long calls = 0, bytes_sent = 0;
};

static_assert(sizeof(ncclUniqueId) == FMI_COMM_ID_BYTES, "fmi_comm id size");

// Non-collective availability check: librccl opens with every symbol and this thread has a current device.  A host calls it on
// EVERY rank and agrees on the result before the first collective call (fmi_comm_create blocks inside ncclCommInitRank until all
// ranks arrive: a rank that failed earlier would leave the healthy ones there until the RCCL timeout).
extern "C" int fmi_comm_probe(void) {
  FMI_TRY(need_rccl());
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return fail(FMI_ERR_HIP, "comm_probe: no current device");
  return FMI_OK;
}

extern "C" int fmi_comm_unique_id(void* id_out) {
  if (!id_out) return fail(FMI_ERR_INVALID, "comm_unique_id: null");
  FMI_TRY(need_rccl());
  ncclUniqueId id;
  FMI_NCCL_TRY("ncclGetUniqueId", g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return FMI_OK;
}

extern "C" int fmi_comm_create(const void* id, int rank, int world_size, fmi_comm** out) {
  if (!id || !out) return fail(FMI_ERR_INVALID, "comm_create: null");
  if (world_size < 1 || rank < 0 || rank >= world_size) return fail(FMI_ERR_INVALID, "comm_create: rank / world_size out of range");
  FMI_TRY(need_rccl());
  fmi_comm* c = new fmi_comm();
  c->rank = rank, c->world = world_size;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    return fail(FMI_ERR_HIP, "comm_create: no current device");
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclResult_t rc = g_rccl.CommInitRank(&c->comm, world_size, uid, rank);  // collective: every rank of the group calls it
  if (rc != ncclSuccess) {
    delete c;
    return nccl_fail("ncclCommInitRank", rc);
  }
  *out = c;
  return FMI_OK;
}

extern "C" void fmi_comm_destroy(fmi_comm* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  delete c;
}

extern "C" int fmi_comm_rank(const fmi_comm* c) { return c ? c->rank : -1; }
extern "C" int fmi_comm_world_size(const fmi_comm* c) { return c ? c->world : 0; }
extern "C" int fmi_comm_stats(const fmi_comm* c, unsigned long long* calls, unsigned long long* bytes_sent) {
  if (!c) return fail(FMI_ERR_INVALID, "comm_stats: null");
  if (calls) *calls = c->calls;
  if (bytes_sent) *bytes_sent = c->bytes_sent;
  return FMI_OK;
}

// fmi_all_to_all_fn: block p of `send` (bytes_per_peer bytes) goes to rank p, block p of `recv` comes from rank p, enqueued on `stream`.
extern "C" int fmi_comm_all_to_all(void* user, const void* send, void* recv, size_t bytes_per_peer, void* stream) {
  fmi_comm* c = (fmi_comm*)user;
  if (!c || !send || !recv) return fail(FMI_ERR_INVALID, "comm_all_to_all: null");
  FMI_NCCL_TRY("ncclAllToAll", g_rccl.AllToAll(send, recv, bytes_per_peer, ncclUint8, c->comm, (hipStream_t)stream));
  c->calls++;
  c->bytes_sent += (unsigned long long)bytes_per_peer * (unsigned long long)(c->world - 1);
  return FMI_OK;
}

// In-place broadcast of a device buffer from `root` (the weight arenas of fmi_flux_state_buffer), enqueued on `stream`.
extern "C" int fmi_comm_broadcast(fmi_comm* c, void* buf, size_t bytes, int root, void* stream) {
  if (!c || (!buf && bytes)) return fail(FMI_ERR_INVALID, "comm_broadcast: null");
  if (root < 0 || root >= c->world) return fail(FMI_ERR_INVALID, "comm_broadcast: root out of range");
  if (!bytes) return FMI_OK;
  FMI_NCCL_TRY("ncclBroadcast", g_rccl.Broadcast(buf, buf, bytes, ncclUint8, root, c->comm, (hipStream_t)stream));
  c->calls++;
  if (c->rank == root) c->bytes_sent += (unsigned long long)bytes;
  return FMI_OK;
}

// Gather `bytes` from every rank into `recv` on `root` (rank r's block at r * bytes; recv may be NULL elsewhere): the decoded u8
// images of a batch.  Grouped point-to-point sends, one message per peer over its own xGMI link.
extern "C" int fmi_comm_gather(fmi_comm* c, const void* send, void* recv, size_t bytes, int root, void* stream) {
  if (!c || (!send && bytes)) return fail(FMI_ERR_INVALID, "comm_gather: null");
  if (root < 0 || root >= c->world) return fail(FMI_ERR_INVALID, "comm_gather: root out of range");
  if (c->rank == root && !recv && bytes) return fail(FMI_ERR_INVALID, "comm_gather: root needs a receive buffer");
  if (!bytes) return FMI_OK;
  hipStream_t s = (hipStream_t)stream;
  FMI_NCCL_TRY("ncclGroupStart", g_rccl.GroupStart());
  // Inside the group nothing returns early: the first error is recorded and ncclGroupEnd is ALWAYS called, so the calling
  // thread never stays in group mode (later RCCL calls on it, torch's included, would be deferred and hang silently).
  ncclResult_t first = ncclSuccess;
  const char* where = nullptr;
  if (c->rank == root) {
    for (int r = 0; r < c->world && first == ncclSuccess; ++r) {
      if (r == root) continue;
      first = g_rccl.Recv((char*)recv + (size_t)r * bytes, bytes, ncclUint8, r, c->comm, s);
      if (first != ncclSuccess) where = "ncclRecv";
    }
  } else {
    first = g_rccl.Send(send, bytes, ncclUint8, root, c->comm, s);
    if (first != ncclSuccess) where = "ncclSend";
  }
  ncclResult_t end = g_rccl.GroupEnd();
  if (first != ncclSuccess) return nccl_fail(where, first);
  if (end != ncclSuccess) return nccl_fail("ncclGroupEnd", end);
  if (c->rank == root) FMI_HIP_TRY(hipMemcpyAsync((char*)recv + (size_t)root * bytes, send, bytes, hipMemcpyDeviceToDevice, s));
  else c->bytes_sent += (unsigned long long)bytes;  // counted only once the send has really been submitted
  c->calls++;
  return FMI_OK;
}
