// prl_wsync: trainer -> inference-worker weight broadcast over RCCL / xGMI.
//
// Replaces the reference's per-parameter NCCL broadcast through vLLM's PyNcclCommunicator on a
// StatelessProcessGroup (pipelinerl/finetune_loop.py:205-292 send side,
// pipelinerl/vllm1.py:62-134 receive side, pipelinerl/torch_utils.py:70-94 bootstrap): the
// reference sends Qwen2.5-7B as 339 separate broadcasts.  Here parameters are flattened into a
// few large byte buckets by the host and each bucket is moved either with one ncclBroadcast or
// with scatter + all-gather: MI355X's xGMI is a point-to-point mesh (7 links x ~153 GB/s per
// GPU), so a 1->N ring broadcast is bound by one link (S / 153 GB/s) while scatter (rank 0 sends
// slice i to receiver i, each on its own link) + all-gather among the receivers (each pair has
// its own link) moves only S / N per link.
//
// RCCL is resolved with dlopen at first use so that libprl.so loads on hosts without it and so
// that inside a PyTorch process the already-loaded librccl.so is shared.
#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "prl_common.h"
#include "prl_wsync_plan.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  const char* override_path = getenv("PRL_RCCL_LIB");
  const char* candidates[] = {override_path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* c : candidates) {
    if (!c) continue;
    g_rccl.handle = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) return;
#define PRL_SYM(field, name)                                                        \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name)); \
  if (!g_rccl.field) return;
  PRL_SYM(GetUniqueId, "ncclGetUniqueId")
  PRL_SYM(CommInitRank, "ncclCommInitRank")
  PRL_SYM(CommDestroy, "ncclCommDestroy")
  PRL_SYM(GetErrorString, "ncclGetErrorString")
  PRL_SYM(Broadcast, "ncclBroadcast")
  PRL_SYM(Send, "ncclSend")
  PRL_SYM(Recv, "ncclRecv")
  PRL_SYM(GroupStart, "ncclGroupStart")
  PRL_SYM(GroupEnd, "ncclGroupEnd")
  PRL_SYM(CommCount, "ncclCommCount")
  PRL_SYM(CommUserRank, "ncclCommUserRank")
#undef PRL_SYM
  g_rccl.ok = true;
}

int need_rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.ok) return prl::set_error(PRL_ENOSYS, "librccl.so could not be loaded: %s", dlerror());
  return PRL_OK;
}

#define PRL_NCCL_CHECK(expr)                                                             \
  do {                                                                                   \
    ncclResult_t _r = (expr);                                                            \
    if (_r != ncclSuccess)                                                               \
      return prl::set_error(PRL_EFAULT, "%s failed: %s", #expr, g_rccl.GetErrorString(_r)); \
  } while (0)

}  // namespace

struct prl_wsync {
  ncclComm_t comm;
  int rank;
  int world;
  int device;
};

extern "C" int prl_wsync_unique_id(uint8_t uid[PRL_WSYNC_UID_BYTES]) {
  static_assert(PRL_WSYNC_UID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size mismatch");
  PRL_CHECK_ARG(uid != nullptr, "uid is null");
  if (int rc = need_rccl()) return rc;
  ncclUniqueId id;
  PRL_NCCL_CHECK(g_rccl.GetUniqueId(&id));
  memcpy(uid, id.internal, NCCL_UNIQUE_ID_BYTES);
  return PRL_OK;
}

extern "C" int prl_wsync_init(const uint8_t uid[PRL_WSYNC_UID_BYTES], int32_t rank,
                              int32_t world_size, int32_t device, prl_wsync** out) {
  PRL_CHECK_ARG(uid && out, "null argument");
  PRL_CHECK_ARG(world_size >= 1 && rank >= 0 && rank < world_size, "bad rank %d / world %d", rank,
                world_size);
  if (int rc = need_rccl()) return rc;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return prl::set_error(PRL_EFAULT, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
  ncclUniqueId id;
  memcpy(id.internal, uid, NCCL_UNIQUE_ID_BYTES);
  auto* w = new (std::nothrow) prl_wsync();
  if (!w) return prl::set_error(PRL_ENOMEM, "out of memory");
  w->rank = rank;
  w->world = world_size;
  w->device = device;
  ncclResult_t r = g_rccl.CommInitRank(&w->comm, world_size, id, rank);
  if (r != ncclSuccess) {
    delete w;
    return prl::set_error(PRL_EFAULT, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
  }
  *out = w;
  return PRL_OK;
}

extern "C" int prl_wsync_bcast_bucket(prl_wsync* w, void* bucket, uint64_t nbytes, int32_t src,
                                      prl_stream_t stream) {
  PRL_CHECK_ARG(w && (bucket || nbytes == 0), "null argument");
  PRL_CHECK_ARG(src >= 0 && src < w->world, "bad src rank %d", src);
  if (nbytes == 0 || w->world == 1) return PRL_OK;
  PRL_NCCL_CHECK(g_rccl.Broadcast(bucket, bucket, (size_t)nbytes, ncclUint8, src, w->comm,
                                  static_cast<hipStream_t>(stream)));
  return PRL_OK;
}

namespace {

// One RCCL group around a list of point-to-point ops.  A failing ncclSend / ncclRecv must not leave the group open
// (every later call on this thread would silently queue into it): the group is closed on every path and the FIRST
// error is the one reported.
template <class Ops>
int run_group(prl_wsync* w, uint8_t* base, hipStream_t s, Ops&& for_each_op) {
  PRL_NCCL_CHECK(g_rccl.GroupStart());
  ncclResult_t first = ncclSuccess;
  const char* what = "";
  for_each_op([&](const prl::wsync::Op& op) {
    if (first != ncclSuccess) return;
    const ncclResult_t r = op.send ? g_rccl.Send(base + op.off, (size_t)op.len, ncclUint8, op.peer, w->comm, s)
                                   : g_rccl.Recv(base + op.off, (size_t)op.len, ncclUint8, op.peer, w->comm, s);
    if (r != ncclSuccess) {
      first = r;
      what = op.send ? "ncclSend" : "ncclRecv";
    }
  });
  const ncclResult_t end = g_rccl.GroupEnd();
  if (first != ncclSuccess) return prl::set_error(PRL_EFAULT, "%s failed: %s", what, g_rccl.GetErrorString(first));
  if (end != ncclSuccess) return prl::set_error(PRL_EFAULT, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(end));
  return PRL_OK;
}

}  // namespace

extern "C" int prl_wsync_bcast_bucket_sag(prl_wsync* w, void* bucket, uint64_t nbytes,
                                          prl_stream_t stream) {
  PRL_CHECK_ARG(w && (bucket || nbytes == 0), "null argument");
  if (nbytes == 0 || w->world == 1) return PRL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint8_t* base = static_cast<uint8_t*>(bucket);
  // the plan (slice boundaries, who sends what to whom) lives in prl_wsync_plan.h and is executed against simulated
  // ranks by tests/harness/wsync_plan_host.cpp
  // phase 1: scatter.  rank 0 -> receiver i+1 gets slice i, every transfer on its own link.
  if (int rc = run_group(w, base, s, [&](auto&& emit) { prl::wsync::scatter_ops(w->rank, w->world, nbytes, emit); })) return rc;
  // phase 2: all-gather among the receivers (stream order makes the received slice visible).
  if (w->world > 2 && w->rank != 0) {
    if (int rc = run_group(w, base, s, [&](auto&& emit) { prl::wsync::allgather_ops(w->rank, w->world, nbytes, emit); })) return rc;
  }
  return PRL_OK;
}

extern "C" int prl_wsync_comm_size(prl_wsync* w, int32_t* world_size, int32_t* rank) {
  PRL_CHECK_ARG(w && world_size && rank, "null argument");
  int n = 0, r = 0;
  PRL_NCCL_CHECK(g_rccl.CommCount(w->comm, &n));
  PRL_NCCL_CHECK(g_rccl.CommUserRank(w->comm, &r));
  *world_size = n;
  *rank = r;
  return PRL_OK;
}

extern "C" int prl_wsync_destroy(prl_wsync* w) {
  if (!w) return PRL_OK;
  if (g_rccl.ok && w->comm) g_rccl.CommDestroy(w->comm);
  delete w;
  return PRL_OK;
}
