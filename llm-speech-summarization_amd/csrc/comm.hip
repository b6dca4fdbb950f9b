// comm.hip — the one collective of the hot path behind the C ABI (SURVEY §8b group 11, §8e): the in-place sum of the encoder's fp32
// gradient arena over the ranks of a data-parallel KD step, RCCL over xGMI on the caller's (side) stream.  The reference has no
// counterpart (ref:README.md:86 "only supports training on a single GPU with a batch size of 1"; its accumulation boundary is
// ref:trainer.py:373-384): this is the build's own contract for lifting that loop to N ranks.
// librccl is bound at run time (dlopen), so libspeechllm.so itself loads on hosts without RCCL / without a GPU; a process that already
// holds librccl (PyTorch's nccl backend) shares that copy.  One process per GPU: the communicator belongs to the device that was
// current at sl_comm_init.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace {
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional: sl_comm_abort falls back to CommDestroy without it
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // what RCCL itself says about the communicator (sl_comm_world / sl_comm_rank)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

const RcclApi& rccl() {
  std::call_once(g_rccl_once, [] {
    RcclApi& a = g_rccl;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.handle) break;
    }
    if (!a.handle) return;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
    a.AllReduce = (decltype(a.AllReduce))dlsym(a.handle, "ncclAllReduce");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
    a.CommAbort = (decltype(a.CommAbort))dlsym(a.handle, "ncclCommAbort");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
    a.CommCount = (decltype(a.CommCount))dlsym(a.handle, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(a.handle, "ncclCommUserRank");
    a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy && a.GetErrorString && a.CommCount && a.CommUserRank;
  });
  return g_rccl;
}

struct SlComm {
  uint32_t magic;
  ncclComm_t comm;
  int rank, world, device;      // rank / world: as RCCL reports them (ncclCommUserRank / ncclCommCount), checked against the caller's at init
};
constexpr uint32_t SL_COMM_MAGIC = 0x534c434du;   // "SLCM"
}  // namespace

static_assert(sizeof(ncclUniqueId) == SL_COMM_ID_BYTES, "speechllm.h: SL_COMM_ID_BYTES is sizeof(ncclUniqueId)");

#define SL_RCCL(call, what)                                                                     \
  do {                                                                                          \
    ncclResult_t r__ = (call);                                                                  \
    if (r__ != ncclSuccess) {                                                                   \
      sl_set_error("%s: %s", what, api.GetErrorString(r__));                                    \
      return SL_ERR_LAUNCH;                                                                     \
    }                                                                                           \
  } while (0)

extern "C" int sl_comm_unique_id(void* id_out) {
  SL_CHECK_ARG(id_out != nullptr, "sl_comm_unique_id: null buffer");
  const RcclApi& api = rccl();
  if (!api.ok) { sl_set_error("sl_comm_unique_id: librccl is not available on this host (%s)", dlerror() ? "dlopen failed" : "missing symbols"); return SL_ERR_UNSUPPORTED; }
  ncclUniqueId id;
  SL_RCCL(api.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int sl_comm_init(sl_comm* comm_out, const void* unique_id, int32_t rank, int32_t world) {
  SL_CHECK_ARG(comm_out != nullptr && unique_id != nullptr, "sl_comm_init: null pointer");
  SL_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "sl_comm_init: rank %d of %d", rank, world);
  *comm_out = nullptr;
  const RcclApi& api = rccl();
  if (!api.ok) { sl_set_error("sl_comm_init: librccl is not available on this host"); return SL_ERR_UNSUPPORTED; }
  int dev = 0;
  SL_HIP(hipGetDevice(&dev));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  SlComm* c = new SlComm{SL_COMM_MAGIC, nullptr, rank, world, dev};
  ncclResult_t r = api.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    sl_set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, api.GetErrorString(r));
    delete c;
    return SL_ERR_LAUNCH;
  }
  // The communicator's own account of itself: sl_comm_world / sl_comm_rank answer with what RCCL counted, never with the arguments
  // above, so a record built from them (bench.py kd_step.comm.rccl_nranks) can disagree with the job that was asked for.
  int n = -1, ur = -1;
  ncclResult_t rc = api.CommCount(c->comm, &n), rr = api.CommUserRank(c->comm, &ur);
  if (rc != ncclSuccess || rr != ncclSuccess || n != world || ur != rank) {
    sl_set_error("sl_comm_init: RCCL reports rank %d of %d for a communicator requested as rank %d of %d (%s)", ur, n, rank, world,
                 api.GetErrorString(rc != ncclSuccess ? rc : rr));
    if (api.CommAbort) (void)api.CommAbort(c->comm); else (void)api.CommDestroy(c->comm);
    delete c;
    return SL_ERR_LAUNCH;
  }
  c->world = n; c->rank = ur;
  *comm_out = (sl_comm)c;
  return 0;
}

extern "C" int sl_allreduce_sum(sl_comm comm, void* buf, int64_t count, int32_t dtype, sl_stream stream) {
  SlComm* c = (SlComm*)comm;
  SL_CHECK_ARG(c != nullptr && c->magic == SL_COMM_MAGIC, "sl_allreduce_sum: not a communicator");
  SL_CHECK_ARG(count >= 0 && (buf != nullptr || count == 0), "sl_allreduce_sum: bad buffer (count %lld)", (long long)count);
  SL_CHECK_ARG(dtype == SL_F32 || dtype == SL_BF16, "sl_allreduce_sum: bad dtype %d", (int)dtype);
  if (count == 0) return 0;
  int dev = 0;
  SL_HIP(hipGetDevice(&dev));
  SL_CHECK_ARG(dev == c->device, "sl_allreduce_sum: the communicator belongs to device %d, current device is %d", c->device, dev);
  const RcclApi& api = rccl();
  SL_RCCL(api.AllReduce(buf, buf, (size_t)count, dtype == SL_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
  return 0;
}

extern "C" int sl_comm_destroy(sl_comm comm) {
  SlComm* c = (SlComm*)comm;
  if (!c) return 0;
  SL_CHECK_ARG(c->magic == SL_COMM_MAGIC, "sl_comm_destroy: not a communicator");
  const RcclApi& api = rccl();
  ncclResult_t r = api.ok ? api.CommDestroy(c->comm) : ncclSuccess;
  c->magic = 0;
  delete c;
  if (r != ncclSuccess) { sl_set_error("ncclCommDestroy: %s", api.GetErrorString(r)); return SL_ERR_LAUNCH; }
  return 0;
}

// Local tear-down of a communicator whose peers may never have made theirs (a start-up that the group voted to abandon, or a
// communicator obtained after this rank had already given up): ncclCommAbort does not wait for the other ranks.
extern "C" int sl_comm_abort(sl_comm comm) {
  SlComm* c = (SlComm*)comm;
  if (!c) return 0;
  SL_CHECK_ARG(c->magic == SL_COMM_MAGIC, "sl_comm_abort: not a communicator");
  const RcclApi& api = rccl();
  ncclResult_t r = ncclSuccess;
  if (api.ok) r = api.CommAbort ? api.CommAbort(c->comm) : api.CommDestroy(c->comm);
  c->magic = 0;
  delete c;
  if (r != ncclSuccess) { sl_set_error("ncclCommAbort: %s", api.GetErrorString(r)); return SL_ERR_LAUNCH; }
  return 0;
}

// Asked of RCCL at every call (ncclCommUserRank / ncclCommCount on the live communicator), -1 when it cannot answer
extern "C" int32_t sl_comm_rank(sl_comm comm) {
  const SlComm* c = (const SlComm*)comm;
  if (!c || c->magic != SL_COMM_MAGIC) return -1;
  const RcclApi& api = rccl();
  int r = -1;
  return (api.ok && api.CommUserRank(c->comm, &r) == ncclSuccess) ? r : -1;
}
extern "C" int32_t sl_comm_world(sl_comm comm) {
  const SlComm* c = (const SlComm*)comm;
  if (!c || c->magic != SL_COMM_MAGIC) return -1;
  const RcclApi& api = rccl();
  int n = -1;
  return (api.ok && api.CommCount(c->comm, &n) == ncclSuccess) ? n : -1;
}
