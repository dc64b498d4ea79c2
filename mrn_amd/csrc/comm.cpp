// Gradient collectives over RCCL / xGMI behind the C ABI (mrn_comm_init / mrn_allreduce_*): what replaces the
// replicate / scatter / gather / reduce traffic of torch.nn.DataParallel (reference il_modules/base.py:68) for a host that is
// not PyTorch.  One communicator and nothing else global per process; every collective is asynchronous on the stream it is given.
//
// RCCL is bound at mrn_comm_init() time with dlopen("librccl.so.1") -- no link-time dependency: a process that already carries an
// RCCL (PyTorch's torch.distributed does) shares that copy, every other host gets the ROCm one.  The Python learners use
// torch.distributed by default (mrn_amd/parallel.py); MRN_COMM=native routes their bucketed all-reduce through these entry points.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "common.hpp"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;

int bind_rccl() {
  if (g_rccl.handle) return MRN_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  MRN_CHECK_ARG(h != nullptr, "mrn_comm: cannot load librccl.so.1 (%s)", dlerror());
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
  g_rccl.Broadcast = (decltype(g_rccl.Broadcast))dlsym(h, "ncclBroadcast");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  MRN_CHECK_ARG(g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.Broadcast && g_rccl.CommDestroy &&
                    g_rccl.GetErrorString, "mrn_comm: librccl lacks a required symbol");
  g_rccl.handle = h;
  return MRN_OK;
}

#define MRN_NCCL(call, what)                                                        \
  do {                                                                              \
    ncclResult_t r__ = (call);                                                      \
    if (r__ != ncclSuccess) {                                                       \
      mrn_set_error("%s: RCCL error %d (%s)", what, (int)r__, g_rccl.GetErrorString(r__)); \
      return MRN_ERR_UNSUPPORTED;                                                       \
    }                                                                               \
  } while (0)

}  // namespace

MRN_EXPORT int64_t mrn_comm_unique_id_bytes(void) { return (int64_t)sizeof(ncclUniqueId); }

// rank 0 creates the rendezvous id (mrn_comm_unique_id_bytes() bytes) and hands it to the other ranks out of band
MRN_EXPORT int mrn_comm_unique_id(void* id_out) {
  MRN_CHECK_ARG(id_out, "mrn_comm_unique_id: null buffer");
  if (int rc = bind_rccl()) return rc;
  ncclUniqueId id;
  MRN_NCCL(g_rccl.GetUniqueId(&id), "mrn_comm_unique_id");
  memcpy(id_out, &id, sizeof(id));
  return MRN_OK;
}

// joins the communicator of `world` ranks (collective: every rank calls it with the same id); the current HIP device is the
// rank's GPU.  One communicator per process.
MRN_EXPORT int mrn_comm_init(int rank, int world, const void* unique_id) {
  MRN_CHECK_ARG(unique_id && world >= 1 && rank >= 0 && rank < world, "mrn_comm_init: bad arguments (rank %d of %d)", rank, world);
  MRN_CHECK_ARG(g_comm == nullptr, "mrn_comm_init: a communicator already exists (mrn_comm_destroy first)");
  if (int rc = bind_rccl()) return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  MRN_NCCL(g_rccl.CommInitRank(&g_comm, world, id, rank), "mrn_comm_init");
  g_world = world;
  g_rank = rank;
  return MRN_OK;
}

MRN_EXPORT int64_t mrn_comm_world(void) { return g_world; }
MRN_EXPORT int64_t mrn_comm_rank(void) { return g_rank; }

// in place: buf = sum over ranks (average = 0) or mean over ranks (average = 1); asynchronous on `stream`
MRN_EXPORT int mrn_allreduce_f32(float* buf, int64_t n, int average, void* stream) {
  MRN_CHECK_ARG(g_comm != nullptr, "mrn_allreduce_f32: no communicator (mrn_comm_init)");
  MRN_CHECK_ARG(buf || n == 0, "mrn_allreduce_f32: null buffer");
  if (n == 0) return MRN_OK;
  MRN_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat, average ? ncclAvg : ncclSum, g_comm, (hipStream_t)stream), "mrn_allreduce_f32");
  return MRN_OK;
}

// in place: every rank's buf = rank root's buf; asynchronous on `stream`
MRN_EXPORT int mrn_broadcast_f32(float* buf, int64_t n, int root, void* stream) {
  MRN_CHECK_ARG(g_comm != nullptr, "mrn_broadcast_f32: no communicator (mrn_comm_init)");
  MRN_CHECK_ARG((buf || n == 0) && root >= 0 && root < g_world, "mrn_broadcast_f32: bad arguments");
  if (n == 0) return MRN_OK;
  MRN_NCCL(g_rccl.Broadcast(buf, buf, (size_t)n, ncclFloat, root, g_comm, (hipStream_t)stream), "mrn_broadcast_f32");
  return MRN_OK;
}

MRN_EXPORT int mrn_comm_destroy(void) {
  if (g_comm) {
    MRN_NCCL(g_rccl.CommDestroy(g_comm), "mrn_comm_destroy");
    g_comm = nullptr;
    g_world = 0;
    g_rank = -1;
  }
  return MRN_OK;
}
