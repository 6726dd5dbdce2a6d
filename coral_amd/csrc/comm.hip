// Data-parallel exchange behind the C ABI: bucket collectives over RCCL (xGMI) on a communication stream owned by an
// explicit context - what accelerate's DDP wrapper / DeepSpeed ZeRO-2 do for the reference
// (accelerate/accelerator.py:1892 prepare_model -> DistributedDataParallel, :2053 backward; R/makefile:79-84), reachable
// from any host language: the Python host (coral_amd/trainer.py) calls these through ctypes, torch.distributed stays for
// the rendezvous (handing rank 0's unique id to the other ranks) and for the gloo CPU tests.
//
// RCCL is bound at run time (dlopen "librccl.so.1" on the first ca_comm_* call): the library loads, and every kernel
// entry point works, on a host without RCCL; a process that already holds RCCL (torch) shares that copy.
#include "common.h"
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>  // types and enums only: no symbol of librccl is linked

struct CaComm {
  ncclComm_t comm;
  hipStream_t stream;  // every collective of this context is enqueued here
  hipEvent_t ev;       // ordering between the communication stream and the caller's streams
  int rank, world, device;
};

namespace {
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional: ca_comm_abort falls back to CommDestroy without it
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.handle) return CA_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    ca_set_error("ca_comm: librccl.so.1 not found (%s)", dlerror());
    return CA_ERR_UNSUPPORTED;
  }
#define CA_SYM(field, name)                                         \
  do {                                                              \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);          \
    if (!g_rccl.field) {                                            \
      ca_set_error("ca_comm: librccl has no symbol %s", name);      \
      return CA_ERR_UNSUPPORTED;                                    \
    }                                                               \
  } while (0)
  CA_SYM(GetUniqueId, "ncclGetUniqueId");
  CA_SYM(CommInitRank, "ncclCommInitRank");
  CA_SYM(CommDestroy, "ncclCommDestroy");
  CA_SYM(AllReduce, "ncclAllReduce");
  CA_SYM(ReduceScatter, "ncclReduceScatter");
  CA_SYM(AllGather, "ncclAllGather");
  CA_SYM(GetErrorString, "ncclGetErrorString");
#undef CA_SYM
  g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(h, "ncclCommAbort");
  g_rccl.handle = h;
  return CA_OK;
}

#define CA_RCCL(call, what)                                                     \
  do {                                                                          \
    ncclResult_t r__ = (call);                                                  \
    if (r__ != ncclSuccess) {                                                   \
      ca_set_error("%s: RCCL: %s", what, g_rccl.GetErrorString(r__));           \
      return CA_ERR_LAUNCH;                                                     \
    }                                                                           \
  } while (0)
#define CA_HIP(call, what)                                                      \
  do {                                                                          \
    hipError_t e__ = (call);                                                    \
    if (e__ != hipSuccess) {                                                    \
      ca_set_error("%s: %s", what, hipGetErrorString(e__));                     \
      return CA_ERR_LAUNCH;                                                     \
    }                                                                           \
  } while (0)

bool dtype_of(int dtype, ncclDataType_t& dt, size_t& esz) {
  if (dtype == CA_COMM_F32) {
    dt = ncclFloat32;
    esz = 4;
    return true;
  }
  if (dtype == CA_COMM_BF16) {
    dt = ncclBfloat16;
    esz = 2;
    return true;
  }
  return false;
}
}  // namespace

extern "C" int ca_comm_unique_id(void* id128) {
  CA_CHECK_ARG(id128, "ca_comm_unique_id: null pointer");
  const int rc = rccl_load();
  if (rc != CA_OK) return rc;
  static_assert(sizeof(ncclUniqueId) == CA_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  CA_RCCL(g_rccl.GetUniqueId(&id), "ca_comm_unique_id");
  memcpy(id128, &id, sizeof(id));
  return CA_OK;
}

extern "C" int ca_comm_init(CaComm** out, const void* id128, int32_t rank, int32_t world) {
  CA_CHECK_ARG(out && id128 && world >= 1 && rank >= 0 && rank < world, "ca_comm_init: bad argument (rank %d of %d)", rank, world);
  const int rc = rccl_load();
  if (rc != CA_OK) return rc;
  CaComm* c = new CaComm();
  c->rank = rank;
  c->world = world;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    ca_set_error("ca_comm_init: no current device");
    return CA_ERR_LAUNCH;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    ca_set_error("ca_comm_init: RCCL: %s", g_rccl.GetErrorString(r));
    delete c;
    return CA_ERR_LAUNCH;
  }
  // (a failure from here on gives the communicator and the context back: nothing of a half-built context survives)
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    e = hipEventCreateWithFlags(&c->ev, hipEventDisableTiming);
    if (e != hipSuccess) hipStreamDestroy(c->stream);
  }
  if (e != hipSuccess) {
    ca_set_error("ca_comm_init: %s", hipGetErrorString(e));
    g_rccl.CommDestroy(c->comm);
    delete c;
    return CA_ERR_LAUNCH;
  }
  *out = c;
  return CA_OK;
}

extern "C" int ca_comm_destroy(CaComm* c) {
  if (!c) return CA_OK;
  hipStreamSynchronize(c->stream);
  if (g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  hipEventDestroy(c->ev);
  hipStreamDestroy(c->stream);
  delete c;
  return CA_OK;
}

// Tear a context down WITHOUT waiting for its stream: for a communicator whose peers never joined a collective (its
// kernel would spin for ever; ncclCommAbort ends it).  The rendezvous fallback of the host side uses it.
extern "C" int ca_comm_abort(CaComm* c) {
  if (!c) return CA_OK;
  if (!g_rccl.CommAbort) {
    // Without ncclCommAbort nothing can end a collective kernel whose peers never joined: destroying the communicator
    // or synchronising its stream would block for ever - the very case this call exists for.  The context (communicator,
    // stream, event) is leaked instead, and the caller is told.
    ca_set_error("ca_comm_abort: librccl has no ncclCommAbort; the context is left behind (not destroyed, not waited for)");
    fprintf(stderr, "coral_amd: ca_comm_abort: librccl has no ncclCommAbort; leaking the communicator context\n");
    return CA_ERR_UNSUPPORTED;
  }
  g_rccl.CommAbort(c->comm);
  hipStreamSynchronize(c->stream);  // (the aborted kernels have left the stream)
  hipEventDestroy(c->ev);
  hipStreamDestroy(c->stream);
  delete c;
  return CA_OK;
}

extern "C" void* ca_comm_stream(CaComm* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int ca_comm_rank(CaComm* c) { return c ? c->rank : -1; }
extern "C" int ca_comm_world(CaComm* c) { return c ? c->world : -1; }

// the communication stream waits for everything enqueued so far on `producer` (the bucket's gradients)
extern "C" int ca_comm_after(CaComm* c, void* producer) {
  CA_CHECK_ARG(c, "ca_comm_after: null context");
  CA_HIP(hipEventRecord(c->ev, (hipStream_t)producer), "ca_comm_after(record)");
  CA_HIP(hipStreamWaitEvent(c->stream, c->ev, 0), "ca_comm_after(wait)");
  return CA_OK;
}
// `consumer` waits for every collective enqueued so far on the communication stream
extern "C" int ca_comm_before(CaComm* c, void* consumer) {
  CA_CHECK_ARG(c, "ca_comm_before: null context");
  CA_HIP(hipEventRecord(c->ev, c->stream), "ca_comm_before(record)");
  CA_HIP(hipStreamWaitEvent((hipStream_t)consumer, c->ev, 0), "ca_comm_before(wait)");
  return CA_OK;
}

extern "C" int ca_allreduce_bucket(CaComm* c, void* buf, int64_t n, int32_t dtype) {
  ncclDataType_t dt;
  size_t esz;
  CA_CHECK_ARG(c && buf && n > 0 && dtype_of(dtype, dt, esz), "ca_allreduce_bucket: bad argument");
  CA_RCCL(g_rccl.AllReduce(buf, buf, (size_t)n, dt, ncclSum, c->comm, c->stream), "ca_allreduce_bucket");
  return CA_OK;
}

extern "C" int ca_reduce_scatter_bucket(CaComm* c, void* buf, int64_t n_per_rank, int32_t dtype) {
  ncclDataType_t dt;
  size_t esz;
  CA_CHECK_ARG(c && buf && n_per_rank > 0 && dtype_of(dtype, dt, esz), "ca_reduce_scatter_bucket: bad argument");
  char* mine = (char*)buf + (size_t)c->rank * (size_t)n_per_rank * esz;  // RCCL's in-place form: recv = send + rank * count
  CA_RCCL(g_rccl.ReduceScatter(buf, mine, (size_t)n_per_rank, dt, ncclSum, c->comm, c->stream), "ca_reduce_scatter_bucket");
  return CA_OK;
}

extern "C" int ca_allgather_bucket(CaComm* c, void* buf, int64_t n_per_rank, int32_t dtype) {
  ncclDataType_t dt;
  size_t esz;
  CA_CHECK_ARG(c && buf && n_per_rank > 0 && dtype_of(dtype, dt, esz), "ca_allgather_bucket: bad argument");
  const char* mine = (const char*)buf + (size_t)c->rank * (size_t)n_per_rank * esz;  // in place: send = recv + rank * count
  CA_RCCL(g_rccl.AllGather(mine, buf, (size_t)n_per_rank, dt, c->comm, c->stream), "ca_allgather_bucket");
  return CA_OK;
}
