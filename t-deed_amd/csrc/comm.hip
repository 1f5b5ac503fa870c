// Data-parallel gradient reduction over RCCL (xGMI), behind the C ABI (SURVEY.md section 8b "tdeed_comm_*", 8e).
//
// The reference has no distributed code at all (single process, single GPU: model/model.py:184-190); data-parallel
// training is new functionality of this build.  One process per GPU owns one communicator and ONE dedicated HIP stream
// for collectives.  A gradient bucket is reduced as
//
//     tdeed_comm_all_reduce(comm, buf, n, dtype, compute_stream)
//         event <- record(compute_stream); comm_stream waits(event); ncclAllReduce(sum) on comm_stream
//     ... the backward keeps running on compute_stream ...
//     tdeed_comm_join(comm, compute_stream)
//         event <- record(comm_stream); compute_stream waits(event)        (before the optimizer reads the gradients)
//
// so the bucket that holds the temporal stack + heads (92 % of the 800MF gradient bytes, produced first by the backward)
// travels over xGMI while the trunk backward runs.  Both calls only enqueue work: no host synchronisation, legal under
// stream capture (the cross-stream event edges become graph dependencies).  RCCL is resolved with dlopen at the first
// tdeed_comm_* call so that single-GPU users need no RCCL at load time.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { NCCL_SUM = 0 };
enum { NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9 };      // ncclDataType_t values of rccl.h (ncclFloat32 = 7, ncclBfloat16 = 9)

struct Api {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Api g_api;

bool load_api() {
  if (g_api.lib) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    tdeed_set_error("tdeed_comm: librccl not found (%s)", dlerror());
    return false;
  }
#define TD_SYM(field, name)                                                        \
  *(void**)(&g_api.field) = dlsym(h, name);                                        \
  if (!g_api.field) { tdeed_set_error("tdeed_comm: symbol %s missing in librccl", name); return false; }
  TD_SYM(GetUniqueId, "ncclGetUniqueId")
  TD_SYM(CommInitRank, "ncclCommInitRank")
  TD_SYM(CommDestroy, "ncclCommDestroy")
  TD_SYM(AllReduce, "ncclAllReduce")
  TD_SYM(ReduceScatter, "ncclReduceScatter")
  TD_SYM(AllGather, "ncclAllGather")
  TD_SYM(GetErrorString, "ncclGetErrorString")
#undef TD_SYM
  g_api.lib = h;
  return true;
}

struct Comm {
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  int world = 1, rank = 0;
  int pending = 0;
};

#define TD_NCCL(call, what)                                                                         \
  do {                                                                                              \
    int r__ = (call);                                                                               \
    if (r__ != 0) {                                                                                 \
      tdeed_set_error("tdeed_comm: %s failed: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r__) : "?"); \
      return TDEED_ERR_RUNTIME;                                                                     \
    }                                                                                               \
  } while (0)
#define TD_HIP(call, what)                                                                          \
  do {                                                                                              \
    hipError_t e__ = (call);                                                                        \
    if (e__ != hipSuccess) {                                                                        \
      tdeed_set_error("tdeed_comm: %s failed: %s", what, hipGetErrorString(e__));                    \
      return TDEED_ERR_RUNTIME;                                                                     \
    }                                                                                               \
  } while (0)

}  // namespace

// rank 0 creates the 128-byte id and hands it to the other ranks through any side channel (torch.distributed
// broadcast, a file, the launcher's environment)
extern "C" int tdeed_comm_unique_id(void* id128) {
  TD_CHECK(id128, "comm_unique_id: null pointer");
  if (!load_api()) return TDEED_ERR_RUNTIME;
  ncclUniqueId id;
  TD_NCCL(g_api.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return TDEED_OK;
}

// collective over all `world` ranks (blocks until every rank has called it); the calling thread's current device is
// the rank's GPU
extern "C" int tdeed_comm_init(void** comm_out, const void* id128, int world, int rank) {
  TD_CHECK(comm_out && id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
  if (!load_api()) return TDEED_ERR_RUNTIME;
  Comm* c = new Comm();
  c->world = world;
  c->rank = rank;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  TD_NCCL(g_api.CommInitRank(&c->nccl, world, id, rank), "ncclCommInitRank");
  int lo = 0, hi = 0;
  TD_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi), "hipDeviceGetStreamPriorityRange");
  TD_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi), "hipStreamCreate");   // highest priority
  TD_HIP(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming), "hipEventCreate");
  TD_HIP(hipEventCreateWithFlags(&c->join, hipEventDisableTiming), "hipEventCreate");
  *comm_out = c;
  return TDEED_OK;
}

extern "C" int tdeed_comm_info(void* comm, int* world, int* rank) {
  TD_CHECK(comm, "comm_info: null communicator");
  Comm* c = (Comm*)comm;
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  return TDEED_OK;
}

// in-place sum over ranks of buf[0, n) (dtype TDEED_F32 or TDEED_BF16), enqueued on the communicator's stream behind
// everything `compute_stream` has been given so far
extern "C" int tdeed_comm_all_reduce(void* comm, void* buf, long n, int dtype, void* compute_stream) {
  TD_CHECK(comm && buf && n > 0, "comm_all_reduce: bad arguments");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "comm_all_reduce: bad dtype %d", dtype);
  Comm* c = (Comm*)comm;
  TD_HIP(hipEventRecord(c->fork, (hipStream_t)compute_stream), "hipEventRecord");
  TD_HIP(hipStreamWaitEvent(c->stream, c->fork, 0), "hipStreamWaitEvent");
  TD_NCCL(g_api.AllReduce(buf, buf, (size_t)n, dtype == TDEED_F32 ? NCCL_FLOAT32 : NCCL_BFLOAT16, NCCL_SUM, c->nccl, c->stream),
          "ncclAllReduce");
  c->pending++;
  return TDEED_OK;
}

// reduce-scatter + all-gather form of the same sum (every link carries 1/world of the buffer twice instead of a ring
// pass over the whole buffer) over the largest prefix of buf that divides evenly over the ranks; the rest
// (< world elements; none when the caller pads its buckets, optim.FlatParams does) goes out as a plain all-reduce, so
// any n is legal and no rank count silently loses the RS+AG form
extern "C" int tdeed_comm_all_reduce_rs_ag(void* comm, void* buf, long n, int dtype, void* compute_stream) {
  TD_CHECK(comm && buf && n > 0, "comm_all_reduce_rs_ag: bad arguments");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "comm_all_reduce_rs_ag: bad dtype %d", dtype);
  Comm* c = (Comm*)comm;
  const size_t per = (size_t)(n / c->world);
  const size_t main_n = per * (size_t)c->world;
  const size_t es = dtype == TDEED_F32 ? 4 : 2;
  const int dt = dtype == TDEED_F32 ? NCCL_FLOAT32 : NCCL_BFLOAT16;
  char* mine = (char*)buf + (size_t)c->rank * per * es;
  TD_HIP(hipEventRecord(c->fork, (hipStream_t)compute_stream), "hipEventRecord");
  TD_HIP(hipStreamWaitEvent(c->stream, c->fork, 0), "hipStreamWaitEvent");
  if (per > 0) {
    TD_NCCL(g_api.ReduceScatter(buf, mine, per, dt, NCCL_SUM, c->nccl, c->stream), "ncclReduceScatter");
    TD_NCCL(g_api.AllGather(mine, buf, per, dt, c->nccl, c->stream), "ncclAllGather");
  }
  if ((size_t)n > main_n) {
    char* tail = (char*)buf + main_n * es;
    TD_NCCL(g_api.AllReduce(tail, tail, (size_t)n - main_n, dt, NCCL_SUM, c->nccl, c->stream), "ncclAllReduce (tail)");
  }
  c->pending++;
  return TDEED_OK;
}

// `compute_stream` waits for every collective enqueued so far (no host wait)
extern "C" int tdeed_comm_join(void* comm, void* compute_stream) {
  TD_CHECK(comm, "comm_join: null communicator");
  Comm* c = (Comm*)comm;
  if (c->pending == 0) return TDEED_OK;
  TD_HIP(hipEventRecord(c->join, c->stream), "hipEventRecord");
  TD_HIP(hipStreamWaitEvent((hipStream_t)compute_stream, c->join, 0), "hipStreamWaitEvent");
  c->pending = 0;
  return TDEED_OK;
}

extern "C" int tdeed_comm_destroy(void* comm) {
  if (!comm) return TDEED_OK;
  Comm* c = (Comm*)comm;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->nccl && g_api.CommDestroy) g_api.CommDestroy(c->nccl);
  if (c->fork) (void)hipEventDestroy(c->fork);
  if (c->join) (void)hipEventDestroy(c->join);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return TDEED_OK;
}
