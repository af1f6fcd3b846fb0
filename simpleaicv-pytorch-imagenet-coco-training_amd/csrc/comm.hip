// Data-parallel gradient exchange over RCCL (xGMI) at the C ABI.
//
// Replaces the DistributedDataParallel wrapper of the reference (tools/utils.py:166-185 build_training_mode;
// process group: tools/train_detection_model.py:50-59) for the gradient all-reduce and DDP's per-forward buffer
// broadcast.  One communicator per process (= per GPU).  All-reduces run on a communicator-owned side HIP stream
// that is event-chained to the producer stream, so a bucket's transfer overlaps with the backward kernels enqueued
// after it; sacv_comm_wait joins the side stream back into the consumer (optimizer) stream.
//
// RCCL is resolved at run time (dlopen of librccl.so.1, preferring a copy the process has already loaded, e.g.
// PyTorch's) so libsacv_hip.so itself has no link-time dependency on it and single-GPU users never touch it.
#include "sacv_common.h"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include <string.h>
#include <string>

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi g_api;
std::once_flag g_api_once;
std::string g_api_error = "(not attempted)";  // written once under g_api_once, read-only afterwards

void load_api() {
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    const char* e = dlerror();  // NULL once consumed: never hand it to a %s
    g_api_error = e ? e : "dlopen failed (no message)";
    return;
  }
  g_api.handle = h;
  g_api.GetUniqueId = (decltype(g_api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_api.CommInitRank = (decltype(g_api.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_api.CommDestroy = (decltype(g_api.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_api.AllReduce = (decltype(g_api.AllReduce))dlsym(h, "ncclAllReduce");
  g_api.Broadcast = (decltype(g_api.Broadcast))dlsym(h, "ncclBroadcast");
  g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(h, "ncclGetErrorString");
  g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllReduce && g_api.Broadcast &&
             g_api.GetErrorString;
  g_api_error = g_api.ok ? "" : "librccl.so.1 loaded but a required symbol is missing";
}

const RcclApi* api() {
  std::call_once(g_api_once, load_api);
  return g_api.ok ? &g_api : nullptr;
}

struct SacvComm {
  ncclComm_t comm = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t ready = nullptr;  // producer -> side
  hipEvent_t done = nullptr;   // side -> consumer
  int rank = 0, world = 1;
  bool pending = false;
};

}  // namespace

#define SACV_CHECK_RCCL(call, what)                                                      \
  do {                                                                                   \
    ncclResult_t r__ = (call);                                                           \
    if (r__ != ncclSuccess) {                                                            \
      sacv_set_error("%s: RCCL error %d: %s", what, (int)r__, A->GetErrorString(r__));   \
      return SACV_ERR_LAUNCH;                                                            \
    }                                                                                    \
  } while (0)
#define SACV_CHECK_HIP(call, what)                                              \
  do {                                                                          \
    hipError_t e__ = (call);                                                    \
    if (e__ != hipSuccess) {                                                    \
      sacv_set_error("%s: HIP error: %s", what, hipGetErrorString(e__));        \
      return SACV_ERR_LAUNCH;                                                   \
    }                                                                           \
  } while (0)

// Stand-in for RCCL's kernels on a box with ONE GPU (sacv_comm_emulate_load): `workgroups` 256-thread workgroups, each
// holding `lds` bytes of LDS, that stay resident for `us` microseconds -- they stream a small buffer (a collective's
// workgroups move bytes, they do not compute) and sleep in between.  Every wave reaches the exit condition: the clock.
__global__ __launch_bounds__(256) void comm_load_kernel(float* scratch, int n4, long long ticks) {
  extern __shared__ __attribute__((aligned(16))) float lds_hold[];
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();  // 100 MHz, constant rate
  float acc = 0.f;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x == 0) lds_hold[0] = 0.f;
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    if (i >= n4) i -= n4;
    if (i < 0 || i >= n4) i = 0;
    const f32x4 v = *reinterpret_cast<const volatile f32x4*>(reinterpret_cast<f32x4*>(scratch) + i);
    acc += v[0];
    i += gridDim.x * 256;
    __builtin_amdgcn_s_sleep(8);
  }
  if (acc == 12345.678f) scratch[0] = acc + lds_hold[0];  // keeps the loads and the LDS allocation alive
}

SACV_API_BEGIN

// Measurement aid: occupy `workgroups` workgroup slots (256 threads, `lds_bytes` of LDS each) for `us`
// microseconds on `stream`, reading the `scratch_bytes` of `scratch` round and round -- what an all-reduce's kernels do to
// the convolution kernels they run beside, on a box that has no second GPU to run RCCL against.
int sacv_comm_emulate_load(int workgroups, int lds_bytes, float us, float* scratch, long long scratch_bytes,
                           void* stream) {
  SACV_CHECK_ARG(workgroups > 0 && workgroups <= 1024 && lds_bytes >= 16 && lds_bytes <= 64 * 1024 && us > 0.f &&
                     us <= 20000.f && scratch && scratch_bytes >= 16,
                 "comm_emulate_load: bad arguments");
  const long long n4 = scratch_bytes / 16;
  hipLaunchKernelGGL(comm_load_kernel, dim3(workgroups), dim3(256), lds_bytes, (hipStream_t)stream, scratch,
                     (int)(n4 > (1 << 30) ? (1 << 30) : n4), (long long)(us * 100.0f));
  SACV_CHECK_LAUNCH("comm_load_kernel");
  return SACV_OK;
}

int sacv_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

// rank 0 creates the rendezvous id (host buffer of sacv_comm_unique_id_bytes() bytes) and hands it to the other
// ranks by any host channel (the launcher's TCP store / environment)
int sacv_comm_unique_id(void* id_out) {
  SACV_CHECK_ARG(id_out, "comm_unique_id: null pointer");
  const RcclApi* A = api();
  if (!A) {
    sacv_set_error("comm_unique_id: RCCL is unavailable: %s", g_api_error.c_str());
    return SACV_ERR_UNSUPPORTED;
  }
  ncclUniqueId id;
  SACV_CHECK_RCCL(A->GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id_out, &id, sizeof(id));
  return SACV_OK;
}

// collective over the `world` ranks; binds the communicator to the calling thread's current HIP device
int sacv_comm_init(const void* id, int rank, int world, void** comm_out) {
  SACV_CHECK_ARG(id && comm_out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad args");
  const RcclApi* A = api();
  if (!A) {
    sacv_set_error("comm_init: RCCL is unavailable: %s", g_api_error.c_str());
    return SACV_ERR_UNSUPPORTED;
  }
  SacvComm* c = new SacvComm();
  c->rank = rank;
  c->world = world;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclResult_t r = A->CommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) {
    sacv_set_error("ncclCommInitRank: RCCL error %d: %s", (int)r, A->GetErrorString(r));
    delete c;
    return SACV_ERR_LAUNCH;
  }
  hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
  if (e != hipSuccess) {
    sacv_set_error("comm_init: HIP error: %s", hipGetErrorString(e));
    A->CommDestroy(c->comm);
    delete c;
    return SACV_ERR_LAUNCH;
  }
  *comm_out = c;
  return SACV_OK;
}

int sacv_comm_destroy(void* comm) {
  if (!comm) return SACV_OK;
  SacvComm* c = (SacvComm*)comm;
  const RcclApi* A = api();
  if (c->side) (void)hipStreamSynchronize(c->side);
  if (A && c->comm) A->CommDestroy(c->comm);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->side) (void)hipStreamDestroy(c->side);
  delete c;
  return SACV_OK;
}

// buf[0..count) <- sum over ranks, in place, fp32, on the communicator's side stream.  The side stream first waits
// for everything enqueued on `producer_stream` so far (the kernels that wrote the bucket); work enqueued on the
// producer stream afterwards overlaps with the transfer.  Call sacv_comm_wait before reading the bucket.
int sacv_allreduce_bucket(void* comm, float* buf, long long count, void* producer_stream) {
  SACV_CHECK_ARG(comm && buf && count > 0, "allreduce_bucket: bad args");
  SacvComm* c = (SacvComm*)comm;
  const RcclApi* A = api();
  SACV_CHECK_ARG(A != nullptr, "allreduce_bucket: RCCL not loaded");
  SACV_CHECK_HIP(hipEventRecord(c->ready, (hipStream_t)producer_stream), "allreduce_bucket");
  SACV_CHECK_HIP(hipStreamWaitEvent(c->side, c->ready, 0), "allreduce_bucket");
  SACV_CHECK_RCCL(A->AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, c->comm, c->side), "ncclAllReduce");
  c->pending = true;
  return SACV_OK;
}

// `consumer_stream` waits for every all-reduce enqueued so far (no host synchronisation)
int sacv_comm_wait(void* comm, void* consumer_stream) {
  SACV_CHECK_ARG(comm, "comm_wait: null communicator");
  SacvComm* c = (SacvComm*)comm;
  if (!c->pending) return SACV_OK;
  SACV_CHECK_HIP(hipEventRecord(c->done, c->side), "comm_wait");
  SACV_CHECK_HIP(hipStreamWaitEvent((hipStream_t)consumer_stream, c->done, 0), "comm_wait");
  c->pending = false;
  return SACV_OK;
}

// DDP's parameter sync at construction / per-forward buffer broadcast: in order on `stream`
int sacv_comm_broadcast(void* comm, float* buf, long long count, int root, void* stream) {
  SACV_CHECK_ARG(comm && buf && count > 0, "comm_broadcast: bad args");
  SacvComm* c = (SacvComm*)comm;
  const RcclApi* A = api();
  SACV_CHECK_ARG(A != nullptr, "comm_broadcast: RCCL not loaded");
  SACV_CHECK_ARG(root >= 0 && root < c->world, "comm_broadcast: root out of range");
  SACV_CHECK_RCCL(A->Broadcast(buf, buf, (size_t)count, ncclFloat, root, c->comm, (hipStream_t)stream),
                  "ncclBroadcast");
  return SACV_OK;
}

SACV_API_END
