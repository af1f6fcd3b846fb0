// Cross-stream ordering inside one GPU, at the C ABI.
//
// The reference gets its overlap from PyTorch's stream / event wrappers (DDP's reducer, tools/utils.py:178-180).  An event
// created the default way performs a SYSTEM-scope fence when it is recorded (hip_runtime_api.h, hipEventDisableSystemFence:
// "the cost of cache writeback and invalidation, and the performance impact of those actions on the execution of following
// work") -- torch.cuda.Event and Stream.wait_stream create theirs that way.  The executor orders its two streams ~160
// times per step and nothing it orders leaves the device: these events are created with hipEventDisableTiming |
// hipEventDisableSystemFence.
#include "sacv_common.h"
#include <mutex>

namespace {
constexpr int RING = 256;  // a slot is reused only after 255 further records: its wait was enqueued long before
struct EventRing {
  hipEvent_t ev[RING];
  int next = 0;
  bool ready = false;
};
EventRing g_ring[16];  // per device
std::mutex g_mu;
}  // namespace

// ---- the same ordering with no queue-level dependency at all: a ticket in device memory.  The signalling stream runs a
// one-lane kernel that stores the ticket (release, agent scope) behind its work; the waiting stream runs a one-lane kernel
// that polls it (relaxed loads + s_sleep, ONE acquire fence at the end).  A barrier packet that depends on another queue's
// signal held the launch stream for ~105 us at the final join of the backward pass although the signal had been there for
// 200 us (scripts/diag_gpu_gaps.py); the poll returns on its first load then.  The wait kernel is always enqueued (host
// order) AFTER its signal kernel, so two streams that share a hardware queue cannot deadlock; a spin gives up after 2 s of
// the 100 MHz clock and raises a sticky device flag (sacv_stream_ticket_timeouts) instead of hanging the GPU.
struct TicketState {
  unsigned long long* flags = nullptr;  // [MAXS + 1]: per signalling stream a monotonic ticket; [MAXS] = timeouts
  void* streams[8];
  unsigned long long next[8];
  int nstreams = 0;
};
TicketState g_tickets[16];
constexpr int MAXS = 8;

__global__ void stream_signal_kernel(unsigned long long* flag, unsigned long long ticket) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void stream_wait_kernel(unsigned long long* flag, unsigned long long ticket, unsigned long long* timeouts) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ticket) {
      __builtin_amdgcn_s_sleep(16);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {  // 2 s at 100 MHz: every wave reaches an exit
        __hip_atomic_fetch_add(timeouts, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}

SACV_API_BEGIN

// sacv_stream_wait_stream through a ticket in device memory (see above).  At most 8 signalling streams per device.
int sacv_stream_wait_stream_ticket(void* waiter, void* signaller) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  SACV_CHECK_ARG(e == hipSuccess && dev >= 0 && dev < 16, "stream_wait_stream_ticket: hipGetDevice: %s", hipGetErrorString(e));
  unsigned long long* flag;
  unsigned long long ticket;
  unsigned long long* timeouts;
  // the lock covers the two launches: tickets of one signalling stream reach its queue in ticket order whatever thread asks
  std::lock_guard<std::mutex> lock(g_mu);
  {
    TicketState& t = g_tickets[dev];
    if (!t.flags) {
      e = hipMalloc(&t.flags, (MAXS + 1) * sizeof(unsigned long long));
      SACV_CHECK_ARG(e == hipSuccess, "stream_wait_stream_ticket: hipMalloc: %s", hipGetErrorString(e));
      e = hipMemset(t.flags, 0, (MAXS + 1) * sizeof(unsigned long long));
      SACV_CHECK_ARG(e == hipSuccess, "stream_wait_stream_ticket: hipMemset: %s", hipGetErrorString(e));
    }
    int i = 0;
    while (i < t.nstreams && t.streams[i] != signaller) ++i;
    if (i == t.nstreams) {
      SACV_CHECK_ARG(t.nstreams < MAXS, "stream_wait_stream_ticket: more than %d signalling streams", MAXS);
      t.streams[i] = signaller;
      t.next[i] = 0;
      ++t.nstreams;
    }
    ticket = ++t.next[i];
    flag = t.flags + i;
    timeouts = t.flags + MAXS;
  }
  hipLaunchKernelGGL(stream_signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)signaller, flag, ticket);
  SACV_CHECK_LAUNCH("stream_signal_kernel");
  hipLaunchKernelGGL(stream_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)waiter, flag, ticket, timeouts);
  SACV_CHECK_LAUNCH("stream_wait_kernel");
  return SACV_OK;
}

// Number of ticket waits that gave up (synchronises the device); 0 in a healthy process.
long long sacv_stream_ticket_timeouts(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || !g_tickets[dev].flags) return 0;
  unsigned long long v = 0;
  if (hipMemcpy(&v, g_tickets[dev].flags + MAXS, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long long)v;
}

// `waiter` (a hipStream_t) waits for everything enqueued on `signaller` so far; both on the current device.
int sacv_stream_wait_stream(void* waiter, void* signaller) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  SACV_CHECK_ARG(e == hipSuccess && dev >= 0 && dev < 16, "stream_wait_stream: hipGetDevice: %s", hipGetErrorString(e));
  hipEvent_t ev;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    EventRing& r = g_ring[dev];
    if (!r.ready) {
      for (int i = 0; i < RING; ++i) {
        e = hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming | hipEventDisableSystemFence);
        SACV_CHECK_ARG(e == hipSuccess, "stream_wait_stream: hipEventCreateWithFlags: %s", hipGetErrorString(e));
      }
      r.ready = true;
    }
    ev = r.ev[r.next];
    r.next = (r.next + 1) % RING;
  }
  e = hipEventRecord(ev, (hipStream_t)signaller);
  SACV_CHECK_ARG(e == hipSuccess, "stream_wait_stream: hipEventRecord: %s", hipGetErrorString(e));
  e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
  SACV_CHECK_ARG(e == hipSuccess, "stream_wait_stream: hipStreamWaitEvent: %s", hipGetErrorString(e));
  return SACV_OK;
}

SACV_API_END
