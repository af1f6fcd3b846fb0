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

SACV_API_BEGIN

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
