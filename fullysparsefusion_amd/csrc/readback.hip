// The library's count / flag read-backs: how the host learns a few words the device just produced.
// hipMemcpyAsync(device -> pageable host) + hipStreamSynchronize costs ~20 us per round trip on this platform beyond the kernels
// themselves (tools/profiling/readback_probe.hip on MI355X: 25.6 us per kernel / read-back / kernel iteration, 17.8 us into pinned memory,
// 5.5 us with no read-back at all); a frame has ~13 of them on its serial path.  Here a one-wave kernel at the end of the producing
// work posts the words and then a sequence number into a 128-byte MAILBOX in mapped pinned host memory (system-scope release), and the
// host thread spins on the sequence number (acquire): 10.3 us per iteration in the same probe.  One mailbox per (host thread, device):
// the two query branches run on two host threads.  A stream that goes idle without the post (a failed launch) ends the spin with
// FSF_ERR_HIP; FSF_READBACK_MAILBOX=0 (A/B switch, latched) or more than 120 bytes take the copy + synchronize route.
// Mailboxes outlive their threads: a thread that exits hands its mailboxes (with their sequence counters) to a process-wide free list
// and the next new thread takes them from there, so a host that starts a worker thread per frame pins one mailbox per CONCURRENT
// thread and device, not one per frame (no HIP call runs in a thread-exit destructor: the list is handed back, never freed).
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace fsf {

struct Mailbox {
  uint64_t seq;
  uint32_t v[30];
};
static_assert(sizeof(Mailbox) == 128, "mailbox = two 64-byte lines");

struct MailboxSlot {
  Mailbox* host;
  Mailbox* dev;
  uint64_t next;
  bool failed;
};
constexpr int RB_MAX_DEVICES = 16;

struct MailboxPool {
  std::mutex mu;
  std::vector<MailboxSlot> idle[RB_MAX_DEVICES];
};
static MailboxPool& rb_pool() {
  static MailboxPool* p = new MailboxPool;  // (never destroyed: thread-exit destructors may run after static destruction began)
  return *p;
}

struct ThreadSlots {
  MailboxSlot s[RB_MAX_DEVICES] = {};
  ~ThreadSlots() {
    MailboxPool& p = rb_pool();
    std::lock_guard<std::mutex> g(p.mu);
    for (int d = 0; d < RB_MAX_DEVICES; ++d)
      if (s[d].host) p.idle[d].push_back(s[d]);
  }
};
static thread_local ThreadSlots t_slots;

__global__ void __launch_bounds__(64) rb_post_kernel(Mailbox* mb, uint64_t seq, const uint32_t* __restrict__ src, int words) {
  if ((int)threadIdx.x < words) mb->v[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();  // (one wave: every lane's words are out before lane 0 publishes the sequence number)
  if (threadIdx.x == 0) __hip_atomic_store(&mb->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static bool rb_mailbox_enabled() {
  static const bool on = !(getenv("FSF_READBACK_MAILBOX") && atoi(getenv("FSF_READBACK_MAILBOX")) == 0);
  return on;
}

int fsf_read_back(void* host_dst, const void* dev_src, size_t bytes, hipStream_t stream) {
  g_host_waits.fetch_add(1, std::memory_order_relaxed);
  int dev = -1;
  MailboxSlot* s = nullptr;
  if (rb_mailbox_enabled() && bytes > 0 && bytes <= sizeof(((Mailbox*)0)->v) && (bytes % 4) == 0 && ((uintptr_t)dev_src % 4) == 0 &&
      hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < RB_MAX_DEVICES) {
    s = &t_slots.s[dev];
    if (!s->host && !s->failed) {
      MailboxPool& p = rb_pool();
      std::lock_guard<std::mutex> g(p.mu);
      if (!p.idle[dev].empty()) {  // a mailbox an exited thread left behind (its sequence counter continues)
        *s = p.idle[dev].back();
        p.idle[dev].pop_back();
      }
    }
    if (!s->host && !s->failed) {
      void* h = nullptr;
      void* d = nullptr;
      if (hipHostMalloc(&h, sizeof(Mailbox), hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
        memset(h, 0, sizeof(Mailbox));
        s->host = (Mailbox*)h;
        s->dev = (Mailbox*)d;
      } else {
        (void)hipGetLastError();
        s->failed = true;
      }
    }
    if (!s->host) s = nullptr;
  }
  if (!s) {
    FSF_HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, stream));
    FSF_HIP_TRY(hipStreamSynchronize(stream));
    return FSF_OK;
  }
  const uint64_t seq = ++s->next;
  hipLaunchKernelGGL(rb_post_kernel, dim3(1), dim3(64), 0, stream, s->dev, seq, (const uint32_t*)dev_src, (int)(bytes / 4));
  FSF_LAUNCH_CHECK();
  for (uint64_t it = 1;; ++it) {
    if (__atomic_load_n(&s->host->seq, __ATOMIC_ACQUIRE) == seq) break;
    if ((it & 0xfff) == 0) {  // every few microseconds: is the stream still busy?
      const hipError_t q = hipStreamQuery(stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(&s->host->seq, __ATOMIC_ACQUIRE) == seq) break;
        return FSF_ERR_HIP;  // the stream drained and the post never ran
      }
      if (q != hipErrorNotReady) return FSF_ERR_HIP;
    }
    __builtin_ia32_pause();
  }
  memcpy(host_dst, s->host->v, bytes);
  return FSF_OK;
}

}  // namespace fsf
