// Shared host-side plumbing of libamdzk: context, error capture, launch wrapper with optional
// per-kernel HIP-event timing, and a growable device workspace.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <chrono>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/amdzk.h"
#include "bn254.cuh"

struct ProfEntry {
  uint64_t launches = 0;
  double ms = 0.0;
};

struct PendingEvt {
  const char* name;
  hipEvent_t a, b;
};

struct TwiddleKey {
  uint32_t log_n;
  uint64_t w[4];
  bool operator<(const TwiddleKey& o) const {
    if (log_n != o.log_n) return log_n < o.log_n;
    for (int i = 0; i < 4; i++)
      if (w[i] != o.w[i]) return w[i] < o.w[i];
    return false;
  }
};

struct amdzk_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // second stream for amdzk_dev_upload_async (created on first use) and the event its fence waits on
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_evt = nullptr;
  bool copy_pending = false;  // an upload has been issued since the last amdzk_upload_fence
  std::string err;
  int num_cu = 256;

  // profiling
  bool prof = false;
  std::map<std::string, ProfEntry> prof_map;
  std::vector<PendingEvt> pending;
  std::vector<hipEvent_t> evt_pool;
  hipEvent_t t0 = nullptr, t1 = nullptr;

  // NTT twiddle tables: omega^i, i < max(1, n/2), resident per (log_n, omega)
  std::map<TwiddleKey, bn254::Fr*> twiddles;

  // workspaces (grow-only). ws[0]: NTT ping-pong; ws[1..]: MSM scratch.
  struct Ws {
    void* p = nullptr;
    size_t cap = 0;
  };
  Ws ws[8];
  // pinned host staging for small results
  void* h_pinned = nullptr;
  size_t h_pinned_cap = 0;

  // How the host waits for the device (zk_host_wait): spinning in hipStreamSynchronize — lowest latency, one busy core
  // per waiting thread — or POLLING a completion event (hipEventQuery, a few microseconds of yielding, then 50-us
  // sleeps) — the thread leaves its core to the other proofs' drivers; each wait then ends up to one sleep quantum late.
  // amdzk_set_host_wait / AMDZK_HOST_WAIT=block; lanes follow their parent. wait_forced: the host process runs the device
  // with hipDeviceScheduleBlockingSync (read in amdzk_init); this library then never calls a blocking wait of the runtime
  // — every host wait polls, whatever amdzk_set_host_wait says (profiles/r03e_host_wait_and_cu_mask.txt: under that flag
  // ten driver threads in hipStreamSynchronize did not finish a 20-proof bench in 300 s).
  bool host_wait_block = false;
  bool wait_forced = false;
  hipEvent_t wait_evt = nullptr;

  // Lanes: auxiliary contexts on the same device (own stream, workspaces, staging) for work of ONE call that is
  // independent of what the call's main stream is doing — create_proof puts the coset transforms of a phase's columns,
  // the lookup products and the random polynomial beside the commitments the transcript is waiting for. Created on
  // first use (zk_lane), ordered against each other with events only (zk_stream_after), freed with the ctx.
  static constexpr int MAX_LANES = 2;
  amdzk_ctx* lanes[MAX_LANES] = {nullptr, nullptr};
  amdzk_ctx* parent = nullptr;
  hipEvent_t order_evt[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned order_next = 0;
  // Pipelined commitments (msm.hip zk_msm_dev_xyzz): a batch's column groups alternate between `stream` and this
  // second stream, level-1 kernels chained through msm_evt[0..5]; [6] / [7] order the two streams at entry / exit.
  // A lone proof's latency against throughput (set by create_proof while a proof runs on lanes, i.e. with a default key):
  // the commitments' latency-bound stages trade instructions for depth — row / column sums of the bucket reduction as
  // shuffle trees (6 dependent additions instead of 11, 4.4 x the instructions), small batches in finer tasks. With
  // several proofs in flight (serial keys) the cheaper forms are the right ones.
  bool msm_latency_mode = false;
  bool msm_pipeline = false;
  hipStream_t msm_stream = nullptr;
  hipEvent_t msm_evt[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // recorded behind the (last) level-1 kernel of every commitment batch on this ctx: what follows on ANOTHER stream and
  // would only compete with that chip-filling kernel waits for it (zk_stream_after_l1) and then runs beside the batch's
  // latency-bound tail instead
  hipEvent_t msm_l1_evt = nullptr;
  bool msm_l1_fresh = false;  // a batch recorded msm_l1_evt and nothing has waited for it yet (zk_stream_after_l1 consumes it)
};

#define ZK_FAIL(ctx, code, ...)                         \
  do {                                                  \
    char _b[512];                                       \
    snprintf(_b, sizeof(_b), __VA_ARGS__);              \
    (ctx)->err = _b;                                    \
    return (code);                                      \
  } while (0)

#define ZK_HIP(ctx, call)                                                                  \
  do {                                                                                     \
    hipError_t _e = (call);                                                                \
    if (_e != hipSuccess) {                                                                \
      char _b[512];                                                                        \
      snprintf(_b, sizeof(_b), "%s:%d %s -> %s", __FILE__, __LINE__, #call,                \
               hipGetErrorString(_e));                                                     \
      (ctx)->err = _b;                                                                     \
      return AMDZK_E_HIP;                                                                  \
    }                                                                                      \
  } while (0)

// HIP's current device is a per-host-thread setting, while an amdzk_ctx belongs to ONE device: every extern "C"
// entry point that can touch the device pins the calling thread to ctx->device for the duration of the call and
// restores the caller's device on the way out. Without this a ctx created on one thread and used from another
// (bench.py's proof workers, a Rust rayon thread) would allocate its workspaces on device 0 while its stream
// lives on ctx->device.
struct ZkDeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit ZkDeviceGuard(const amdzk_ctx* ctx) {
    if (!ctx) return;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess && prev >= 0;
  }
  ~ZkDeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  ZkDeviceGuard(const ZkDeviceGuard&) = delete;
  ZkDeviceGuard& operator=(const ZkDeviceGuard&) = delete;
};
#define ZK_ENTER(ctx) ZkDeviceGuard _zk_device_guard(ctx)

#define ZK_TRY(expr)            \
  do {                          \
    int _r = (expr);            \
    if (_r != AMDZK_OK) return _r; \
  } while (0)

// Host waits until everything enqueued on `s` (a stream of ctx) has completed. Spin mode: hipStreamSynchronize (the
// runtime busy-waits: lowest latency, one core per waiting thread). Block mode: an event recorded behind the work is
// POLLED — a few microseconds of yielding, then 50-microsecond sleeps — so that a waiting thread leaves its core to
// the threads that have kernels to launch. (The runtime's own blocking waits are not usable here: events created with
// hipEventBlockingSync still spin in hipEventSynchronize on this ROCm, and the device-wide
// hipDeviceScheduleBlockingSync flag stalled the ten-proofs-in-flight bench outright — measured, round 3. A host that
// sets that flag itself is detected in amdzk_init and gets polling waits throughout: amdzk_ctx::wait_forced.)
inline hipError_t zk_event_poll(hipEvent_t evt) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    hipError_t e = hipEventQuery(evt);
    if (e != hipErrorNotReady) return e;
    if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(20)) {
      std::this_thread::yield();
    } else {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
}
inline bool zk_waits_poll(const amdzk_ctx* ctx) {
  const amdzk_ctx* root = ctx->parent ? ctx->parent : ctx;
  return root->host_wait_block || root->wait_forced;
}
inline hipError_t zk_host_wait(amdzk_ctx* ctx, hipStream_t s) {
  if (!zk_waits_poll(ctx)) return hipStreamSynchronize(s);
  hipError_t e = hipSuccess;
  if (!ctx->wait_evt) e = hipEventCreateWithFlags(&ctx->wait_evt, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(ctx->wait_evt, s);
  if (e != hipSuccess) return e;
  return zk_event_poll(ctx->wait_evt);
}
// Host waits for an event already recorded (the timer's stop event).
inline hipError_t zk_host_wait_event(amdzk_ctx* ctx, hipEvent_t evt) { return zk_waits_poll(ctx) ? zk_event_poll(evt) : hipEventSynchronize(evt); }

int zk_ws_reserve(amdzk_ctx* ctx, int slot, size_t bytes, void** out);
int zk_pinned_reserve(amdzk_ctx* ctx, size_t bytes, void** out);
int zk_ptr_on_device(amdzk_ctx* ctx, const void* p, const char* what);
hipEvent_t zk_evt_get(amdzk_ctx* ctx);
// Lane i of ctx (created on first use). While per-kernel profiling is on, a lane IS the ctx: one stream, so that the
// event-bracketed kernel times are those of kernels running alone.
int zk_lane(amdzk_ctx* ctx, int i, amdzk_ctx** out);
hipError_t zk_stream_create(hipStream_t* s, bool low_priority);
// Everything enqueued on `waiter`'s stream after this call runs after everything enqueued on `signaler`'s stream
// before it (event record + stream wait; no host synchronisation). No-op when both are the same context.
int zk_stream_after(amdzk_ctx* waiter, amdzk_ctx* signaler);
// Everything enqueued on `waiter` after this call runs after the level-1 kernel of the last commitment batch launched on
// `signaler` — or, when no batch has recorded that event since the last call (no columns, nothing launched), after
// everything enqueued on `signaler` (zk_stream_after): never a stale event, never no ordering at all.
int zk_stream_after_l1(amdzk_ctx* waiter, amdzk_ctx* signaler);
// Host waits for the ctx's stream and all its lanes.
int zk_sync_all(amdzk_ctx* ctx);
void zk_prof_drain(amdzk_ctx* ctx);

// Launch wrapper: kernel<<<grid, block, shmem, ctx->stream>>>(args...), optionally event-bracketed.
#define ZK_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                        \
  do {                                                                               \
    hipEvent_t _ea = nullptr, _eb = nullptr;                                         \
    if ((ctx)->prof) {                                                               \
      _ea = zk_evt_get(ctx);                                                         \
      _eb = zk_evt_get(ctx);                                                         \
      (void)hipEventRecord(_ea, (ctx)->stream);                                          \
    }                                                                                \
    hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);      \
    if ((ctx)->prof) {                                                               \
      (void)hipEventRecord(_eb, (ctx)->stream);                                          \
      (ctx)->pending.push_back(PendingEvt{name, _ea, _eb});                          \
    }                                                                                \
    ZK_HIP(ctx, hipGetLastError());                                                  \
  } while (0)

// Optional per-element multiplier tables of zk_ntt_ex (ntt.hip): entries are constants in radix 2^261 (packed
// canonical), one per element of a column. nz > 1 runs nz transforms of every input column, transform z with the
// tables at + z * tab_z_stride and its output at + z * out_z_stride (the cosets of the quotient domain, poly.hip).
struct NttTables {
  const bn254::Fr* in_tab = nullptr;   // input element i *= in_tab[i]
  const bn254::Fr* out_tab = nullptr;  // output element j *= out_tab[j]
  size_t tab_col_stride = 0;           // elements the tables advance per column
  size_t tab_z_stride = 0;             // ... and per z
  uint32_t nz = 1;
  size_t out_z_stride = 0;
};

// entry points implemented in the kernel translation units
int zk_ntt_dev(amdzk_ctx* ctx, bn254::Fr* d_a, uint32_t log_n, const uint64_t omega[4],
               uint32_t flags, size_t ncols, size_t col_stride);
