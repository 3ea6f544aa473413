// C ABI of libamdzk (include/amdzk.h): context, device memory, and the host-pointer forms of the
// hot-path entry points. Kernels live in ntt.hip / msm.hip. There is no CPU fallback here: every
// entry point either runs on the gfx950 device or fails with a status code.
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "common.hpp"
#include "build_stamp.h"  // AMDZK_SRC_HASH: written by the Makefile from the kernel sources (tools/src_hash.py)

using namespace bn254;

struct amdzk_srs;
int zk_srs_upload(amdzk_ctx* ctx, const uint64_t* g, const uint64_t* g_lagrange, uint32_t k, amdzk_srs** out);
void zk_srs_free(amdzk_ctx*, amdzk_srs* s);
int zk_msm_dev_xyzz(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, const Fr* d_scalars, size_t ncols,
                    size_t len, size_t col_stride, G1X** d_out);
int zk_msm_finish(amdzk_ctx* ctx, const G1X* d_res, size_t ncols, uint64_t* out_jac);
int zk_srs_setup(amdzk_ctx* ctx, uint32_t k, const uint64_t s_mont[4], const uint64_t omega_mont[4], amdzk_srs** out, uint64_t* g_out,
                 uint64_t* g_lagrange_out);
size_t zk_srs_serialized_size(uint32_t k);
int zk_g_to_lagrange(amdzk_ctx* ctx, const uint64_t* g, uint32_t k, const uint64_t omega_inv[4], const uint64_t n_inv[4], uint64_t* out);
int zk_srs_downsize(amdzk_ctx* ctx, const amdzk_srs* srs, uint32_t new_k, const uint64_t omega_inv[4], const uint64_t n_inv[4], amdzk_srs** out);
int zk_srs_get(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, uint64_t* out);
int zk_srs_write(amdzk_ctx* ctx, const amdzk_srs* s, const uint8_t g2[64], const uint8_t s_g2[64], uint8_t* out, size_t cap);
int zk_srs_read(amdzk_ctx* ctx, const uint8_t* data, size_t len, amdzk_srs** out, uint8_t g2_out[64], uint8_t s_g2_out[64]);
extern "C" int amdzk_domain_new(amdzk_ctx* ctx, uint32_t j, uint32_t k, struct amdzk_domain** out);
extern "C" void amdzk_domain_free(amdzk_ctx* ctx, struct amdzk_domain* d);
extern "C" int amdzk_domain_constant(const struct amdzk_domain* d, int what, uint64_t out[4]);

static thread_local std::string g_init_err;

int zk_ws_reserve(amdzk_ctx* ctx, int slot, size_t bytes, void** out) {
  amdzk_ctx::Ws& w = ctx->ws[slot];
  if (w.cap < bytes) {
    if (w.p) {
      ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
      ZK_HIP(ctx, hipFree(w.p));
      w.p = nullptr;
      w.cap = 0;
    }
    size_t want = bytes + bytes / 8;
    hipError_t e = hipMalloc(&w.p, want);
    if (e != hipSuccess) {
      want = bytes;
      e = hipMalloc(&w.p, want);
    }
    if (e != hipSuccess) ZK_FAIL(ctx, AMDZK_E_NOMEM, "workspace %d: hipMalloc(%zu) failed: %s", slot, bytes, hipGetErrorString(e));
    w.cap = want;
  }
  *out = w.p;
  return AMDZK_OK;
}

int zk_pinned_reserve(amdzk_ctx* ctx, size_t bytes, void** out) {
  if (ctx->h_pinned_cap < bytes) {
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    ctx->h_pinned = nullptr;
    ctx->h_pinned_cap = 0;
    size_t want = bytes < 65536 ? 65536 : bytes;
    ZK_HIP(ctx, hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault));
    ctx->h_pinned_cap = want;
  }
  *out = ctx->h_pinned;
  return AMDZK_OK;
}

hipEvent_t zk_evt_get(amdzk_ctx* ctx) {
  if (!ctx->evt_pool.empty()) {
    hipEvent_t e = ctx->evt_pool.back();
    ctx->evt_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

void zk_prof_drain(amdzk_ctx* ctx) {
  if (ctx->pending.empty()) return;
  zk_host_wait(ctx, ctx->stream);
  for (auto& p : ctx->pending) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, p.a, p.b);
    ProfEntry& e = ctx->prof_map[p.name];
    e.launches++;
    e.ms += ms;
    ctx->evt_pool.push_back(p.a);
    ctx->evt_pool.push_back(p.b);
  }
  ctx->pending.clear();
}

// Every device allocation this ctx owns (workspaces, twiddle tables) must live on ctx->device and its pinned
// staging must be host memory: the check behind bench.py's per-rank assertion and tests/test_gpu_affinity.py.
int zk_ptr_on_device(amdzk_ctx* ctx, const void* p, const char* what) {
  if (!p) return AMDZK_OK;
  hipPointerAttribute_t at;
  hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) ZK_FAIL(ctx, AMDZK_E_HIP, "affinity: hipPointerGetAttributes(%s) -> %s", what, hipGetErrorString(e));
  if (at.type != hipMemoryTypeDevice || at.device != ctx->device)
    ZK_FAIL(ctx, AMDZK_E_INVALID, "affinity: %s lives on device %d (memory type %d), ctx is on device %d", what, at.device, (int)at.type,
            ctx->device);
  return AMDZK_OK;
}

// Every stream of the library is a plain non-blocking stream. (Measured and rejected: hipStreamCreateWithPriority —
// the caller's stream at the device's highest priority, the lanes at its lowest, so that the commitment chain's
// latency-bound kernels would not queue behind a chip-filling transform — made EVERYTHING slower on this runtime:
// 56 ms instead of 21 for one proof, 57 instead of 72 proofs/s with ten in flight, 39 ms even for a serial-mode key;
// profiles/r03b_stream_priorities_and_gating.txt.)
// Also measured and rejected: keeping the lanes' streams off every 2nd / 4th / 8th compute unit (a CU mask,
// hipExtStreamCreateWithCUMask) so that the caller's stream always finds free compute units: 19.8-19.95 ms per proof
// against 19.35 without (profiles/r03e_host_wait_and_cu_mask.txt).
void zk_note_streams(int delta);
hipError_t zk_stream_create(hipStream_t* s, bool /*lane*/) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }

int zk_lane(amdzk_ctx* ctx, int i, amdzk_ctx** out) {
  if (i < 0 || i >= amdzk_ctx::MAX_LANES) ZK_FAIL(ctx, AMDZK_E_INVALID, "lane %d out of range", i);
  if (ctx->prof || ctx->parent) {  // profiling: one stream; a lane has no lanes of its own
    *out = ctx;
    return AMDZK_OK;
  }
  if (!ctx->lanes[i]) {
    amdzk_ctx* l = new amdzk_ctx();
    l->device = ctx->device;
    l->num_cu = ctx->num_cu;
    l->parent = ctx;
    if (zk_stream_create(&l->own_stream, true) != hipSuccess) {
      delete l;
      ZK_FAIL(ctx, AMDZK_E_HIP, "lane %d: hipStreamCreate failed", i);
    }
    l->stream = l->own_stream;
    ctx->lanes[i] = l;
    zk_note_streams(1);
  }
  *out = ctx->lanes[i];
  return AMDZK_OK;
}

// Failures of the ordering calls are reported on the ROOT context (the one the caller holds and asks amdzk_last_error
// about), whichever lane waits.
int zk_stream_after(amdzk_ctx* waiter, amdzk_ctx* signaler) {
  if (waiter == signaler || waiter->stream == signaler->stream) return AMDZK_OK;
  amdzk_ctx* rep = waiter->parent ? waiter->parent : waiter;
  hipEvent_t& e = signaler->order_evt[signaler->order_next++ % 8];
  if (!e) ZK_HIP(rep, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  ZK_HIP(rep, hipEventRecord(e, signaler->stream));
  ZK_HIP(rep, hipStreamWaitEvent(waiter->stream, e, 0));
  return AMDZK_OK;
}

int zk_stream_after_l1(amdzk_ctx* waiter, amdzk_ctx* signaler) {
  if (waiter == signaler || waiter->stream == signaler->stream) return AMDZK_OK;
  if (!signaler->msm_l1_evt || !signaler->msm_l1_fresh) return zk_stream_after(waiter, signaler);
  amdzk_ctx* rep = waiter->parent ? waiter->parent : waiter;
  signaler->msm_l1_fresh = false;
  ZK_HIP(rep, hipStreamWaitEvent(waiter->stream, signaler->msm_l1_evt, 0));
  return AMDZK_OK;
}

int zk_sync_all(amdzk_ctx* ctx) {
  amdzk_ctx* root = ctx->parent ? ctx->parent : ctx;
  ZK_HIP(ctx, zk_host_wait(root, root->stream));
  if (root->msm_stream) ZK_HIP(ctx, zk_host_wait(root, root->msm_stream));
  for (amdzk_ctx* l : root->lanes)
    if (l) {
      ZK_HIP(ctx, zk_host_wait(l, l->stream));
      if (l->msm_stream) ZK_HIP(ctx, zk_host_wait(l, l->msm_stream));
    }
  return AMDZK_OK;
}

static void ctx_release(amdzk_ctx* ctx) {
  zk_host_wait(ctx, ctx->stream);
  zk_prof_drain(ctx);
  for (auto& kv : ctx->twiddles) hipFree(kv.second);
  for (auto& w : ctx->ws)
    if (w.p) hipFree(w.p);
  if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
  for (auto e : ctx->evt_pool) hipEventDestroy(e);
  for (auto e : ctx->order_evt)
    if (e) hipEventDestroy(e);
  for (auto e : ctx->msm_evt)
    if (e) hipEventDestroy(e);
  if (ctx->msm_l1_evt) hipEventDestroy(ctx->msm_l1_evt);
  if (ctx->msm_stream) {
    zk_host_wait(ctx, ctx->msm_stream);
    hipStreamDestroy(ctx->msm_stream);
  }
  if (ctx->t0) hipEventDestroy(ctx->t0);
  if (ctx->t1) hipEventDestroy(ctx->t1);
  if (ctx->copy_stream) {
    zk_host_wait(ctx, ctx->copy_stream);
    hipStreamDestroy(ctx->copy_stream);
  }
  if (ctx->copy_evt) hipEventDestroy(ctx->copy_evt);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  if (ctx->wait_evt) hipEventDestroy(ctx->wait_evt);
  delete ctx;
}

// One context = one HIP stream, and a host that keeps several proofs in flight uses several contexts. The HIP runtime
// maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues, 4 unless the environment says otherwise, and kernels
// of streams that share a queue wait for each other: 8 proofs in flight on 4 queues measure no better than 4
// (profiles/r02j_hw_queues_sweep.txt). The variable is read when the runtime initialises, so it is set — never
// overridden — when this library is loaded; a host that initialised HIP earlier sets it itself.
namespace {
struct HwQueuesDefault {
  HwQueuesDefault() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }
} hw_queues_default;
}  // namespace

// Streams this library keeps busy in this process — one per context, two more per context whose proofs run on lanes
// (default keys; AMDZK_KEYGEN_SERIAL keys stay on the context's stream) — against the hardware queues the HIP runtime
// was (most likely) initialised with: more busy streams than queues queue up behind each other silently (6-9 % fewer
// proofs per second with 8-12 proofs in flight on the default 4), e.g. when the host initialised HIP before this library
// could set the variable. Said once, on stderr.
static std::atomic<int> g_live_streams{0};
static std::atomic<bool> g_queue_note_given{false};
void zk_note_streams(int delta) {
  const int live = (g_live_streams += delta);
  if (delta <= 0) return;
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  const int queues = e && atoi(e) > 0 ? atoi(e) : 4;
  if (live > queues && !g_queue_note_given.exchange(true))
    fprintf(stderr,
            "[amdzk] note: %d streams of this library in this process (one per context, two more per context that proves on lanes) but "
            "GPU_MAX_HW_QUEUES=%d hardware queues: kernels of different streams will wait for each other. Export GPU_MAX_HW_QUEUES >= the "
            "number of proofs in flight before the process's first HIP call, and with 4 or more proofs in flight make the keys with "
            "AMDZK_KEYGEN_SERIAL (one stream per proof; the proofs fill the chip between them).\n",
            live, queues);
}
static std::atomic<bool> g_blocking_note_given{false};

extern "C" {

int amdzk_version(void) { return 1001; }

// "amdzk <abi> src=<hash of the comment-stripped kernel sources and the Makefile> arch=gfx950": what this binary was built
// from. bench.py refuses a library whose stamp is not its tree's bench.kernel_src_hash().
const char* amdzk_build_info(void) { return "amdzk 1001 src=" AMDZK_SRC_HASH " arch=gfx950"; }

int amdzk_init(int device_id, amdzk_ctx** out) {
  if (!out) return AMDZK_E_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return AMDZK_E_NO_DEVICE;
  amdzk_ctx probe;  // only `device` is read: pins this thread to device_id until return, then restores the caller's device
  probe.device = device_id;
  ZK_ENTER(&probe);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device_id) return AMDZK_E_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return AMDZK_E_NO_DEVICE;
  // This library carries gfx950 code objects only.
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return AMDZK_E_NO_DEVICE;
  amdzk_ctx* c = new amdzk_ctx();
  c->device = device_id;
  c->num_cu = prop.multiProcessorCount;
  if (zk_stream_create(&c->own_stream, false) != hipSuccess) {
    delete c;
    return AMDZK_E_HIP;
  }
  c->stream = c->own_stream;
  if (const char* e = getenv("AMDZK_HOST_WAIT")) c->host_wait_block = strcmp(e, "block") == 0;
  // A host that runs this device with hipDeviceScheduleBlockingSync: the runtime's own waits (hipStreamSynchronize from
  // one driver thread per proof in flight, on streams that share hardware queues) are the path that did not come back in
  // round 3's experiment, so this library then polls on every host wait and never enters them.
  unsigned dflags = 0;
  // (hip_runtime_api.h: "on ROCm, hipDeviceScheduleBlockingSync is a synonym for hipDeviceScheduleYield" — both select the
  // runtime's non-spinning wait, so both are treated alike.)
  if (hipGetDeviceFlags(&dflags) == hipSuccess &&
      ((dflags & hipDeviceScheduleMask) == hipDeviceScheduleBlockingSync || (dflags & hipDeviceScheduleMask) == hipDeviceScheduleYield)) {
    c->wait_forced = true;
    if (!g_blocking_note_given.exchange(true))
      fprintf(stderr, "[amdzk] note: the device runs with hipDeviceScheduleBlockingSync / Yield: every host wait of this library polls a "
                      "completion event (50-us sleeps) instead of calling the runtime's blocking waits.\n");
  }
  hipEventCreate(&c->t0);
  hipEventCreate(&c->t1);
  zk_note_streams(1);
  *out = c;
  return AMDZK_OK;
}

void amdzk_destroy(amdzk_ctx* ctx) {
  ZK_ENTER(ctx);
  if (!ctx) return;
  for (amdzk_ctx*& l : ctx->lanes)
    if (l) {
      ctx_release(l);
      zk_note_streams(-1);
      l = nullptr;
    }
  ctx_release(ctx);
  zk_note_streams(-1);
}

const char* amdzk_last_error(const amdzk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int amdzk_set_stream(amdzk_ctx* ctx, void* hip_stream) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return AMDZK_OK;
}

int amdzk_set_host_wait(amdzk_ctx* ctx, int mode) {
  if (!ctx) return AMDZK_E_INVALID;
  if (mode != AMDZK_WAIT_SPIN && mode != AMDZK_WAIT_BLOCK) ZK_FAIL(ctx, AMDZK_E_INVALID, "set_host_wait: unknown mode %d", mode);
  ctx->host_wait_block = mode == AMDZK_WAIT_BLOCK;  // (with wait_forced the waits poll either way)
  return AMDZK_OK;
}

int amdzk_sync(amdzk_ctx* ctx) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  return zk_sync_all(ctx);
}

int amdzk_dev_alloc(amdzk_ctx* ctx, size_t bytes, void** dptr) {
  ZK_ENTER(ctx);
  if (!ctx || !dptr) return AMDZK_E_INVALID;
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
  if (e != hipSuccess) ZK_FAIL(ctx, AMDZK_E_NOMEM, "dev_alloc(%zu): %s", bytes, hipGetErrorString(e));
  return AMDZK_OK;
}
int amdzk_dev_free(amdzk_ctx* ctx, void* dptr) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!dptr) return AMDZK_OK;
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  ZK_HIP(ctx, hipFree(dptr));
  return AMDZK_OK;
}
int amdzk_dev_upload(amdzk_ctx* ctx, void* dptr, const void* host, size_t bytes) {
  ZK_ENTER(ctx);
  if (!ctx || (!dptr && bytes) || (!host && bytes)) return AMDZK_E_INVALID;
  ZK_HIP(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return AMDZK_OK;
}
int amdzk_dev_download(amdzk_ctx* ctx, void* host, const void* dptr, size_t bytes) {
  ZK_ENTER(ctx);
  if (!ctx || (!dptr && bytes) || (!host && bytes)) return AMDZK_E_INVALID;
  ZK_HIP(ctx, hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return AMDZK_OK;
}
// ---- streaming inputs: pinned host memory + uploads on a copy stream of the ctx's own
int amdzk_host_alloc(amdzk_ctx* ctx, size_t bytes, void** hptr) {
  ZK_ENTER(ctx);
  if (!ctx || !hptr) return AMDZK_E_INVALID;
  hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) ZK_FAIL(ctx, AMDZK_E_NOMEM, "host_alloc(%zu): %s", bytes, hipGetErrorString(e));
  return AMDZK_OK;
}
int amdzk_host_free(amdzk_ctx* ctx, void* hptr) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!hptr) return AMDZK_OK;
  if (ctx->copy_stream) ZK_HIP(ctx, zk_host_wait(ctx, ctx->copy_stream));
  ZK_HIP(ctx, hipHostFree(hptr));
  return AMDZK_OK;
}
int amdzk_dev_upload_async(amdzk_ctx* ctx, void* dptr, const void* host, size_t bytes) {
  ZK_ENTER(ctx);
  if (!ctx || (!dptr && bytes) || (!host && bytes)) return AMDZK_E_INVALID;
  if (!ctx->copy_stream) {
    ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->copy_evt, hipEventDisableTiming));
  }
  if (bytes) {
    ZK_HIP(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    // the completion marker travels with the copy (not with the fence): by the time a pipelined caller fences, the copy
    // queue has long gone idle, and a marker sent to an idle queue completes only when that queue is next serviced
    ZK_HIP(ctx, hipEventRecord(ctx->copy_evt, ctx->copy_stream));
    ctx->copy_pending = true;
  }
  return AMDZK_OK;
}
// Order the context's stream behind its uploads. A caller that overlaps the next witness's upload with the current proof
// (feeder.py) fences a copy that finished long ago: the host sees that with one event query and the proof's stream gets NO
// cross-queue dependency. (With one in front of every proof, eight proofs that run in step all waited on such a marker at the
// same moment and some timed regions ran 5-10 % low with the chip partly idle: profiles/r04y_streamed_regions_in_step.txt.)
// Only a copy still in flight is waited for, on the device.
int amdzk_upload_fence(amdzk_ctx* ctx) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!ctx->copy_stream || !ctx->copy_pending) return AMDZK_OK;  // nothing uploaded asynchronously since the last fence
  hipError_t q = hipEventQuery(ctx->copy_evt);
  if (q == hipErrorNotReady) {
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->copy_evt, 0));
  } else {
    ZK_HIP(ctx, q);
  }
  ctx->copy_pending = false;
  return AMDZK_OK;
}

int amdzk_dev_memset(amdzk_ctx* ctx, void* dptr, int byte, size_t bytes) {
  ZK_ENTER(ctx);
  if (!ctx || (!dptr && bytes)) return AMDZK_E_INVALID;
  ZK_HIP(ctx, hipMemsetAsync(dptr, byte, bytes, ctx->stream));
  return AMDZK_OK;
}

int amdzk_srs_upload(amdzk_ctx* ctx, const uint64_t* g, const uint64_t* g_lagrange, uint32_t k, amdzk_srs** out) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  return zk_srs_upload(ctx, g, g_lagrange, k, out);
}
int amdzk_srs_setup(amdzk_ctx* ctx, uint32_t k, const uint64_t s[4], amdzk_srs** out, uint64_t* g_out, uint64_t* g_lagrange_out) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!s || !out) ZK_FAIL(ctx, AMDZK_E_INVALID, "srs_setup: null argument");
  amdzk_domain* dom = nullptr;
  ZK_TRY(amdzk_domain_new(ctx, 3, k, &dom));
  uint64_t omega[4];
  amdzk_domain_constant(dom, 0, omega);
  amdzk_domain_free(ctx, dom);
  return zk_srs_setup(ctx, k, s, omega, out, g_out, g_lagrange_out);
}
static int fft_constants(amdzk_ctx* ctx, uint32_t k, uint64_t omega_inv[4], uint64_t n_inv[4]) {
  amdzk_domain* dom = nullptr;
  ZK_TRY(amdzk_domain_new(ctx, 3, k, &dom));
  amdzk_domain_constant(dom, 1, omega_inv);
  amdzk_domain_constant(dom, 6, n_inv);
  amdzk_domain_free(ctx, dom);
  return AMDZK_OK;
}
int amdzk_g_to_lagrange(amdzk_ctx* ctx, const uint64_t* g, uint32_t k, uint64_t* g_lagrange_out) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  uint64_t wi[4], ni[4];
  ZK_TRY(fft_constants(ctx, k, wi, ni));
  return zk_g_to_lagrange(ctx, g, k, wi, ni, g_lagrange_out);
}
int amdzk_srs_downsize(amdzk_ctx* ctx, const amdzk_srs* srs, uint32_t new_k, amdzk_srs** out) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  uint64_t wi[4], ni[4];
  ZK_TRY(fft_constants(ctx, new_k, wi, ni));
  return zk_srs_downsize(ctx, srs, new_k, wi, ni, out);
}
int amdzk_srs_get(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, uint64_t* out) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  return zk_srs_get(ctx, srs, basis, out);
}
size_t amdzk_srs_serialized_size(uint32_t k) { return zk_srs_serialized_size(k); }
int amdzk_srs_write(amdzk_ctx* ctx, const amdzk_srs* srs, const uint8_t g2[64], const uint8_t s_g2[64], uint8_t* out, size_t cap) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  return zk_srs_write(ctx, srs, g2, s_g2, out, cap);
}
int amdzk_srs_read(amdzk_ctx* ctx, const uint8_t* data, size_t len, amdzk_srs** out, uint8_t g2_out[64], uint8_t s_g2_out[64]) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  return zk_srs_read(ctx, data, len, out, g2_out, s_g2_out);
}
void amdzk_srs_free(amdzk_ctx* ctx, amdzk_srs* srs) {
  ZK_ENTER(ctx);
  if (ctx) zk_host_wait(ctx, ctx->stream);
  zk_srs_free(ctx, srs);
}

int amdzk_msm_g1_dev(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, const void* d_scalars, size_t ncols,
                     size_t len, size_t col_stride, uint64_t* out_jacobian) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!d_scalars || !out_jacobian) ZK_FAIL(ctx, AMDZK_E_INVALID, "msm: null pointer");
  G1X* d_res = nullptr;
  ZK_TRY(zk_msm_dev_xyzz(ctx, srs, basis, (const Fr*)d_scalars, ncols, len, col_stride, &d_res));
  return zk_msm_finish(ctx, d_res, ncols, out_jacobian);
}

int amdzk_msm_g1_batch(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, const uint64_t* const* scalars,
                       size_t ncols, size_t len, uint64_t* out_jacobian) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!scalars || !out_jacobian || ncols == 0) ZK_FAIL(ctx, AMDZK_E_INVALID, "msm_batch: null pointer or ncols == 0");
  Fr* d = nullptr;
  size_t stride = len ? len : 1;
  ZK_TRY(zk_ws_reserve(ctx, 2, ncols * stride * sizeof(Fr), (void**)&d));
  for (size_t c = 0; c < ncols; c++) {
    if (!scalars[c] && len) ZK_FAIL(ctx, AMDZK_E_INVALID, "msm_batch: column %zu is null", c);
    ZK_HIP(ctx, hipMemcpyAsync(d + c * stride, scalars[c], len * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
  }
  return amdzk_msm_g1_dev(ctx, srs, basis, d, ncols, len, stride, out_jacobian);
}

int amdzk_msm_g1(amdzk_ctx* ctx, const amdzk_srs* srs, int basis, const uint64_t* scalars, size_t len,
                 uint64_t out_jacobian[12]) {
  ZK_ENTER(ctx);
  const uint64_t* cols[1] = {scalars};
  return amdzk_msm_g1_batch(ctx, srs, basis, cols, 1, len, out_jacobian);
}

int amdzk_ntt_fr_dev(amdzk_ctx* ctx, void* d_a, uint32_t log_n, const uint64_t omega[4], uint32_t flags,
                     size_t ncols, size_t col_stride) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!d_a || !omega) ZK_FAIL(ctx, AMDZK_E_INVALID, "ntt: null pointer");
  return zk_ntt_dev(ctx, (Fr*)d_a, log_n, omega, flags, ncols, col_stride);
}

int amdzk_ntt_fr(amdzk_ctx* ctx, uint64_t* a, uint32_t log_n, const uint64_t omega[4], uint32_t flags) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!a || !omega) ZK_FAIL(ctx, AMDZK_E_INVALID, "ntt: null pointer");
  if (log_n > 27) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "ntt: log_n %u > 27", log_n);
  size_t bytes = ((size_t)1 << log_n) * sizeof(Fr);
  Fr* d = nullptr;
  ZK_TRY(zk_ws_reserve(ctx, 2, bytes, (void**)&d));
  ZK_HIP(ctx, hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, ctx->stream));
  ZK_TRY(zk_ntt_dev(ctx, d, log_n, omega, flags, 1, (size_t)1 << log_n));
  ZK_HIP(ctx, hipMemcpyAsync(a, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return AMDZK_OK;
}

int amdzk_ctx_check_affinity(amdzk_ctx* ctx) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  int dev = -1;
  ZK_HIP(ctx, hipStreamGetDevice(ctx->stream, &dev));
  if (dev != ctx->device) ZK_FAIL(ctx, AMDZK_E_INVALID, "affinity: stream is on device %d, ctx on %d", dev, ctx->device);
  for (int i = 0; i < 8; i++) ZK_TRY(zk_ptr_on_device(ctx, ctx->ws[i].p, "workspace"));
  for (auto& kv : ctx->twiddles) ZK_TRY(zk_ptr_on_device(ctx, kv.second, "twiddle table"));
  for (amdzk_ctx* l : ctx->lanes)
    if (l) {
      ZK_HIP(ctx, hipStreamGetDevice(l->stream, &dev));
      if (dev != ctx->device) ZK_FAIL(ctx, AMDZK_E_INVALID, "affinity: a lane's stream is on device %d, ctx on %d", dev, ctx->device);
      for (int i = 0; i < 8; i++) ZK_TRY(zk_ptr_on_device(ctx, l->ws[i].p, "lane workspace"));
      for (auto& kv : l->twiddles) ZK_TRY(zk_ptr_on_device(ctx, kv.second, "lane twiddle table"));
    }
  return AMDZK_OK;
}
int amdzk_ptr_check_affinity(amdzk_ctx* ctx, const void* dptr) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  return zk_ptr_on_device(ctx, dptr, "pointer");
}
int amdzk_ctx_device(const amdzk_ctx* ctx) { return ctx ? ctx->device : -1; }

// best_fft on ncols host vectors of 2^log_n in one submission (one transfer each way per column).
int amdzk_ntt_fr_batch(amdzk_ctx* ctx, uint64_t* const* cols, size_t ncols, uint32_t log_n, const uint64_t omega[4], uint32_t flags) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  if (!cols || !omega || ncols == 0) ZK_FAIL(ctx, AMDZK_E_INVALID, "ntt_batch: null pointer or ncols == 0");
  if (log_n > 27) ZK_FAIL(ctx, AMDZK_E_UNSUPPORTED, "ntt: log_n %u > 27", log_n);
  const size_t n = (size_t)1 << log_n;
  Fr* d = nullptr;
  ZK_TRY(zk_ws_reserve(ctx, 2, ncols * n * sizeof(Fr), (void**)&d));
  for (size_t c = 0; c < ncols; c++) {
    if (!cols[c]) ZK_FAIL(ctx, AMDZK_E_INVALID, "ntt_batch: column %zu is null", c);
    ZK_HIP(ctx, hipMemcpyAsync(d + c * n, cols[c], n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
  }
  ZK_TRY(zk_ntt_dev(ctx, d, log_n, omega, flags, ncols, n));
  for (size_t c = 0; c < ncols; c++) ZK_HIP(ctx, hipMemcpyAsync(cols[c], d + c * n, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_host_wait(ctx, ctx->stream));
  return AMDZK_OK;
}

int amdzk_timer_start(amdzk_ctx* ctx) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  ZK_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
  return AMDZK_OK;
}
int amdzk_timer_stop(amdzk_ctx* ctx, float* ms) {
  ZK_ENTER(ctx);
  if (!ctx || !ms) return AMDZK_E_INVALID;
  ZK_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
  ZK_HIP(ctx, zk_host_wait_event(ctx, ctx->t1));
  ZK_HIP(ctx, hipEventElapsedTime(ms, ctx->t0, ctx->t1));
  return AMDZK_OK;
}
int amdzk_prof_enable(amdzk_ctx* ctx, int on) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  zk_prof_drain(ctx);
  ctx->prof = on != 0;
  return AMDZK_OK;
}
int amdzk_prof_reset(amdzk_ctx* ctx) {
  ZK_ENTER(ctx);
  if (!ctx) return AMDZK_E_INVALID;
  zk_prof_drain(ctx);
  ctx->prof_map.clear();
  return AMDZK_OK;
}
int amdzk_prof_get(amdzk_ctx* ctx, const char* kernel_name, uint64_t* launches, double* total_ms) {
  ZK_ENTER(ctx);
  if (!ctx || !kernel_name) return AMDZK_E_INVALID;
  zk_prof_drain(ctx);
  auto it = ctx->prof_map.find(kernel_name);
  if (launches) *launches = it == ctx->prof_map.end() ? 0 : it->second.launches;
  if (total_ms) *total_ms = it == ctx->prof_map.end() ? 0.0 : it->second.ms;
  return AMDZK_OK;
}
size_t amdzk_prof_dump(amdzk_ctx* ctx, char* buf, size_t cap) {
  ZK_ENTER(ctx);
  if (!ctx) return 0;
  zk_prof_drain(ctx);
  std::string s;
  char line[256];
  for (auto& kv : ctx->prof_map) {
    snprintf(line, sizeof(line), "%s %llu %.6f\n", kv.first.c_str(), (unsigned long long)kv.second.launches, kv.second.ms);
    s += line;
  }
  if (buf && cap) {
    size_t m = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(buf, s.data(), m);
    buf[m] = 0;
  }
  return s.size() + 1;
}

}  // extern "C"
