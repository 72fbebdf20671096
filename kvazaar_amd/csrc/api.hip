// api.hip -- library contexts (one per device, all reachable from one process), device-memory helpers and
// HIP-event timing of the C ABI declared in include/kvz_hip.h.
//
// Device model (the reference is ONE process whose strategy pointers are process-global,
// strategies/strategies-picture.c:33-64, and whose workers are pthreads, threadqueue.c:263): a context per device;
// every host thread has a current device (kvz_hip_init / kvz_hip_set_device bind it, exactly like hipSetDevice);
// a thread that never chose one uses the process default = the first device initialised.  Every entry runs on the
// calling thread's current device: its default stream, its CU count, its staging buffers.
#include "kvz_hip_internal.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>

namespace kvzhip {

constexpr int KVZ_MAX_DEVICES = 64;
struct dev_ctx {
  std::atomic<bool> ready{false};
  int num_cus = 256;
  hipStream_t stream = nullptr;      // the device's default stream of this library (what a NULL kvz_hip_stream means)
  char name[256] = "";
};
static std::mutex g_mu;
static dev_ctx g_ctx[KVZ_MAX_DEVICES];
static std::atomic<int> g_default_dev{-1};    // first device initialised: the current device of threads that never chose one
static thread_local int t_dev = -1;           // the calling thread's current device (kvz_hip_init / kvz_hip_set_device)
static thread_local char g_err[512] = "";     // last error of the calling thread

struct tune_entry { const char *key; std::atomic<int> value; };
static tune_entry g_tune[] = { { "sad_wgs_per_cu", {-1} }, { "satd8_wgs_per_cu", {-1} }, { "dct32_wgs_per_cu", {-1} },
                               { "idct32_wgs_per_cu", {-1} }, { "dct_wgs_per_cu", {-1} }, { "qr32_wgs_per_cu", {-1} },
                               { "qr_wgs_per_cu", {-1} }, { "dct16_wgs_per_cu", {-1} }, { "idct16_wgs_per_cu", {-1} },
                               { "qr16_wgs_per_cu", {-1} }, { "qr4_lane_kernel", {-1} },
                               { "sao_edge_fast", {-1} }, 
                               { "intra_rough_waves", {-1} }, { "pair_wave_kernel", {-1} }, { "qr4_wgs_per_cu", {-1} }, { "quant_wgs_per_cu", {-1} },
                               { "qr8_reg_kernel", {-1} }, { "qr8_wgs_per_cu", {-1} }, { "qr_tile_kernel", {-1} }, { "dct4_tile", {-1} }, { "dct4_wgs_per_cu", {-1} }, { "idct4_wgs_per_cu", {-1} },
                               { "wg_chunk_min_wgs", {-1} }, { "pair_satd_threads", {-1} }, { "qr8_tile_kernel", {-1} }, { "qr_tile_pipe", {-1} }, { "pipe", {-1} }, { "dct_pipe", {-1} }, { "sample8_wave", {-1} }, { "pair_satd16_lanes", {-1} }, { "full_qsad", {-1} }, { "service_streams", {-1} }, { "service_inflight", {-1} }, { "service_workers", {-1} }, { "service_linger_us", {-1} }, { "service_life_ms", {-1} }, { "service_spin_us", {-1} }, { "service_ticket_base_k", {-1} }, { "service_push", {-1} }, { "service_nap_us", {-1} }, { "service_spin_crowded_us", {-1} } };
int tuning(const char *key, int dflt)
{
  for (auto &e : g_tune) if (!std::strcmp(e.key, key)) { const int v = e.value.load(std::memory_order_relaxed); return v >= 0 ? v : dflt; }
  return dflt;
}

int ctx_device() { return t_dev >= 0 ? t_dev : g_default_dev.load(std::memory_order_acquire); }
bool ctx_ready()
{
  const int d = ctx_device();
  return d >= 0 && g_ctx[d].ready.load(std::memory_order_acquire);
}

// HIP's current device is a per-thread setting that starts at device 0 and that the host may change behind our back
// (a framework's set_device): every entry makes the HIP device of the calling thread the thread's kvz_hip device.
bool ctx_enter()
{
  const int d = ctx_device();
  if (d < 0 || !g_ctx[d].ready.load(std::memory_order_acquire)) return false;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != d) {
    if (hipSetDevice(d) != hipSuccess) return false;
  }
  return true;
}
hipStream_t ctx_stream(kvz_hip_stream s) { return s ? (hipStream_t)s : g_ctx[ctx_device()].stream; }
int num_cus() { const int d = ctx_device(); return d >= 0 ? g_ctx[d].num_cus : 256; }

void set_error(const char *what, hipError_t e)
{
  std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
void set_error_msg(const char *what) { std::snprintf(g_err, sizeof(g_err), "%s", what); }
int invalid_arg(const char *entry)
{
  std::snprintf(g_err, sizeof(g_err), "%s: invalid argument (null or misaligned buffer, size or parameter out of range)", entry);
  return KVZ_HIP_ERR_INVALID;
}

// brings up the context of one device (idempotent); g_mu held by the caller
static int ctx_create_locked(int device)
{
  dev_ctx &c = g_ctx[device];
  if (c.ready.load(std::memory_order_acquire)) return KVZ_HIP_OK;
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) { set_error("hipSetDevice", e); return KVZ_HIP_ERR_NO_DEVICE; }
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { set_error("hipGetDeviceProperties", e); return KVZ_HIP_ERR_NO_DEVICE; }
  std::snprintf(c.name, sizeof(c.name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::snprintf(g_err, sizeof(g_err), "kvz_hip_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return KVZ_HIP_ERR_NO_DEVICE;
  }
  c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // A BLOCKING stream: work enqueued with a NULL kvz_hip_stream is ordered after everything the host put on the legacy
  // default stream before the call (hipMemcpy, a framework's default stream) and before what it puts there afterwards --
  // what a host that passes NULL expects from a HIP API.  Streams from kvz_hip_stream_create() are non-blocking.
  if ((e = hipStreamCreateWithFlags(&c.stream, hipStreamDefault)) != hipSuccess) { set_error("hipStreamCreate", e); return KVZ_HIP_ERR_RUNTIME; }
  c.ready.store(true, std::memory_order_release);
  return KVZ_HIP_OK;
}

}  // namespace kvzhip

using namespace kvzhip;

#define HIP_TRY(call, what)                                  \
  do {                                                       \
    hipError_t e__ = (call);                                 \
    if (e__ != hipSuccess) { set_error(what, e__); return KVZ_HIP_ERR_RUNTIME; } \
  } while (0)

extern "C" {

int kvz_hip_set_tuning(const char *key, int value)
{
  if (!key) return kvzhip::invalid_arg(__func__);
  for (auto &e : g_tune) if (!std::strcmp(e.key, key)) { e.value.store(value, std::memory_order_relaxed); return KVZ_HIP_OK; }
  return kvzhip::invalid_arg(__func__);
}

int kvz_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int kvz_hip_init(int device)
{
  if (device < 0) {
    // "whatever the process uses": the calling thread's current device if it has one, else the process default,
    // else $KVZ_HIP_DEVICE, else device 0
    if (ctx_ready()) return KVZ_HIP_OK;
    const int dflt = g_default_dev.load(std::memory_order_acquire);
    const char *env = std::getenv("KVZ_HIP_DEVICE");
    device = dflt >= 0 ? dflt : (env ? std::atoi(env) : 0);
  }
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error_msg("kvz_hip_init: no HIP device visible (this library has no CPU fallback)");
    return KVZ_HIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n || device >= KVZ_MAX_DEVICES) {      // < 0: a negative $KVZ_HIP_DEVICE
    std::snprintf(g_err, sizeof(g_err), "kvz_hip_init: device index %d out of range (%d device(s) visible)", device, n);
    return KVZ_HIP_ERR_INVALID;
  }
  const bool first = g_default_dev.load(std::memory_order_acquire) < 0;
  const int rc = ctx_create_locked(device);
  if (rc != KVZ_HIP_OK) return rc;
  if (first) {
    // A/B runs of unmodified hosts: KVZ_HIP_TUNE="key=value,key=value" presets kvz_hip_set_tuning knobs
    if (const char *env = std::getenv("KVZ_HIP_TUNE")) {
      std::string all(env);
      size_t pos = 0;
      while (pos < all.size()) {
        size_t end = all.find(',', pos);
        if (end == std::string::npos) end = all.size();
        const std::string kv = all.substr(pos, end - pos);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos) {
          const std::string key = kv.substr(0, eq);
          for (auto &t : g_tune) if (key == t.key) t.value.store(std::atoi(kv.c_str() + eq + 1), std::memory_order_relaxed);
        }
        pos = end + 1;
      }
    }
    g_default_dev.store(device, std::memory_order_release);
  }
  t_dev = device;             // like hipSetDevice: the initialising thread now works on this device
  return KVZ_HIP_OK;
}

int kvz_hip_set_device(int device)
{
  if (device < 0) return kvzhip::invalid_arg(__func__);
  if (device >= KVZ_MAX_DEVICES || !g_ctx[device].ready.load(std::memory_order_acquire)) {
    const int saved = t_dev;
    const int rc = kvz_hip_init(device);          // binds on success
    if (rc != KVZ_HIP_OK) { t_dev = saved; return rc; }
  }
  t_dev = device;
  if (hipSetDevice(device) != hipSuccess) { set_error_msg("kvz_hip_set_device: hipSetDevice failed"); return KVZ_HIP_ERR_RUNTIME; }
  return KVZ_HIP_OK;
}

int kvz_hip_get_device(void) { return ctx_device(); }

void kvz_hip_shutdown(void)
{
  std::lock_guard<std::mutex> lk(g_mu);
  for (int d = 0; d < KVZ_MAX_DEVICES; ++d) {
    dev_ctx &c = g_ctx[d];
    if (!c.ready.load(std::memory_order_acquire)) continue;
    c.ready.store(false, std::memory_order_release);
    if (hipSetDevice(d) == hipSuccess) {
      (void)hipStreamSynchronize(c.stream);
      (void)hipStreamDestroy(c.stream);
    }
    c.stream = nullptr;
  }
  g_default_dev.store(-1, std::memory_order_release);
  t_dev = -1;
}

const char *kvz_hip_last_error(void) { return g_err; }
const char *kvz_hip_device_name(void) { const int d = ctx_device(); return d >= 0 ? g_ctx[d].name : ""; }
int kvz_hip_abi_version(void) { return KVZ_HIP_ABI_VERSION; }

void *kvz_hip_malloc(size_t bytes)
{
  if (!ctx_enter() && (kvz_hip_init(-1) != KVZ_HIP_OK || !ctx_enter())) return nullptr;
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) { set_error("hipMalloc", e); return nullptr; }
  return p;
}
void kvz_hip_free(void *dptr) { if (dptr) (void)hipFree(dptr); }
void *kvz_hip_malloc_host(size_t bytes)
{
  if (!ctx_enter() && (kvz_hip_init(-1) != KVZ_HIP_OK || !ctx_enter())) return nullptr;
  void *p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) { set_error("hipHostMalloc", e); return nullptr; }
  return p;
}
void kvz_hip_free_host(void *hptr) { if (hptr) (void)hipHostFree(hptr); }

int kvz_hip_memcpy_h2d(void *dst, const void *src, size_t bytes, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx_stream(s)), "hipMemcpyAsync(H2D)");
  return KVZ_HIP_OK;
}
int kvz_hip_memcpy_d2h(void *dst, const void *src, size_t bytes, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx_stream(s)), "hipMemcpyAsync(D2H)");
  HIP_TRY(hipStreamSynchronize(ctx_stream(s)), "hipStreamSynchronize");
  return KVZ_HIP_OK;
}
int kvz_hip_memset(void *dst, int value, size_t bytes, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  HIP_TRY(hipMemsetAsync(dst, value, bytes, ctx_stream(s)), "hipMemsetAsync");
  return KVZ_HIP_OK;
}
int kvz_hip_memcpy_d2d(void *dst, const void *src, size_t bytes, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx_stream(s)), "hipMemcpyAsync(D2D)");
  return KVZ_HIP_OK;
}
// Rows of a reconstructed plane from one device's shard into a neighbour's halo, inside one process:
// an asynchronous peer copy over xGMI on a stream of the calling thread's current device.
static std::atomic<bool> g_peer_on[KVZ_MAX_DEVICES][KVZ_MAX_DEVICES];      // [from][to]: peer access enabled

// the calling thread's current device `cur` may read / write memory of device `other` (once per ordered pair)
static int enable_peer(int cur, int other)
{
  if (cur == other || g_peer_on[cur][other].load(std::memory_order_acquire)) return KVZ_HIP_OK;
  int can = 0;
  hipError_t e = hipDeviceCanAccessPeer(&can, cur, other);
  if (e != hipSuccess || !can) {
    std::snprintf(g_err, sizeof(g_err), "kvz_hip_memcpy_peer: device %d cannot access device %d", cur, other);
    return KVZ_HIP_ERR_RUNTIME;
  }
  e = hipDeviceEnablePeerAccess(other, 0);
  if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { set_error("hipDeviceEnablePeerAccess", e); return KVZ_HIP_ERR_RUNTIME; }
  (void)hipGetLastError();              // "already enabled" is not a failure
  g_peer_on[cur][other].store(true, std::memory_order_release);
  return KVZ_HIP_OK;
}

// a stream handed in by the caller must be a stream of the calling thread's current device
static int stream_on_current_device(kvz_hip_stream s, const char *entry)
{
  if (!s) return KVZ_HIP_OK;
  hipDevice_t dev = -1;
  if (hipStreamGetDevice((hipStream_t)s, &dev) != hipSuccess) { (void)hipGetLastError(); return KVZ_HIP_OK; }   // cannot tell: the copy itself will
  if ((int)dev != ctx_device()) {
    std::snprintf(g_err, sizeof(g_err), "%s: the stream belongs to device %d, the calling thread works on device %d", entry, (int)dev, ctx_device());
    return KVZ_HIP_ERR_INVALID;
  }
  return KVZ_HIP_OK;
}

int kvz_hip_memcpy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!dst || !src || dst_device < 0 || src_device < 0 || dst_device >= KVZ_MAX_DEVICES || src_device >= KVZ_MAX_DEVICES ||
      !g_ctx[dst_device].ready.load(std::memory_order_acquire) || !g_ctx[src_device].ready.load(std::memory_order_acquire))
    return kvzhip::invalid_arg(__func__);
  const int cur = ctx_device();
  if (cur != dst_device && cur != src_device) {
    set_error_msg("kvz_hip_memcpy_peer: the calling thread's current device must be the source or the destination");
    return KVZ_HIP_ERR_INVALID;
  }
  int rc = stream_on_current_device(s, "kvz_hip_memcpy_peer");
  if (rc != KVZ_HIP_OK) return rc;
  if (bytes == 0) return KVZ_HIP_OK;
  if ((rc = enable_peer(cur, cur == dst_device ? src_device : dst_device)) != KVZ_HIP_OK) return rc;
  HIP_TRY(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, ctx_stream(s)), "hipMemcpyPeerAsync");
  return KVZ_HIP_OK;
}

// The halo exchange of kvazaar_amd/shard.py (exchange_halo_into) for a host that drives its shards from ONE process: the
// calling thread's shard pushes the `margin` rows at the top / bottom edge of its own rows into the halo rows of the shard
// above / below.  See include/kvz_hip.h.
int kvz_hip_halo_exchange(const kvz_hip_shard_plane *self, const kvz_hip_shard_plane *up, const kvz_hip_shard_plane *down,
                          uint32_t stride, int margin, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!self || !self->ext || margin < 0 || stride == 0 || self->top < 0 || self->rows < margin || self->device != ctx_device()) {
    set_error_msg("kvz_hip_halo_exchange: the calling thread must work on self->device, and a shard cannot be thinner than the margin");
    return KVZ_HIP_ERR_INVALID;
  }
  if ((up && (!up->ext || up->top < 0 || up->rows < 0)) || (down && (!down->ext || down->top < margin))) return kvzhip::invalid_arg(__func__);
  if (margin == 0) return KVZ_HIP_OK;
  const size_t bytes = (size_t)margin * stride;
  const u8 *own = (const u8 *)self->ext + (size_t)self->top * stride;
  if (up) {                  // my first rows -> the halo below the upper shard's own rows
    const int rc = kvz_hip_memcpy_peer((u8 *)up->ext + (size_t)(up->top + up->rows) * stride, up->device, own, self->device, bytes, s);
    if (rc != KVZ_HIP_OK) return rc;
  }
  if (down) {                // my last rows -> the halo above the lower shard's own rows
    const int rc = kvz_hip_memcpy_peer((u8 *)down->ext + (size_t)(down->top - margin) * stride, down->device,
                                       own + (size_t)(self->rows - margin) * stride, self->device, bytes, s);
    if (rc != KVZ_HIP_OK) return rc;
  }
  return KVZ_HIP_OK;
}
kvz_hip_stream kvz_hip_stream_create(void)
{
  if (!ctx_enter() && (kvz_hip_init(-1) != KVZ_HIP_OK || !ctx_enter())) return nullptr;
  hipStream_t st = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  if (e != hipSuccess) { set_error("hipStreamCreate", e); return nullptr; }
  return (kvz_hip_stream)st;
}
void kvz_hip_stream_destroy(kvz_hip_stream s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
int kvz_hip_stream_sync(kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  HIP_TRY(hipStreamSynchronize(ctx_stream(s)), "hipStreamSynchronize");
  return KVZ_HIP_OK;
}

void *kvz_hip_event_create(void)
{
  if (!ctx_enter() && (kvz_hip_init(-1) != KVZ_HIP_OK || !ctx_enter())) return nullptr;
  hipEvent_t ev = nullptr;
  if (hipEventCreate(&ev) != hipSuccess) return nullptr;
  return (void *)ev;
}
void kvz_hip_event_destroy(void *ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
int kvz_hip_event_record(void *ev, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  HIP_TRY(hipEventRecord((hipEvent_t)ev, ctx_stream(s)), "hipEventRecord");
  return KVZ_HIP_OK;
}
int kvz_hip_event_elapsed_ms(void *start, void *stop, float *ms)
{
  KVZ_CHECK_CTX();
  HIP_TRY(hipEventSynchronize((hipEvent_t)stop), "hipEventSynchronize");
  HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop), "hipEventElapsedTime");
  return KVZ_HIP_OK;
}

int kvz_hip_stream_wait_event(kvz_hip_stream s, void *ev)
{
  KVZ_CHECK_CTX();
  if (!ev) return kvzhip::invalid_arg(__func__);
  HIP_TRY(hipStreamWaitEvent(ctx_stream(s), (hipEvent_t)ev, 0), "hipStreamWaitEvent");
  return KVZ_HIP_OK;
}

// Graph capture.  ThreadLocal mode: other threads of the host (the per-call strategies run on every threadqueue
// worker) keep allocating / synchronising on their own streams while this thread captures.
int kvz_hip_graph_begin(kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  HIP_TRY(hipStreamBeginCapture(ctx_stream(s), hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
  return KVZ_HIP_OK;
}
int kvz_hip_graph_end(kvz_hip_stream s, kvz_hip_graph *graph_out)
{
  KVZ_CHECK_CTX();
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(ctx_stream(s), &g);     // always called: it also ends a capture an entry invalidated
  if (!graph_out) { if (g) (void)hipGraphDestroy(g); return kvzhip::invalid_arg(__func__); }
  *graph_out = nullptr;
  if (e != hipSuccess || !g) { set_error("hipStreamEndCapture", e); return KVZ_HIP_ERR_RUNTIME; }
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) { set_error("hipGraphInstantiate", e); return KVZ_HIP_ERR_RUNTIME; }
  *graph_out = (kvz_hip_graph)ex;
  return KVZ_HIP_OK;
}
int kvz_hip_graph_launch(kvz_hip_graph graph, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!graph) return kvzhip::invalid_arg(__func__);
  HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph, ctx_stream(s)), "hipGraphLaunch");
  return KVZ_HIP_OK;
}
void kvz_hip_graph_destroy(kvz_hip_graph graph) { if (graph) (void)hipGraphExecDestroy((hipGraphExec_t)graph); }

}  // extern "C"
