// api.hip -- library context, device-memory helpers and HIP-event timing of the
// C ABI declared in include/kvz_hip.h.
#include "kvz_hip_internal.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>

namespace kvzhip {

static std::mutex g_mu;
static std::atomic<bool> g_ready{false};
static int g_device = -1;
static int g_num_cus = 256;
static hipStream_t g_stream = nullptr;
static thread_local char g_err[512] = "";     // last error of the calling thread
static char g_name[256] = "";

struct tune_entry { const char *key; int value; };
static tune_entry g_tune[] = { { "sad_wgs_per_cu", -1 }, { "satd8_wgs_per_cu", -1 }, { "dct32_wgs_per_cu", -1 },
                               { "idct32_wgs_per_cu", -1 }, { "dct_wgs_per_cu", -1 }, { "qr32_wgs_per_cu", -1 },
                               { "qr_wgs_per_cu", -1 }, { "dct16_wgs_per_cu", -1 }, { "idct16_wgs_per_cu", -1 }, { "idct16_use_mfma", -1 },
                               { "qr16_wgs_per_cu", -1 }, { "qr16_use_mfma", -1 }, { "qr4_lane_kernel", -1 },
                               { "me_big_threads", -1 }, { "sao_edge_fast", -1 }, { "me_medium_threads", -1 },
                               { "intra_rough_waves", -1 }, { "pair_wave_kernel", -1 }, { "qr4_wgs_per_cu", -1 }, { "quant_wgs_per_cu", -1 } };
int tuning(const char *key, int dflt)
{
  for (auto &e : g_tune) if (!std::strcmp(e.key, key)) return e.value >= 0 ? e.value : dflt;
  return dflt;
}

bool ctx_ready() { return g_ready.load(std::memory_order_acquire); }

// HIP's current device is a per-thread setting that starts at device 0: every thread that enters the library
// (encoder worker threads call the strategy functions directly) is bound to the context's device once.
bool ctx_enter()
{
  if (!ctx_ready()) return false;
  static thread_local int bound = -1;
  if (bound != g_device) {
    if (hipSetDevice(g_device) != hipSuccess) return false;
    bound = g_device;
  }
  return true;
}
hipStream_t ctx_stream(kvz_hip_stream s) { return s ? (hipStream_t)s : g_stream; }
int num_cus() { return g_num_cus; }

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
  for (auto &e : g_tune) if (!std::strcmp(e.key, key)) { e.value = value; return KVZ_HIP_OK; }
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
  if (ctx_ready()) return KVZ_HIP_OK;
  std::lock_guard<std::mutex> lk(g_mu);
  if (ctx_ready()) return KVZ_HIP_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error_msg("kvz_hip_init: no HIP device visible (this library has no CPU fallback)");
    return KVZ_HIP_ERR_NO_DEVICE;
  }
  if (device < 0) {
    const char *env = std::getenv("KVZ_HIP_DEVICE");
    device = env ? std::atoi(env) : 0;
  }
  if (device >= n) { set_error_msg("kvz_hip_init: device index out of range"); return KVZ_HIP_ERR_INVALID; }
  if ((e = hipSetDevice(device)) != hipSuccess) { set_error("hipSetDevice", e); return KVZ_HIP_ERR_NO_DEVICE; }
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { set_error("hipGetDeviceProperties", e); return KVZ_HIP_ERR_NO_DEVICE; }
  std::snprintf(g_name, sizeof(g_name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::snprintf(g_err, sizeof(g_err), "kvz_hip_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return KVZ_HIP_ERR_NO_DEVICE;
  }
  g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if ((e = hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking)) != hipSuccess) { set_error("hipStreamCreate", e); return KVZ_HIP_ERR_RUNTIME; }
  g_device = device;
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
        for (auto &e : g_tune) if (key == e.key) e.value = std::atoi(kv.c_str() + eq + 1);
      }
      pos = end + 1;
    }
  }
  g_ready.store(true, std::memory_order_release);
  return KVZ_HIP_OK;
}

void kvz_hip_shutdown(void)
{
  std::lock_guard<std::mutex> lk(g_mu);
  if (!ctx_ready()) return;
  (void)hipStreamSynchronize(g_stream);
  (void)hipStreamDestroy(g_stream);
  g_stream = nullptr;
  g_ready.store(false, std::memory_order_release);
}

const char *kvz_hip_last_error(void) { return g_err; }
const char *kvz_hip_device_name(void) { return g_name; }
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
