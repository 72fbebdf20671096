// serve.hip -- the search service: motion searches posted concurrently by many host threads leave in shared launches.
//
// Reference: the encoder runs one CTU job per threadqueue worker (src/encoderstate.c:777-828; workers are pthreads,
// threadqueue.c:263; frames in flight under --owf, encoder.c:59-119), and each worker reaches search_pu_inter
// (search_inter.c:1451-1520) with one PU at a time and walks its reference pictures in order (:1502-1507).  This file is the
// piece between those workers and the kernels of me_search.hip (declared in include/kvz_hip.h, "search service").
//
//   * No dispatcher thread.  A caller appends its request to the pending list and then either finds the launch path free --
//     it takes EVERYTHING pending (flat combining) and launches it -- or waits for its results while another caller
//     launches; a waiting caller that finds the path free launches what has queued up meanwhile, its own request or not.
//     A launch call takes a few microseconds, which is exactly the window in which other workers' requests pile up:
//     batches grow with the load, an idle service costs no wait.
//   * A batch is copied into a ring of page-locked buffers the kernels read directly (one 176-byte unit per PU and
//     reference picture); results come back the same way, each unit signalling its own `done` word with a system-scope
//     release store.  There is no copy command and no stream synchronisation on the request path.
//   * Launches rotate over a few non-blocking streams so that consecutive batches overlap on the device; at most `service_inflight`
//     batches are in the air (the runtime has four hardware queues: more launches than that wait behind each other one by one,
//     fewer and larger ones do not).
//
// Resident workers (tuning "service_workers" > 0): the same requests without a launch on their way.  Workgroups of
// me_search.hip's serve_worker_kernel stay on the device and take units by ticket from a ring of slots in page-locked memory:
//   host, under ring_mu:  wait until slot.seq == 0, write the unit, slot.seq = serve_seq(ticket), ... ctl->tail += n      (publish)
//   worker:               ticket = head++ while head < tail (device atomics; ctl->tail is read across PCIe by one worker at a
//                         time and mirrored in device memory), copy the unit, slot.seq = 0, search, write the results and `done`.
//   Workers leave when they have found no work for `service_linger_us`, or at the first idle moment after `service_life_ms`, or when
//   the service is destroyed.  Who makes sure somebody is there: a worker says "gone" (ctl->alive[w] = 0) BEFORE its last look at
//   ctl->tail, and the look is a PCIe read that cannot overtake that store; a caller publishes BEFORE it looks at alive[] (and keeps
//   looking while it waits) and starts the workers that are missing.  So a published unit is seen either by a worker's last look or
//   by a caller that finds the worker gone.  alive[w] has one writer at a time: the host sets it when it launches worker w, which it
//   only does when it reads 0; the worker clears it once.
#include "kvz_hip_internal.h"
#include "serve_seq.h"

#include <immintrin.h>
#include <sched.h>
#include <sys/prctl.h>
#include <time.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

using namespace kvzhip;

namespace {

constexpr int N_STREAMS = 32;         // launch streams at most (tuning "service_streams", default 8)
constexpr int N_UPLOAD = 8;           // upload streams (put_rect)
constexpr int N_WSTREAMS = 4;         // streams the worker generations rotate over (their own priority level, hence their own hardware queues)
constexpr unsigned WRING_SLOTS = 4096;    // a power of two, more than any number of units in flight (max_threads x 16 is checked)
constexpr int BATCH_CAP = 256;        // units per batch buffer (a larger batch leaves as several launches)
// How a caller waits for its answer (a search takes 25-45 us on the device).  With a core to itself it polls: 256 pauses, then
// sched_yield until SPIN_NS ("service_spin_us"; 1080p --me full16, 16 threads on 16 cores: 6.6 -> 7.5 frames/s against 40 us).  With
// more callers than cores the core is needed by somebody else: 16 pauses, sched_yield until SPIN_CROWDED_NS ("service_spin_crowded_us"),
// then naps of NAP_NS ("service_nap_us"; the kernel adds the thread's timer slack, set to 1 us) -- measured with 32 encoder threads on
// 16 cores: 8.3 frames/s polling for 40 us, 9.7 for 10 us, 11.6-13.1 for 0-5 us with naps of 5-10 us; 48 threads: 8.6 -> 10.6-11.2.
constexpr uint64_t SPIN_NS = 100 * 1000, SPIN_CROWDED_NS = 5 * 1000;
constexpr long NAP_NS = 10 * 1000;
constexpr uint64_t WAIT_LIMIT_NS = 20ull * 1000 * 1000 * 1000;    // a request that is not answered in 20 s is a failure

inline uint64_t now_ns()
{
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec;
}

constexpr int MAX_TABLE_RANGE = 32;   // kvz_hip_me_service_sad_tables: +-range, (2 * range + 1)^2 candidates per CTU and picture
struct thread_area {                  // one per calling thread, page-locked
  serve_result res[KVZ_HIP_SERVICE_MAX_REFS];
  // SAD tables: one descriptor per row of the window and picture (the row's candidates are its mv_offsets), the offsets of a row
  kvz_hip_ctu_search ctus[KVZ_HIP_SERVICE_MAX_REFS][2 * MAX_TABLE_RANGE + 1];
  int16_t offs[2 * MAX_TABLE_RANGE + 1][2];
};
struct thread_tables {                // per calling thread, host side: the page-locked table buffer (grown on demand) and its stream
  uint32_t *buf = nullptr;
  size_t bytes = 0;
  hipStream_t stream = nullptr;
};

struct pending_req {
  const kvz_hip_me_request *req;      // the caller's request (alive until its results are there)
  int thread_slot;
};

}  // namespace

struct kvz_hip_me_service {
  int device = 0;
  int w = 0, h = 0, n_slots = 0, max_threads = 0;
  size_t plane_bytes = 0;
  u8 *planes = nullptr;                                 // device: n_slots planes, stride = w
  thread_area *areas = nullptr;                         // page-locked host, device-visible
  serve_unit *ring = nullptr;                           // page-locked host: n_batch x BATCH_CAP units
  hipStream_t streams[N_STREAMS] = {};
  int n_streams = 8;
  int inflight_cap = 4;                                  // batches in the air at most (0: no limit); tuning "service_inflight"
  std::atomic<int> inflight{0};
  hipStream_t up_streams[N_UPLOAD] = {};
  std::mutex up_mu[N_UPLOAD];
  int n_batch = 0;                                      // ring of batch buffers: max_threads + 8
  std::atomic<int> *batch_open = nullptr;               // requests of the batch in that ring buffer not answered yet
  std::atomic<int> *slot_batch = nullptr;               // per calling thread: the ring buffer its request went out in
  u8 *up_stage[N_UPLOAD] = {};                          // page-locked staging of put_rect, one per upload stream
  thread_tables *tables = nullptr;                      // [max_threads], each touched by its own thread only
  std::atomic<uint64_t> st_tables{0}, st_table_bytes{0}, st_table_ns{0};
  uint64_t next_batch = 0;                              // touched under launch_mu only
  // resident workers
  int n_workers = 0;                                    // 0: a launch per batch
  uint64_t spin_ns = SPIN_NS;
  long nap_ns = NAP_NS;
  uint64_t spin_crowded_ns = SPIN_CROWDED_NS;
  bool spin_tuned = false;
  int host_cpus = 1;                                    // CPUs this process may run on
  serve_slot *wring = nullptr;                          // page-locked: WRING_SLOTS slots
  serve_ring_ctl *ctl = nullptr;                        // page-locked
  serve_ring_dev *wdev = nullptr;                       // device
  serve_slot *dring = nullptr;                          // push mode: the ring's copy in fine-grained device memory, written through the BAR
  serve_push *dpush = nullptr;                          // push mode: tail and quit for the workers, same
  std::mutex ring_mu;
  unsigned long long wtail = 0;                         // under ring_mu
  bool debug = false;                                   // KVZ_HIP_SERVICE_DEBUG: a line of worker statistics on stderr when the service is destroyed
  std::atomic<long long> dbg_sum_pick{0}, dbg_n_pick{0}, dbg_alive_sum{0}, dbg_alive_n{0}, dbg_hist[16] = {};
  hipStream_t wstreams[N_WSTREAMS] = {};
  int wnext = 0;                                        // under launch_mu
  unsigned long long linger_ticks = 0, life_ticks = 0;
  std::mutex pend_mu;
  std::vector<pending_req> pending;
  std::atomic<int> n_pending{0};
  std::mutex launch_mu;
  std::atomic<int> next_thread{0};
  std::atomic<int> failed{0};
  uint64_t id = 0;
  // statistics
  std::atomic<uint64_t> st_requests{0}, st_units{0}, st_batches{0}, st_launches{0}, st_max_batch{0}, st_rects{0}, st_rect_bytes{0}, st_wait_ns{0};
};

namespace {

std::atomic<uint64_t> g_service_ids{1};
// A calling thread's place in each service it has used (a result area of its own, a row of statistics): remembered per thread for
// the last few services, so that a thread that alternates between two encoder instances does not take a new place at every switch.
struct thread_binding { uint64_t id; int slot; };
constexpr int N_BINDINGS = 8;
thread_local thread_binding t_bind[N_BINDINGS] = {};
thread_local unsigned t_bind_next = 0;

int thread_slot(kvz_hip_me_service *svc)
{
  for (int i = 0; i < N_BINDINGS; ++i)
    if (t_bind[i].id == svc->id) return t_bind[i].slot;
  const int s = svc->next_thread.fetch_add(1);
  if (s >= svc->max_threads) return -1;
  thread_binding &b = t_bind[t_bind_next++ % N_BINDINGS];      // the oldest remembered service gives way (its place stays taken there)
  b.id = svc->id; b.slot = s;
  prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);          // this thread's naps (see kvz_hip_me_service_search) end on time: 1 us of slack instead of 50
  return s;
}

// takes everything pending and launches it; launch_mu held by the caller
int drain_and_launch(kvz_hip_me_service *svc)
{
  std::vector<pending_req> grabbed;
  {
    std::lock_guard<std::mutex> lk(svc->pend_mu);
    grabbed.swap(svc->pending);
    svc->n_pending.store(0, std::memory_order_relaxed);
  }
  if (grabbed.empty()) return KVZ_HIP_OK;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != svc->device) {
    if (hipSetDevice(svc->device) != hipSuccess) { set_error_msg("kvz_hip_me_service: hipSetDevice failed"); return KVZ_HIP_ERR_RUNTIME; }
  }
  size_t at = 0;
  int rc = KVZ_HIP_OK;
  while (at < grabbed.size() && rc == KVZ_HIP_OK) {
    // A ring buffer is free once every caller of the batch it carried has been answered (they count themselves out).  A
    // caller has one request in flight, so at most max_threads buffers are taken and the ring has more: the next free
    // one is used.  (Waiting for a particular buffer here could wait for the launching thread's OWN unanswered request.)
    int b = -1;
    for (int k = 0; k < svc->n_batch && b < 0; ++k) {
      const int c = (int)((svc->next_batch + (uint64_t)k) % (uint64_t)svc->n_batch);
      if (svc->batch_open[c].load(std::memory_order_acquire) == 0) b = c;
    }
    if (b < 0) { set_error_msg("kvz_hip_me_service: no free batch buffer (more callers than max_threads?)"); rc = KVZ_HIP_ERR_RUNTIME; break; }
    svc->next_batch = (uint64_t)b;
    serve_unit *buf = svc->ring + (size_t)b * BATCH_CAP;
    size_t end = at;
    int total = 0;
    bool constrained = false;
    while (end < grabbed.size() && total + grabbed[end].req->n_refs <= BATCH_CAP) {
      const kvz_hip_me_request *r = grabbed[end].req;
      thread_area *area = svc->areas + grabbed[end].thread_slot;
      constrained = constrained || r->params.wpp_owf != 0 || r->params.mv_constraint != 0;
      for (int i = 0; i < r->n_refs; ++i) {
        serve_unit &u = buf[total++];
        u.pic_slot = r->pic_slot; u.ref_slot = r->ref_slot[i];
        u.result = &area->res[i];
        u.pu = r->pu[i];
        u.pu.width = r->pu[0].width; u.pu.height = r->pu[0].height;     // every picture sees the same PU
        u.prm = r->params;
        u.prm.cost_to_beat = nullptr; u.prm.cabac = nullptr; u.prm.mv_rdo = 0; u.prm.size_classes = 0;
        if (u.prm.tile_w == 0 && u.prm.tile_h == 0) { u.prm.tile_x = 0; u.prm.tile_y = 0; u.prm.tile_w = svc->w; u.prm.tile_h = svc->h; }
      }
      svc->slot_batch[grabbed[end].thread_slot].store(b, std::memory_order_relaxed);
      ++end;
    }
    svc->batch_open[b].store((int)(end - at), std::memory_order_release);
    svc->inflight.fetch_add(1, std::memory_order_relaxed);
    rc = serve_launch(constrained, svc->planes, svc->plane_bytes, svc->n_slots, (u32)svc->w, svc->w, svc->h, buf, total, svc->streams[svc->st_batches.load(std::memory_order_relaxed) % (unsigned)svc->n_streams]);
    svc->st_launches.fetch_add(1, std::memory_order_relaxed);
    svc->next_batch = (uint64_t)((b + 1) % svc->n_batch);
    svc->st_batches.fetch_add(1, std::memory_order_relaxed);
    svc->st_units.fetch_add((uint64_t)total, std::memory_order_relaxed);
    uint64_t m = svc->st_max_batch.load(std::memory_order_relaxed);
    while ((uint64_t)total > m && !svc->st_max_batch.compare_exchange_weak(m, (uint64_t)total)) {}
    at = end;
  }
  if (rc != KVZ_HIP_OK) svc->failed.store(1);
  return rc;
}

// starts the workers that are missing; cheap when none is (alive[] is 16 flags to the cache line and only changes when a worker comes or goes)
int ensure_workers(kvz_hip_me_service *svc)
{
  serve_ring_ctl *ctl = svc->ctl;
  int alive = 0;
  for (int w = 0; w < svc->n_workers; ++w) alive += __atomic_load_n(&ctl->alive[w], __ATOMIC_RELAXED) != 0u;
  // Workers come and go as a crowd: those of one launch were born together and leave together (end of life, or the service idle), and
  // a launch is one kernel that ends with its last worker -- the next kernel on its stream waits for that, so launches must be few and
  // whole.  More than half there: enough.  Otherwise everybody who is missing is started as one launch, on the stream the previous
  // launch did not use.
  if (svc->debug) { svc->dbg_alive_sum.fetch_add(alive, std::memory_order_relaxed); svc->dbg_alive_n.fetch_add(1, std::memory_order_relaxed); }
  if (alive > svc->n_workers / 2) return KVZ_HIP_OK;
  if (!svc->launch_mu.try_lock()) return KVZ_HIP_OK;    // somebody else is starting them
  serve_worker_ids ids;
  int n = 0;
  for (int w = 0; w < svc->n_workers; ++w)
    if (__atomic_load_n(&ctl->alive[w], __ATOMIC_ACQUIRE) == 0u) { ids.id[n++] = (unsigned char)w; __atomic_store_n(&ctl->alive[w], 2u, __ATOMIC_RELAXED); }
  if (n < svc->n_workers / 2) {                          // they came back meanwhile (another caller's launch)
    for (int i = 0; i < n; ++i) __atomic_store_n(&ctl->alive[ids.id[i]], 0u, __ATOMIC_RELAXED);
    n = 0;
  }
  int rc = KVZ_HIP_OK;
  if (n > 0) {
    std::atomic_thread_fence(std::memory_order_seq_cst);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != svc->device) (void)hipSetDevice(svc->device);
    // poll period per worker: with the ticket count in device memory every worker looks every 0.5 us; across PCIe they take turns
    rc = serve_workers_launch_push(svc->planes, svc->plane_bytes, svc->n_slots, (u32)svc->w, svc->w, svc->h, svc->dring ? svc->dring : svc->wring, svc->wring,
                                   svc->dpush, WRING_SLOTS - 1, ctl, svc->wdev, ids, n, svc->linger_ticks, svc->life_ticks,
                                   svc->dpush ? 50ull : 50ull * (unsigned long long)svc->n_workers, svc->wstreams[svc->wnext++ % N_WSTREAMS]);
    svc->st_launches.fetch_add(1, std::memory_order_relaxed);
    if (rc != KVZ_HIP_OK) {
      for (int i = 0; i < n; ++i) __atomic_store_n(&ctl->alive[ids.id[i]], 0u, __ATOMIC_RELAXED);
      svc->failed.store(1);
    }
  }
  svc->launch_mu.unlock();
  return rc;
}

// writes the units of a request into the ring and publishes them
int post_to_ring(kvz_hip_me_service *svc, const kvz_hip_me_request *r, thread_area *area)
{
  std::lock_guard<std::mutex> lk(svc->ring_mu);
  const unsigned long long t0 = svc->wtail;
  for (int i = 0; i < r->n_refs; ++i) {
    serve_slot *slot = svc->wring + ((t0 + (unsigned)i) & (WRING_SLOTS - 1));
    // free unless the ring has gone round while a worker was still copying the unit of 4096 tickets ago
    for (uint64_t t = 0; __atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE) != 0u;) {
      __builtin_ia32_pause();
      if (t == 0) t = now_ns();
      else if (now_ns() - t > WAIT_LIMIT_NS) { svc->failed.store(1); set_error_msg("kvz_hip_me_service_search: the ring does not drain"); return KVZ_HIP_ERR_RUNTIME; }
    }
    serve_unit &u = slot->u;
    u.pic_slot = r->pic_slot; u.ref_slot = r->ref_slot[i];
    u.result = &area->res[i];
    u.pu = r->pu[i];
    u.pu.width = r->pu[0].width; u.pu.height = r->pu[0].height;     // every picture sees the same PU
    u.prm = r->params;
    u.prm.cost_to_beat = nullptr; u.prm.cabac = nullptr; u.prm.mv_rdo = 0; u.prm.size_classes = 0;
    if (u.prm.tile_w == 0 && u.prm.tile_h == 0) { u.prm.tile_x = 0; u.prm.tile_y = 0; u.prm.tile_w = svc->w; u.prm.tile_h = svc->h; }
    if (svc->dring) {                                     // push: the unit and its sequence word straight into device memory
      serve_slot *ds = svc->dring + ((t0 + (unsigned)i) & (WRING_SLOTS - 1));
      std::memcpy(&ds->u, &u, sizeof(serve_unit));
      ds->seq = serve_seq(t0 + (unsigned)i);
    }
    __atomic_store_n(&slot->seq, serve_seq(t0 + (unsigned)i), __ATOMIC_RELEASE);
  }
  svc->wtail = t0 + (unsigned)r->n_refs;
  __atomic_store_n(&svc->ctl->tail, svc->wtail, __ATOMIC_RELEASE);
  if (svc->dpush) {
    _mm_sfence();                                         // the units have left the write-combining buffers before the count that publishes them
    *reinterpret_cast<volatile unsigned long long *>(&svc->dpush->tail) = svc->wtail;
    _mm_sfence();
  }
  svc->st_batches.fetch_add(1, std::memory_order_relaxed);
  svc->st_units.fetch_add((uint64_t)r->n_refs, std::memory_order_relaxed);
  return KVZ_HIP_OK;
}

bool request_ok(const kvz_hip_me_service *svc, const kvz_hip_me_request *r)
{
  if (r->n_refs < 1 || r->n_refs > KVZ_HIP_SERVICE_MAX_REFS || r->pic_slot < 0 || r->pic_slot >= svc->n_slots) return false;
  for (int i = 0; i < r->n_refs; ++i)
    if (r->ref_slot[i] < 0 || r->ref_slot[i] >= svc->n_slots) return false;
  const kvz_hip_me_params &p = r->params;
  if (p.lambda_cost < 0 || p.lambda_cost > (1 << 20) || p.fme_level < 0 || p.fme_level > 4 || p.early_termination < 0 || p.early_termination > 2 ||
      p.algorithm < 0 || p.algorithm > 3 || (p.algorithm == 3 && (p.search_range < 1 || p.search_range > 64)) || p.mv_rdo) return false;
  if (p.mv_constraint < 0 || p.mv_constraint > 4) return false;
  // the exhaustive search reads the current block with scalar loads: dword-aligned addresses, i.e. PUs at multiples of 4 (every HEVC PU is)
  if (p.algorithm == 3)
    for (int i = 0; i < r->n_refs; ++i)
      if (r->pu[i].x & 3) return false;
  if (!(p.tile_w == 0 && p.tile_h == 0) &&
      (p.tile_x < 0 || p.tile_y < 0 || p.tile_w <= 0 || p.tile_h <= 0 || p.tile_x + p.tile_w > svc->w || p.tile_y + p.tile_h > svc->h ||
       (p.wpp_owf && ((p.tile_x & 63) || (p.tile_y & 63))))) return false;
  return true;
}

}  // namespace

extern "C" {

kvz_hip_me_service *kvz_hip_me_service_create(const kvz_hip_me_service_config *cfg)
{
  if (!cfg || cfg->width < 8 || cfg->height < 8 || cfg->width > 16384 || cfg->height > 16384 || (cfg->width & 3) ||
      cfg->max_pictures < 1 || cfg->max_pictures > 256 || cfg->max_threads < 1 || cfg->max_threads > 1024) {
    kvzhip::invalid_arg(__func__);
    return nullptr;
  }
  if (!ctx_enter() && (kvz_hip_init(-1) != KVZ_HIP_OK || !ctx_enter())) return nullptr;
  kvz_hip_me_service *svc = new (std::nothrow) kvz_hip_me_service;
  if (!svc) return nullptr;
  svc->device = ctx_device();
  svc->w = cfg->width; svc->h = cfg->height; svc->n_slots = cfg->max_pictures; svc->max_threads = cfg->max_threads;
  svc->n_batch = cfg->max_threads + 8;
  svc->plane_bytes = ((size_t)cfg->width * cfg->height + 255) & ~(size_t)255;
  svc->id = g_service_ids.fetch_add(1);
  bool ok = hipMalloc((void **)&svc->planes, svc->plane_bytes * svc->n_slots + 64) == hipSuccess;
  ok = ok && hipMemset(svc->planes, 0, svc->plane_bytes * svc->n_slots + 64) == hipSuccess;
  // page-locked and device-visible: the kernels read the ring and write the result areas across PCIe
  ok = ok && hipHostMalloc((void **)&svc->areas, sizeof(thread_area) * svc->max_threads, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess;
  ok = ok && hipHostMalloc((void **)&svc->ring, sizeof(serve_unit) * (size_t)svc->n_batch * BATCH_CAP, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess;
  if (ok) std::memset(svc->areas, 0, sizeof(thread_area) * svc->max_threads);
  svc->n_streams = kvzhip::tuning("service_streams", 8);
  svc->inflight_cap = kvzhip::tuning("service_inflight", 4);       // measured: 1080p full16 all served 2.41 -> 3.76 frames/s, medium 3.74 -> 4.31 (same box)
  if (svc->n_streams < 1) svc->n_streams = 1;
  if (svc->n_streams > N_STREAMS) svc->n_streams = N_STREAMS;
  for (int i = 0; i < svc->n_streams && ok; ++i) ok = hipStreamCreateWithFlags(&svc->streams[i], hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < N_UPLOAD && ok; ++i) ok = hipStreamCreateWithFlags(&svc->up_streams[i], hipStreamNonBlocking) == hipSuccess;
  // Resident workers are the default way to the device (measured against a launch per batch on the same box, 1080p, 16 encoder
  // threads: exhaustive +-16 search 4.3 -> 6.8 frames/s, preset medium 5.8 -> 6.3, 16 closed-loop C threads 211k -> 317k requests/s);
  // "service_workers" 0 selects the launches, and so does a service with more calling threads than the ring is sized for.
  svc->n_workers = kvzhip::tuning("service_workers", 64);
  svc->debug = getenv("KVZ_HIP_SERVICE_DEBUG") != nullptr;
  svc->spin_crowded_ns = 1000ull * (uint64_t)kvzhip::tuning("service_spin_crowded_us", (int)(SPIN_CROWDED_NS / 1000));
  svc->nap_ns = 1000L * kvzhip::tuning("service_nap_us", (int)(NAP_NS / 1000));
  if (svc->nap_ns < 1000) svc->nap_ns = 1000;
  svc->spin_tuned = kvzhip::tuning("service_spin_us", -1) >= 0;
  svc->spin_ns = 1000ull * (uint64_t)kvzhip::tuning("service_spin_us", (int)(SPIN_NS / 1000));
  {
    cpu_set_t set;
    CPU_ZERO(&set);
    svc->host_cpus = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : 1;
    // a container's share of the machine (cgroup v2 "cpu.max", v1 quota / period) when it is smaller than the affinity mask
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = "";
      if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
      fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (fscanf(g, "%lld", &quota) != 1) quota = -1;
      fclose(g);
      if (FILE *p = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(p, "%lld", &period) != 1) period = 0; fclose(p); }
    }
    if (quota > 0 && period > 0 && (quota + period - 1) / period < svc->host_cpus) svc->host_cpus = (int)((quota + period - 1) / period);
  }
  if (svc->n_workers > SERVE_MAX_WORKERS) svc->n_workers = SERVE_MAX_WORKERS;
  if ((unsigned)svc->max_threads * KVZ_HIP_SERVICE_MAX_REFS > WRING_SLOTS / 2) svc->n_workers = 0;
  if (svc->n_workers > 0 && ok) {
    // the wall clock counts 10 ns.  Idle workers cost a few atomics per microsecond; a relaunch costs the first request after a
    // pause ~15 us, so they stay through pauses of a couple of milliseconds (an encoder that serves only its largest PUs posts that rarely)
    svc->linger_ticks = 100ull * (unsigned long long)kvzhip::tuning("service_linger_us", 2000);
    svc->life_ticks = 100000ull * (unsigned long long)kvzhip::tuning("service_life_ms", 20);
    ok = ok && hipHostMalloc((void **)&svc->wring, sizeof(serve_slot) * WRING_SLOTS, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&svc->ctl, sizeof(serve_ring_ctl), hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess;
    ok = ok && hipMalloc((void **)&svc->wdev, sizeof(serve_ring_dev)) == hipSuccess;
    ok = ok && hipMemset(svc->wdev, 0, sizeof(serve_ring_dev)) == hipSuccess;
    if (ok) { std::memset(svc->wring, 0, sizeof(serve_slot) * WRING_SLOTS); std::memset(svc->ctl, 0, sizeof(serve_ring_ctl)); }
    // tickets normally start at 0; "service_ticket_base_k" (x 1024) starts them elsewhere, so that a test can walk the counters across
    // 2^32 in seconds instead of an hour of full load
    const unsigned long long base = 1024ull * (unsigned long long)kvzhip::tuning("service_ticket_base_k", 0);
    if (ok && base) {
      serve_ring_dev d0;
      std::memset(&d0, 0, sizeof(d0));
      d0.head = base; d0.tail = base;
      ok = hipMemcpy(svc->wdev, &d0, sizeof(d0), hipMemcpyHostToDevice) == hipSuccess;
      svc->wtail = base; svc->ctl->tail = base;
      if (svc->dpush) { *reinterpret_cast<volatile unsigned long long *>(&svc->dpush->tail) = base; _mm_sfence(); }
    }
    // push mode when the host can write device memory (large BAR); tuning "service_push" 0 keeps the workers reading host memory
    int large_bar = 0;
    if (ok && kvzhip::tuning("service_push", 1) != 0 && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, svc->device) == hipSuccess && large_bar) {
      bool got = hipExtMallocWithFlags((void **)&svc->dring, sizeof(serve_slot) * WRING_SLOTS, hipDeviceMallocFinegrained) == hipSuccess &&
                 hipExtMallocWithFlags((void **)&svc->dpush, sizeof(serve_push), hipDeviceMallocFinegrained) == hipSuccess &&
                 hipMemset(svc->dring, 0, sizeof(serve_slot) * WRING_SLOTS) == hipSuccess && hipMemset(svc->dpush, 0, sizeof(serve_push)) == hipSuccess &&
                 hipDeviceSynchronize() == hipSuccess;
      if (!got) {
        (void)hipGetLastError();
        if (svc->dring) (void)hipFree(svc->dring);
        if (svc->dpush) (void)hipFree(svc->dpush);
        svc->dring = nullptr; svc->dpush = nullptr;
      }
    }
    int least = 0, greatest = 0;
    ok = ok && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
    for (int i = 0; i < N_WSTREAMS && ok; ++i) ok = hipStreamCreateWithPriority(&svc->wstreams[i], hipStreamNonBlocking, greatest) == hipSuccess;
  }
  svc->batch_open = new (std::nothrow) std::atomic<int>[svc->n_batch];
  ok = ok && svc->batch_open != nullptr;
  for (int i = 0; i < svc->n_batch && ok; ++i) svc->batch_open[i].store(0);
  svc->tables = new (std::nothrow) thread_tables[svc->max_threads];
  ok = ok && svc->tables != nullptr;
  svc->slot_batch = new (std::nothrow) std::atomic<int>[svc->max_threads];
  ok = ok && svc->slot_batch != nullptr;
  for (int i = 0; i < N_UPLOAD && ok; ++i) ok = hipHostMalloc((void **)&svc->up_stage[i], svc->plane_bytes, hipHostMallocPortable) == hipSuccess;
  if (!ok) {
    set_error("kvz_hip_me_service_create", hipGetLastError());
    kvz_hip_me_service_destroy(svc);
    return nullptr;
  }
  (void)hipStreamSynchronize(nullptr);                 // the memsets above; not hipDeviceSynchronize: another service's resident workers may be running
  return svc;
}

void kvz_hip_me_service_destroy(kvz_hip_me_service *svc)
{
  if (!svc) return;
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur != svc->device) (void)hipSetDevice(svc->device);
  if (svc->ctl) {                                       // the workers leave at their next idle moment
    __atomic_store_n(&svc->ctl->quit, 1u, __ATOMIC_SEQ_CST);
    if (svc->dpush) { *reinterpret_cast<volatile uint32_t *>(&svc->dpush->quit) = 1u; _mm_sfence(); }
    for (int i = 0; i < N_WSTREAMS; ++i) if (svc->wstreams[i]) { (void)hipStreamSynchronize(svc->wstreams[i]); (void)hipStreamDestroy(svc->wstreams[i]); }
    if (svc->debug && svc->wdev) {
      if (svc->dbg_n_pick.load()) {
        fprintf(stderr, "kvz_hip service workers: pick-up delay histogram (bins of 1 us x 2^b):");
        for (int b = 0; b < 16; ++b) fprintf(stderr, " %lld", svc->dbg_hist[b].load());
        fprintf(stderr, "\n");
      }
      if (svc->dbg_n_pick.load())
        fprintf(stderr, "kvz_hip service workers: post -> last ticket of the request taken, above the smallest seen: %.2f us (mean of %lld requests), request wait %.2f us\n",
                svc->dbg_sum_pick.load() / (double)svc->dbg_n_pick.load() / 1e3, svc->dbg_n_pick.load(),
                svc->st_wait_ns.load() / 1e3 / (double)svc->st_requests.load());
      serve_ring_dev d;
      if (hipMemcpy(&d, svc->wdev, sizeof(d), hipMemcpyDeviceToHost) == hipSuccess && d.units_served)
        fprintf(stderr, "kvz_hip service workers: %llu units, ticket -> unit copied %.2f us, ticket -> results written %.2f us, idle before a ticket %.2f us, units waiting behind a taken ticket %.2f (means), %llu worker launches, workers alive when a request was posted %.1f\n",
                d.units_served, d.fetch_ticks / 100.0 / d.units_served, d.busy_ticks / 100.0 / d.units_served, d.idle_ticks / 100.0 / d.units_served,
                (double)d.backlog / d.units_served, (unsigned long long)svc->st_launches.load(), svc->dbg_alive_sum.load() / (double)(svc->dbg_alive_n.load() ? svc->dbg_alive_n.load() : 1));
    }
  }
  for (int i = 0; i < N_STREAMS; ++i) if (svc->streams[i]) { (void)hipStreamSynchronize(svc->streams[i]); (void)hipStreamDestroy(svc->streams[i]); }
  for (int i = 0; i < N_UPLOAD; ++i) if (svc->up_streams[i]) { (void)hipStreamSynchronize(svc->up_streams[i]); (void)hipStreamDestroy(svc->up_streams[i]); }
  for (int i = 0; i < N_UPLOAD; ++i) if (svc->up_stage[i]) (void)hipHostFree(svc->up_stage[i]);
  if (svc->tables)
    for (int i = 0; i < svc->max_threads; ++i) {
      if (svc->tables[i].stream) { (void)hipStreamSynchronize(svc->tables[i].stream); (void)hipStreamDestroy(svc->tables[i].stream); }
      if (svc->tables[i].buf) (void)hipHostFree(svc->tables[i].buf);
    }
  delete[] svc->tables;
  delete[] svc->slot_batch;
  delete[] svc->batch_open;
  if (svc->planes) (void)hipFree(svc->planes);
  if (svc->areas) (void)hipHostFree(svc->areas);
  if (svc->ring) (void)hipHostFree(svc->ring);
  if (svc->wring) (void)hipHostFree(svc->wring);
  if (svc->ctl) (void)hipHostFree(svc->ctl);
  if (svc->wdev) (void)hipFree(svc->wdev);
  if (svc->dring) (void)hipFree(svc->dring);
  if (svc->dpush) (void)hipFree(svc->dpush);
  delete svc;
}

int kvz_hip_me_service_put_rect(kvz_hip_me_service *svc, int slot, const kvz_hip_pixel *host, uint32_t host_stride, int x, int y, int w, int h)
{
  if (!svc || !host || slot < 0 || slot >= svc->n_slots || x < 0 || y < 0 || w <= 0 || h <= 0 || x + w > svc->w || y + h > svc->h || host_stride < (uint32_t)w)
    return kvzhip::invalid_arg(__func__);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != svc->device) {
    if (hipSetDevice(svc->device) != hipSuccess) { set_error_msg("kvz_hip_me_service_put_rect: hipSetDevice failed"); return KVZ_HIP_ERR_RUNTIME; }
  }
  const int ts = thread_slot(svc);
  const int k = (ts < 0 ? 0 : ts) % N_UPLOAD;
  u8 *dst = svc->planes + (size_t)slot * svc->plane_bytes + (size_t)y * svc->w + x;
  hipError_t e;
  {
    std::lock_guard<std::mutex> lk(svc->up_mu[k]);
    // rows gathered into page-locked memory by the CPU, then ONE DMA: a 2-D copy from pageable memory is staged row by row
    u8 *stage = svc->up_stage[k];
    for (int r = 0; r < h; ++r) std::memcpy(stage + (size_t)r * w, host + (size_t)r * host_stride, (size_t)w);
    e = hipMemcpy2DAsync(dst, (size_t)svc->w, stage, (size_t)w, (size_t)w, (size_t)h, hipMemcpyHostToDevice, svc->up_streams[k]);
    if (e == hipSuccess) e = hipStreamSynchronize(svc->up_streams[k]);
  }
  if (e != hipSuccess) { set_error("kvz_hip_me_service_put_rect", e); return KVZ_HIP_ERR_RUNTIME; }
  svc->st_rects.fetch_add(1, std::memory_order_relaxed);
  svc->st_rect_bytes.fetch_add((uint64_t)w * h, std::memory_order_relaxed);
  return KVZ_HIP_OK;
}

int kvz_hip_me_service_search(kvz_hip_me_service *svc, const kvz_hip_me_request *req, kvz_hip_me_result *results)
{
  if (!svc || !req || !results) return kvzhip::invalid_arg(__func__);
  if (!request_ok(svc, req)) { set_error_msg("kvz_hip_me_service_search: bad request (slots, n_refs, parameters; mv_rdo is not served; the exhaustive search wants pu.x a multiple of 4)"); return KVZ_HIP_ERR_INVALID; }
  if (svc->failed.load()) { set_error_msg("kvz_hip_me_service_search: the service has failed earlier"); return KVZ_HIP_ERR_RUNTIME; }
  const int ts = thread_slot(svc);
  if (ts < 0) { set_error_msg("kvz_hip_me_service_search: more calling threads than max_threads"); return KVZ_HIP_ERR_INVALID; }
  thread_area *area = svc->areas + ts;
  const int n = req->n_refs;
  for (int i = 0; i < n; ++i) __atomic_store_n(&area->res[i].done, 0u, __ATOMIC_RELAXED);
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  const uint64_t t0 = now_ns();
  const bool workers = svc->n_workers > 0;
  if (workers) {
    const int rc = post_to_ring(svc, req, area);
    if (rc != KVZ_HIP_OK) return rc;
    std::atomic_thread_fence(std::memory_order_seq_cst);      // published; now look who is there
  } else {
    std::lock_guard<std::mutex> lk(svc->pend_mu);
    svc->pending.push_back(pending_req{ req, ts });
    svc->n_pending.fetch_add(1, std::memory_order_relaxed);
  }
  svc->st_requests.fetch_add(1, std::memory_order_relaxed);
  int spins = 0;
  for (;;) {
    bool all = true;
    for (int i = 0; i < n; ++i)
      if (__atomic_load_n(&area->res[i].done, __ATOMIC_ACQUIRE) == 0u) { all = false; break; }
    if (all) break;
    if (workers) {
      if ((spins & 31) == 0 || spins >= 256) {
        const int rc = ensure_workers(svc);
        if (rc != KVZ_HIP_OK) return rc;
        if (__atomic_load_n(&svc->ctl->failed, __ATOMIC_RELAXED)) svc->failed.store(1);
      }
    } else
    // With a cap on the batches in the air a request that finds them all taken waits for one to come back and then shares its
    // launch with everything that arrived meanwhile: the device runs only as many kernels side by side as the runtime has hardware
    // queues (four unless GPU_MAX_HW_QUEUES says otherwise), more launches than that queue up behind each other one request at a time.
    if (svc->n_pending.load(std::memory_order_relaxed) > 0 && (svc->inflight_cap <= 0 || svc->inflight.load(std::memory_order_relaxed) < svc->inflight_cap) &&
        svc->launch_mu.try_lock()) {
      const int rc = drain_and_launch(svc);
      svc->launch_mu.unlock();
      if (rc != KVZ_HIP_OK) return rc;
      continue;
    }
    if (svc->failed.load()) { set_error_msg("kvz_hip_me_service_search: a launch failed"); return KVZ_HIP_ERR_RUNTIME; }
    // A search takes tens of microseconds.  Spin through the first ones (a host with a core per worker loses nothing by it);
    // a worker that is still waiting then sleeps in short naps, so that on a host with more workers than cores the core
    // goes to a worker that has CPU work to do instead of to a poll loop.
    const bool crowded = !svc->spin_tuned && svc->next_thread.load(std::memory_order_relaxed) > svc->host_cpus;
    if (++spins < (crowded ? 16 : 256)) {
      __builtin_ia32_pause();
    } else if (now_ns() - t0 < (crowded ? svc->spin_crowded_ns : svc->spin_ns)) {
      sched_yield();
    } else {
      timespec nap = { 0, svc->nap_ns };
      nanosleep(&nap, nullptr);
      if ((spins & 255) == 0 && now_ns() - t0 > WAIT_LIMIT_NS) {
        svc->failed.store(1);
        set_error_msg("kvz_hip_me_service_search: no answer from the device within 20 s");
        return KVZ_HIP_ERR_RUNTIME;
      }
    }
  }
  if (workers) {
    if (svc->debug) {
      // post -> ticket taken, on two clocks: the smallest difference seen stands for "no delay" (it is a PCIe read or two)
      unsigned long long last = 0;
      for (int i = 0; i < n; ++i) { const unsigned long long c = ((unsigned long long)area->res[i].pad[1] << 32) | area->res[i].pad[0]; if (c > last) last = c; }
      const long long d = (long long)(last * 10ull) - (long long)t0;
      // per calling thread, over windows of 128 requests (the two clocks drift apart by microseconds per second)
      static thread_local long long w_min = (long long)1 << 62, w_d[128];
      static thread_local int w_n = 0;
      if (d < w_min) w_min = d;
      w_d[w_n] = d;
      if (++w_n == 128) {
        long long sum = 0;
        for (int i = 0; i < 128; ++i) {
          const long long e = w_d[i] - w_min;
          sum += e;
          int b = 0;
          while (b < 15 && (e >> (10 + b)) > 0) ++b;      // bin 0: < 1.02 us, bin b: < 1.02 us * 2^b
          svc->dbg_hist[b].fetch_add(1, std::memory_order_relaxed);
        }
        svc->dbg_sum_pick.fetch_add(sum, std::memory_order_relaxed);
        svc->dbg_n_pick.fetch_add(128, std::memory_order_relaxed);
        w_min = (long long)1 << 62; w_n = 0;
      }
    }
  }
  if (!workers && svc->batch_open[svc->slot_batch[ts].load(std::memory_order_relaxed)].fetch_sub(1, std::memory_order_release) == 1)
    svc->inflight.fetch_sub(1, std::memory_order_relaxed);          // the last caller of its batch
  svc->st_wait_ns.fetch_add(now_ns() - t0, std::memory_order_relaxed);
  // the sequential rule of search_pu_inter's loop (search_inter.c:1502-1507 with :1239-1252 and :1275-1290)
  uint32_t running = req->cost_to_beat;
  for (int i = 0; i < n; ++i) {
    const serve_result &sr = area->res[i];
    const kvz_hip_me_result &pick = (req->params.fme_level > 0 && sr.integer_search_cost < running) ? sr.frac : sr.integer;
    results[i] = pick;
    if (pick.reserved == -1) { set_error_msg("kvz_hip_me_service_search: malformed PU descriptor"); return KVZ_HIP_ERR_INVALID; }
    if (pick.cost < running) running = pick.cost;
  }
  return KVZ_HIP_OK;
}

// The candidate-independent half of the integer search (include/kvz_hip.h): every SAD check_mv_cost could ask
// kvz_image_calc_sad for inside +-range of a CTU, for every square PU of the CTU and every listed reference picture.
const uint32_t *kvz_hip_me_service_sad_tables(kvz_hip_me_service *svc, int pic_slot, int n_refs, const int32_t *ref_slots,
                                              int ctu_x, int ctu_y, int range)
{
  if (!svc || !ref_slots || n_refs < 1 || n_refs > KVZ_HIP_SERVICE_MAX_REFS || pic_slot < 0 || pic_slot >= svc->n_slots || range < 1 ||
      range > MAX_TABLE_RANGE || ctu_x < 0 || ctu_y < 0 || ctu_x >= svc->w || ctu_y >= svc->h || (ctu_x & 63) || (ctu_y & 63)) {
    kvzhip::invalid_arg(__func__);
    return nullptr;
  }
  for (int i = 0; i < n_refs; ++i)
    if (ref_slots[i] < 0 || ref_slots[i] >= svc->n_slots) { kvzhip::invalid_arg(__func__); return nullptr; }
  const int ts = thread_slot(svc);
  if (ts < 0) { set_error_msg("kvz_hip_me_service_sad_tables: more calling threads than max_threads"); return nullptr; }
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != svc->device) {
    if (hipSetDevice(svc->device) != hipSuccess) { set_error_msg("kvz_hip_me_service_sad_tables: hipSetDevice failed"); return nullptr; }
  }
  const uint64_t t0 = now_ns();
  thread_tables &tt = svc->tables[ts];
  const int side = 2 * range + 1;
  const size_t per_ref = (size_t)side * side * KVZ_HIP_CTU_PUS, need = per_ref * (size_t)n_refs * sizeof(uint32_t);
  if (!tt.stream && hipStreamCreateWithFlags(&tt.stream, hipStreamNonBlocking) != hipSuccess) { set_error("kvz_hip_me_service_sad_tables: hipStreamCreate", hipGetLastError()); return nullptr; }
  if (tt.bytes < need) {
    if (tt.buf) (void)hipHostFree(tt.buf);
    tt.buf = nullptr; tt.bytes = 0;
    if (hipHostMalloc((void **)&tt.buf, need, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { set_error("kvz_hip_me_service_sad_tables: hipHostMalloc", hipGetLastError()); return nullptr; }
    tt.bytes = need;
  }
  thread_area *area = svc->areas + ts;
  for (int k = 0; k < side; ++k) { area->offs[k][0] = (int16_t)(k - range); area->offs[k][1] = 0; }
  for (int i = 0; i < n_refs; ++i) {
    for (int k = 0; k < side; ++k) { kvz_hip_ctu_search &c = area->ctus[i][k]; c.x = ctu_x; c.y = ctu_y; c.mvx = 0; c.mvy = k - range; }
    // one workgroup per row of the window: `side` rows of `side` candidates each, all 85 PU sums per candidate, written
    // straight into the caller's page-locked table
    const int rc = kvz_hip_ctu_sad_grid_batch(svc->planes + (size_t)pic_slot * svc->plane_bytes, (uint32_t)svc->w, svc->w, svc->h,
                                              svc->planes + (size_t)ref_slots[i] * svc->plane_bytes, (uint32_t)svc->w, svc->w, svc->h,
                                              area->ctus[i], (size_t)side, &area->offs[0][0], side, tt.buf + per_ref * (size_t)i, (kvz_hip_stream)tt.stream);
    if (rc != KVZ_HIP_OK) return nullptr;
  }
  const hipError_t e = hipStreamSynchronize(tt.stream);
  if (e != hipSuccess) { set_error("kvz_hip_me_service_sad_tables: hipStreamSynchronize", e); return nullptr; }
  svc->st_tables.fetch_add((uint64_t)n_refs, std::memory_order_relaxed);
  svc->st_table_bytes.fetch_add((uint64_t)need, std::memory_order_relaxed);
  svc->st_table_ns.fetch_add(now_ns() - t0, std::memory_order_relaxed);
  return tt.buf;
}

const kvz_hip_pixel *kvz_hip_me_service_plane(kvz_hip_me_service *svc, int slot)
{
  if (!svc || slot < 0 || slot >= svc->n_slots) { kvzhip::invalid_arg(__func__); return nullptr; }
  return svc->planes + (size_t)slot * svc->plane_bytes;
}

int kvz_hip_me_service_get_stats(kvz_hip_me_service *svc, kvz_hip_me_service_stats *out)
{
  if (!svc || !out) return kvzhip::invalid_arg(__func__);
  out->requests = svc->st_requests.load(); out->units = svc->st_units.load(); out->batches = svc->st_batches.load();
  out->launches = svc->st_launches.load(); out->max_batch_units = svc->st_max_batch.load(); out->rects = svc->st_rects.load();
  out->rect_bytes = svc->st_rect_bytes.load(); out->wait_ns = svc->st_wait_ns.load();
  out->tables = svc->st_tables.load(); out->table_bytes = svc->st_table_bytes.load(); out->table_ns = svc->st_table_ns.load();
  return KVZ_HIP_OK;
}

}  // extern "C"
