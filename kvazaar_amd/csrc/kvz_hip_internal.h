// Internal declarations shared by the HIP translation units of libkvzhip.so.
// gfx950 (MI355X, CDNA4) only: wave64, 256 CUs, 160 KiB LDS/CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/kvz_hip.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef int16_t i16;
typedef int32_t i32;

namespace kvzhip {

// ---- context (api.hip) ----
int ctx_device();                               // the calling thread's current device (-1: none initialised)
bool ctx_ready();
bool ctx_enter();                               // ready, and the calling thread's HIP device = its kvz_hip device
hipStream_t ctx_stream(kvz_hip_stream s);      // NULL -> library default stream
void set_error(const char *what, hipError_t e);
void set_error_msg(const char *what);
int invalid_arg(const char *entry);             // records "<entry>: invalid argument" and returns KVZ_HIP_ERR_INVALID
int num_cus();
int tuning(const char *key, int dflt);     // kvz_hip_set_tuning override or dflt

#define KVZ_CHECK_CTX()                         \
  do {                                          \
    if (!kvzhip::ctx_enter()) {                 \
      if (kvz_hip_init(-1) != KVZ_HIP_OK || !kvzhip::ctx_enter()) return KVZ_HIP_ERR_NO_DEVICE; \
    }                                           \
  } while (0)

#define KVZ_CHECK_LAUNCH(name)                               \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) {                                 \
      kvzhip::set_error(name, e__);                          \
      return KVZ_HIP_ERR_RUNTIME;                            \
    }                                                        \
  } while (0)

// ---- search service (serve.hip <-> me_search.hip) ----
// One (PU, reference picture) unit of a batch, as the kernels read it from page-locked host memory.
struct serve_unit {                       // 176 bytes
  int32_t pic_slot, ref_slot;
  void *result;                           // this unit's serve_result (page-locked host memory)
  kvz_hip_me_pu pu;
  kvz_hip_me_params prm;
};
struct serve_result {                     // 80 bytes
  kvz_hip_me_result frac;                 // integer + fractional search: search_pu_inter_ref when info->best_cost < *inter_cost
  kvz_hip_me_result integer;              // otherwise: the integer vector with its SATD cost (search_inter.c:1242-1252)
  uint32_t integer_search_cost;           // info->best_cost after the integer search, what :1239 compares with *inter_cost
  uint32_t done;                          // written last
  uint32_t pad[2];
};
// Resident workers (the service's other way to the device; serve.hip explains the protocol): units travel through a ring of slots in
// page-locked host memory, workgroups that stay on the device take them by ticket.
constexpr int SERVE_MAX_WORKERS = 256;
struct serve_slot {                       // 192 bytes
  serve_unit u;
  uint32_t seq;                           // ticket + 1 once the unit is written (host), 0 once a worker has copied it (device)
  uint32_t pad[3];
};
struct serve_ring_ctl {                   // page-locked host memory
  unsigned long long tail;                // units published so far (host writes, workers read across PCIe)
  uint32_t quit;                          // host: every worker leaves when it next finds no work
  uint32_t failed;                        // a worker gave up waiting for a slot to be written
  uint32_t pad[12];
  uint32_t alive[SERVE_MAX_WORKERS];      // worker w: 0 = gone (or going and no longer counted), else launched / running
};
struct serve_ring_dev {                   // device memory
  unsigned long long head;                // next ticket to take
  unsigned long long tail;                // the workers' copy of ctl->tail
  unsigned long long reserved0;
  unsigned long long last_claim;          // wall clock of the last ticket taken by anybody: the workers leave together when the service falls idle
  unsigned long long pad[4];
  // what the workers measured (10 ns ticks, summed over units): ticket -> unit copied, ticket -> results written; units served
  unsigned long long fetch_ticks, busy_ticks, units_served, backlog, idle_ticks, pad2[3];   // backlog: units published and not taken, summed at every ticket
};
struct serve_worker_ids { unsigned char id[SERVE_MAX_WORKERS]; };
// starts `count` workers (512 threads each) named ids.id[0 .. count); they leave after linger_ticks without work, or at the first
// idle moment after life_ticks (wall-clock ticks of 10 ns), or when ctl->quit is set
int serve_workers_launch(const u8 *planes, size_t plane_bytes, int n_slots, u32 stride, int w, int h, serve_slot *ring, u32 ring_mask,
                         serve_ring_ctl *ctl, serve_ring_dev *dev, const serve_worker_ids &ids, int count,
                         unsigned long long linger_ticks, unsigned long long life_ticks, unsigned long long poll_period_ticks, hipStream_t st);
// `constrained`: some fracmv_within_tile rule is active in the batch (wpp_owf or an mv_constraint)
int serve_launch(bool constrained, const u8 *planes, size_t plane_bytes, int n_slots, u32 stride, int w, int h,
                 const serve_unit *units, int count, hipStream_t st);

// Grid sizing for streaming kernels: enough workgroups to fill 256 CUs several
// times over, capped so that grid-stride loops amortise the launch.
static inline unsigned stream_grid(size_t work_items, unsigned items_per_block, unsigned max_blocks_per_cu = 128)
{
  size_t need = (work_items + items_per_block - 1) / items_per_block;
  size_t cap = (size_t)num_cus() * max_blocks_per_cu;
  if (need < 1) need = 1;
  return (unsigned)(need < cap ? need : cap);
}

// ---- device helpers ----
#if defined(__HIPCC__)

// DPP cross-lane moves (no LDS traffic).  quad_perm / row_half_mirror / row_mirror.
template <int CTRL>
__device__ __forceinline__ u32 dpp_mov(u32 v)
{
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// Sum over aligned groups of L consecutive lanes (L power of two <= 64); every
// lane of the group ends up with the group's sum.
template <int L>
__device__ __forceinline__ u32 group_sum(u32 v)
{
  if (L >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  if (L >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  if (L >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
  if (L >= 16) v += dpp_mov<0x140>(v);  // row_mirror
  if (L >= 32) v += (u32)__shfl_xor((int)v, 16, 64);
  if (L >= 64) v += (u32)__shfl_xor((int)v, 32, 64);
  return v;
}

// Ordering point between two phases of ONE wave that exchange data through its private LDS slice.
// The hardware executes a wave's DS operations in order, so no s_barrier is needed; this only stops
// the compiler from moving LDS accesses across the phase boundary.
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Streaming (touched once) 16-byte global accesses.  The nontemporal policy keeps once-used lines from displacing
// useful ones: measured on MI355X (tools/bw_probe.hip) a copy-shaped kernel moves 6.0-6.4 TB/s with both hints
// against 5.5-5.75 TB/s without, a read-only stream 6.9-7.0 against 6.3.
typedef unsigned int kvz_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream_u4(const void *p)
{
  const kvz_u32x4 v = __builtin_nontemporal_load((const kvz_u32x4 *)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream_u4(void *p, uint4 v)
{
  const kvz_u32x4 w = { v.x, v.y, v.z, v.w };
  __builtin_nontemporal_store(w, (kvz_u32x4 *)p);
}

// Every outstanding vector-memory operation of the wave has completed; nothing is scheduled across it.
// Why the streaming kernels place this by hand: gfx950 counts vector loads AND stores in one counter (vmcnt), and loads and
// stores complete out of order with each other -- once a store is in flight, "wait for that earlier load" can only be
// expressed as vmcnt(0), which also waits for the stores and for any younger prefetch.  Left to the compiler, a
// load-compute-store loop with a prefetch waits at its top for the stores it has just issued (or even for the prefetch it
// has just issued).  The kernels put the iteration's one vmcnt(0) right BEFORE its first store instead: what is
// outstanding there is the prefetch, issued a stretch of arithmetic earlier, and the previous iteration's stores.
__device__ __forceinline__ void wait_vmem_all()
{
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0); expcnt and lgkmcnt left alone (gfx9 encoding)
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// kvz_fast_clip_16bit_to_pixel (picture-generic.c:30-48): int16 argument,
// any bit outside 0..255 => low byte of (-v >> 15).
__device__ __forceinline__ u8 fast_clip16(i16 v)
{
  int x = v;
  return (x & ~255) ? (u8)((-x) >> 15) : (u8)x;
}
// kvz_fast_clip_32bit_to_pixel (picture-generic.c:52-70)
__device__ __forceinline__ u8 fast_clip32(i32 v)
{
  return (v & ~255) ? (u8)(((i32)(0u - (u32)v)) >> 31) : (u8)v;
}

#endif  // __HIPCC__

}  // namespace kvzhip
