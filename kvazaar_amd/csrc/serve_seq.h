// serve_seq.h -- the sequence word of a ring slot (kvz_hip_internal.h: serve_slot.seq), shared by the host side (serve.hip) and the
// resident workers (me_search.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace kvzhip {

// serve_slot.seq of the unit with this ticket: never 0 (0 = "free: a worker has copied the unit"), also when the 64-bit ticket count
// passes a multiple of 2^32 -- at a million units a second that is every 71 minutes.
__host__ __device__ inline uint32_t serve_seq(unsigned long long ticket) { return (uint32_t)(ticket % 0xFFFFFFFFull) + 1u; }

// "Push" mode (a device whose memory the host can write through a large PCIe BAR): the host writes the units and the published
// ticket count straight into fine-grained DEVICE memory, so a resident worker never reads host memory on the request path (a PCIe read
// is ~2 us, and there were two of them per unit).  The slot's sequence word in HOST memory stays the "slot taken / free" handshake, and
// the ticket count in host memory (serve_ring_ctl.tail) stays what a LEAVING worker looks at last -- that look must be a PCIe read
// behind its "gone" store (serve.hip, "who makes sure somebody is there").
struct serve_push {                       // fine-grained device memory, written by the host through the BAR
  unsigned long long tail;                // = serve_ring_ctl.tail
  uint32_t quit;                          // = serve_ring_ctl.quit
  uint32_t pad[13];
};

struct serve_slot;
struct serve_ring_ctl;
struct serve_ring_dev;
struct serve_worker_ids;
// serve_workers_launch (kvz_hip_internal.h) with the ring the workers READ units from (`ring`: the host ring, or its copy in device
// memory), the host ring whose sequence words they clear, and the push block (nullptr: they poll serve_ring_ctl in host memory)
int serve_workers_launch_push(const unsigned char *planes, size_t plane_bytes, int n_slots, unsigned stride, int w, int h, serve_slot *ring,
                              serve_slot *host_ring, const serve_push *push, unsigned ring_mask, serve_ring_ctl *ctl, serve_ring_dev *dev,
                              const serve_worker_ids &ids, int count, unsigned long long linger_ticks, unsigned long long life_ticks,
                              unsigned long long poll_period_ticks, hipStream_t st);

}  // namespace kvzhip
