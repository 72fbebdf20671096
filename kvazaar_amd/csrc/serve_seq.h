// serve_seq.h -- the sequence word of a ring slot (kvz_hip_internal.h: serve_slot.seq), shared by the host side (serve.hip) and the
// resident workers (me_search.hip).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace kvzhip {

// serve_slot.seq of the unit with this ticket: never 0 (0 = "free: a worker has copied the unit"), also when the 64-bit ticket count
// passes a multiple of 2^32 -- at a million units a second that is every 71 minutes.
__host__ __device__ inline uint32_t serve_seq(unsigned long long ticket) { return (uint32_t)(ticket % 0xFFFFFFFFull) + 1u; }

}  // namespace kvzhip
