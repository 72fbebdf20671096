// dct.hip -- batched 4/8/16/32 integer DCT / IDCT and the 4x4 DST for gfx950.
// Reference: src/strategies/generic/dct-generic.c:567-617 (dct_func typedef:
// src/strategies/strategies-dct.h:31).
//
// Layout: `count` contiguous row-major N x N int16 blocks in, same out.
// One 256-thread workgroup transforms 256/N blocks per iteration: blocks are
// staged into LDS with coalesced 16-byte loads, N threads per block run the two
// 1-D passes in registers (exact int32 even/odd butterflies, transform_core.h)
// exchanging the intermediate through LDS, and the result leaves with coalesced
// 16-byte stores.  HBM traffic = 4*N*N bytes per block, the algorithmic minimum.
#include "kvz_hip_internal.h"
#include "transform_core.h"

#include <cstdlib>

using namespace kvzhip;

template <int N, int KIND>
__global__ __launch_bounds__(256) void transform_kernel(const i16 *__restrict__ in, i16 *__restrict__ out, size_t count)
{
  constexpr int TPB = 256 / N;                       // blocks (TUs) per workgroup iteration
  constexpr int LD = lds_tile_ld(N);                 // LDS row stride (int16): an odd number of dwords, bank-conflict free
  constexpr int CPB = N * N / 8;                     // 16-byte chunks per block
  constexpr int CHUNKS = TPB * CPB;                  // per iteration
  __shared__ __attribute__((aligned(16))) i16 sa[TPB * N * LD];
  __shared__ __attribute__((aligned(16))) i16 sb[TPB * N * LD];

  const int tid = threadIdx.x;
  const int tu = tid / N, row = tid % N;
  const size_t ngroups = (count + TPB - 1) / TPB;

  for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const size_t first = g * TPB;
    // stage in
#pragma unroll
    for (int c = tid; c < CHUNKS; c += 256) {
      const int t = c / CPB, e = (c % CPB) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (first + t < count) v = ld_stream_u4(in + (first + t) * (size_t)(N * N) + e);
      lds_tile_store8<N, LD>(sa + t * N * LD, e, v);
    }
    __syncthreads();
    transform_2d_lds<N, KIND, LD>(sa + tu * N * LD, sb + tu * N * LD, row);
    __syncthreads();
    // stage out
#pragma unroll
    for (int c = tid; c < CHUNKS; c += 256) {
      const int t = c / CPB, e = (c % CPB) * 8;
      if (first + t < count) {
        const uint4 v = lds_tile_load8<N, LD>(sa + t * N * LD, e);
        st_stream_u4(out + (first + t) * (size_t)(N * N) + e, v);
      }
    }
    __syncthreads();
  }
}

namespace kvzhip {
int launch_dct32_mfma(bool inverse, const i16 *in, i16 *out, size_t count, hipStream_t st);
int launch_dct16_tile(bool inverse, const i16 *in, i16 *out, size_t count, hipStream_t st);
int launch_dct4_tile(bool inverse, bool dst, const i16 *in, i16 *out, size_t count, hipStream_t st);
}

// KVZ_HIP_DCT32_VALU=1 selects the VALU/LDS butterfly kernel for 32x32 and 16x16 (A/B comparison only)
static bool dct32_use_valu()
{
  static const bool v = [] { const char *e = getenv("KVZ_HIP_DCT32_VALU"); return e && e[0] == '1'; }();
  return v;
}

// kvz_transformskip / kvz_itransformskip (transform.c:150-180): coeff = (int16)(block << shift);
// block = (int16)((coeff + (1 << (shift - 1))) >> shift), shift = 15 - 8 - log2(n).  8 values (16 bytes) per lane.
template <bool INV>
__global__ __launch_bounds__(256) void transform_skip_kernel(const i16 *__restrict__ in, i16 *__restrict__ out, size_t total, int shift)
{
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
  const int offset = 1 << (shift - 1);
  for (size_t i = tid * 8; i < total; i += nthreads * 8) {
    union { uint4 v; i16 s[8]; } a, b;
    a.v = ld_stream_u4(in + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) b.s[j] = INV ? (i16)(((int)a.s[j] + offset) >> shift) : (i16)((int)a.s[j] << shift);
    st_stream_u4(out + i, b.v);
  }
}

template <int N, int KIND>
static int launch_transform(const i16 *in, i16 *out, size_t count, hipStream_t st)
{
  constexpr int TPB = 256 / N;
  // workgroups per CU of the grid-stride launch, measured per size (0.5 GiB arrays): 4x4 is best at <= 48 (5.8 TB/s, 5.2 at 96),
  // 8x8 keeps gaining up to ~192 (6.5 TB/s against 5.8 at 16), the 16x16 inverse peaks around 48
  const unsigned grid = stream_grid(count, TPB, (unsigned)tuning("dct_wgs_per_cu", N == 8 ? 160 : N == 16 ? 48 : 16));
  hipLaunchKernelGGL((transform_kernel<N, KIND>), dim3(grid), dim3(256), 0, st, in, out, count);
  KVZ_CHECK_LAUNCH("transform_kernel");
  return KVZ_HIP_OK;
}

extern "C" int kvz_hip_transform_batch(int kind, int n, const int16_t *in, int16_t *out, size_t count, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!in || !out || (((uintptr_t)in | (uintptr_t)out) & 15)) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  hipStream_t st = ctx_stream(s);
  switch (kind) {
    case KVZ_HIP_DCT:
      switch (n) {
        case 4: return tuning("dct4_tile", 1) ? launch_dct4_tile(false, false, in, out, count, st) : launch_transform<4, 0>(in, out, count, st);
        case 8: return launch_transform<8, 0>(in, out, count, st);
        case 16: return dct32_use_valu() ? launch_transform<16, 0>(in, out, count, st) : launch_dct16_tile(false, in, out, count, st);
        case 32: return dct32_use_valu() ? launch_transform<32, 0>(in, out, count, st) : launch_dct32_mfma(false, in, out, count, st);
      }
      break;
    case KVZ_HIP_IDCT:
      switch (n) {
        case 4: return tuning("dct4_tile", 1) ? launch_dct4_tile(true, false, in, out, count, st) : launch_transform<4, 1>(in, out, count, st);
        case 8: return launch_transform<8, 1>(in, out, count, st);
        // four blocks per MFMA tile: 6.1-6.2 TB/s against 5.2 for the butterflies and less for the earlier two-blocks-per-tile MFMA kernel
        case 16: return dct32_use_valu() ? launch_transform<16, 1>(in, out, count, st) : launch_dct16_tile(true, in, out, count, st);
        case 32: return dct32_use_valu() ? launch_transform<32, 1>(in, out, count, st) : launch_dct32_mfma(true, in, out, count, st);
      }
      break;
    case KVZ_HIP_DST:
      if (n == 4) return tuning("dct4_tile", 1) ? launch_dct4_tile(false, true, in, out, count, st) : launch_transform<4, 2>(in, out, count, st);
      break;
    case KVZ_HIP_IDST:
      if (n == 4) return tuning("dct4_tile", 1) ? launch_dct4_tile(true, true, in, out, count, st) : launch_transform<4, 3>(in, out, count, st);
      break;
    case KVZ_HIP_TRSKIP:
    case KVZ_HIP_ITRSKIP:
      if (n == 4 || n == 8 || n == 16 || n == 32) {
        const size_t total = count * (size_t)n * n;
        const int shift = 15 - 8 - (n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5);
        const unsigned grid = stream_grid(total, 2048, 256);
        if (kind == KVZ_HIP_TRSKIP) hipLaunchKernelGGL(transform_skip_kernel<false>, dim3(grid), dim3(256), 0, st, in, out, total, shift);
        else hipLaunchKernelGGL(transform_skip_kernel<true>, dim3(grid), dim3(256), 0, st, in, out, total, shift);
        KVZ_CHECK_LAUNCH("transform_skip_kernel");
        return KVZ_HIP_OK;
      }
      break;
  }
  return kvzhip::invalid_arg(__func__);
}
