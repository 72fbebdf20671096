// dct16_mfma.hip -- 16x16 forward / inverse integer DCT on the matrix cores, two
// blocks per v_mfma_i32_32x32x32_i8.
//
// Reference: src/strategies/generic/dct-generic.c:368-455, :567-597 (N = 16).  The method and the two
// directions live in dct16_mfma_core.h; this file is the streaming kernel around them.
// Same method as dct32_mfma.hip (byte planes X = 256*Xh + Xl' + 128, exact int32
// partial sums, accumulator tile re-used as the next operand).  Two 16x16 blocks
// a, b are stacked into the 32 rows of the MFMA tile:
//   pass 1   D1 = [S_a; S_b] * M16^T             (K: 16 live of 32, dead K operands are 0)
//   pass 2   D2[k][x] = sum_j T'[j][k] * B2[j][x], B2 block diagonal in (block of j, block of x)
// so that lane x of the result holds row (x & 15) of block (x >> 4).  The blocks are
// independent in HBM; a lane loads / stores exactly one 16-byte chunk of the pair's
// 1 KiB (chunk 2r + h: row r, columns 8h .. 8h+7): every global access instruction
// covers a dense 1 KiB, no LDS staging is needed; v_permlane32_swap turns the
// accumulator's column order into contiguous columns before the store.
#include "dct16_mfma_core.h"

using namespace kvzhip;

template <bool INVERSE>
__global__ __launch_bounds__(256, 4) void dct16_mfma_kernel(const i16 *__restrict__ in, i16 *__restrict__ out, size_t count)
{
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const size_t npairs = (count + 1) >> 1;

  dct16_lane k;
  dct16_setup<INVERSE>(r, h, k);
  __shared__ int s_c2[2][8];
  if (INVERSE) {
    dct16_fill_c2(s_c2);
    __syncthreads();
  }

  const int chunk = 2 * r + h;
  // a single trailing block (odd count): the second block's 32 chunks are clamped onto the first
  auto load = [&](size_t p, u32x4v &c) {
    const bool tail = (2 * p + 1 >= count);
    const int ch = (tail && chunk >= 32) ? chunk - 32 : chunk;
    c = __builtin_nontemporal_load((const u32x4v *)(in + p * 512) + ch);
  };

  size_t p = wave;
  constexpr int DEPTH = 5;                             // pairs in flight per wave (1 KiB each)
  u32x4v q[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH - 1; ++i)
    if (p + i * nwaves < npairs) load(p + i * nwaves, q[i]);
  for (; p < npairs; p += nwaves) {
    if (p + (DEPTH - 1) * nwaves < npairs) load(p + (DEPTH - 1) * nwaves, q[DEPTH - 1]);
    int o[8];
    if (!INVERSE) dct16_fwd_pair(q[0], k, o);
    else dct16_inv_pair(q[0], k, s_c2[h], o);
    const u32x4v ov = acc16_to_chunk(o);
    if (2 * p + 1 < count || chunk < 32) __builtin_nontemporal_store(ov, (u32x4v *)(out + p * 512) + chunk);
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) q[i] = q[i + 1];
  }
}

namespace kvzhip {
int launch_dct16_mfma(bool inverse, const i16 *in, i16 *out, size_t count, hipStream_t st)
{
  const size_t npairs = (count + 1) / 2;
  size_t wgs = (npairs + 3) / 4;
  const size_t cap = (size_t)num_cus() * (size_t)tuning(inverse ? "idct16_wgs_per_cu" : "dct16_wgs_per_cu", inverse ? 64 : 64)       /* per-lane constants precomputed: forward 3: 5.52, 32: 5.55, 64: 5.86, 128: 5.70 TB/s */;
  if (wgs > cap) wgs = cap;
  if (inverse) hipLaunchKernelGGL((dct16_mfma_kernel<true>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  else hipLaunchKernelGGL((dct16_mfma_kernel<false>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  KVZ_CHECK_LAUNCH("dct16_mfma_kernel");
  return KVZ_HIP_OK;
}
}  // namespace kvzhip
