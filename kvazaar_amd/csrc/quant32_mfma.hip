// quant32_mfma.hip -- fused kvz_quantize_residual for 32x32 TUs on the matrix cores.
//
// Reference: src/strategies/generic/quant-generic.c:180-273 (rdoq off, no transform
// skip, sign hiding off -- the other variants stay on quantize_residual_kernel in
// quant.hip).  One wave per TU, no barrier:
//   ref/pred rows (16 B per lane, lane (r, h) = row r, columns 16h..16h+15)
//   -> v_permlane32_swap into the accumulator's column order kappa(h, .)
//   -> residual (packed int16) -> byte planes -> forward DCT (4 MFMA)
//   -> quant (registers) -> coeff_out through the wave's LDS tile, coalesced
//   -> if any coefficient: dequant -> inverse DCT (6 MFMA) -> + pred, clip
//   -> permlane32 swap back -> rec_out, 16 B per lane.
// HBM traffic per TU: 1 KiB ref + 1 KiB pred + 1 KiB rec + 2 KiB coeff = 5*N*N.
#include "dct32_mfma_core.h"

using namespace kvzhip;

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

struct q32_consts {
  int q_bits, add, flat_qc;
  const int32_t *qtable;
  int dq_mode, dq_shift, dq_add, dq_scale;
  const int32_t *dqtable;
};

// natural (lane (r,h): columns 16h .. 16h+15 as 4 dwords) <-> kappa order (dword q = columns 8q+4h .. +3).
// An involution: swap the upper-half lanes of a[0]/a[2] with the lower-half lanes of a[1]/a[3].
__device__ __forceinline__ void kappa_swap(u32 (&a)[4])
{
  u32x2v p = __builtin_amdgcn_permlane32_swap(a[0], a[1], false, false);
  u32x2v q = __builtin_amdgcn_permlane32_swap(a[2], a[3], false, false);
  // p = (a0' , a1') with a0' = (d0 | d1), a1' = (e0 | e1);  kappa order: [d0|d1], [d2|d3], [e0|e1], [e2|e3]
  a[0] = p.x; a[2] = p.y; a[1] = q.x; a[3] = q.y;
}
__device__ __forceinline__ void kappa_unswap(u32 (&a)[4])
{
  u32x2v p = __builtin_amdgcn_permlane32_swap(a[0], a[2], false, false);
  u32x2v q = __builtin_amdgcn_permlane32_swap(a[1], a[3], false, false);
  a[0] = p.x; a[1] = p.y; a[2] = q.x; a[3] = q.y;
}

__device__ __forceinline__ int q32_quant(int c, int n, const q32_consts &k)
{
  const int a = c < 0 ? -c : c;
  int level;
  if (k.qtable) level = (int)(((long long)a * k.qtable[n] + k.add) >> k.q_bits);
  else level = (int)((__umul24((u32)a, (u32)k.flat_qc) + (u32)k.add) >> k.q_bits);      // < 2^31: |c| <= 2^15, qc < 2^15, add < 2^23
  level = c < 0 ? -level : level;
  return clip16(level);
}
__device__ __forceinline__ int q32_dequant(int q, int n, const q32_consts &k)
{
  if (k.dq_mode == 0) return clip16((int)((u32)__mul24(q, k.dq_scale) + (u32)k.dq_add) >> k.dq_shift);
  const int d = k.dqtable[n];
  if (k.dq_mode == 1) return clip16((q * d + k.dq_add) >> k.dq_shift);
  return clip16((int)((u32)clip16(q * d) << k.dq_shift));
}

__global__ __launch_bounds__(256, 3) void quantize_residual32_mfma_kernel(const u8 *__restrict__ ref_in, const u8 *pred_in, u8 *rec_out,
                                                                           i16 *__restrict__ coeff_out, i32 *__restrict__ has_coeffs,
                                                                           size_t count, q32_consts k,
                                                                           u32 *__restrict__ ssd_out, u32 *__restrict__ abs_sum_out)
{
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  // constant operands: one precomputed record per lane (dct32_mfma_core.h)
  op16 t_kap, t_col, t_idk;
  const dct32_lane_consts &lc = c_dct32_lanes.l[lane];
#pragma unroll
  for (int q = 0; q < 4; ++q) { t_kap.w[q] = lc.t_kap[q]; t_col.w[q] = lc.t_col[q]; t_idk.w[q] = lc.t_idk[q]; }
  const int rowsum = lc.rowsum, colsum = lc.colsum;
  __shared__ __attribute__((aligned(16))) u8 s_tile[4][2048];
  __shared__ __attribute__((aligned(16))) int s_c2[2][16];
  u8 *tile = s_tile[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
  fill_inv_c2(s_c2);
  __syncthreads();

  const int chunk = 2 * r + h;                       // 16-byte chunk of the TU's 1 KiB pixel block owned by this lane
  size_t t = wave;
  u32x4v rv, pv, rn, pn;
  if (t < count) {
    rv = __builtin_nontemporal_load((const u32x4v *)(ref_in + t * 1024) + chunk);
    pv = *((const u32x4v *)(pred_in + t * 1024) + chunk);
  }
  for (; t < count; t += nwaves) {
    const size_t tn = t + nwaves;
    if (tn < count) {                                // prefetch the wave's next TU (never the one being written: tn != t)
      rn = __builtin_nontemporal_load((const u32x4v *)(ref_in + tn * 1024) + chunk);
      pn = *((const u32x4v *)(pred_in + tn * 1024) + chunk);
    }
    u32 rf[4] = { rv.x, rv.y, rv.z, rv.w }, pr[4] = { pv.x, pv.y, pv.z, pv.w };
    kappa_swap(rf);
    kappa_swap(pr);
    // residual, element e = 4q + i <-> column kappa(h, e); packed int16 pairs
    u32 d[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v2s a0 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, rf[q], 0x0c010c00u)), a1 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, rf[q], 0x0c030c02u));
      const v2s b0 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, pr[q], 0x0c010c00u)), b1 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, pr[q], 0x0c030c02u));
      d[2 * q] = __builtin_bit_cast(u32, a0 - b0);
      d[2 * q + 1] = __builtin_bit_cast(u32, a1 - b1);
    }
    op16 hi, lo;
    planes_from_rows(d, hi, lo);
    int c[16];
    fwd32_core(hi, lo, t_kap, t_kap, rowsum, c);
    int qv[16], any = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      qv[g] = q32_quant((int)(short)c[g], r * 32 + kappa(h, g), k);
      any |= qv[g];
    }
    const bool has = __ballot(any != 0) != 0ull;
    rows_to_chunks_store(tile, lane, r, h, qv, coeff_out + t * 1024);
    u32 out[4] = { pr[0], pr[1], pr[2], pr[3] };
    if (has) {                                       // wave-uniform
      int dq[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) dq[g] = q32_dequant(qv[g], r * 32 + kappa(h, g), k);
      op16 h2, l2;
      planes_from_regs(dq, h2, l2, 0x80808080u);
      int res[16];
      inv32_core(h2, l2, t_idk, t_col, colsum, s_c2[h], res);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32 w = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = (int)((pr[q] >> (8 * i)) & 255u);
          const int val = (int)(short)(res[4 * q + i] + p);           // int16_t val = residual + pred (quant-generic.c:255)
          const int px = val < 0 ? 0 : (val > 255 ? 255 : val);
          w |= (u32)px << (8 * i);
        }
        out[q] = w;
      }
    }
    if (ssd_out) {
      // rd=0 TU cost inputs from the registers: kvz_pixels_calc_ssd(ref, rec) (search.c:291; rf and out hold the
      // same 16 pixels in the same order) and kvz_coeff_abs_sum (rdo.c:219)
      u32 sq2 = 0, sab = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int dd = (int)((rf[q] >> (8 * i)) & 255u) - (int)((out[q] >> (8 * i)) & 255u);
          sq2 += (u32)(dd * dd);
        }
#pragma unroll
      for (int g = 0; g < 16; ++g) sab += (u32)(qv[g] < 0 ? -qv[g] : qv[g]);
      sq2 = group_sum<64>(sq2);
      sab = group_sum<64>(sab);
      if (lane == 0) { ssd_out[t] = sq2; abs_sum_out[t] = sab; }
    }
    kappa_unswap(out);
    u32x4v ov = { out[0], out[1], out[2], out[3] };
    *((u32x4v *)(rec_out + t * 1024) + chunk) = ov;
    if (lane == 0) has_coeffs[t] = has ? 1 : 0;
    rv = rn; pv = pn;
  }
}

namespace kvzhip {
// consts are produced by quant.hip's make_consts (same field meaning)
int launch_quantize_residual32_mfma(const u8 *ref_in, const u8 *pred_in, u8 *rec_out, i16 *coeff_out, i32 *has_coeffs, size_t count,
                                    int q_bits, int add, int flat_qc, const int32_t *qtable, int dq_mode, int dq_shift, int dq_add,
                                    int dq_scale, const int32_t *dqtable, u32 *ssd_out, u32 *abs_sum_out, hipStream_t st)
{
  q32_consts k = { q_bits, add, flat_qc, qtable, dq_mode, dq_shift, dq_add, dq_scale, dqtable };
  size_t wgs = (count + 3) / 4;
  const size_t cap = (size_t)num_cus() * (size_t)tuning("qr32_wgs_per_cu", 16)       /* with the per-lane constants precomputed: 3: 3.44, 6: 3.74, 12: 3.93, 32: 3.90, 64: 3.75 TB/s */;
  if (wgs > cap) wgs = cap;
  hipLaunchKernelGGL(quantize_residual32_mfma_kernel, dim3((unsigned)wgs), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, k, ssd_out, abs_sum_out);
  KVZ_CHECK_LAUNCH("quantize_residual32_mfma_kernel");
  return KVZ_HIP_OK;
}
}  // namespace kvzhip
