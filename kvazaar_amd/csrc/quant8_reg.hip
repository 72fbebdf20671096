// quant8_reg.hip -- fused kvz_quantize_residual for 8x8 TUs with every value in registers; no workgroup barrier.
//
// Reference: src/strategies/generic/quant-generic.c:180-273 (rdoq off, no transform skip, sign hiding off, flat
// scaling -- every other variant stays on quantize_residual_kernel in quant.hip), with the transform pair of
// src/strategies/generic/dct-generic.c:567-587 (partial_butterfly_8 / _inverse_8: shifts 2, 9 and 7, 12).
//
// Mapping: 8 lanes per TU, 8 TUs per wave.  A lane's global accesses are its natural chunk of the TU -- 8 pixels
// (one row) of ref / pred / rec, 8 coefficients (one row, 16 B) of coeff_out -- so a wave reads and writes contiguous
// 512 B / 1 KiB pieces.  The arithmetic is on PACKED int16 pairs:
//   residual row (4 dwords) -> pass 1 in the lane: even/odd butterflies as v_pk_add/sub_i16 (|sum| <= 1020), the
//   products as v_dot2_i32_i16 against packed coefficient pairs (two MACs per instruction, exact in int32)
//   -> 8x8 transpose over the TU's 8 lanes -> pass 2 in the lane = column k of the coefficient block, matrix form:
//   4 dot2 per output (the butterfly's first sums would overflow int16) -> quant -> [transpose -> coeff_out] -> dequant
//   -> inverse pass 1 in the same lane (the reference's first inverse pass runs down the columns) -> transpose ->
//   inverse pass 2 = the lane's row -> + pred, clip -> rec_out.
// The kernel is bound by vector-instruction ISSUE (rocprofv3 PMC, profiles/r02_a_qr_pmc_before.txt: one vector
// instruction per ~4 cycles per SIMD, 70 % of the wave-cycles issue-stalled), so everything is shaped to spend few of them:
//   * the three transposes go through a wave-private LDS tile -- one ds_write_b128 + eight ds_read_u16 per lane,
//     conflict-free (TU stride 144 B), no barrier: LDS issue slots are free here, the ~26 DPP / v_cndmask / v_perm
//     instructions each transpose took in the first version were not;
//   * the dot products are the three-operand v_dot2_i32_i16 with the coefficient pair in an SGPR (the compiler's
//     v_dot2c form needed a v_mov per accumulator chain);
//   * quantisation is signed: level = (c * qc + (c < 0 ? 2^q - 1 - add : add)) >> q, which equals
//     sign(c) * ((|c| * qc + add) >> q) for every c (floor / ceil identity), 4 instructions per coefficient.
// The LDS kernel this replaces (3.1-3.2 TB/s) spent 42 % of its wave-cycles parked at eight barriers per 32 TUs.
// HBM traffic per TU: 64 B ref + 64 B pred + 64 B rec + 128 B coeff (+ 4 B flag) = 5*N*N.
#include "kvz_hip_internal.h"
#include "transform_core.h"

using namespace kvzhip;

namespace {

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2s S2(u32 v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ u32 U2(v2s v) { return __builtin_bit_cast(u32, v); }
__host__ __device__ constexpr u32 pk(int lo, int hi) { return ((u32)lo & 0xffffu) | (((u32)hi & 0xffffu) << 16); }
// a (VGPR) . coefficient pair (compile-time constant, kept in an SGPR) + acc (VGPR): the VOP3P form, no accumulator move
__device__ __forceinline__ int dot2k(u32 a, u32 coef, int acc)
{
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(coef), "v"(acc));
  return d;
}
__device__ __forceinline__ int dot2k0(u32 a, u32 coef)
{
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "s"(coef));
  return d;
}
__device__ __forceinline__ u32 swap16(u32 d) { return __builtin_amdgcn_alignbit(d, d, 16); }
// low halves of two int32 -> one packed dword (the forward passes' truncating (short) cast)
__device__ __forceinline__ u32 pack_lo(int lo, int hi) { return __builtin_amdgcn_perm((u32)hi, (u32)lo, 0x05040100u); }
// two int32 -> packed int16 with saturation = clip16 + pack (the inverse passes), one v_cvt_pk_i16_i32
__device__ __forceinline__ u32 pack_sat(int lo, int hi) { return U2(__builtin_amdgcn_cvt_pk_i16(lo, hi)); }

// 8x8 int16 transpose over the 8 lanes of a TU through the wave's LDS tile.  In: lane r holds row r as
// w[d] = (a[r][2d], a[r][2d+1]).  Out: lane c holds column c as w[m] = (a[2m][c], a[2m+1][c]).
// Tile: TU t at t * 144 bytes (128 + 16 of padding: the eight TUs' 16-byte rows then fall on eight different groups of 4
// banks), row r at + 16 r.  wr = this lane's row, rd = this lane's column.  DS operations of one wave execute in order,
// so the fences only stop the compiler from moving accesses across the phases.
constexpr int TU_STRIDE = 144;
__device__ __forceinline__ void transpose8(u32 (&w)[4], u8 *tile, int wr, int rd)
{
  *(uint4 *)(tile + wr) = make_uint4(w[0], w[1], w[2], w[3]);
  wave_lds_fence();
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const u32 lo = *(const unsigned short *)(tile + rd + 32 * m);
    const u32 hi = *(const unsigned short *)(tile + rd + 32 * m + 16);
    w[m] = lo | (hi << 16);
  }
  wave_lds_fence();
}

constexpr int C8(int k, int i) { return dct_coef(8, k, i); }

// forward pass on 8 packed values whose pairwise sums stay inside int16 (pixel differences): partial_butterfly_8,
// dct-generic.c:275-306.  out[k] = (sum_i M[k][i] x[i] + add) >> shift, untruncated int32
__device__ __forceinline__ void fwd8_small(const u32 (&x)[4], int (&y)[8], int add, int shift)
{
  const v2s x01 = S2(x[0]), x23 = S2(x[1]), x54 = S2(swap16(x[2])), x76 = S2(swap16(x[3]));
  const v2s e01 = x01 + x76, e23 = x23 + x54, o01 = x01 - x76, o23 = x23 - x54;
  const v2s e32 = S2(swap16(U2(e23)));
  const u32 ee = U2(e01 + e32), eo = U2(e01 - e32);    // (e0+e3, e1+e2), (e0-e3, e1-e2)
  y[0] = dot2k(ee, pk(64, 64), add) >> shift;
  y[4] = dot2k(ee, pk(64, -64), add) >> shift;
  y[2] = dot2k(eo, pk(C8(2, 0), C8(2, 1)), add) >> shift;
  y[6] = dot2k(eo, pk(C8(6, 0), C8(6, 1)), add) >> shift;
#pragma unroll
  for (int k = 1; k < 8; k += 2)
    y[k] = dot2k(U2(o01), pk(C8(k, 0), C8(k, 1)), dot2k(U2(o23), pk(C8(k, 2), C8(k, 3)), add)) >> shift;
}
// forward pass, matrix form, on 8 packed int16 of any magnitude
__device__ __forceinline__ void fwd8_full(const u32 (&x)[4], int (&y)[8], int add, int shift)
{
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int acc = add;
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = dot2k(x[m], pk(C8(k, 2 * m), C8(k, 2 * m + 1)), acc);
    y[k] = acc >> shift;
  }
}
// inverse pass (partial_butterfly_inverse_8, dct-generic.c:308-340): x[i] = (sum_k M[k][i] y[k] + add) >> shift, int32
__device__ __forceinline__ void inv8(const u32 (&y)[4], int (&x)[8], int add, int shift)
{
  const u32 y13 = __builtin_amdgcn_perm(y[1], y[0], 0x07060302u);   // (y1, y3)
  const u32 y57 = __builtin_amdgcn_perm(y[3], y[2], 0x07060302u);   // (y5, y7)
  const u32 y04 = __builtin_amdgcn_perm(y[2], y[0], 0x05040100u);   // (y0, y4)
  const u32 y26 = __builtin_amdgcn_perm(y[3], y[1], 0x05040100u);   // (y2, y6)
  const int ee0 = dot2k(y04, pk(64, 64), add), ee1 = dot2k(y04, pk(64, -64), add);
  const int eo0 = dot2k0(y26, pk(C8(2, 0), C8(6, 0))), eo1 = dot2k0(y26, pk(C8(2, 1), C8(6, 1)));
  const int e[4] = { ee0 + eo0, ee1 + eo1, ee1 - eo1, ee0 - eo0 };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = dot2k(y13, pk(C8(1, i), C8(3, i)), dot2k0(y57, pk(C8(5, i), C8(7, i))));
    x[i] = (e[i] + o) >> shift;
    x[7 - i] = (e[i] - o) >> shift;
  }
}

struct q8_consts { int q_bits, add, flat_qc, dq_shift, dq_add, dq_scale; };

template <bool COST, bool PIPE = false>
__global__ __launch_bounds__(256) void quantize_residual8_reg_kernel(const u8 *__restrict__ ref_in, const u8 *pred_in, u8 *rec_out,
                                                                     i16 *__restrict__ coeff_out, i32 *__restrict__ has_coeffs,
                                                                     size_t count, q8_consts k,
                                                                     u32 *__restrict__ ssd_out, u32 *__restrict__ abs_sum_out)
{
  __shared__ __attribute__((aligned(16))) u8 s_tile[4][8 * TU_STRIDE];
  const int lane = threadIdx.x & 63, j = lane & 7;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)wv;
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const size_t ngroups = (count + 7) >> 3;             // 8 TUs per wave step
  u8 *tile = s_tile[wv];
  const int wr = (lane >> 3) * TU_STRIDE + j * 16, rd = (lane >> 3) * TU_STRIDE + j * 2;
  // quant-generic.c:55-67 with the sign carried through: bias for c >= 0 and for c < 0
  const int bias_pos = k.add, bias_neg = (int)((1u << k.q_bits) - 1u) - k.add;

  auto load = [&](size_t g, u32x2v &rv, u32x2v &pv, bool &live) {
    const size_t tu = g * 8 + (size_t)(lane >> 3);
    live = tu < count;
    const size_t t = live ? tu : count - 1;            // lanes of a missing TU mirror the last one and store nothing
    rv = __builtin_nontemporal_load((const u32x2v *)(ref_in + t * 64) + j);
    pv = *((const u32x2v *)(pred_in + t * 64) + j);
  };

  size_t g = wave;
  u32x2v rv, pv, rn, pn;
  bool live = false, live_n = false;
  if (g < ngroups) load(g, rv, pv, live);
  if (PIPE) wait_vmem_all();
  for (; g < ngroups; g += nwaves) {
    const size_t gn = g + nwaves;
    if (gn < ngroups) load(gn, rn, pn, live_n);        // prefetch the wave's next group (never one being written)
    const size_t tu = g * 8 + (size_t)(lane >> 3);

    // residual row, packed: ref - pred
    const u32 p16[4] = { __builtin_amdgcn_perm(0u, pv.x, 0x0c010c00u), __builtin_amdgcn_perm(0u, pv.x, 0x0c030c02u),
                         __builtin_amdgcn_perm(0u, pv.y, 0x0c010c00u), __builtin_amdgcn_perm(0u, pv.y, 0x0c030c02u) };
    u32 w[4] = { U2(S2(__builtin_amdgcn_perm(0u, rv.x, 0x0c010c00u)) - S2(p16[0])), U2(S2(__builtin_amdgcn_perm(0u, rv.x, 0x0c030c02u)) - S2(p16[1])),
                 U2(S2(__builtin_amdgcn_perm(0u, rv.y, 0x0c010c00u)) - S2(p16[2])), U2(S2(__builtin_amdgcn_perm(0u, rv.y, 0x0c030c02u)) - S2(p16[3])) };

    // forward: rows (shift 2), transpose, columns (shift 9); lane j now holds column j of the coefficient block.
    // |coefficient| <= 255 * 512 * 512 / 4 / 512 = 32640: the reference's (short) casts never truncate here.
    int y[8];
    fwd8_small(w, y, 2, 2);
#pragma unroll
    for (int m = 0; m < 4; ++m) w[m] = pack_lo(y[2 * m], y[2 * m + 1]);
    transpose8(w, tile, wr, rd);
    fwd8_full(w, y, 256, 9);

    // quant (quant-generic.c:55-67, flat), signed form; |level| < 2^12, so the clip to int16 never acts
    int lv[8];
    u32 q[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) lv[i] = (__mul24(y[i], k.flat_qc) + (y[i] < 0 ? bias_neg : bias_pos)) >> k.q_bits;
    int any = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) { q[m] = pack_lo(lv[2 * m], lv[2 * m + 1]); any |= (int)q[m]; }
    // has_coeffs: OR over the TU's 8 lanes
    any |= (int)dpp_mov<0xB1>((u32)any);
    any |= (int)dpp_mov<0x4E>((u32)any);
    any |= (int)dpp_mov<0x141>((u32)any);              // row_half_mirror
    const bool has = any != 0;

    u32 sab = 0;
    if (COST) {
#pragma unroll
      for (int i = 0; i < 8; ++i) sab += (u32)(lv[i] < 0 ? -lv[i] : lv[i]);
    }
    u32 qr[4] = { q[0], q[1], q[2], q[3] };
    transpose8(qr, tile, wr, rd);                        // lane j: row j of the quantized block
    // the iteration's one wait on vector memory, just before its first store (wait_vmem_all, kvz_hip_internal.h): the prefetch
    // is half an iteration old, the previous iteration's stores a whole one.  (The compiler's own placement was vmcnt(0) at the
    // loop top, i.e. on the stores the iteration before had issued last.)
    if (PIPE) wait_vmem_all();
    const u32x2v rv_next = rn, pv_next = pn;
    const bool live_next = live_n;
    if (live) {
      const u32x4r qrow = { qr[0], qr[1], qr[2], qr[3] };
      __builtin_nontemporal_store(qrow, (u32x4r *)(coeff_out + tu * 64) + j);
    }

    u32x2v out = pv;                                     // a TU without coefficients keeps its prediction (:262-271)
    if (__ballot(has) != 0ull) {                         // wave-uniform: some TU of the group has coefficients
      // dequant (quant-generic.c:290-320, flat) in the column layout
      u32 dq[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int d0 = (__mul24(lv[2 * m], k.dq_scale) + k.dq_add) >> k.dq_shift;
        const int d1 = (__mul24(lv[2 * m + 1], k.dq_scale) + k.dq_add) >> k.dq_shift;
        dq[m] = pack_sat(d0, d1);
      }
      // inverse: columns (shift 7) in the same lane, transpose, rows (shift 12)
      int x[8];
      inv8(dq, x, 64, 7);
#pragma unroll
      for (int m = 0; m < 4; ++m) w[m] = pack_sat(x[2 * m], x[2 * m + 1]);
      transpose8(w, tile, wr, rd);
      inv8(w, x, 2048, 12);
      // reconstruction: (int16)(residual + pred) clipped to a pixel (quant-generic.c:253-259)
      u32 o16[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const v2s r = S2(pack_sat(x[2 * m], x[2 * m + 1])) + S2(p16[m]);
        const v2s lo = { 0, 0 }, hi = { 255, 255 };
        o16[m] = U2(__builtin_elementwise_min(__builtin_elementwise_max(r, lo), hi));
      }
      if (has) {
        out.x = __builtin_amdgcn_perm(o16[1], o16[0], 0x06040200u);
        out.y = __builtin_amdgcn_perm(o16[3], o16[2], 0x06040200u);
      }
    }
    if (COST) {
      // rd=0 TU cost inputs (search.c:291, rdo.c:219): SSD(ref, rec) and sum |coeff| over the TU's 8 lanes
      const u32 rr[2] = { rv.x, rv.y }, oo[2] = { out.x, out.y };
      int sq2 = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const u32 selb = (m & 1) ? 0x0c030c02u : 0x0c010c00u;
        const u32 d = U2(S2(__builtin_amdgcn_perm(0u, rr[m >> 1], selb)) - S2(__builtin_amdgcn_perm(0u, oo[m >> 1], selb)));
        sq2 = __builtin_amdgcn_sdot2(S2(d), S2(d), sq2, false);
      }
      const u32 ssd = group_sum<8>((u32)sq2), asum = group_sum<8>(sab);
      if (live && j == 0) { ssd_out[tu] = ssd; abs_sum_out[tu] = asum; }
    }
    if (live) {
      *((u32x2v *)(rec_out + tu * 64) + j) = out;
      if (j == 0) has_coeffs[tu] = has ? 1 : 0;
    }
    rv = rv_next; pv = pv_next; live = live_next;
  }
}

}  // namespace

namespace kvzhip {
// flat quantisation only (quant.hip routes scaling lists, sign hiding and transform skip to the LDS kernel)
int launch_quantize_residual8_reg(const u8 *ref_in, const u8 *pred_in, u8 *rec_out, i16 *coeff_out, i32 *has_coeffs, size_t count,
                                  int q_bits, int add, int flat_qc, int dq_shift, int dq_add, int dq_scale,
                                  u32 *ssd_out, u32 *abs_sum_out, hipStream_t st)
{
  const q8_consts k = { q_bits, add, flat_qc, dq_shift, dq_add, dq_scale };
  const size_t ngroups = (count + 7) / 8;
  size_t wgs = (ngroups + 3) / 4;                       // 4 waves per workgroup, one group of 8 TUs per wave step
  const size_t cap = (size_t)num_cus() * (size_t)tuning("qr8_wgs_per_cu", 32);   // measured (0.5 GiB operands): 32: 4.95, 64: 4.84, 128: 4.79, 192: 4.67 TB/s
  if (wgs > cap) wgs = cap;
  // "pipe" 1: the iteration's wait on vector memory placed by hand before its first store (wait_vmem_all).  Measured A/B on one
  // box: 4.59 TB/s with it, 4.73 without -- at this kernel's 8 waves per SIMD the wave interleaving hides the latency by itself;
  // the matrix-core tile kernels (4 waves per SIMD) gain 10-15 % from the same placement.
  if (!ssd_out && tuning("pipe", 0))
    hipLaunchKernelGGL((quantize_residual8_reg_kernel<false, true>), dim3((unsigned)wgs), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out,
                       has_coeffs, count, k, ssd_out, abs_sum_out);
  else if (ssd_out)
    hipLaunchKernelGGL(quantize_residual8_reg_kernel<true>, dim3((unsigned)wgs), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out,
                       has_coeffs, count, k, ssd_out, abs_sum_out);
  else
    hipLaunchKernelGGL(quantize_residual8_reg_kernel<false>, dim3((unsigned)wgs), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out,
                       has_coeffs, count, k, ssd_out, abs_sum_out);
  KVZ_CHECK_LAUNCH("quantize_residual8_reg_kernel");
  return KVZ_HIP_OK;
}
}  // namespace kvzhip
