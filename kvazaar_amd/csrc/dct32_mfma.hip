// dct32_mfma.hip -- 32x32 forward / inverse integer DCT on the CDNA4 matrix cores.
//
// Reference: src/strategies/generic/dct-generic.c:458-597 (partial_butterfly_32 /
// partial_butterfly_inverse_32, DCT_NXN_GENERIC / IDCT_NXN_GENERIC).  Each 1-D pass
// is an exact integer matrix product with the 32x32 coefficient matrix M
// (|M| <= 90, fits int8).  The int16 operand X is split into byte planes,
//     X = 256 * Xh + Xl' + 128,   Xh = X >> 8,  Xl' = (X & 255) - 128  (both int8),
// so that   M*X = 256 * (M*Xh) + (M*Xl') + 128 * (M*1)
// is two v_mfma_i32_32x32x32_i8 per pass plus a per-row constant; every partial
// sum is exact in int32, so the results are bit-identical to the butterflies.
// rocprof showed the VALU/LDS butterfly kernel (dct.hip) at 3.3 TB/s, VALU- and
// LDS-issue bound; the MFMA form leaves both nearly idle and runs at HBM speed.
//
// One wave owns one 32x32 block at a time and needs no barrier:
//   * HBM is touched only with fully coalesced 16-byte-per-lane accesses (1 KiB per
//     wave instruction); a 2 KiB wave-private LDS tile (XOR-swizzled 16-byte slots,
//     conflict-free for ds_write_b128 / ds_read_b128 / ds_write_b64) turns the
//     linear chunks into the A-operand layout (lane (r, h) = row r, columns
//     16h..16h+15) and the row-per-lane result back into linear chunks.  Direct
//     operand-shaped global accesses (16 B at a 64 B stride) measured 4.0 TB/s,
//     the staged form reaches the streaming rate;
//   * the 32x32 i32 accumulator tile of pass 1 (column on the lane, 16 rows in
//     registers) is re-used in place as the operand of pass 2, whose contraction
//     index is exactly the accumulator's row index -- no data movement;
//   * the operand orientation of pass 2 is chosen so that each lane ends up with
//     one output ROW in 4 runs of 4 consecutive int16, stored as 4 x 8 bytes.
// K-index permutation: an MFMA computes sum_k A[i][k]*B[k][j] for whatever
// bijection maps (lane half h, element e) to k, as long as A and B use the same
// one; kappa(h, e) below is the accumulator's row map, natural order 16h+e is
// used where the operand comes from memory.  (Lane maps verified with exact
// integer data on MI355X: tools/mfma_probe.hip.)
#include "dct32_mfma_core.h"

using namespace kvzhip;

// N = 32: one block per tile.  N = 16: FOUR blocks per tile, arranged 2 x 2, with the block-diagonal coefficient matrix
// diag(M16, M16) (dct32_mfma_core.h): the same instruction stream, every lane and accumulator register live -- the round-1
// two-blocks-per-tile 16x16 kernel left half of K dead (5.7 TB/s forward) and its inverse lost to the VALU butterflies (5.2).
// `count` is in blocks; a tile's chunks are permuted between memory order and tile order on the way through LDS.
// PIPE: the iteration's wait on vector memory placed by hand before its stores (wait_vmem_all, kvz_hip_internal.h) -- for grids
// whose waves take several tiles each; with one tile per wave there is no loop to pipeline.
template <int N, bool INVERSE, bool DST = false, bool PIPE = false>
__global__ __launch_bounds__(256, 4) void dct32_mfma_kernel(const i16 *__restrict__ in, i16 *__restrict__ out, size_t count)
{
  constexpr int LOG2N = N == 32 ? 5 : N == 16 ? 4 : 2;
  constexpr int BPT = (32 / N) * (32 / N);             // blocks per tile: 1, 4 or 64 (4x4, DCT or DST)
  constexpr int CPB = N * N / 8;                       // 16-byte chunks per block: 128, 32 or 2
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const size_t ntiles = (count + BPT - 1) / BPT;
  // constant operands: one precomputed record per lane (dct32_mfma_core.h)
  op16 t_nat, t_kap, t_col, t_id;
  const dct32_lane_consts &lc = tile_lane_consts<N, DST>(lane);
#pragma unroll
  for (int q = 0; q < 4; ++q) { t_nat.w[q] = lc.t_nat[q]; t_kap.w[q] = lc.t_kap[q]; t_col.w[q] = lc.t_col[q]; t_id.w[q] = lc.t_id[q]; }
  const int rowsum = lc.rowsum, colsum = lc.colsum;
  __shared__ __attribute__((aligned(16))) u8 s_tile[4][2048];
  u8 *tile = s_tile[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];               // wave-private: DS ops of one wave execute in order, no barrier

  // inverse pass 2: the plane offset 128 * (column sum of M) + rounding depends on the output ROW, i.e. on
  // (lane half, register): a 2 x 16 table in LDS, read back as broadcasts, instead of 16 live registers
  __shared__ __attribute__((aligned(16))) int s_c2[2][16];
  if (INVERSE) {
    if (N == 32) fill_inv_c2(s_c2); else fill_inv_c2_tile<N, DST>(s_c2);
    __syncthreads();
  }
  // this lane's two linear chunks of a tile (l, 64 + l) and where they sit in the tile; 16x16: which block they belong to
  const int mc0 = lane, mc1 = 64 + lane;
  const int tc0 = N == 16 ? tile_chunk16(mc0) : mc0, tc1 = N == 16 ? tile_chunk16(mc1) : mc1;
  auto load = [&](size_t t, u32x4v (&c)[2]) {
    if (N == 32) { load_chunks(in + t * 1024, lane, c); return; }
    const u32x4v z = { 0u, 0u, 0u, 0u };
    c[0] = t * BPT + (size_t)(mc0 / CPB) < count ? __builtin_nontemporal_load((const u32x4v *)(in + t * 1024) + mc0) : z;   // a tile past the last
    c[1] = t * BPT + (size_t)(mc1 / CPB) < count ? __builtin_nontemporal_load((const u32x4v *)(in + t * 1024) + mc1) : z;   // block is padded with zeros
  };

  size_t t = wave;
  u32x4v cur[2], nx1[2], nx2[2];
  if (t < ntiles) load(t, cur);
  if (t + nwaves < ntiles) load(t + nwaves, nx1);
  if (PIPE) wait_vmem_all();
  for (; t < ntiles; t += nwaves) {
    const size_t tn = t + 2 * nwaves;
    if (tn < ntiles) load(tn, nx2);                    // keep two of the wave's next tiles in flight
    u32 d[8];
    if (N == 32) chunks_to_rows(tile, lane, r, h, cur, d);
    else {
      if (N == 16) {
        *(u32x4v *)(tile + slot_of(tc0) * 16) = cur[0];
        *(u32x4v *)(tile + slot_of(tc1) * 16) = cur[1];
      } else {                                         // 4x4: a chunk is two rows of a block = two 8-byte pieces of the tile
        *(uint2 *)(tile + tile_piece4(mc0, 0)) = make_uint2(cur[0].x, cur[0].y);
        *(uint2 *)(tile + tile_piece4(mc0, 1)) = make_uint2(cur[0].z, cur[0].w);
        *(uint2 *)(tile + tile_piece4(mc1, 0)) = make_uint2(cur[1].x, cur[1].y);
        *(uint2 *)(tile + tile_piece4(mc1, 1)) = make_uint2(cur[1].z, cur[1].w);
      }
      wave_lds_fence();
      const u32x4v a = *(const u32x4v *)(tile + slot_of(4 * r + 2 * h) * 16);
      const u32x4v b = *(const u32x4v *)(tile + slot_of(4 * r + 2 * h + 1) * 16);
      wave_lds_fence();
      d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
    }
    op16 hi, lo;
    planes_from_rows(d, hi, lo);
    int o[16];
    if (!INVERSE) fwd32_core<LOG2N>(hi, lo, t_nat, t_kap, rowsum, o);
    else inv32_core(hi, lo, t_id, t_col, colsum, s_c2[h], o);
    if (PIPE) wait_vmem_all();
    if (N == 32) rows_to_chunks_store(tile, lane, r, h, o, out + t * 1024);
    else {
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        uint2 v;
        v.x = __builtin_amdgcn_perm((u32)o[4 * gg + 1], (u32)o[4 * gg], 0x05040100u);
        v.y = __builtin_amdgcn_perm((u32)o[4 * gg + 3], (u32)o[4 * gg + 2], 0x05040100u);
        *(uint2 *)(tile + slot_of(4 * r + gg) * 16 + 8 * h) = v;
      }
      wave_lds_fence();
      u32x4v a, b;
      if (N == 16) {
        a = *(const u32x4v *)(tile + slot_of(tc0) * 16);
        b = *(const u32x4v *)(tile + slot_of(tc1) * 16);
      } else {
        const uint2 a0 = *(const uint2 *)(tile + tile_piece4(mc0, 0)), a1 = *(const uint2 *)(tile + tile_piece4(mc0, 1));
        const uint2 b0 = *(const uint2 *)(tile + tile_piece4(mc1, 0)), b1 = *(const uint2 *)(tile + tile_piece4(mc1, 1));
        a.x = a0.x; a.y = a0.y; a.z = a1.x; a.w = a1.y; b.x = b0.x; b.y = b0.y; b.z = b1.x; b.w = b1.y;
      }
      wave_lds_fence();
      if (t * BPT + (size_t)(mc0 / CPB) < count) __builtin_nontemporal_store(a, (u32x4v *)(out + t * 1024) + mc0);
      if (t * BPT + (size_t)(mc1 / CPB) < count) __builtin_nontemporal_store(b, (u32x4v *)(out + t * 1024) + mc1);
    }
    cur[0] = nx1[0]; cur[1] = nx1[1]; nx1[0] = nx2[0]; nx1[1] = nx2[1];
  }
}

namespace kvzhip {
int launch_dct32_mfma(bool inverse, const i16 *in, i16 *out, size_t count, hipStream_t st)
{
  // 4 waves per workgroup, each wave strides over blocks; enough workgroups for ~4 waves per SIMD
  size_t wgs = (count + 3) / 4;
  // Workgroups per CU.  Few persistent workgroups (2 .. 12 per CU, long grid-stride loops) gave 5.3 - 5.9 TB/s depending on the
  // box and on how the count divided over them; many short-lived ones -- two to three blocks per wave, handed out by the
  // dispatcher as waves retire -- are faster and steadier: forward 4: 5.81, 32: 5.94, 64: 6.24, 96: 6.35, 128: 6.33, 160: 6.00 TB/s;
  // inverse 8: 5.53, 32: 5.84, 64: 5.69, 128: 5.48 (0.5 GiB arrays, tools/bench_all.py --tune).  With the per-lane constants
  // precomputed (five vector loads per wave) the optimum moved out further: forward 96: 6.27, 192: 6.39, 256: 6.41 (one block
  // per wave at this size); inverse 32: 5.83, 64: 6.07, 128: 6.17, 256: 5.29.  On the four times larger 4K batch of bench.py's shard
  // leg (1 029 120 blocks) the forward kernel confirmed that what it wants is ONE BLOCK PER WAVE at any size, not a number of
  // workgroups per CU: cap 192: 5.77, 384: 5.90, 768: 6.07, 1100 (no wave takes a second block): 6.14 TB/s -- so the forward cap only
  // bounds the grid for lists beyond 4 M blocks.
  const size_t cap = (size_t)num_cus() * (size_t)(inverse ? tuning("idct32_wgs_per_cu", 96) : tuning("dct32_wgs_per_cu", 4096));
  if (wgs > cap) wgs = cap;
  if (tuning("dct_pipe", 0)) {
    if (inverse) hipLaunchKernelGGL((dct32_mfma_kernel<32, true, false, true>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
    else hipLaunchKernelGGL((dct32_mfma_kernel<32, false, false, true>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  } else if (inverse) hipLaunchKernelGGL((dct32_mfma_kernel<32, true>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  else hipLaunchKernelGGL((dct32_mfma_kernel<32, false>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  KVZ_CHECK_LAUNCH("dct32_mfma_kernel");
  return KVZ_HIP_OK;
}
// 4x4 blocks (DCT or the DST-VII of intra luma), sixty-four per tile
int launch_dct4_tile(bool inverse, bool dst, const i16 *in, i16 *out, size_t count, hipStream_t st)
{
  const size_t ntiles = (count + 63) / 64;
  size_t wgs = (ntiles + 3) / 4;
  // workgroups per CU (0.5 GiB arrays): forward 64: 6.20, 128: 6.37, 192: 6.42, 256: 6.29 TB/s; inverse 32: 5.89, 64: 6.18, 96: 6.05, 128: 5.80
  // (the LDS butterfly kernel these replace: 5.4 / 5.55)
  const size_t cap = (size_t)num_cus() * (size_t)(inverse ? tuning("idct4_wgs_per_cu", 64) : tuning("dct4_wgs_per_cu", 192));
  if (wgs > cap) wgs = cap;
  const dim3 g((unsigned)wgs), b(256);
  if (dst) {
    if (inverse) hipLaunchKernelGGL((dct32_mfma_kernel<4, true, true>), g, b, 0, st, in, out, count);
    else hipLaunchKernelGGL((dct32_mfma_kernel<4, false, true>), g, b, 0, st, in, out, count);
  } else {
    if (inverse) hipLaunchKernelGGL((dct32_mfma_kernel<4, true, false>), g, b, 0, st, in, out, count);
    else hipLaunchKernelGGL((dct32_mfma_kernel<4, false, false>), g, b, 0, st, in, out, count);
  }
  KVZ_CHECK_LAUNCH("dct32_mfma_kernel<4>");
  return KVZ_HIP_OK;
}
// 16x16 blocks, four per tile
int launch_dct16_tile(bool inverse, const i16 *in, i16 *out, size_t count, hipStream_t st)
{
  const size_t ntiles = (count + 3) / 4;
  size_t wgs = (ntiles + 3) / 4;
  // workgroups per CU (0.5 GiB arrays): forward 64: 6.43, 96: 6.50, 128: 6.52, 192: 6.42, 256: 6.34 TB/s; inverse 32: 6.16, 64: 6.17, 96: 5.98, 128: 5.81
  const size_t cap = (size_t)num_cus() * (size_t)(inverse ? tuning("idct16_wgs_per_cu", 64) : tuning("dct16_wgs_per_cu", 128));
  if (wgs > cap) wgs = cap;
  if (inverse) hipLaunchKernelGGL((dct32_mfma_kernel<16, true>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  else hipLaunchKernelGGL((dct32_mfma_kernel<16, false>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  KVZ_CHECK_LAUNCH("dct32_mfma_kernel<16>");
  return KVZ_HIP_OK;
}
}  // namespace kvzhip
