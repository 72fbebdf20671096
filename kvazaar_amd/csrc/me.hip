// me.hip -- batched integer motion-estimation costs, one CTU per workgroup.
//
// Reference callers: check_mv_cost (src/search_inter.c:195-232) evaluates one
// candidate at a time through kvz_image_calc_sad (src/image.c:455-486) ->
// kvz_reg_sad; the search patterns (hexbs :690, diamond :796, full :886) visit
// tens to hundreds of candidates per PU and the quadtree (src/search.c:424)
// repeats that for every PU size.  Here one launch evaluates every candidate of a
// pattern for all 85 square PUs (64x64 .. 8x8) of every CTU:
//   * the CTU's 64x64 source block and the search window (64 + offset range,
//     edge replicated exactly like image_interpolated_sad, image.c:320-444) are
//     read from HBM once and staged in LDS;
//   * thread = one 8x8 block (source rows held in registers), wave = one
//     candidate: 8x8 SADs with v_sad_u8 on dwords re-aligned by v_alignbyte;
//   * 16/32/64 PUs are sums of their 8x8 SADs (wave shuffles), bit-identical to
//     the reference because SAD is additive over disjoint pixels.
// HBM traffic per CTU: 4 KiB + window + 340 B per candidate, against
// 2 * 85-PU-areas * candidates for per-candidate kernels: the arithmetic, not
// HBM, bounds this kernel (LDS reads + VALU).
#include "kvz_hip_internal.h"

using namespace kvzhip;

#define ME_RANGE 64
#define ME_WS 196                      /* window row stride: 192 + 4, 49 dwords -> rows 8 apart land 8 banks apart */
#define ME_WROWS 192

struct me_plane { const u8 *p; u32 stride; int w, h; };

__global__ __launch_bounds__(256) void ctu_sad_grid_kernel(me_plane pic, me_plane ref, const kvz_hip_ctu_search *__restrict__ ctus,
                                                           const short *__restrict__ mv_offsets, int n_mv, u32 *__restrict__ costs)
{
  __shared__ __attribute__((aligned(16))) u8 s_cur[64 * 64];
  __shared__ __attribute__((aligned(16))) u8 s_win[ME_WROWS * ME_WS];
  __shared__ int s_box[4];

  const int tid = threadIdx.x;
  const kvz_hip_ctu_search c = ctus[blockIdx.x];
  u32 *out = costs + (size_t)blockIdx.x * n_mv * KVZ_HIP_CTU_PUS;

  // bounding box of the (valid) candidate offsets
  if (tid == 0) { s_box[0] = ME_RANGE + 1; s_box[1] = -ME_RANGE - 1; s_box[2] = ME_RANGE + 1; s_box[3] = -ME_RANGE - 1; }
  __syncthreads();
  for (int m = tid; m < n_mv; m += 256) {
    const int dx = mv_offsets[2 * m], dy = mv_offsets[2 * m + 1];
    if (dx >= -ME_RANGE && dx <= ME_RANGE && dy >= -ME_RANGE && dy <= ME_RANGE) {
      atomicMin(&s_box[0], dx); atomicMax(&s_box[1], dx); atomicMin(&s_box[2], dy); atomicMax(&s_box[3], dy);
    }
  }
  __syncthreads();
  const int x0 = s_box[0], x1 = s_box[1], y0 = s_box[2], y1 = s_box[3];
  const bool any_valid = x1 >= x0;
  const int ww = any_valid ? 64 + (x1 - x0) : 0, wh = any_valid ? 64 + (y1 - y0) : 0;

  // source block (coordinates clamped so that a ragged CTU never reads outside the picture)
  for (int i = tid; i < 64 * 16; i += 256) {
    const int r = i >> 4, q = (i & 15) << 2;
    const int yy = clampi(c.y + r, 0, pic.h - 1);
    u32 v;
    if (c.x + q >= 0 && c.x + q + 4 <= pic.w) __builtin_memcpy(&v, pic.p + (size_t)yy * pic.stride + c.x + q, 4);
    else {
      u8 b[4];
      for (int k = 0; k < 4; ++k) b[k] = pic.p[(size_t)yy * pic.stride + clampi(c.x + q + k, 0, pic.w - 1)];
      v = b[0] | (b[1] << 8) | (b[2] << 16) | ((u32)b[3] << 24);
    }
    *(u32 *)(s_cur + r * 64 + q) = v;
  }
  // search window: ref rows (c.y + mvy + y0 ..), cols (c.x + mvx + x0 ..), edge replicated
  {
    const int rx = c.x + c.mvx + x0, ry = c.y + c.mvy + y0;
    const bool inside = rx >= 0 && ry >= 0 && rx + ww <= ref.w && ry + wh <= ref.h;
    const int wq = (ww + 3) >> 2;
    for (int i = tid; i < wh * wq; i += 256) {
      const int r = i / wq, q = (i - r * wq) << 2;
      u32 v;
      if (inside && q + 4 <= ww) __builtin_memcpy(&v, ref.p + (size_t)(ry + r) * ref.stride + rx + q, 4);
      else {
        const u8 *row = ref.p + (size_t)clampi(ry + r, 0, ref.h - 1) * ref.stride;
        u8 b[4];
        for (int k = 0; k < 4; ++k) b[k] = row[clampi(rx + q + k, 0, ref.w - 1)];
        v = b[0] | (b[1] << 8) | (b[2] << 16) | ((u32)b[3] << 24);
      }
      *(u32 *)(s_win + r * ME_WS + q) = v;
    }
  }
  __syncthreads();

  const int b = tid & 63, bx = b & 7, by = b >> 3, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool blk_valid = (c.x + bx * 8 + 8 <= pic.w) && (c.y + by * 8 + 8 <= pic.h);
  u32 cur[16];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint2 v = *(const uint2 *)(s_cur + (by * 8 + r) * 64 + bx * 8);
    cur[2 * r] = v.x; cur[2 * r + 1] = v.y;
  }

  for (int m = wv; m < n_mv; m += 4) {            // wave-uniform candidate
    const int dx = mv_offsets[2 * m], dy = mv_offsets[2 * m + 1];
    u32 *o = out + (size_t)m * KVZ_HIP_CTU_PUS;
    if (dx < -ME_RANGE || dx > ME_RANGE || dy < -ME_RANGE || dy > ME_RANGE) {
      for (int k = b; k < KVZ_HIP_CTU_PUS; k += 64) o[k] = 0xffffffffu;
      continue;
    }
    const int wx = bx * 8 + dx - x0, wy = by * 8 + dy - y0;
    const int sh = wx & 3;
    const u8 *p = s_win + wy * ME_WS + (wx & ~3);
    u32 sad = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const u32 d0 = *(const u32 *)(p + r * ME_WS), d1 = *(const u32 *)(p + r * ME_WS + 4), d2 = *(const u32 *)(p + r * ME_WS + 8);
      sad = __builtin_amdgcn_sad_u8(cur[2 * r], __builtin_amdgcn_alignbyte(d1, d0, sh), sad);
      sad = __builtin_amdgcn_sad_u8(cur[2 * r + 1], __builtin_amdgcn_alignbyte(d2, d1, sh), sad);
    }
    // SAD <= 64*64*255 < 2^20; bits 24.. count the valid 8x8 blocks under each sum
    u32 v8 = blk_valid ? (sad | (1u << 24)) : 0u;
    u32 t = v8 + (u32)__shfl_xor((int)v8, 1, 64);
    const u32 v16 = t + (u32)__shfl_xor((int)t, 8, 64);
    t = v16 + (u32)__shfl_xor((int)v16, 2, 64);
    const u32 v32 = t + (u32)__shfl_xor((int)t, 16, 64);
    t = v32 + (u32)__shfl_xor((int)v32, 4, 64);
    const u32 v64 = t + (u32)__shfl_xor((int)t, 32, 64);
    o[21 + b] = blk_valid ? sad : 0xffffffffu;
    if (!(bx & 1) && !(by & 1)) o[5 + (by >> 1) * 4 + (bx >> 1)] = (v16 >> 24) == 4 ? (v16 & 0xffffffu) : 0xffffffffu;
    if (!(bx & 3) && !(by & 3)) o[1 + (by >> 2) * 2 + (bx >> 2)] = (v32 >> 24) == 16 ? (v32 & 0xffffffu) : 0xffffffffu;
    if (b == 0) o[0] = (v64 >> 24) == 64 ? (v64 & 0xffffffu) : 0xffffffffu;
  }
}

extern "C" int kvz_hip_ctu_sad_grid_batch(const kvz_hip_pixel *pic, uint32_t pic_stride, int pic_w, int pic_h,
                                          const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                                          const kvz_hip_ctu_search *ctus, size_t count, const int16_t *mv_offsets, int n_mv,
                                          uint32_t *costs, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!pic || !ref || !ctus || !mv_offsets || !costs || pic_w <= 0 || pic_h <= 0 || ref_w <= 0 || ref_h <= 0 || n_mv < 0) return kvzhip::invalid_arg(__func__);
  if (count == 0 || n_mv == 0) return KVZ_HIP_OK;
  if (count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  me_plane p = { pic, pic_stride, pic_w, pic_h }, r = { ref, ref_stride, ref_w, ref_h };
  hipLaunchKernelGGL(ctu_sad_grid_kernel, dim3((unsigned)count), dim3(256), 0, ctx_stream(s), p, r, ctus, mv_offsets, n_mv, costs);
  KVZ_CHECK_LAUNCH("ctu_sad_grid_kernel");
  return KVZ_HIP_OK;
}
