// sao.hip -- sample adaptive offset: statistics, distortion deltas and reconstruction.
//
// Reference: src/strategies/generic/sao-generic.c:34-183 (the four strategies of
// strategies-sao.h:36-57), src/sao.c:164-180 (kvz_calc_sao_offset_array) and :247-261
// (calc_sao_bands).  SURVEY.md section 8(f) row 4.
//
// The statistics / distortion entries take `count` contiguous bw x bh blocks (stride = bw), the
// way sao.c blits an LCU before it calls the strategies.  One workgroup per block: both blocks are
// staged in LDS once, and ALL FOUR edge classes are evaluated from that one copy (the reference
// makes one pass over the data per class).  Per-thread partial sums are selected into fixed
// registers by category (no dynamically indexed accumulators), reduced with DPP inside the wave
// and with a handful of LDS atomics across the four waves.  HBM-bound in principle (2 bw bh bytes
// per block); everything is integer.
#include "kvz_hip_internal.h"

using namespace kvzhip;

namespace {

constexpr int MAX_PX = 64 * 64;

// g_sao_edge_offsets (sao.h:58-63): neighbours a, b of c as index deltas for a given row stride
__device__ __forceinline__ void eo_deltas(int eo_class, int stride, int &da, int &db)
{
  switch (eo_class) {
    case 0: da = -1; db = 1; break;
    case 1: da = -stride; db = stride; break;
    case 2: da = -stride - 1; db = stride + 1; break;
    default: da = -stride + 1; db = stride - 1; break;
  }
}
// sao_calc_eo_cat (sao-generic.c:34-43)
__device__ __forceinline__ int eo_cat(int a, int b, int c)
{
  const int idx = 2 + ((c > a) - (c < a)) + ((c > b) - (c < b));
  // {1, 2, 0, 3, 4} packed in nibbles
  return (int)((0x43021u >> (4 * idx)) & 15u);
}

// stage `n` bytes of two contiguous arrays into LDS (dword loads when the arrays allow it)
__device__ __forceinline__ void stage_pair(u8 *s_a, u8 *s_b, const u8 *a, const u8 *b, int n, int tid)
{
  const bool fast = ((((uintptr_t)a | (uintptr_t)b) & 3) == 0);
  const int n4 = fast ? n >> 2 : 0;
  for (int i = tid; i < n4; i += 256) {
    ((u32 *)s_a)[i] = ((const u32 *)a)[i];
    ((u32 *)s_b)[i] = ((const u32 *)b)[i];
  }
  for (int i = 4 * n4 + tid; i < n; i += 256) { s_a[i] = a[i]; s_b[i] = b[i]; }
}

__device__ __forceinline__ int wave_sum(int v) { return (int)group_sum<64>((u32)v); }

// MODE 0: calc_sao_edge_dir for the four classes -> out[blk][4][2][5]
// MODE 1: sao_edge_ddistortion for the four classes with offsets[blk][4][5] -> out[blk][4]
template <int MODE>
__global__ __launch_bounds__(256) void sao_edge_kernel(const u8 *__restrict__ orig, const u8 *__restrict__ rec, int bw, int bh,
                                                      const int *__restrict__ offsets, int *__restrict__ out)
{
  __shared__ __attribute__((aligned(16))) u8 s_o[MAX_PX], s_r[MAX_PX];
  __shared__ int s_acc[40];
  __shared__ int s_off[20];
  const int tid = threadIdx.x, n = bw * bh;
  const size_t blk = blockIdx.x;
  stage_pair(s_o, s_r, orig + blk * (size_t)n, rec + blk * (size_t)n, n, tid);
  if (tid < 40) s_acc[tid] = 0;
  if (MODE == 1 && tid < 20) s_off[tid] = offsets[blk * 20 + tid];
  __syncthreads();

  const int iw = bw - 2, ih = bh - 2, interior = iw > 0 && ih > 0 ? iw * ih : 0;
  int acc_sum[4][5], acc_cnt[4][5], acc_dd[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc_dd[e] = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { acc_sum[e][k] = 0; acc_cnt[e][k] = 0; }
  }
  for (int i = tid; i < interior; i += 256) {
    const int y = 1 + i / iw, x = 1 + (i - (y - 1) * iw), p = y * bw + x;
    const int c = s_r[p], diff = (int)s_o[p] - c;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int da, db;
      eo_deltas(e, bw, da, db);
      const int cat = eo_cat(s_r[p + da], s_r[p + db], c);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          acc_sum[e][k] += cat == k ? diff : 0;
          acc_cnt[e][k] += cat == k ? 1 : 0;
        }
      } else {
        const int off = s_off[e * 5 + cat];
        // (diff - off)^2 - diff^2 (sao-generic.c:67-71); zero for off == 0
        acc_dd[e] += off * off - 2 * diff * off;
      }
    }
  }
  const int lane = tid & 63;
  if (MODE == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int s = wave_sum(acc_sum[e][k]), c = wave_sum(acc_cnt[e][k]);
        if (lane == 0) { atomicAdd(&s_acc[e * 10 + k], s); atomicAdd(&s_acc[e * 10 + 5 + k], c); }
      }
    __syncthreads();
    if (tid < 40) out[blk * 40 + tid] = s_acc[tid];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s = wave_sum(acc_dd[e]);
      if (lane == 0) atomicAdd(&s_acc[e], s);
    }
    __syncthreads();
    if (tid < 4) out[blk * 4 + tid] = s_acc[tid];
  }
}

// The same two entries for the shapes sao.c really blits (width a multiple of 4, dword-aligned arrays).
// ONE WAVE per block.  A work item is one dword of rec = 4 pixels with the 3 x 3 dwords around it, so a pixel's eight
// neighbours come from registers.  Per (class, pixel) the category index sgn(c-a) + sgn(c-b) is two v_sub + two v_med3,
// and instead of selecting it into 5 sums + 5 counts, the pixel is added into PACKED accumulators at a bit position
// looked up with v_perm: counts in 7-bit fields of a dword, (orig - rec + 255) in 15-bit fields of a 64-bit register
// (a lane sees at most 64 pixels).  The middle category (index 0) and the excluded border columns go to a scratch
// field: its sum and count are what is left of the totals.  One wave per block keeps the unpack + cross-lane reduction
// (34 values) to once per block -- with four waves it was a third of the kernel.  ddistortion is evaluated from the
// statistics: the sum over a category of (diff - off)^2 - diff^2 is cnt * off^2 - 2 * off * sum (sao-generic.c:67-71).
template <int MODE>
__global__ __launch_bounds__(64) void sao_edge_fast_kernel(const u8 *__restrict__ orig, const u8 *__restrict__ rec, int bw, int bh,
                                                          const int *__restrict__ offsets, int *__restrict__ out)
{
  // only rec is staged (its pixels are read nine times); orig is read once per item, straight from memory -- with both
  // blocks in LDS a CU held 19 of its 32 waves
  __shared__ __attribute__((aligned(16))) u32 s_r[MAX_PX / 4];
  __shared__ int s_stat[4][2][5];
  const int tid = threadIdx.x, n4 = (bw * bh) >> 2, g4 = bw >> 2;
  const size_t blk = blockIdx.x;
  const u32 *go = (const u32 *)(orig + blk * (size_t)(bw * bh));
  {
    const u32 *gr = (const u32 *)(rec + blk * (size_t)(bw * bh));
    for (int i = tid; i < n4; i += 64) s_r[i] = gr[i];
  }
  wave_lds_fence();

  unsigned long long sum[4] = { 0, 0, 0, 0 };
  u32 cnt[4] = { 0, 0, 0, 0 };
  int tot = 0, npx = 0;
  const int items = (bh - 2) * g4;
  const u32 recip = (65536u + (u32)g4 - 1u) / (u32)g4;        // it / g4 == (it * recip) >> 16 for it < 1024, g4 <= 16
  for (int it = tid; it < items; it += 64) {
    const int yy = (int)(((u32)it * recip) >> 16), xg = it - yy * g4, at = (yy + 1) * g4 + xg;
    // rows y-1, y, y+1 as 6-pixel windows: [last byte of the left dword, the dword, first byte of the right dword]
    int win[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      // the first / last item's outer neighbour dword lies just outside the block: those pixels are excluded border
      // columns, any value will do, but the read stays inside the array
      const int il = at + (r - 1) * g4 - 1, ir = at + (r - 1) * g4 + 1;
      const u32 l = s_r[r == 0 ? (il < 0 ? 0 : il) : il], c = s_r[at + (r - 1) * g4], rt = s_r[r == 2 ? (ir < n4 ? ir : n4 - 1) : ir];
      win[r][0] = (int)(l >> 24);
#pragma unroll
      for (int k = 0; k < 4; ++k) win[r][1 + k] = (int)((c >> (8 * k)) & 255u);
      win[r][5] = (int)(rt & 255u);
    }
    const u32 od = go[at];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool valid = !((k == 0 && xg == 0) || (k == 3 && xg == g4 - 1));
      const int c = win[1][1 + k];
      const int diff = (int)((od >> (8 * k)) & 255u) - c;
      tot += valid ? diff : 0;
      npx += valid ? 1 : 0;
      const u32 v = (u32)(diff + 255);
      // neighbour pairs of the four classes (g_sao_edge_offsets, sao.h:58-63)
      const int na[4] = { win[1][k], win[0][1 + k], win[0][k], win[0][2 + k] };
      const int nb[4] = { win[1][2 + k], win[2][1 + k], win[2][2 + k], win[2][k] };
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s1 = clampi(c - na[e], -1, 1), s2 = clampi(c - nb[e], -1, 1);
        u32 sel = (u32)(s1 + s2) + 0x0c0c0c02u;                         // byte 0 = index + 2, other selector bytes = constant zero
        if (k == 0 || k == 3) sel = valid ? sel : 0x0c0c0c02u;
        const u32 shs = __builtin_amdgcn_perm(0x0000002Du, 0x1E3C0F00u, sel);    // index -2, -1, 0, 1, 2 -> bit 0, 15, 60, 30, 45
        const u32 shc = __builtin_amdgcn_perm(0x00000015u, 0x0E1C0700u, sel);    //                        -> bit 0, 7, 28, 14, 21
        sum[e] += (unsigned long long)v << shs;
        cnt[e] += 1u << shc;
      }
    }
  }
  // unpack and reduce over the wave: slots -2, -1, +1, +2 are categories 1, 2, 3, 4 (sao_calc_eo_cat, sao-generic.c:34-43)
  const int wt = wave_sum(tot), wn = wave_sum(npx);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int rest_s = wt, rest_c = wn;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = (int)((cnt[e] >> (7 * k)) & 127u);
      const int sm = (int)((sum[e] >> (15 * k)) & 32767u) - 255 * c;
      const int ws = wave_sum(sm), wc = wave_sum(c);
      rest_s -= ws; rest_c -= wc;
      if (tid == 0) { s_stat[e][0][1 + k] = ws; s_stat[e][1][1 + k] = wc; }
    }
    if (tid == 0) { s_stat[e][0][0] = rest_s; s_stat[e][1][0] = rest_c; }      // category 0 is the rest
  }
  wave_lds_fence();
  if (MODE == 0) {
    if (tid < 40) out[blk * 40 + tid] = (&s_stat[0][0][0])[tid];
  } else if (tid < 4) {
    int dd = 0;
#pragma unroll
    for (int cat = 0; cat < 5; ++cat) {
      const int off = offsets[blk * 20 + tid * 5 + cat];
      dd += s_stat[tid][1][cat] * off * off - 2 * off * s_stat[tid][0][cat];
    }
    out[blk * 4 + tid] = dd;
  }
}

// MODE 0: calc_sao_bands (sao.c:247-261) -> out[blk][2][32]
// MODE 1: sao_band_ddistortion (sao-generic.c:157-183) with band_pos[blk], bands[blk][4] -> out[blk]
template <int MODE>
__global__ __launch_bounds__(256) void sao_band_kernel(const u8 *__restrict__ orig, const u8 *__restrict__ rec, int bw, int bh,
                                                      const int *__restrict__ band_pos, const int *__restrict__ bands, int *__restrict__ out)
{
  __shared__ int s_hist[4][64];                        // one private histogram per wave: [sum 0..31 | count 0..31]
  __shared__ int s_dd;
  const int tid = threadIdx.x, n = bw * bh, wv = tid >> 6;
  const size_t blk = blockIdx.x;
  const u8 *o = orig + blk * (size_t)n, *r = rec + blk * (size_t)n;
  s_hist[wv][tid & 63] = 0;
  if (tid == 0) s_dd = 0;
  __syncthreads();
  if (MODE == 0) {
    for (int i = tid; i < n; i += 256) {
      const int rv = r[i], band = rv >> 3;               // bitdepth 8: shift = 3
      atomicAdd(&s_hist[wv][band], (int)o[i] - rv);
      atomicAdd(&s_hist[wv][32 + band], 1);
    }
    __syncthreads();
    if (tid < 64) out[blk * 64 + tid] = s_hist[0][tid] + s_hist[1][tid] + s_hist[2][tid] + s_hist[3][tid];
  } else {
    const int bp = band_pos[blk];
    int offs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) offs[k] = bands[blk * 4 + k];
    int acc = 0;
    for (int i = tid; i < n; i += 256) {
      const int rv = r[i], band = (rv >> 3) - bp;
      const int off = band == 0 ? offs[0] : band == 1 ? offs[1] : band == 2 ? offs[2] : band == 3 ? offs[3] : 0;
      const int diff = (int)o[i] - rv;
      acc += off * off - 2 * diff * off;
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) atomicAdd(&s_dd, acc);
    __syncthreads();
    if (tid == 0) out[blk] = s_dd;
  }
}

// sao_reconstruct_color (sao-generic.c:112-154): one workgroup per block descriptor of the plane
__global__ __launch_bounds__(256) void sao_reconstruct_kernel(const u8 *__restrict__ rec, u32 stride, u8 *__restrict__ dst, u32 dst_stride,
                                                             const kvz_hip_sao_block *__restrict__ blocks,
                                                             const kvz_hip_sao_info *__restrict__ infos, int n_infos, int plane_w, int plane_h,
                                                             int color)
{
  const kvz_hip_sao_block &b = blocks[blockIdx.x];
  if (b.sao_index < 0 || b.sao_index >= n_infos || b.width < 1 || b.height < 1) return;
  const kvz_hip_sao_info &sao = infos[b.sao_index];
  // the edge filter reads the ring around the block (the caller trims blocks at the picture border the way
  // kvz_sao_reconstruct does, sao.c:296-318); a descriptor that would read outside the plane is skipped
  const int rx = (sao.type == 2 && (sao.eo_class & 3) != 1) ? 1 : 0, ry = (sao.type == 2 && (sao.eo_class & 3) != 0) ? 1 : 0;
  if (b.x - rx < 0 || b.y - ry < 0 || b.x + b.width + rx > plane_w || b.y + b.height + ry > plane_h) return;
  const int is_v = color == 2, n = b.width * b.height;
  if (sao.type == 1) {
    // kvz_calc_sao_offset_array (sao.c:164-180) applied per pixel
    const int bp = sao.band_position[is_v];
    for (int i = threadIdx.x; i < n; i += 256) {
      const int y = i / b.width, x = i - y * b.width;
      const int val = rec[(size_t)(b.y + y) * stride + b.x + x], band = (val >> 3) - bp;
      int v = val;
      if (band >= 0 && band < 4) v = clampi(val + sao.offsets[band + 1 + 5 * is_v], 0, 255);
      dst[(size_t)(b.y + y) * dst_stride + b.x + x] = (u8)v;
    }
  } else if (sao.type == 2) {
    int da, db;
    eo_deltas(sao.eo_class & 3, (int)stride, da, db);
    for (int i = threadIdx.x; i < n; i += 256) {
      const int y = i / b.width, x = i - y * b.width;
      const u8 *c = rec + (size_t)(b.y + y) * stride + b.x + x;
      const int cat = eo_cat(c[da], c[db], c[0]);
      dst[(size_t)(b.y + y) * dst_stride + b.x + x] = (u8)clampi((int)c[0] + sao.offsets[cat + 5 * is_v], 0, 255);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) {
      const int y = i / b.width, x = i - y * b.width;
      dst[(size_t)(b.y + y) * dst_stride + b.x + x] = rec[(size_t)(b.y + y) * stride + b.x + x];
    }
  }
}

// the packed-accumulator kernel: whole dwords per row, dword-aligned arrays, an interior to work on
bool edge_fast_ok(const void *orig, const void *rec, int bw, int bh)
{
  return tuning("sao_edge_fast", 1) && (bw & 3) == 0 && bh >= 3 && (((uintptr_t)orig | (uintptr_t)rec) & 3) == 0;
}

bool block_dims_ok(int bw, int bh)
{
  if (bw < 1 || bh < 1 || bw > 64 || bh > 64) {
    set_error_msg("SAO entries take blocks of 1..64 x 1..64 pixels (an LCU plane)");
    return false;
  }
  return true;
}

}  // namespace

extern "C" {

int kvz_hip_sao_edge_stats_batch(const kvz_hip_pixel *orig, const kvz_hip_pixel *rec, int block_width, int block_height, size_t count,
                                 int32_t *cat_sum_cnt, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!block_dims_ok(block_width, block_height)) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  if (!orig || !rec || !cat_sum_cnt || count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  if (edge_fast_ok(orig, rec, block_width, block_height))
    hipLaunchKernelGGL((sao_edge_fast_kernel<0>), dim3((unsigned)count), dim3(64), 0, ctx_stream(s), orig, rec, block_width, block_height,
                       (const int *)nullptr, cat_sum_cnt);
  else
    hipLaunchKernelGGL((sao_edge_kernel<0>), dim3((unsigned)count), dim3(256), 0, ctx_stream(s), orig, rec, block_width, block_height,
                       (const int *)nullptr, cat_sum_cnt);
  KVZ_CHECK_LAUNCH("sao_edge_kernel<stats>");
  return KVZ_HIP_OK;
}

int kvz_hip_sao_edge_ddistortion_batch(const kvz_hip_pixel *orig, const kvz_hip_pixel *rec, int block_width, int block_height, size_t count,
                                       const int32_t *offsets, int32_t *ddistortion, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!block_dims_ok(block_width, block_height)) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  if (!orig || !rec || !offsets || !ddistortion || count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  if (edge_fast_ok(orig, rec, block_width, block_height))
    hipLaunchKernelGGL((sao_edge_fast_kernel<1>), dim3((unsigned)count), dim3(64), 0, ctx_stream(s), orig, rec, block_width, block_height,
                       offsets, ddistortion);
  else
    hipLaunchKernelGGL((sao_edge_kernel<1>), dim3((unsigned)count), dim3(256), 0, ctx_stream(s), orig, rec, block_width, block_height,
                       offsets, ddistortion);
  KVZ_CHECK_LAUNCH("sao_edge_kernel<ddistortion>");
  return KVZ_HIP_OK;
}

int kvz_hip_sao_band_stats_batch(const kvz_hip_pixel *orig, const kvz_hip_pixel *rec, int block_width, int block_height, size_t count,
                                 int32_t *sao_bands, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!block_dims_ok(block_width, block_height)) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  if (!orig || !rec || !sao_bands || count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  hipLaunchKernelGGL((sao_band_kernel<0>), dim3((unsigned)count), dim3(256), 0, ctx_stream(s), orig, rec, block_width, block_height,
                     (const int *)nullptr, (const int *)nullptr, sao_bands);
  KVZ_CHECK_LAUNCH("sao_band_kernel<stats>");
  return KVZ_HIP_OK;
}

int kvz_hip_sao_band_ddistortion_batch(const kvz_hip_pixel *orig, const kvz_hip_pixel *rec, int block_width, int block_height, size_t count,
                                       const int32_t *band_pos, const int32_t *sao_bands, int32_t *ddistortion, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!block_dims_ok(block_width, block_height)) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  if (!orig || !rec || !band_pos || !sao_bands || !ddistortion || count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  hipLaunchKernelGGL((sao_band_kernel<1>), dim3((unsigned)count), dim3(256), 0, ctx_stream(s), orig, rec, block_width, block_height,
                     band_pos, sao_bands, ddistortion);
  KVZ_CHECK_LAUNCH("sao_band_kernel<ddistortion>");
  return KVZ_HIP_OK;
}

int kvz_hip_sao_reconstruct_color_batch(const kvz_hip_pixel *rec, uint32_t stride, int plane_w, int plane_h,
                                        kvz_hip_pixel *new_rec, uint32_t new_stride,
                                        const kvz_hip_sao_block *blocks, size_t count, const kvz_hip_sao_info *infos, int n_infos, int color,
                                        kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (color < 0 || color > 2 || plane_w < 1 || plane_h < 1 || n_infos < 0) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  if (!rec || !new_rec || !blocks || !infos || count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  hipLaunchKernelGGL(sao_reconstruct_kernel, dim3((unsigned)count), dim3(256), 0, ctx_stream(s), rec, stride, new_rec, new_stride, blocks, infos,
                     n_infos, plane_w, plane_h, color);
  KVZ_CHECK_LAUNCH("sao_reconstruct_kernel");
  return KVZ_HIP_OK;
}

}  // extern "C"
