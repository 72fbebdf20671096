// intra.hip -- intra prediction (angular / planar / DC) and the fused rough mode search.
//
// Reference: src/strategies/generic/intra-generic.c:37-189 (angular_pred, intra_pred_planar),
// src/intra.c:164-331 (reference smoothing, DC, edge filters, kvz_intra_predict) and the
// cost loop of search_intra_rough, src/search_intra.c:404-520.  SURVEY.md section 8(f) row 2.
//
// Three kernels:
//   intra_build_reference_kernel  one wave per PU: the kvz_intra_ref of kvz_intra_build_reference
//                          (intra.c:334-588) gathered from the reconstruction plane in HBM, so the
//                          references of a wavefront of PUs never visit the host.
//   intra_predict_kernel   one wave per (PU, mode): writes the N x N prediction (the drop-in
//                          strategies and kvz_intra_predict for a list of modes).
//   intra_rough_kernel     all 35 modes of a PU against its original block, SATD (and SAD) per
//                          mode, predictions never leave the registers.  Unlike the streaming
//                          kernels this one is VALU bound: ~240 B of HBM traffic per 8x8 PU
//                          against 35 predictions + 35 Hadamard transforms.
//
// Angular modes are evaluated in the reference's "vertical" orientation (rows advance along the
// prediction direction) from the main reference as staged in LDS -- extended below index -1 by
// the projected side reference only for the modes that need it; horizontal modes are the
// transpose, and because SATD / SAD are invariant under transposing both blocks the rough
// kernel compares them against the transposed original instead of flipping.
#include "kvz_hip_internal.h"
#include "satd_regs.h"

using namespace kvzhip;

namespace {

typedef unsigned short v2us __attribute__((ext_vector_type(2)));

constexpr int RS = 68;   // bytes per staged reference array: entries 0 .. 2N (<= 64), zero padded

// intra-generic.c:46-47
__constant__ int c_ang_disp[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__constant__ int c_ang_inv[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };

struct ang_t { int vertical, disp, inv; };
__device__ __forceinline__ ang_t ang_of(int mode)
{
  ang_t a;
  a.vertical = mode >= 18;
  const int md = a.vertical ? mode - 26 : 10 - mode;
  const int amd = md < 0 ? -md : md;
  a.disp = md < 0 ? -c_ang_disp[amd] : c_ang_disp[amd];
  a.inv = c_ang_inv[amd];
  return a;
}

// intra.c:289-306: which reference kvz_intra_predict hands to the predictor
__device__ __forceinline__ bool use_filtered(int mode, int log2_width, int flags)
{
  if ((flags & KVZ_HIP_INTRA_RAW) || !(flags & KVZ_HIP_INTRA_LUMA) || mode == 1 || log2_width == 2) return false;
  if (mode == 0) return true;
  const int dv = mode > 26 ? mode - 26 : 26 - mode, dh = mode > 10 ? mode - 10 : 10 - mode;
  const int thres = log2_width == 3 ? 7 : (log2_width == 4 ? 1 : 0);
  return (dv < dh ? dv : dh) > thres;
}
// intra.c:314-329: DC edge filter / boundary post-process apply to luma blocks narrower than 32
__device__ __forceinline__ bool luma_edge_filters(int log2_width, int flags)
{
  return !(flags & KVZ_HIP_INTRA_RAW) && (flags & KVZ_HIP_INTRA_LUMA) && log2_width < 5;
}

// Stage the reference pixels of `npu` PUs: s_ref[p][0] = left, [1] = top (unfiltered, entry 0 = corner),
// [2], [3] = smoothed left / top (intra.c:164-192).  Entries beyond 2N are zero.  Ends with a barrier.
template <int N>
__device__ __forceinline__ void stage_refs(u8 (*s_ref)[4][RS], int npu, const kvz_hip_intra_ref *refs, size_t pu0, size_t count,
                                           int tid, int nthreads)
{
  u32 *z = (u32 *)s_ref;
  for (int i = tid; i < npu * 4 * RS / 4; i += nthreads) z[i] = 0u;
  __syncthreads();
  const u8 *g = (const u8 *)(refs + pu0);
  const size_t left = count - pu0;
  const int avail = (int)(left < (size_t)npu ? left : (size_t)npu) * 130;
  for (int i = tid; i < avail; i += nthreads) {
    const int p = i / 130, o = i - p * 130, a = o >= 65, k = o - 65 * a;
    if (k <= 2 * N) s_ref[p][a][k] = g[i];
  }
  __syncthreads();
  constexpr int RW = 2 * N + 1;
  for (int i = tid; i < npu * 2 * RW; i += nthreads) {
    const int p = i / (2 * RW), r = i - p * 2 * RW, a = r >= RW, k = r - RW * a;
    const u8 *src = s_ref[p][a];
    int v;
    if (k == 0) v = (s_ref[p][0][1] + 2 * s_ref[p][0][0] + s_ref[p][1][1] + 2) >> 2;
    else if (k == 2 * N) v = src[k];
    else v = (src[k - 1] + 2 * src[k] + src[k + 1] + 2) >> 2;
    s_ref[p][2 + a][k] = (u8)v;
  }
  __syncthreads();
}

// Extended main reference of one PU for an angular mode: e[k], k = idx + N, idx = -N .. 2N+1.
// idx >= -1: main[idx + 1]; below: the side reference projected with the inverse angle
// (intra-generic.c:78-93).  Entries the mode never reads are still filled (index clamped).
template <int N>
__device__ __forceinline__ u8 ext_entry(const u8 (*ref)[RS], bool fil, const ang_t &a, int idx)
{
  const u8 *mainr = ref[2 * fil + (a.vertical ? 1 : 0)];
  const u8 *side = ref[2 * fil + (a.vertical ? 0 : 1)];
  if (idx >= -1) return mainr[idx + 1];
  int si = (128 + (-idx - 1) * a.inv) >> 8;
  if (si > 2 * N) si = 2 * N;
  return side[si];
}

__device__ __forceinline__ int dc_value(const u8 (*ref)[RS], int n, int log2_width)
{
  int sum = n;
  for (int i = 1; i <= n; ++i) sum += ref[0][i] + ref[1][i];
  return sum >> (log2_width + 1);
}

// one pixel of the DC prediction (intra.c:217-278)
__device__ __forceinline__ int dc_px(const u8 (*ref)[RS], int dc, bool edge, int x, int y)
{
  if (!edge || (x > 0 && y > 0)) return dc;
  if (x == 0 && y == 0) return (ref[0][1] + 2 * dc + ref[1][1] + 2) >> 2;
  if (y == 0) return (ref[1][x + 1] + 3 * dc + 2) >> 2;
  return (ref[0][y + 1] + 3 * dc + 2) >> 2;
}

// one pixel of the planar prediction (intra-generic.c:155-189, closed form :167-175)
__device__ __forceinline__ int planar_px(const u8 *left, const u8 *top, int n, int log2_width, int x, int y)
{
  const int hor = (n - 1 - x) * left[y + 1] + (x + 1) * top[n + 1];
  const int ver = (n - 1 - y) * top[x + 1] + (y + 1) * left[n + 1];
  return (hor + ver + n) >> (log2_width + 1);
}

// one pixel of an angular prediction in the vertical orientation: row r, column c; e = &ext[N]
__device__ __forceinline__ int ang_px(const u8 *e, int disp, int r, int c)
{
  const int pos = (r + 1) * disp, di = pos >> 5, f = pos & 31;
  return ((32 - f) * e[c + di] + f * e[c + di + 1] + 16) >> 5;
}

// intra_post_process_angular (intra.c:195-208) on column 0 of row r (vertical orientation)
__device__ __forceinline__ int post_px(int v, const u8 *side, int r)
{
  return clampi(v + (((int)side[r + 1] - (int)side[0]) >> 1), 0, 255);
}

struct mode_list { signed char m[36]; };

// --------------------------------------------------------------------------------------------
// predictions to memory: one wave per (PU, mode slot)
// --------------------------------------------------------------------------------------------
template <int LOG2>
__global__ __launch_bounds__(256) void intra_predict_kernel(const kvz_hip_intra_ref *__restrict__ refs, size_t count, mode_list modes,
                                                           int num_modes, int flags, u8 *__restrict__ dst)
{
  constexpr int N = 1 << LOG2, ES = 3 * N + 4;
  __shared__ __align__(16) u8 s_ref[4][4][RS];
  __shared__ __align__(16) u8 s_ext[4][ES];
  // the wave index is uniform: say so, or every mode-dependent value becomes a per-lane quantity
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t items = count * (size_t)num_modes;
  for (size_t base = (size_t)blockIdx.x * 4; base < items; base += (size_t)gridDim.x * 4) {
    // the four waves of the workgroup take four consecutive items; stage their PUs' references
    __syncthreads();
    {
      u32 *z = (u32 *)s_ref;
      for (int i = threadIdx.x; i < 4 * 4 * RS / 4; i += 256) z[i] = 0u;
    }
    __syncthreads();
    const size_t item = base + w;
    const bool live = item < items;
    const size_t pu = live ? item / num_modes : 0;
    const int mode = live ? modes.m[item - pu * num_modes] : 0;
    {
      const u8 *g = (const u8 *)(refs + pu);
      for (int i = lane; i < 130; i += 64) {
        const int a = i >= 65, k = i - 65 * a;
        if (k <= 2 * N) s_ref[w][a][k] = g[i];
      }
    }
    wave_lds_fence();
    for (int i = lane; i < 2 * (2 * N + 1); i += 64) {
      const int a = i >= 2 * N + 1, k = i - (2 * N + 1) * a;
      const u8 *src = s_ref[w][a];
      int v;
      if (k == 0) v = (s_ref[w][0][1] + 2 * s_ref[w][0][0] + s_ref[w][1][1] + 2) >> 2;
      else if (k == 2 * N) v = src[k];
      else v = (src[k - 1] + 2 * src[k] + src[k + 1] + 2) >> 2;
      s_ref[w][2 + a][k] = (u8)v;
    }
    wave_lds_fence();
    const bool fil = use_filtered(mode, LOG2, flags);
    const bool edge = luma_edge_filters(LOG2, flags);
    u8 *out = dst + item * (size_t)(N * N);
    if (mode >= 2) {
      const ang_t a = ang_of(mode);
      for (int k = lane; k < 3 * N + 2; k += 64) s_ext[w][k] = ext_entry<N>(s_ref[w], fil, a, k - N);
      wave_lds_fence();
      const u8 *e = &s_ext[w][N];
      const u8 *side = s_ref[w][2 * fil + (a.vertical ? 0 : 1)];
      const bool pp = edge && (flags & KVZ_HIP_INTRA_FILTER_BOUNDARY) && a.disp == 0;
      if (live)
        for (int px = lane; px < N * N; px += 64) {
          const int y = px >> LOG2, x = px & (N - 1);
          const int r = a.vertical ? y : x, c = a.vertical ? x : y;
          int v = ang_px(e, a.disp, r, c);
          if (pp && c == 0) v = post_px(v, side, r);
          out[px] = (u8)v;
        }
    } else if (mode == 1) {
      const int dc = dc_value(s_ref[w], N, LOG2);
      if (live)
        for (int px = lane; px < N * N; px += 64) out[px] = (u8)dc_px(s_ref[w], dc, edge, px & (N - 1), px >> LOG2);
    } else {
      const u8 *left = s_ref[w][2 * fil], *top = s_ref[w][2 * fil + 1];
      if (live)
        for (int px = lane; px < N * N; px += 64) out[px] = (u8)planar_px(left, top, N, LOG2, px & (N - 1), px >> LOG2);
    }
  }
}

// --------------------------------------------------------------------------------------------
// rough search: 35 mode costs per PU.  A lane owns one NB x NB sub-block (NB = 8, or 4 for 4x4
// PUs) of one PU for the whole kernel and keeps the matching original pixels (and their
// transpose) in registers; the four waves of a workgroup walk the modes 4 apart, so the mode --
// and with it every branch and the extended-reference build -- is uniform across a wave.
// --------------------------------------------------------------------------------------------
template <int NB> struct sub_block;
template <> struct sub_block<8> {
  v2s x[8][4];
  __device__ __forceinline__ u32 satd() { return satd8x8_diff(x); }
};
template <> struct sub_block<4> {
  v2s x[4][2];
  __device__ __forceinline__ u32 satd() { return satd4x4_diff(x); }
};

template <int NB>
__device__ __forceinline__ u32 sad_of(const sub_block<NB> &b)
{
  v2us acc = { 0, 0 };
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) {
      const v2s n = -b.x[r][q];
      acc += __builtin_bit_cast(v2us, __builtin_elementwise_max(b.x[r][q], n));
    }
  return (u32)acc.x + (u32)acc.y;
}

// NW waves per workgroup walk the 35 modes NW apart
template <int LOG2, bool WITH_SAD, int NW>
__global__ __launch_bounds__(64 * NW) void intra_rough_kernel(const kvz_hip_intra_ref *__restrict__ refs, const u8 *__restrict__ orig,
                                                         size_t count, int flags, u32 *__restrict__ satd_out, u32 *__restrict__ sad_out)
{
  constexpr int N = 1 << LOG2, NB = N < 8 ? 4 : 8, SB = N / NB, S = SB * SB, G = 64 / S;
  constexpr int ES = 2 * N + 4;          // per-wave projected reference e[-N .. N-1] (+ pad: odd dword stride)
  constexpr int OS = N * N + 8;
  constexpr int NQ = NB / 2;             // packed pairs per sub-block row
  __shared__ __align__(16) u8 s_ref[G][4][RS];
  __shared__ __align__(16) u8 s_orig[G][OS];
  __shared__ __align__(16) u8 s_ext[NW][G][ES];
  __shared__ int s_dc[G];
  __shared__ u32 s_cost[WITH_SAD ? 2 : 1][G][35];

  // the wave index (and with it the mode) is uniform: say so, or mode-dependent values, branches and table
  // look-ups are compiled per lane
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t pu0 = (size_t)blockIdx.x * G;
  stage_refs<N>(s_ref, G, refs, pu0, count, tid, 64 * NW);
  {
    const size_t left = count - pu0;
    const int avail = (int)(left < (size_t)G ? left : (size_t)G);
    const uint2 *g = (const uint2 *)(orig + pu0 * (size_t)(N * N));
    for (int i = tid; i < G * N * N / 8; i += 64 * NW) {
      const int p = i / (N * N / 8), o = (i - p * (N * N / 8)) * 8;
      uint2 v = { 0u, 0u };
      if (p < avail) v = g[i];
      *(uint2 *)&s_orig[p][o] = v;
    }
    if (tid < G) s_dc[tid] = dc_value(s_ref[tid], N, LOG2);
  }
  __syncthreads();

  const int p = lane / S, sub = lane % S, bx = sub % SB, by = sub / SB;
  const int gx0 = bx * NB, gy0 = by * NB;
  // original pixels of the lane's sub-block, and of the transposed block's sub-block at the same place
  u32 o[NB * NB / 4], ot[NB * NB / 4];
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const u8 *row = &s_orig[p][(gy0 + r) * N + gx0];
    if (NB == 8) { const uint2 v = *(const uint2 *)row; o[2 * r] = v.x; o[2 * r + 1] = v.y; }
    else o[r] = *(const u32 *)row;
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) {
      u32 d = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) d |= (u32)s_orig[p][(gx0 + 4 * q + k) * N + gy0 + r] << (8 * k);
      ot[r * (NB / 4) + q] = d;
    }
  }
  const bool edge = luma_edge_filters(LOG2, flags);
  const u8 (*ref)[RS] = s_ref[p];

  for (int mode = w; mode < 35; mode += NW) {
    sub_block<NB> b;
    const bool fil = use_filtered(mode, LOG2, flags);
    if (mode >= 2) {
      const ang_t a = ang_of(mode);
      // Modes that lean away from the side reference (disp >= 0) read the main reference as it
      // is; only the others need the side reference projected below index -1, and they never
      // read beyond index N-1: a 2N-entry per-wave array e[-N .. N-1].
      // The row reads below fetch NB+1 consecutive bytes at an arbitrary byte offset.  Unaligned wide DS
      // reads are replayed by the hardware (SQ_LDS_UNALIGNED_STALL was 2/3 of the LDS time), so aligned
      // dwords are read and shifted into place with v_alignbyte.  Both arrays are 16-byte aligned.
      const u32 *words = (const u32 *)&s_ref[0][0][0];
      int boff = (p * 4 + 2 * fil + (a.vertical ? 1 : 0)) * RS + 1 + gx0;
      if (a.disp < 0) {
        wave_lds_fence();
        for (int i = lane; i < G * 2 * N; i += 64) {
          const int pp = i / (2 * N), k = i - pp * (2 * N);
          s_ext[w][pp][k] = ext_entry<N>(s_ref[pp], fil, a, k - N);
        }
        wave_lds_fence();
        words = (const u32 *)&s_ext[0][0][0];
        boff = (w * G + p) * ES + N + gx0;
      }
      const u8 *side = ref[2 * fil + (a.vertical ? 0 : 1)];
      const bool post = edge && (flags & KVZ_HIP_INTRA_FILTER_BOUNDARY) && a.disp == 0 && gx0 == 0;
      const u32 *src = a.vertical ? o : ot;
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const int pos = (gy0 + r + 1) * a.disp, di = pos >> 5, f = pos & 31;
        const int at = boff + di;
        const u32 *q0 = words + (at >> 2);
        const u32 sh = (u32)at & 3u;
        u32 raw[NB / 4 + 1], d[NB / 4 + 1];
#pragma unroll
        for (int k = 0; k <= NB / 4; ++k) raw[k] = q0[k];
#pragma unroll
        for (int k = 0; k < NB / 4; ++k) d[k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], sh);
        d[NB / 4] = __builtin_amdgcn_alignbyte(0u, raw[NB / 4], sh);
        const u32 wf = (u32)f * 0x10001u;
        const v2us w1 = __builtin_bit_cast(v2us, wf), w0 = __builtin_bit_cast(v2us, 0x00200020u - wf);
        const v2us rnd = { 16, 16 };
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const u32 lo = d[q >> 1], hi = d[(q >> 1) + 1];
          const v2us pa = __builtin_bit_cast(v2us, (q & 1) ? __builtin_amdgcn_perm(0u, lo, 0x0c030c02u) : __builtin_amdgcn_perm(0u, lo, 0x0c010c00u));
          const v2us pb = __builtin_bit_cast(v2us, (q & 1) ? __builtin_amdgcn_perm(hi, lo, 0x0c040c03u) : __builtin_amdgcn_perm(0u, lo, 0x0c020c01u));
          v2us v = (pa * w0 + pb * w1 + rnd) >> 5;
          if (q == 0 && post) v.x = (unsigned short)post_px(v.x, side, gy0 + r);
          const u32 od = src[(r * NB + 2 * q) >> 2];
          const v2s ov = (q & 1) ? unpack_hi(od) : unpack_lo(od);
          b.x[r][q] = __builtin_bit_cast(v2s, v) - ov;
        }
      }
    } else {
      const int dc = s_dc[p];
      const u8 *left = ref[2 * fil], *top = ref[2 * fil + 1];
#pragma unroll
      for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int x = gx0 + 2 * q, y = gy0 + r;
          int v0, v1;
          if (mode == 1) { v0 = dc_px(ref, dc, edge, x, y); v1 = dc_px(ref, dc, edge, x + 1, y); }
          else { v0 = planar_px(left, top, N, LOG2, x, y); v1 = planar_px(left, top, N, LOG2, x + 1, y); }
          const u32 od = o[(r * NB + 2 * q) >> 2];
          const v2s ov = (q & 1) ? unpack_hi(od) : unpack_lo(od);
          const v2s pv = { (short)v0, (short)v1 };
          b.x[r][q] = pv - ov;
        }
    }
    if (WITH_SAD) {
      const u32 sd = group_sum<S>(sad_of<NB>(b));
      if (sub == 0) s_cost[WITH_SAD ? 1 : 0][p][mode] = sd;
    }
    const u32 c = group_sum<S>(b.satd());
    if (sub == 0) s_cost[0][p][mode] = c;
  }
  __syncthreads();
  const size_t total = count * 35, base = pu0 * 35;
  for (int i = tid; i < G * 35; i += 64 * NW)
    if (base + i < total) {
      satd_out[base + i] = (&s_cost[0][0][0])[i];
      if (WITH_SAD) sad_out[base + i] = (&s_cost[WITH_SAD ? 1 : 0][0][0])[i];
    }
}

template <int LOG2>
int launch_rough(const kvz_hip_intra_ref *refs, const u8 *orig, size_t count, int flags, u32 *satd, u32 *sad, hipStream_t st)
{
  constexpr int N = 1 << LOG2, NB = N < 8 ? 4 : 8, S = (N / NB) * (N / NB), G = 64 / S;
  const size_t wgs = (count + G - 1) / G;
  if (wgs > 0x7fffffffu) return kvzhip::invalid_arg("kvz_hip_intra_rough_batch");
  // 4x4 PUs: 64 PUs per workgroup make it LDS-limited (31 KB), eight waves per workgroup fill the CU's wave slots
  // (3.4 against 2.7 G PUs/s); the larger sizes are register-limited and lose a third with eight (measured)
  const int nw = tuning("intra_rough_waves", LOG2 == 2 ? 8 : 4);
  if (nw == 8) {
    if (sad) hipLaunchKernelGGL((intra_rough_kernel<LOG2, true, 8>), dim3((unsigned)wgs), dim3(512), 0, st, refs, orig, count, flags, satd, sad);
    else hipLaunchKernelGGL((intra_rough_kernel<LOG2, false, 8>), dim3((unsigned)wgs), dim3(512), 0, st, refs, orig, count, flags, satd, sad);
  } else {
    if (sad) hipLaunchKernelGGL((intra_rough_kernel<LOG2, true, 4>), dim3((unsigned)wgs), dim3(256), 0, st, refs, orig, count, flags, satd, sad);
    else hipLaunchKernelGGL((intra_rough_kernel<LOG2, false, 4>), dim3((unsigned)wgs), dim3(256), 0, st, refs, orig, count, flags, satd, sad);
  }
  KVZ_CHECK_LAUNCH("intra_rough_kernel");
  return KVZ_HIP_OK;
}

template <int LOG2>
int launch_predict(const kvz_hip_intra_ref *refs, size_t count, const mode_list &ml, int num_modes, int flags, u8 *dst, hipStream_t st)
{
  const size_t items = count * (size_t)num_modes;
  const unsigned grid = stream_grid(items, 4, 16);
  hipLaunchKernelGGL((intra_predict_kernel<LOG2>), dim3(grid), dim3(256), 0, st, refs, count, ml, num_modes, flags, dst);
  KVZ_CHECK_LAUNCH("intra_predict_kernel");
  return KVZ_HIP_OK;
}

// ---- kvz_intra_build_reference (intra.c:334-588) from the reconstruction plane ----
// num_ref_pixels_left / num_ref_pixels_top (intra.c:35-70) in closed form.  (ux, uy) = the PU's 4x4 unit inside
// its LCU; s = the largest power of two dividing the coordinate (16 on the LCU border).  Everything left of
// the unit down to the end of its s-aligned group was coded before it, and so was everything above it up to
// the end of the enclosing 2s-aligned group, never more than 64 pixels.
__device__ __forceinline__ int intra_coded_left(int ux, int uy)
{
  const int s = ux ? (ux & -ux) : 16;
  return 4 * (s - (uy & (s - 1)));
}
__device__ __forceinline__ int intra_coded_above(int ux, int uy)
{
  const int s2 = uy ? 2 * (uy & -uy) : 32;
  const int n = 4 * (s2 - (ux & (s2 - 1)));
  return n < 64 ? n : 64;
}

// L = max(2N, 16) lanes per PU, 64 / L PUs per wave (a wave per PU spent 25 us on the 129 600 4x4 PUs of a frame: 129 600
// waves for 18 useful bytes each).  Lane l < 2N of a PU's group owns left[1 + l] and top[1 + l], lane 0 also the corner;
// entries past 2N are written as zero so a record is a function of its inputs alone.  A position outside the picture, or
// off the 4-pixel grid, gives an all-zero record and reads nothing.
template <int LOG2>
__global__ __launch_bounds__(256) void intra_build_reference_kernel(const u8 *__restrict__ rec, int stride, int pic_w, int pic_h,
                                                                    const kvz_hip_intra_pos *__restrict__ pus, size_t count,
                                                                    int chroma, kvz_hip_intra_ref *__restrict__ refs)
{
  constexpr int n2 = 2 << LOG2, L = n2 < 16 ? 16 : n2, PER_WAVE = 64 / L;
  const int lane = threadIdx.x & 63, l = lane & (L - 1);
  const size_t i = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PER_WAVE + lane / L;
  if (i >= count) return;
  const int lx = pus[i].x, ly = pus[i].y;
  u8 *out = reinterpret_cast<u8 *>(refs + i);
  const int span = (1 << LOG2) << chroma;
  const bool ok = lx >= 0 && ly >= 0 && ((lx | ly) & 3) == 0 && lx + span <= pic_w && ly + span <= pic_h;
  u8 left = 0, top = 0, corner = 0;
  if (ok && l < n2) {
    const int x = lx >> chroma, y = ly >> chroma;
    const int ux = (lx & 63) >> 2, uy = (ly & 63) >> 2;
    const bool has_left = lx > 0, has_top = ly > 0;
    const u8 *above = rec + (size_t)(has_top ? y - 1 : 0) * stride;   // the row over the PU
    // intra.c:386-412 / :519-543
    if (has_left) {
      int avail = intra_coded_left(ux, uy) >> chroma;
      avail = min(avail, min(n2, (pic_h - ly) >> chroma));
      left = rec[(size_t)(y + min(l, avail - 1)) * stride + x - 1];
    } else {
      left = has_top ? above[x] : 128;
    }
    // intra.c:430-455 / :545-571
    if (has_top) {
      int avail = intra_coded_above(ux, uy) >> chroma;
      avail = min(avail, min(n2, (pic_w - lx) >> chroma));
      top = above[x + min(l, avail - 1)];
    } else {
      top = has_left ? rec[(size_t)y * stride + x - 1] : 128;
    }
    // intra.c:414-428 / :504-517; left[1] is the same value in every lane wherever it stands in for the corner
    if (l == 0) corner = (has_left && has_top) ? above[x - 1] : (has_left ? rec[(size_t)y * stride + x - 1] : left);
  }
  // entries 1 .. 64 of both arrays: the group's lanes stride over them (values for e < 2N, zeros behind)
  for (int e = l; e < 64; e += L) {
    out[1 + e] = e == l ? left : 0;
    out[65 + 1 + e] = e == l ? top : 0;
  }
  if (l == 0) { out[0] = corner; out[65] = corner; }
}

}  // namespace

extern "C" {

int kvz_hip_intra_build_reference_batch(int log2_width, int color, const kvz_hip_pixel *rec, int stride, int pic_width, int pic_height,
                                        const kvz_hip_intra_pos *pus, size_t count, kvz_hip_intra_ref *refs, kvz_hip_stream stream)
{
  KVZ_CHECK_CTX();
  if (log2_width < 2 || log2_width > 5 || color < 0 || color > 2) {
    set_error_msg("kvz_hip_intra_build_reference_batch: log2_width 2..5 and color 0..2 required");
    return KVZ_HIP_ERR_INVALID;
  }
  const int chroma = color != 0;
  if (pic_width <= 0 || pic_height <= 0 || ((pic_width | pic_height) & 7) || stride < (pic_width >> chroma)) {
    set_error_msg("kvz_hip_intra_build_reference_batch: picture size must be a positive multiple of 8 and fit the stride");
    return KVZ_HIP_ERR_INVALID;
  }
  if (count == 0) return KVZ_HIP_OK;
  if (!rec || !pus || !refs) { set_error_msg("kvz_hip_intra_build_reference_batch: null buffer"); return KVZ_HIP_ERR_INVALID; }
  if (count > (size_t)0x7fffffff * 4) { set_error_msg("kvz_hip_intra_build_reference_batch: count too large"); return KVZ_HIP_ERR_INVALID; }
  hipStream_t st = ctx_stream(stream);
  const unsigned per_wg = 4u * (64u / (unsigned)((2 << log2_width) < 16 ? 16 : (2 << log2_width)));   // PUs per workgroup of four waves
  const dim3 grid((unsigned)((count + per_wg - 1) / per_wg)), block(256);
  switch (log2_width) {
    case 2: hipLaunchKernelGGL(intra_build_reference_kernel<2>, grid, block, 0, st, rec, stride, pic_width, pic_height, pus, count, chroma, refs); break;
    case 3: hipLaunchKernelGGL(intra_build_reference_kernel<3>, grid, block, 0, st, rec, stride, pic_width, pic_height, pus, count, chroma, refs); break;
    case 4: hipLaunchKernelGGL(intra_build_reference_kernel<4>, grid, block, 0, st, rec, stride, pic_width, pic_height, pus, count, chroma, refs); break;
    default: hipLaunchKernelGGL(intra_build_reference_kernel<5>, grid, block, 0, st, rec, stride, pic_width, pic_height, pus, count, chroma, refs); break;
  }
  KVZ_CHECK_LAUNCH("intra_build_reference_kernel");
  return KVZ_HIP_OK;
}


int kvz_hip_intra_predict_batch(int log2_width, int flags, const kvz_hip_intra_ref *refs, size_t count, const int8_t *modes,
                                int num_modes, kvz_hip_pixel *dst, kvz_hip_stream stream)
{
  KVZ_CHECK_CTX();
  if (log2_width < 2 || log2_width > 5 || num_modes < 1 || num_modes > 35 || !modes) {
    set_error_msg("kvz_hip_intra_predict_batch: log2_width 2..5 and 1..35 modes required");
    return KVZ_HIP_ERR_INVALID;
  }
  mode_list ml;
  for (int i = 0; i < num_modes; ++i) {
    if (modes[i] < 0 || modes[i] > 34) { set_error_msg("kvz_hip_intra_predict_batch: mode outside 0..34"); return KVZ_HIP_ERR_INVALID; }
    ml.m[i] = modes[i];
  }
  if (count == 0) return KVZ_HIP_OK;
  if (!refs || !dst) { set_error_msg("kvz_hip_intra_predict_batch: null buffer"); return KVZ_HIP_ERR_INVALID; }
  hipStream_t st = ctx_stream(stream);
  switch (log2_width) {
    case 2: return launch_predict<2>(refs, count, ml, num_modes, flags, dst, st);
    case 3: return launch_predict<3>(refs, count, ml, num_modes, flags, dst, st);
    case 4: return launch_predict<4>(refs, count, ml, num_modes, flags, dst, st);
    default: return launch_predict<5>(refs, count, ml, num_modes, flags, dst, st);
  }
}

int kvz_hip_intra_rough_batch(int log2_width, int flags, const kvz_hip_intra_ref *refs, const kvz_hip_pixel *orig, size_t count,
                              uint32_t *satd_costs, uint32_t *sad_costs, kvz_hip_stream stream)
{
  KVZ_CHECK_CTX();
  if (log2_width < 2 || log2_width > 5) { set_error_msg("kvz_hip_intra_rough_batch: log2_width must be 2..5"); return KVZ_HIP_ERR_INVALID; }
  if (flags & KVZ_HIP_INTRA_RAW) { set_error_msg("kvz_hip_intra_rough_batch: KVZ_HIP_INTRA_RAW is not a search mode"); return KVZ_HIP_ERR_INVALID; }
  if (count == 0) return KVZ_HIP_OK;
  if (!refs || !orig || !satd_costs) { set_error_msg("kvz_hip_intra_rough_batch: null buffer"); return KVZ_HIP_ERR_INVALID; }
  if (((uintptr_t)orig) & 7) { set_error_msg("kvz_hip_intra_rough_batch: orig must be 8-byte aligned"); return KVZ_HIP_ERR_INVALID; }
  hipStream_t st = ctx_stream(stream);
  switch (log2_width) {
    case 2: return launch_rough<2>(refs, orig, count, flags, satd_costs, sad_costs, st);
    case 3: return launch_rough<3>(refs, orig, count, flags, satd_costs, sad_costs, st);
    case 4: return launch_rough<4>(refs, orig, count, flags, satd_costs, sad_costs, st);
    default: return launch_rough<5>(refs, orig, count, flags, satd_costs, sad_costs, st);
  }
}

}  // extern "C"
