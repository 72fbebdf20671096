// strategy.hip -- the drop-in boundary: per-call strategy functions with the
// reference's exact typedefs (host pointers in, result visible on return) and
// the registration hooks kvz_strategy_register_<group>_hip.
//
// Reference interface replaced: src/strategyselector.h:86-87,
// strategies/strategies-picture.h:102-130, strategies-dct.h:31,
// strategies-quant.h:36-48, strategies-ipol.h:34-47; model for the hooks:
// strategies/avx2/picture-avx2.c:1224-1258.
//
// Each call stages its operands through a per-thread pinned buffer to HBM,
// launches the same kernels as the batched entries with a batch of one on a
// per-thread stream, and waits: correct and thread-safe (the function pointers
// are called concurrently from all threadqueue workers, encoderstate.c:781) but
// launch-latency bound -- throughput comes from the batched entries.
#include "kvz_hip_internal.h"

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>

using namespace kvzhip;

// the host encoder's registry function; resolved at load time when present
extern "C" int kvz_strategyselector_register(void *opaque, const char *type, const char *strategy_name, int priority, void *fptr)
    __attribute__((weak, visibility("default")));

namespace kvzhip {
int launch_frac_step(const u8 *win, int w, int h, int step, int fme_level, int hx, int hy,
                     u8 *filtered, i16 *hor_out, i16 *cols_out, hipStream_t st);
int launch_extend_block(const u8 *rect, int rw, int rh, int ox, int oy, u8 *out, int ow, int oh, hipStream_t st);
}

namespace {

kvz_hip_register_fn g_registrar = nullptr;
kvz_hip_state_accessors g_acc;
bool g_have_acc = false;

struct call_ctx {
  hipStream_t st = nullptr;
  u8 *h = nullptr;       // pinned host staging
  u8 *d = nullptr;       // device staging (same layout)
  size_t cap = 0;
  bool ok = false;
  bool zero_copy = false;
  call_ctx()
  {
    if (!ctx_enter() && (kvz_hip_init(-1) != KVZ_HIP_OK || !ctx_enter())) return;   // also binds this thread to the device
    cap = 1u << 20;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return;
    if (hipHostMalloc((void **)&h, cap, hipHostMallocDefault) != hipSuccess) return;
    // By default the kernels read and write the pinned staging buffer directly (no staging copies): a per-call operand
    // set is a few hundred bytes, so the PCIe accesses are latency, not bandwidth -- 14 us per call against 20 us with
    // explicit copies (tools/percall_latency.py).  KVZ_HIP_ZEROCOPY=0 restores the copies into device memory.
    const char *zc = std::getenv("KVZ_HIP_ZEROCOPY");
    zero_copy = !(zc && zc[0] == '0');
    if (zero_copy) d = h;
    else if (hipMalloc((void **)&d, cap) != hipSuccess) return;
    ok = true;
  }
  ~call_ctx()
  {
    if (d && !zero_copy) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    if (st) (void)hipStreamDestroy(st);
  }
};

// one staging context per (host thread, device): a worker that switches devices with kvz_hip_set_device keeps both
call_ctx &tls()
{
  static thread_local call_ctx *per_dev[64] = { nullptr };
  struct reaper { call_ctx **v; ~reaper() { for (int i = 0; i < 64; ++i) delete v[i]; } };
  static thread_local reaper r{ per_dev };
  if (!ctx_enter()) (void)kvz_hip_init(-1);
  const int d = ctx_device();
  if (d < 0 || d >= 64) {
    std::fprintf(stderr, "kvzhip: no GPU context in a strategy call: %s\n", kvz_hip_last_error());
    std::abort();
  }
  if (!per_dev[d]) per_dev[d] = new call_ctx();
  call_ctx &c = *per_dev[d];
  if (!c.ok) {
    // A strategy function has no error channel: never return garbage.
    std::fprintf(stderr, "kvzhip: GPU context unavailable in a strategy call: %s\n", kvz_hip_last_error());
    std::abort();
  }
  return c;
}

void die(const char *what, int rc)
{
  std::fprintf(stderr, "kvzhip: %s failed (rc=%d): %s\n", what, rc, kvz_hip_last_error());
  std::abort();
}
#define MUST(call) do { int rc__ = (call); if (rc__ != KVZ_HIP_OK) die(#call, rc__); } while (0)
#define HMUST(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { set_error(#call, e__); die(#call, KVZ_HIP_ERR_RUNTIME); } } while (0)

inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }
inline int clampi_host(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

std::atomic<unsigned long long> g_dropin_calls{0};

// bump allocator over the staging buffer; one per strategy call that reaches the GPU
struct stage {
  call_ctx &c;
  size_t off = 0;
  explicit stage(call_ctx &cc) : c(cc) { g_dropin_calls.fetch_add(1, std::memory_order_relaxed); }
  size_t take(size_t bytes)
  {
    size_t o = off;
    off = up16(off + bytes);
    if (off > c.cap) die("staging buffer overflow", KVZ_HIP_ERR_INVALID);
    return o;
  }
  void h2d(size_t o, size_t bytes) { if (!c.zero_copy) HMUST(hipMemcpyAsync(c.d + o, c.h + o, bytes, hipMemcpyHostToDevice, c.st)); }
  void d2h(size_t o, size_t bytes) { if (!c.zero_copy) HMUST(hipMemcpyAsync(c.h + o, c.d + o, bytes, hipMemcpyDeviceToHost, c.st)); }
  void sync() { HMUST(hipStreamSynchronize(c.st)); }
};

void pack_rows(u8 *dst, const u8 *src, int w, int h, size_t stride)
{
  for (int y = 0; y < h; ++y) std::memcpy(dst + (size_t)y * w, src + (size_t)y * stride, (size_t)w);
}

// ---------------- picture group ----------------
unsigned hip_reg_sad(const kvz_hip_pixel *d1, const kvz_hip_pixel *d2, int w, int h, unsigned s1, unsigned s2)
{
  if (w <= 0 || h <= 0) return 0;
  call_ctx &c = tls(); stage s(c);
  size_t oa = s.take((size_t)w * h), ob = s.take((size_t)w * h), od = s.take(sizeof(kvz_hip_block_pair)), in_end = s.off, oc = s.take(4);
  pack_rows(c.h + oa, d1, w, h, s1); pack_rows(c.h + ob, d2, w, h, s2);
  kvz_hip_block_pair bp = { 0, 0, 0, 0, w, h };
  std::memcpy(c.h + od, &bp, sizeof(bp));
  s.h2d(0, in_end);
  MUST(kvz_hip_reg_sad_batch(c.d + oa, (u32)w, c.d + ob, (u32)w, (const kvz_hip_block_pair *)(c.d + od), 1, (u32 *)(c.d + oc), c.st));
  s.d2h(oc, 4); s.sync();
  return *(u32 *)(c.h + oc);
}

template <int N, bool SATD>
unsigned hip_cost_nxn(const kvz_hip_pixel *b1, const kvz_hip_pixel *b2)
{
  call_ctx &c = tls(); stage s(c);
  size_t oa = s.take(N * N), ob = s.take(N * N), in_end = s.off, oc = s.take(4);
  std::memcpy(c.h + oa, b1, N * N); std::memcpy(c.h + ob, b2, N * N);
  s.h2d(0, in_end);
  if (SATD) MUST(kvz_hip_satd_nxn_batch(N, c.d + oa, c.d + ob, 1, (u32 *)(c.d + oc), c.st));
  else MUST(kvz_hip_sad_nxn_batch(N, c.d + oa, c.d + ob, 1, (u32 *)(c.d + oc), c.st));
  s.d2h(oc, 4); s.sync();
  return *(u32 *)(c.h + oc);
}

// cost_pixel_nxn_multi_func: preds is kvz_pixel (*)[32*32]; preds[1] is 1024 bytes after preds[0]
template <int N, bool SATD>
void hip_cost_nxn_dual(const kvz_hip_pixel (*preds)[32 * 32], const kvz_hip_pixel *orig, unsigned num_modes, unsigned *costs_out)
{
  (void)num_modes;
  call_ctx &c = tls(); stage s(c);
  // N == 64 overlaps the two predictions in the reference's buffer (64*64 > 1024); copy what the generic code reads
  const size_t pbytes = 1024 + (size_t)N * N;
  size_t op = s.take(pbytes), oo = s.take(N * N), in_end = s.off, oc = s.take(8);
  std::memcpy(c.h + op, preds, pbytes); std::memcpy(c.h + oo, orig, N * N);
  s.h2d(0, in_end);
  if (SATD) MUST(kvz_hip_satd_nxn_dual_batch(N, c.d + op, 1024, up16(pbytes), c.d + oo, 1, (u32 *)(c.d + oc), c.st));
  else MUST(kvz_hip_sad_nxn_dual_batch(N, c.d + op, 1024, up16(pbytes), c.d + oo, 1, (u32 *)(c.d + oc), c.st));
  s.d2h(oc, 8); s.sync();
  costs_out[0] = ((u32 *)(c.h + oc))[0]; costs_out[1] = ((u32 *)(c.h + oc))[1];
}

unsigned hip_satd_any_size(int w, int h, const kvz_hip_pixel *b1, int s1, const kvz_hip_pixel *b2, int s2)
{
  if (w <= 0 || h <= 0) return 0;
  call_ctx &c = tls(); stage s(c);
  size_t oa = s.take((size_t)w * h), ob = s.take((size_t)w * h), od = s.take(sizeof(kvz_hip_block_pair)), in_end = s.off, oc = s.take(4);
  pack_rows(c.h + oa, b1, w, h, (size_t)s1); pack_rows(c.h + ob, b2, w, h, (size_t)s2);
  kvz_hip_block_pair bp = { 0, 0, 0, 0, w, h };
  std::memcpy(c.h + od, &bp, sizeof(bp));
  s.h2d(0, in_end);
  // both planes are the packed blocks themselves: clamp extents == block => never clamps
  MUST(kvz_hip_image_calc_satd_batch(c.d + oa, (u32)w, c.d + ob, (u32)w, w, h, (const kvz_hip_block_pair *)(c.d + od), 1, (u32 *)(c.d + oc), c.st));
  s.d2h(oc, 4); s.sync();
  return *(u32 *)(c.h + oc);
}

void hip_satd_any_size_quad(int w, int h, const kvz_hip_pixel **preds, const int stride, const kvz_hip_pixel *orig,
                            const int orig_stride, unsigned num_modes, unsigned *costs_out, int8_t *valid)
{
  (void)num_modes; (void)valid;
  costs_out[0] = costs_out[1] = costs_out[2] = costs_out[3] = 0;
  if (w <= 0 || h <= 0) return;
  call_ctx &c = tls(); stage s(c);
  const size_t bsz = up16((size_t)w * h);
  size_t op = s.take(bsz * 4), oo = s.take((size_t)w * h), od = s.take(sizeof(kvz_hip_block_pair)), in_end = s.off, oc = s.take(16);
  for (int k = 0; k < 4; ++k) pack_rows(c.h + op + k * bsz, preds[k], w, h, (size_t)stride);
  pack_rows(c.h + oo, orig, w, h, (size_t)orig_stride);
  kvz_hip_block_pair bp = { 0, 0, 0, 0, w, h };
  std::memcpy(c.h + od, &bp, sizeof(bp));
  s.h2d(0, in_end);
  MUST(kvz_hip_satd_any_size_quad_batch(c.d + op, (u32)w, bsz, c.d + oo, (u32)w, (const kvz_hip_block_pair *)(c.d + od), 1, (u32 *)(c.d + oc), c.st));
  s.d2h(oc, 16); s.sync();
  for (int k = 0; k < 4; ++k) costs_out[k] = ((u32 *)(c.h + oc))[k];
}

unsigned hip_pixels_calc_ssd(const kvz_hip_pixel *ref, const kvz_hip_pixel *rec, const int ref_stride, const int rec_stride, const int width)
{
  if (width <= 0) return 0;
  call_ctx &c = tls(); stage s(c);
  const int w = width;
  size_t oa = s.take((size_t)w * w), ob = s.take((size_t)w * w), od = s.take(sizeof(kvz_hip_block_pair)), in_end = s.off, oc = s.take(4);
  pack_rows(c.h + oa, ref, w, w, (size_t)ref_stride); pack_rows(c.h + ob, rec, w, w, (size_t)rec_stride);
  kvz_hip_block_pair bp = { 0, 0, 0, 0, w, w };
  std::memcpy(c.h + od, &bp, sizeof(bp));
  s.h2d(0, in_end);
  MUST(kvz_hip_pixels_calc_ssd_batch(c.d + oa, (u32)w, c.d + ob, (u32)w, (const kvz_hip_block_pair *)(c.d + od), 1, (u32 *)(c.d + oc), c.st));
  s.d2h(oc, 4); s.sync();
  return *(u32 *)(c.h + oc);
}

// ---------------- dct group ----------------
template <int KIND, int N>
unsigned hip_transform(int8_t bitdepth, const int16_t *input, int16_t *output)
{
  (void)bitdepth;           // registered for bitdepth 8 only
  call_ctx &c = tls(); stage s(c);
  size_t oi = s.take(N * N * 2), in_end = s.off, oo = s.take(N * N * 2);
  std::memcpy(c.h + oi, input, N * N * 2);
  s.h2d(0, in_end);
  MUST(kvz_hip_transform_batch(KIND, N, (const int16_t *)(c.d + oi), (int16_t *)(c.d + oo), 1, c.st));
  s.d2h(oo, N * N * 2); s.sync();
  std::memcpy(output, c.h + oo, N * N * 2);
  return 0;
}

// ---------------- quant group ----------------
// flatten encoder_state_t through the accessors supplied by the host glue
int log2w(int w) { int l = 0; while ((1 << l) < w) ++l; return l; }

// stages the per-coefficient tables when the scaling list is enabled; returns params with DEVICE table pointers
kvz_hip_quant_params flatten_state(const void *state, int width, int type_q, int type_dq, int block_is_intra, stage &s, call_ctx &c)
{
  kvz_hip_quant_params p;
  std::memset(&p, 0, sizeof(p));
  p.qp = g_acc.qp(state);
  p.slice_is_intra = g_acc.slice_is_intra(state);
  p.signhide = g_acc.signhide_enable(state);
  p.scaling_list = g_acc.scaling_list_enable ? g_acc.scaling_list_enable(state) : 0;
  if (p.scaling_list) {
    static const unsigned char chroma_scale[58] = {
       0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,
      33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51 };
    auto sqp = [&](int type) { if (type == 0) return p.qp; int q = p.qp < 0 ? 0 : (p.qp > 57 ? 57 : p.qp); return (int)chroma_scale[q]; };
    static const int map[4] = { 0, 3, 1, 2 };        // "\0\3\1\2"[type], quant-generic.c:46
    const size_t bytes = (size_t)width * width * 4;
    const int l2 = log2w(width);
    const int32_t *qt = g_acc.quant_coeff(state, l2, (block_is_intra ? 0 : 3) + map[type_q], sqp(type_q) % 6);
    const int32_t *dt = g_acc.dequant_coeff(state, l2, (block_is_intra ? 0 : 3) + map[type_dq], sqp(type_dq) % 6);
    size_t oq = s.take(bytes), odq = s.take(bytes);
    std::memcpy(c.h + oq, qt, bytes); std::memcpy(c.h + odq, dt, bytes);
    p.quant_coeff = (const int32_t *)(c.d + oq);
    p.dequant_coeff = (const int32_t *)(c.d + odq);
  }
  return p;
}

unsigned hip_quant(const void *state, kvz_hip_coeff *coef, kvz_hip_coeff *q_coef, int32_t width, int32_t height,
                   int8_t type, int8_t scan_idx, int8_t block_type)
{
  (void)height;
  call_ctx &c = tls(); stage s(c);
  const size_t bytes = (size_t)width * width * 2;
  const int intra = (block_type == 1);          // CU_INTRA == 1 (cu.h:39-41)
  kvz_hip_quant_params p = flatten_state(state, width, type, type, intra, s, c);
  size_t oi = s.take(bytes), in_end = s.off, oo = s.take(bytes);
  std::memcpy(c.h + oi, coef, bytes);
  s.h2d(0, in_end);
  MUST(kvz_hip_quant_batch(&p, (const kvz_hip_coeff *)(c.d + oi), (kvz_hip_coeff *)(c.d + oo), width, type, scan_idx, 1, c.st));
  s.d2h(oo, bytes); s.sync();
  std::memcpy(q_coef, c.h + oo, bytes);
  return 0;
}

unsigned hip_dequant(const void *state, kvz_hip_coeff *q_coef, kvz_hip_coeff *coef, int32_t width, int32_t height,
                     int8_t type, int8_t block_type)
{
  (void)height;
  call_ctx &c = tls(); stage s(c);
  const size_t bytes = (size_t)width * width * 2;
  const int intra = (block_type == 1);
  kvz_hip_quant_params p = flatten_state(state, width, type, type, intra, s, c);
  size_t oi = s.take(bytes), in_end = s.off, oo = s.take(bytes);
  std::memcpy(c.h + oi, q_coef, bytes);
  s.h2d(0, in_end);
  MUST(kvz_hip_dequant_batch(&p, (const kvz_hip_coeff *)(c.d + oi), (kvz_hip_coeff *)(c.d + oo), width, type, 1, c.st));
  s.d2h(oo, bytes); s.sync();
  std::memcpy(coef, c.h + oo, bytes);
  return 0;
}

uint32_t hip_coeff_abs_sum(const kvz_hip_coeff *coeffs, size_t length)
{
  if (length == 0) return 0;
  call_ctx &c = tls(); stage s(c);
  size_t oi = s.take(length * 2), in_end = s.off, oc = s.take(4);
  std::memcpy(c.h + oi, coeffs, length * 2);
  s.h2d(0, in_end);
  MUST(kvz_hip_coeff_abs_sum_batch((const kvz_hip_coeff *)(c.d + oi), length, 1, (u32 *)(c.d + oc), c.st));
  s.d2h(oc, 4); s.sync();
  return *(u32 *)(c.h + oc);
}

// quant_residual_func; color_t and coeff_scan_order_t are int-sized enums
unsigned hip_quantize_residual(void *state, const void *cur_cu, const int width, const int color, const int scan_order,
                               const int use_trskip, const int in_stride, const int out_stride,
                               const kvz_hip_pixel *ref_in, const kvz_hip_pixel *pred_in, kvz_hip_pixel *rec_out,
                               kvz_hip_coeff *coeff_out)
{
  call_ctx &c = tls(); stage s(c);
  const int w = width;
  const int intra = g_acc.cu_is_intra(cur_cu);
  const int tq = color == 0 ? 0 : 2, tdq = color == 0 ? 0 : (color == 1 ? 2 : 3);
  if (g_acc.rdoq_enable && g_acc.rdoq_enable(state) && (w > 4 || !(g_acc.rdoq_skip && g_acc.rdoq_skip(state)))) {
    // quant-generic.c:214-221: the quantiser is the host's kvz_rdoq (CABAC-context dependent control plane, called
    // by every strategy's quantize_residual); residual, transforms, dequantisation and reconstruction stay on the GPU
    // (registration guarantees the four rdoq accessors: kvz_strategy_register_quant_hip)
    const size_t n = (size_t)w * w;
    const bool dst = (w == 4 && color == 0 && intra);               // strategies-dct.c:66-85
    // quant-generic.c:206-212: transform skip replaces the transform pair, RDOQ still quantises
    const int fwd_kind = use_trskip ? KVZ_HIP_TRSKIP : (dst ? KVZ_HIP_DST : KVZ_HIP_DCT);
    const int inv_kind = use_trskip ? KVZ_HIP_ITRSKIP : (dst ? KVZ_HIP_IDST : KVZ_HIP_IDCT);
    kvz_hip_quant_params p = flatten_state(state, w, tq, tdq, intra, s, c);
    size_t orf = s.take(n), opr = s.take(n), in_end = s.off;
    size_t ors = s.take(n * 2), oco = s.take(n * 2);
    pack_rows(c.h + orf, ref_in, w, w, (size_t)in_stride); pack_rows(c.h + opr, pred_in, w, w, (size_t)in_stride);
    s.h2d(0, in_end);
    MUST(kvz_hip_residual_batch(c.d + orf, c.d + opr, (kvz_hip_coeff *)(c.d + ors), n, c.st));
    MUST(kvz_hip_transform_batch(fwd_kind, w, (const kvz_hip_coeff *)(c.d + ors), (kvz_hip_coeff *)(c.d + oco), 1, c.st));
    s.d2h(oco, n * 2); s.sync();
    g_acc.rdoq(state, (kvz_hip_coeff *)(c.h + oco), coeff_out, w, w, (int8_t)tq, (int8_t)scan_order, (int8_t)g_acc.cu_type(cur_cu),
               (int8_t)g_acc.cu_rdoq_tr_depth(cur_cu));
    int has = 0;
    for (size_t i = 0; i < n; ++i) has |= coeff_out[i] != 0;
    if (has) {
      size_t oq = s.take(n * 2), q_end = s.off, odq = s.take(n * 2), ore = s.take(n);
      std::memcpy(c.h + oq, coeff_out, n * 2);
      s.h2d(oq, q_end - oq);
      MUST(kvz_hip_dequant_batch(&p, (const kvz_hip_coeff *)(c.d + oq), (kvz_hip_coeff *)(c.d + odq), w, tdq, 1, c.st));
      MUST(kvz_hip_transform_batch(inv_kind, w, (const kvz_hip_coeff *)(c.d + odq), (kvz_hip_coeff *)(c.d + ors), 1, c.st));
      MUST(kvz_hip_reconstruct_batch((const kvz_hip_coeff *)(c.d + ors), c.d + opr, c.d + ore, n, c.st));
      s.d2h(ore, n); s.sync();
      for (int y = 0; y < w; ++y) std::memcpy(rec_out + (size_t)y * out_stride, c.h + ore + (size_t)y * w, (size_t)w);
    } else if (rec_out != pred_in) {
      for (int y = 0; y < w; ++y) std::memcpy(rec_out + (size_t)y * out_stride, pred_in + (size_t)y * in_stride, (size_t)w);
    }
    return (unsigned)has;
  }
  kvz_hip_quant_params p = flatten_state(state, w, tq, tdq, intra, s, c);
  size_t orf = s.take((size_t)w * w), opr = s.take((size_t)w * w), in_end = s.off;
  size_t ore = s.take((size_t)w * w), oco = s.take((size_t)w * w * 2), oha = s.take(4), out_end = s.off;
  pack_rows(c.h + orf, ref_in, w, w, (size_t)in_stride); pack_rows(c.h + opr, pred_in, w, w, (size_t)in_stride);
  s.h2d(0, in_end);
  MUST(kvz_hip_quantize_residual_batch(&p, intra, w, color, scan_order, use_trskip, c.d + orf, c.d + opr, c.d + ore,
                                       (kvz_hip_coeff *)(c.d + oco), (int32_t *)(c.d + oha), 1, c.st));
  s.d2h(ore, out_end - ore); s.sync();
  const int has = *(int32_t *)(c.h + oha);
  std::memcpy(coeff_out, c.h + oco, (size_t)w * w * 2);
  if (has || rec_out != pred_in)
    for (int y = 0; y < w; ++y) std::memcpy(rec_out + (size_t)y * out_stride, c.h + ore + (size_t)y * w, (size_t)w);
  return (unsigned)has;
}

// ---------------- ipol group ----------------
template <bool LUMA, bool OUT14>
void hip_sample(const void *encoder, kvz_hip_pixel *src, int16_t src_stride, int width, int height, void *dst,
                int16_t dst_stride, int8_t hor_flag, int8_t ver_flag, const int16_t mv[2])
{
  (void)encoder; (void)hor_flag; (void)ver_flag;
  constexpr int TAPS = LUMA ? 8 : 4, OFF = TAPS / 2 - 1;
  call_ctx &c = tls(); stage s(c);
  const int ww = width + TAPS - 1, wh = height + TAPS - 1;
  size_t ow = s.take((size_t)ww * wh), ob = s.take(sizeof(kvz_hip_ipol_block)), oof = s.take(8), in_end = s.off;
  const size_t obytes = (size_t)width * height * (OUT14 ? 2 : 1);
  size_t oo = s.take(obytes);
  pack_rows(c.h + ow, src - (ptrdiff_t)OFF * src_stride - OFF, ww, wh, (size_t)src_stride);
  kvz_hip_ipol_block b = { OFF, OFF, mv[0] & (LUMA ? 3 : 7), mv[1] & (LUMA ? 3 : 7), width, height };
  std::memcpy(c.h + ob, &b, sizeof(b));
  uint64_t zero = 0; std::memcpy(c.h + oof, &zero, 8);
  s.h2d(0, in_end);
  if (LUMA) MUST(kvz_hip_sample_luma_batch(c.d + ow, (u32)ww, ww, wh, (const kvz_hip_ipol_block *)(c.d + ob), (const uint64_t *)(c.d + oof), 1, OUT14, c.d + oo, c.st));
  else MUST(kvz_hip_sample_chroma_batch(c.d + ow, (u32)ww, ww, wh, (const kvz_hip_ipol_block *)(c.d + ob), (const uint64_t *)(c.d + oof), 1, OUT14, c.d + oo, c.st));
  s.d2h(oo, obytes); s.sync();
  const size_t esz = OUT14 ? 2 : 1;
  for (int y = 0; y < height; ++y)
    std::memcpy((u8 *)dst + (size_t)y * dst_stride * esz, c.h + oo + (size_t)y * width * esz, (size_t)width * esz);
}

// epol_func (strategies-ipol.h:41-42), kvz_get_extended_block_generic (ipol-generic.c:731-784): a window that lies inside the
// reference plane is returned as pointers into it, exactly like generic (no copy, no launch); one that leaves the plane
// is built with edge replication in a malloc'ed buffer the caller frees (malloc_used = 1).  The replication runs on the
// GPU: the part of the plane the window overlaps is staged, extend_block_kernel clamps, the result is copied back.
struct ext_block { kvz_hip_pixel *buffer; kvz_hip_pixel *orig_topleft; unsigned stride; unsigned malloc_used; };   // kvz_extended_block, strategies-ipol.h:35
unsigned hip_get_extended_block(int xpos, int ypos, int mv_x, int mv_y, int off_x, int off_y, kvz_hip_pixel *ref, int ref_width,
                                int ref_height, int filter_size, int width, int height, void *out_v)
{
  ext_block *out = (ext_block *)out_v;
  const int half = filter_size >> 1;
  const int min_y = ypos - half + off_y + mv_y, max_y = min_y + height + filter_size;
  const int min_x = xpos - half + off_x + mv_x, max_x = min_x + width + filter_size;
  out->buffer = ref + (ptrdiff_t)min_y * ref_width + min_x;
  out->stride = (unsigned)ref_width;
  out->orig_topleft = out->buffer + (ptrdiff_t)out->stride * half + half;
  out->malloc_used = 0;
  const bool oob = (min_y < 0) || (max_y >= ref_height) || (min_x < 0) || (max_x >= ref_width);
  if (!oob) return 0;
  // rows / columns the loops of :759-783 visit: ypos - half .. ypos + height + half - 1 (+ offsets), i.e. height + 2 * half
  const int ow = width + filter_size, oh = height + filter_size;
  const int nrows = height + 2 * half, ncols = width + 2 * half;
  out->buffer = (kvz_hip_pixel *)std::malloc((size_t)ow * oh);
  if (!out->buffer) { std::fprintf(stderr, "kvzhip: get_extended_block: out of memory\n"); std::abort(); }
  out->stride = (unsigned)ow;
  out->orig_topleft = out->buffer + (size_t)out->stride * half + half;
  out->malloc_used = 1;
  // the clamped source rectangle [cx0, cx1] x [cy0, cy1] (never empty: clamping maps every coordinate into the plane)
  const int cy0 = clampi_host(min_y, 0, ref_height - 1), cy1 = clampi_host(min_y + nrows - 1, 0, ref_height - 1);
  const int cx0 = clampi_host(min_x, 0, ref_width - 1), cx1 = clampi_host(min_x + ncols - 1, 0, ref_width - 1);
  const int rw = cx1 - cx0 + 1, rh = cy1 - cy0 + 1;
  call_ctx &c = tls(); stage s(c);
  size_t oi = s.take((size_t)rw * rh), in_end = s.off, oo = s.take((size_t)ncols * nrows);
  pack_rows(c.h + oi, ref + (size_t)cy0 * ref_width + cx0, rw, rh, (size_t)ref_width);
  s.h2d(0, in_end);
  MUST(launch_extend_block(c.d + oi, rw, rh, cx0 - min_x, cy0 - min_y, c.d + oo, ncols, nrows, c.st));
  s.d2h(oo, (size_t)ncols * nrows); s.sync();
  for (int y = 0; y < nrows; ++y) std::memcpy(out->buffer + (size_t)y * out->stride, c.h + oo + (size_t)y * ncols, (size_t)ncols);
  return 0;
}

// ipol_blocks_func (strategies-ipol.h:36-38): one of the four frac-search filter steps.  The caller owns
// `filtered`, `hor_intermediate`, `hor_first_cols`; every element the generic step defines is written.
template <int STEP>
void hip_filter_step(const void *encoder, kvz_hip_pixel *src, int16_t src_stride, int width, int height,
                     kvz_hip_pixel filtered[4][64 * 64], int16_t hor_intermediate[5][72 * 64], int8_t fme_level,
                     int16_t hor_first_cols[5][72], int8_t hpel_off_x, int8_t hpel_off_y)
{
  (void)encoder;
  const int w = width, h = height, ph = h + 8, pw = w + 9;
  call_ctx &c = tls(); stage s(c);
  size_t ow = s.take((size_t)pw * ph), in_end = s.off;
  size_t of = s.take((size_t)4 * w * h), oh = s.take((size_t)2 * ph * w * 2), oc = s.take((size_t)2 * ph * 2), out_end = s.off;
  pack_rows(c.h + ow, src - (ptrdiff_t)3 * src_stride - 3, pw, ph, (size_t)src_stride);
  s.h2d(0, in_end);
  MUST(launch_frac_step(c.d + ow, w, h, STEP, fme_level, hpel_off_x, hpel_off_y, c.d + of, (i16 *)(c.d + oh), (i16 *)(c.d + oc), c.st));
  s.d2h(of, out_end - of); s.sync();
  for (int k = 0; k < 4; ++k)
    for (int y = 0; y < h; ++y) std::memcpy(&filtered[k][y * 64], c.h + of + ((size_t)k * h + y) * w, (size_t)w);
  if (STEP == 0 || STEP == 2) {
    // step 0 fills hor_intermediate[0],[1] and hor_first_cols[0],[2]; step 2 fills [3],[4] and [1],[3]
    const int hslot[2] = { STEP == 0 ? 0 : 3, STEP == 0 ? 1 : 4 }, cslot[2] = { STEP == 0 ? 0 : 1, STEP == 0 ? 2 : 3 };
    const i16 *hsrc = (const i16 *)(c.h + oh), *csrc = (const i16 *)(c.h + oc);
    for (int p = 0; p < 2; ++p) {
      const int first_y = (STEP == 0 && p == 1 && fme_level <= 1) ? 1 : 0;
      for (int y = first_y; y < ph; ++y) {
        std::memcpy(&hor_intermediate[hslot[p]][y * 64], hsrc + ((size_t)p * ph + y) * w, (size_t)w * 2);
        hor_first_cols[cslot[p]][y] = csrc[p * ph + y];
      }
    }
  }
}

// ---------------- intra group ----------------
// angular_pred_func / intra_pred_planar_func (strategies-intra.h:33-45): the bare predictors on the caller's
// reference rows (entry 0 = corner, 2N+1 entries each), N*N contiguous pixels out.
void intra_call(int log2_width, int mode, const kvz_hip_pixel *ref_above, const kvz_hip_pixel *ref_left, kvz_hip_pixel *dst)
{
  const int n = 1 << log2_width;
  call_ctx &c = tls(); stage s(c);
  size_t orf = s.take(sizeof(kvz_hip_intra_ref)), in_end = s.off, oo = s.take((size_t)n * n);
  kvz_hip_intra_ref r;
  std::memset(&r, 0, sizeof(r));
  std::memcpy(r.left, ref_left, (size_t)2 * n + 1);
  std::memcpy(r.top, ref_above, (size_t)2 * n + 1);
  std::memcpy(c.h + orf, &r, sizeof(r));
  s.h2d(0, in_end);
  const int8_t m = (int8_t)mode;
  MUST(kvz_hip_intra_predict_batch(log2_width, KVZ_HIP_INTRA_RAW, (const kvz_hip_intra_ref *)(c.d + orf), 1, &m, 1, c.d + oo, c.st));
  s.d2h(oo, (size_t)n * n); s.sync();
  std::memcpy(dst, c.h + oo, (size_t)n * n);
}

void hip_angular_pred(const int_fast8_t log2_width, const int_fast8_t intra_mode, const kvz_hip_pixel *const in_ref_above,
                      const kvz_hip_pixel *const in_ref_left, kvz_hip_pixel *const dst)
{
  if (log2_width < 2 || log2_width > 5 || intra_mode < 2 || intra_mode > 34) die("angular_pred: bad arguments", KVZ_HIP_ERR_INVALID);
  intra_call(log2_width, intra_mode, in_ref_above, in_ref_left, dst);
}

void hip_intra_pred_planar(const int_fast8_t log2_width, const kvz_hip_pixel *const ref_top, const kvz_hip_pixel *const ref_left,
                           kvz_hip_pixel *const dst)
{
  if (log2_width < 2 || log2_width > 5) die("intra_pred_planar: bad arguments", KVZ_HIP_ERR_INVALID);
  intra_call(log2_width, 0, ref_top, ref_left, dst);
}

// ---------------- sao group ----------------
// sao_info_t as laid out by sao.h:42-50 (two enums and 15 ints; tests/test_oracle_vs_ref.py checks the size
// against the compiled reference)
struct sao_info_mirror { int type, eo_class, ddistortion, merge_left_flag, merge_up_flag, band_position[2], offsets[10]; };

// orig / rec are contiguous bw x bh blocks (sao-generic.c:46-109)
static void sao_stage_pair(call_ctx &c, stage &s, const kvz_hip_pixel *orig, const kvz_hip_pixel *rec, int n, size_t &oo, size_t &orr)
{
  oo = s.take((size_t)n); orr = s.take((size_t)n);
  std::memcpy(c.h + oo, orig, (size_t)n);
  std::memcpy(c.h + orr, rec, (size_t)n);
}

int hip_sao_edge_ddistortion(const kvz_hip_pixel *orig_data, const kvz_hip_pixel *rec_data, int block_width, int block_height,
                             int eo_class, int offsets[5])
{
  if (block_width < 1 || block_height < 1 || block_width > 64 || block_height > 64 || eo_class < 0 || eo_class > 3)
    die("sao_edge_ddistortion: bad arguments", KVZ_HIP_ERR_INVALID);
  call_ctx &c = tls(); stage s(c);
  size_t oo, orr;
  sao_stage_pair(c, s, orig_data, rec_data, block_width * block_height, oo, orr);
  size_t of = s.take(80), in_end = s.off, od = s.take(16);
  int32_t offs[20] = { 0 };
  for (int k = 0; k < 5; ++k) offs[eo_class * 5 + k] = offsets[k];
  std::memcpy(c.h + of, offs, sizeof(offs));
  s.h2d(0, in_end);
  MUST(kvz_hip_sao_edge_ddistortion_batch(c.d + oo, c.d + orr, block_width, block_height, 1, (const int32_t *)(c.d + of), (int32_t *)(c.d + od), c.st));
  s.d2h(od, 16); s.sync();
  return ((const int32_t *)(c.h + od))[eo_class];
}

void hip_calc_sao_edge_dir(const kvz_hip_pixel *orig_data, const kvz_hip_pixel *rec_data, int eo_class, int block_width, int block_height,
                           int cat_sum_cnt[2][5])
{
  if (block_width < 1 || block_height < 1 || block_width > 64 || block_height > 64 || eo_class < 0 || eo_class > 3)
    die("calc_sao_edge_dir: bad arguments", KVZ_HIP_ERR_INVALID);
  call_ctx &c = tls(); stage s(c);
  size_t oo, orr;
  sao_stage_pair(c, s, orig_data, rec_data, block_width * block_height, oo, orr);
  size_t in_end = s.off, od = s.take(160);
  s.h2d(0, in_end);
  MUST(kvz_hip_sao_edge_stats_batch(c.d + oo, c.d + orr, block_width, block_height, 1, (int32_t *)(c.d + od), c.st));
  s.d2h(od, 160); s.sync();
  const int32_t *r = (const int32_t *)(c.h + od) + eo_class * 10;
  for (int k = 0; k < 5; ++k) { cat_sum_cnt[0][k] += r[k]; cat_sum_cnt[1][k] += r[5 + k]; }   // the reference accumulates
}

int hip_sao_band_ddistortion(const void *state, const kvz_hip_pixel *orig_data, const kvz_hip_pixel *rec_data, int block_width,
                             int block_height, int band_pos, int sao_bands[4])
{
  (void)state;                                         // only encoder_control->bitdepth is read; the hooks register for 8 bits
  if (block_width < 1 || block_height < 1 || block_width > 64 || block_height > 64) die("sao_band_ddistortion: bad arguments", KVZ_HIP_ERR_INVALID);
  call_ctx &c = tls(); stage s(c);
  size_t oo, orr;
  sao_stage_pair(c, s, orig_data, rec_data, block_width * block_height, oo, orr);
  size_t ob = s.take(32), in_end = s.off, od = s.take(16);
  int32_t v[5] = { band_pos, sao_bands[0], sao_bands[1], sao_bands[2], sao_bands[3] };
  std::memcpy(c.h + ob, v, sizeof(v));
  s.h2d(0, in_end);
  MUST(kvz_hip_sao_band_ddistortion_batch(c.d + oo, c.d + orr, block_width, block_height, 1, (const int32_t *)(c.d + ob),
                                          (const int32_t *)(c.d + ob) + 1, (int32_t *)(c.d + od), c.st));
  s.d2h(od, 16); s.sync();
  return *(const int32_t *)(c.h + od);
}

void hip_sao_reconstruct_color(const void *encoder, const kvz_hip_pixel *rec_data, kvz_hip_pixel *new_rec_data, const void *sao_in,
                               int stride, int new_stride, int block_width, int block_height, int color_i)
{
  (void)encoder;
  if (block_width < 1 || block_height < 1) return;
  const sao_info_mirror *sao = (const sao_info_mirror *)sao_in;
  // like the reference, everything that is not band is filtered as edge (sao-generic.c:124-153); only the edge
  // filter reads around the block, and only in the directions of its class (the caller guarantees no more)
  const int band = sao->type == 1, cls = sao->eo_class & 3;
  const int rx = (!band && cls != 1) ? 1 : 0, ry = (!band && cls != 0) ? 1 : 0;
  const int ww = block_width + 2 * rx, wh = block_height + 2 * ry;
  call_ctx &c = tls(); stage s(c);
  size_t ow = s.take((size_t)ww * wh), ob = s.take(sizeof(kvz_hip_sao_block)), oi = s.take(sizeof(kvz_hip_sao_info)), in_end = s.off;
  size_t oo = s.take((size_t)ww * wh);
  pack_rows(c.h + ow, rec_data - (ptrdiff_t)ry * stride - rx, ww, wh, (size_t)stride);
  kvz_hip_sao_block b = { rx, ry, block_width, block_height, 0 };
  kvz_hip_sao_info info;
  info.type = band ? 1 : 2; info.eo_class = sao->eo_class;
  info.band_position[0] = sao->band_position[0]; info.band_position[1] = sao->band_position[1];
  for (int k = 0; k < 10; ++k) info.offsets[k] = sao->offsets[k];
  std::memcpy(c.h + ob, &b, sizeof(b));
  std::memcpy(c.h + oi, &info, sizeof(info));
  s.h2d(0, in_end);
  MUST(kvz_hip_sao_reconstruct_color_batch(c.d + ow, (u32)ww, ww, wh, c.d + oo, (u32)ww, (const kvz_hip_sao_block *)(c.d + ob), 1,
                                           (const kvz_hip_sao_info *)(c.d + oi), 1, color_i, c.st));
  s.d2h(oo, (size_t)ww * wh); s.sync();
  for (int y = 0; y < block_height; ++y)
    std::memcpy(new_rec_data + (size_t)y * new_stride, c.h + oo + (size_t)(y + ry) * ww + rx, (size_t)block_width);
}

// inter_recon_bipred_func (strategies-picture.h:117-130, picture-generic.c:538-588): the lcu_t / hi_prec_buf_t
// planes are reached through the accessor glue; luma w x h at (xpos, ypos) & 63 and chroma w/2 x h/2.
void hip_inter_recon_bipred(const int hi_prec_luma_rec0, const int hi_prec_luma_rec1, const int hi_prec_chroma_rec0,
                            const int hi_prec_chroma_rec1, int height, int width, int ypos, int xpos,
                            const void *hp0, const void *hp1, void *lcu, kvz_hip_pixel *temp_lcu_y,
                            kvz_hip_pixel *temp_lcu_u, kvz_hip_pixel *temp_lcu_v)
{
  // the caller allocates a hi_prec_buf_t only for a fractional MV (inter.c:455-458): hp0 / hp1 may be NULL
  struct plane { int hi0, hi1; const int16_t *h0, *h1; const u8 *t0; u8 *rec; int w, h, x, y, stride; };
  const plane pl[3] = {
    { hi_prec_luma_rec0, hi_prec_luma_rec1, (hp0 ? g_acc.hi_prec_y(hp0) : nullptr), (hp1 ? g_acc.hi_prec_y(hp1) : nullptr), temp_lcu_y, g_acc.lcu_rec_y(lcu), width, height, xpos & 63, ypos & 63, 64 },
    { hi_prec_chroma_rec0, hi_prec_chroma_rec1, (hp0 ? g_acc.hi_prec_u(hp0) : nullptr), (hp1 ? g_acc.hi_prec_u(hp1) : nullptr), temp_lcu_u, g_acc.lcu_rec_u(lcu), width >> 1, height >> 1, (xpos >> 1) & 31, (ypos >> 1) & 31, 32 },
    { hi_prec_chroma_rec0, hi_prec_chroma_rec1, (hp0 ? g_acc.hi_prec_v(hp0) : nullptr), (hp1 ? g_acc.hi_prec_v(hp1) : nullptr), temp_lcu_v, g_acc.lcu_rec_v(lcu), width >> 1, height >> 1, (xpos >> 1) & 31, (ypos >> 1) & 31, 32 } };
  for (const plane &p : pl) {
    if (p.w <= 0 || p.h <= 0) continue;
    call_ctx &c = tls(); stage s(c);
    const size_t n = (size_t)p.w * p.h;
    size_t o0 = s.take(n * 2), o1 = s.take(n * 2), in_end = s.off, od = s.take(n);
    // the reference wraps coordinates inside the LCU: (pos + t) & (LCU_WIDTH - 1)
    for (int y = 0; y < p.h; ++y)
      for (int x = 0; x < p.w; ++x) {
        const int yy = (p.y + y) & (p.stride - 1), xx = (p.x + x) & (p.stride - 1), i = yy * p.stride + xx;
        if (p.hi0) ((int16_t *)(c.h + o0))[y * p.w + x] = p.h0[i]; else (c.h + o0)[y * p.w + x] = p.t0[i];
        if (p.hi1) ((int16_t *)(c.h + o1))[y * p.w + x] = p.h1[i]; else (c.h + o1)[y * p.w + x] = p.rec[i];
      }
    s.h2d(0, in_end);
    MUST(kvz_hip_bipred_blend_batch(p.w, p.h, p.hi0, c.d + o0, p.hi1, c.d + o1, c.d + od, 1, c.st));
    s.d2h(od, n); s.sync();
    for (int y = 0; y < p.h; ++y)
      for (int x = 0; x < p.w; ++x) {
        const int yy = (p.y + y) & (p.stride - 1), xx = (p.x + x) & (p.stride - 1);
        p.rec[yy * p.stride + xx] = (c.h + od)[y * p.w + x];
      }
  }
}

int reg(void *opaque, const char *type, void *fptr)
{
  kvz_hip_register_fn f = g_registrar ? g_registrar : (kvz_hip_register_fn)kvz_strategyselector_register;
  if (!f) { set_error_msg("no strategy registrar: the host has no kvz_strategyselector_register and kvz_hip_set_registrar was not called"); return 0; }
  return f(opaque, type, KVZ_HIP_STRATEGY_NAME, KVZ_HIP_STRATEGY_PRIORITY, fptr);
}

bool hook_ready(uint8_t bitdepth)
{
  if (bitdepth != 8) return false;
  return kvz_hip_init(-1) == KVZ_HIP_OK;
}

}  // namespace

extern "C" {

void kvz_hip_set_registrar(kvz_hip_register_fn fn) { g_registrar = fn; }

unsigned long long kvz_hip_dropin_calls(void) { return g_dropin_calls.load(std::memory_order_relaxed); }

void kvz_hip_set_state_accessors(const kvz_hip_state_accessors *acc)
{
  if (acc) { g_acc = *acc; g_have_acc = acc->qp && acc->slice_is_intra && acc->signhide_enable && acc->cu_is_intra; }
  else g_have_acc = false;
}

// STRATEGIES_PICTURE_EXPORTS, strategies-picture.h:174-199.  inter_recon_bipred takes
// lcu_t / hi_prec_buf_t encoder structs, so it registers only when the host glue
// supplied the plane accessors.
int kvz_strategy_register_picture_hip(void *opaque, uint8_t bitdepth)
{
  if (!hook_ready(bitdepth)) return 0;
  int ok = 1;
  ok &= reg(opaque, "reg_sad", (void *)&hip_reg_sad);
  ok &= reg(opaque, "sad_4x4", (void *)&hip_cost_nxn<4, false>);
  ok &= reg(opaque, "sad_8x8", (void *)&hip_cost_nxn<8, false>);
  ok &= reg(opaque, "sad_16x16", (void *)&hip_cost_nxn<16, false>);
  ok &= reg(opaque, "sad_32x32", (void *)&hip_cost_nxn<32, false>);
  ok &= reg(opaque, "sad_64x64", (void *)&hip_cost_nxn<64, false>);
  ok &= reg(opaque, "satd_4x4", (void *)&hip_cost_nxn<4, true>);
  ok &= reg(opaque, "satd_8x8", (void *)&hip_cost_nxn<8, true>);
  ok &= reg(opaque, "satd_16x16", (void *)&hip_cost_nxn<16, true>);
  ok &= reg(opaque, "satd_32x32", (void *)&hip_cost_nxn<32, true>);
  ok &= reg(opaque, "satd_64x64", (void *)&hip_cost_nxn<64, true>);
  ok &= reg(opaque, "satd_any_size", (void *)&hip_satd_any_size);
  ok &= reg(opaque, "sad_4x4_dual", (void *)&hip_cost_nxn_dual<4, false>);
  ok &= reg(opaque, "sad_8x8_dual", (void *)&hip_cost_nxn_dual<8, false>);
  ok &= reg(opaque, "sad_16x16_dual", (void *)&hip_cost_nxn_dual<16, false>);
  ok &= reg(opaque, "sad_32x32_dual", (void *)&hip_cost_nxn_dual<32, false>);
  ok &= reg(opaque, "sad_64x64_dual", (void *)&hip_cost_nxn_dual<64, false>);
  ok &= reg(opaque, "satd_4x4_dual", (void *)&hip_cost_nxn_dual<4, true>);
  ok &= reg(opaque, "satd_8x8_dual", (void *)&hip_cost_nxn_dual<8, true>);
  ok &= reg(opaque, "satd_16x16_dual", (void *)&hip_cost_nxn_dual<16, true>);
  ok &= reg(opaque, "satd_32x32_dual", (void *)&hip_cost_nxn_dual<32, true>);
  ok &= reg(opaque, "satd_64x64_dual", (void *)&hip_cost_nxn_dual<64, true>);
  ok &= reg(opaque, "satd_any_size_quad", (void *)&hip_satd_any_size_quad);
  ok &= reg(opaque, "pixels_calc_ssd", (void *)&hip_pixels_calc_ssd);
  if (g_have_acc && g_acc.hi_prec_y && g_acc.hi_prec_u && g_acc.hi_prec_v && g_acc.lcu_rec_y && g_acc.lcu_rec_u && g_acc.lcu_rec_v)
    ok &= reg(opaque, "inter_recon_bipred", (void *)&hip_inter_recon_bipred);
  return ok;
}

// STRATEGIES_DCT_EXPORTS, strategies-dct.h:55-69
int kvz_strategy_register_dct_hip(void *opaque, uint8_t bitdepth)
{
  if (!hook_ready(bitdepth)) return 0;
  int ok = 1;
  ok &= reg(opaque, "fast_forward_dst_4x4", (void *)&hip_transform<KVZ_HIP_DST, 4>);
  ok &= reg(opaque, "dct_4x4", (void *)&hip_transform<KVZ_HIP_DCT, 4>);
  ok &= reg(opaque, "dct_8x8", (void *)&hip_transform<KVZ_HIP_DCT, 8>);
  ok &= reg(opaque, "dct_16x16", (void *)&hip_transform<KVZ_HIP_DCT, 16>);
  ok &= reg(opaque, "dct_32x32", (void *)&hip_transform<KVZ_HIP_DCT, 32>);
  ok &= reg(opaque, "fast_inverse_dst_4x4", (void *)&hip_transform<KVZ_HIP_IDST, 4>);
  ok &= reg(opaque, "idct_4x4", (void *)&hip_transform<KVZ_HIP_IDCT, 4>);
  ok &= reg(opaque, "idct_8x8", (void *)&hip_transform<KVZ_HIP_IDCT, 8>);
  ok &= reg(opaque, "idct_16x16", (void *)&hip_transform<KVZ_HIP_IDCT, 16>);
  ok &= reg(opaque, "idct_32x32", (void *)&hip_transform<KVZ_HIP_IDCT, 32>);
  return ok;
}

// STRATEGIES_QUANT_EXPORTS, strategies-quant.h:58-62.  quant / dequant /
// quantize_residual read encoder_state_t, so they register only when the host
// glue supplied accessors (kvz_hip_set_state_accessors).
int kvz_strategy_register_quant_hip(void *opaque, uint8_t bitdepth)
{
  if (!hook_ready(bitdepth)) return 0;
  int ok = 1;
  ok &= reg(opaque, "coeff_abs_sum", (void *)&hip_coeff_abs_sum);
  // quant / dequant read the state through the accessors; with a scaling-list switch they also need the two tables
  const bool tables_ok = !g_acc.scaling_list_enable || (g_acc.quant_coeff && g_acc.dequant_coeff);
  // quantize_residual additionally calls the host's kvz_rdoq when RDOQ is on: all of its accessors, or no RDOQ switch at all
  const bool rdoq_ok = !g_acc.rdoq_enable || (g_acc.rdoq && g_acc.rdoq_skip && g_acc.cu_rdoq_tr_depth && g_acc.cu_type);
  if (g_have_acc && tables_ok) {
    ok &= reg(opaque, "quant", (void *)&hip_quant);
    ok &= reg(opaque, "dequant", (void *)&hip_dequant);
    if (rdoq_ok) ok &= reg(opaque, "quantize_residual", (void *)&hip_quantize_residual);
  }
  return ok;
}

// STRATEGIES_IPOL_EXPORTS, strategies-ipol.h:65-74: the four sample filters and the four
// frac-search filter steps (each writes the caller's scratch exactly like generic; the
// throughput form is the fused kvz_hip_search_frac_batch), and get_extended_block (in-plane windows are pointers into the
// caller's plane like generic's; the edge replication of the others runs on the GPU).
int kvz_strategy_register_ipol_hip(void *opaque, uint8_t bitdepth)
{
  if (!hook_ready(bitdepth)) return 0;
  int ok = 1;
  ok &= reg(opaque, "sample_quarterpel_luma", (void *)&hip_sample<true, false>);
  ok &= reg(opaque, "sample_octpel_chroma", (void *)&hip_sample<false, false>);
  ok &= reg(opaque, "sample_14bit_quarterpel_luma", (void *)&hip_sample<true, true>);
  ok &= reg(opaque, "sample_14bit_octpel_chroma", (void *)&hip_sample<false, true>);
  ok &= reg(opaque, "filter_hpel_blocks_hor_ver_luma", (void *)&hip_filter_step<0>);
  ok &= reg(opaque, "filter_hpel_blocks_diag_luma", (void *)&hip_filter_step<1>);
  ok &= reg(opaque, "filter_qpel_blocks_hor_ver_luma", (void *)&hip_filter_step<2>);
  ok &= reg(opaque, "filter_qpel_blocks_diag_luma", (void *)&hip_filter_step<3>);
  ok &= reg(opaque, "get_extended_block", (void *)&hip_get_extended_block);
  return ok;
}

// STRATEGIES_INTRA_EXPORTS, strategies-intra.h:52-55
int kvz_strategy_register_intra_hip(void *opaque, uint8_t bitdepth)
{
  if (!hook_ready(bitdepth)) return 0;
  int ok = 1;
  ok &= reg(opaque, "angular_pred", (void *)&hip_angular_pred);
  ok &= reg(opaque, "intra_pred_planar", (void *)&hip_intra_pred_planar);
  return ok;
}

// STRATEGIES_SAO_EXPORTS, strategies-sao.h:64-69
int kvz_strategy_register_sao_hip(void *opaque, uint8_t bitdepth)
{
  if (!hook_ready(bitdepth)) return 0;
  int ok = 1;
  ok &= reg(opaque, "sao_edge_ddistortion", (void *)&hip_sao_edge_ddistortion);
  ok &= reg(opaque, "calc_sao_edge_dir", (void *)&hip_calc_sao_edge_dir);
  ok &= reg(opaque, "sao_reconstruct_color", (void *)&hip_sao_reconstruct_color);
  ok &= reg(opaque, "sao_band_ddistortion", (void *)&hip_sao_band_ddistortion);
  return ok;
}

}  // extern "C"
