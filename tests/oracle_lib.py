"""ctypes binding of oracle/libkvzoracle.so (our CPU restatement of the
reference's generic strategy).  TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

u8p = C.POINTER(C.c_uint8)
i16p = C.POINTER(C.c_int16)
u32p = C.POINTER(C.c_uint32)


class QuantParams(C.Structure):
    _fields_ = [("qp", C.c_int32), ("slice_is_intra", C.c_int32), ("signhide", C.c_int32),
                ("scaling_list", C.c_int32), ("quant_coeff", C.POINTER(C.c_int32)),
                ("dequant_coeff", C.POINTER(C.c_int32))]


def build():
    so = os.path.join(ORACLE_DIR, "libkvzoracle.so")
    src = os.path.join(ORACLE_DIR, "kvz_oracle.c")
    if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_reg_sad.restype = C.c_uint
        L.orc_reg_sad.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_uint, C.c_uint]
        for f in ("orc_sad_nxn", "orc_satd_nxn"):
            getattr(L, f).restype = C.c_uint
            getattr(L, f).argtypes = [C.c_int, u8p, u8p]
        L.orc_satd_any_size.restype = C.c_uint
        L.orc_satd_any_size.argtypes = [C.c_int, C.c_int, u8p, C.c_int, u8p, C.c_int]
        for f in ("orc_sad_nxn_dual", "orc_satd_nxn_dual"):
            getattr(L, f).restype = None
            getattr(L, f).argtypes = [C.c_int, u8p, C.c_size_t, u8p, u32p]
        L.orc_satd_any_size_quad.restype = None
        L.orc_satd_any_size_quad.argtypes = [C.c_int, C.c_int, C.POINTER(u8p), C.c_int, u8p, C.c_int, u32p]
        L.orc_pixels_calc_ssd.restype = C.c_uint
        L.orc_pixels_calc_ssd.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
        L.orc_bipred_blend_plane.restype = None
        L.orc_bipred_blend_plane.argtypes = [C.c_int, C.c_int, C.c_int, i16p, u8p, C.c_int,
                                             C.c_int, i16p, u8p, C.c_int, u8p, C.c_int]
        for f in ("orc_image_calc_sad", "orc_image_calc_satd"):
            getattr(L, f).restype = C.c_uint
            getattr(L, f).argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int] + [C.c_int] * 6
        L.orc_ctu_sad_grid.restype = None
        L.orc_ctu_sad_grid.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int] + [C.c_int] * 4 + [i16p, C.c_int, u32p]
        L.orc_transform.restype = None
        L.orc_transform.argtypes = [C.c_int, C.c_int, i16p, i16p]
        L.orc_dct_matrix.restype = i16p
        L.orc_dct_matrix.argtypes = [C.c_int]
        L.orc_scan_order.restype = u32p
        L.orc_scan_order.argtypes = [C.c_int, C.c_int]
        L.orc_get_scaled_qp.restype = C.c_int32
        L.orc_get_scaled_qp.argtypes = [C.c_int] * 3
        L.orc_quant.restype = None
        L.orc_quant.argtypes = [C.POINTER(QuantParams), i16p, i16p] + [C.c_int] * 5
        L.orc_dequant.restype = None
        L.orc_dequant.argtypes = [C.POINTER(QuantParams), i16p, i16p] + [C.c_int] * 4
        L.orc_coeff_abs_sum.restype = C.c_uint32
        L.orc_coeff_abs_sum.argtypes = [i16p, C.c_size_t]
        L.orc_quantize_residual.restype = C.c_int
        L.orc_quantize_residual.argtypes = [C.POINTER(QuantParams)] + [C.c_int] * 7 + [u8p, u8p, u8p, i16p]
        for f, dt in (("orc_sample_quarterpel_luma", u8p), ("orc_sample_14bit_quarterpel_luma", i16p),
                      ("orc_sample_octpel_chroma", u8p), ("orc_sample_14bit_octpel_chroma", i16p)):
            getattr(L, f).restype = None
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dt, C.c_int, i16p]
        L.orc_filter_frac_blocks.restype = None
        L.orc_filter_frac_blocks.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, u8p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_search_frac_costs.restype = None
        L.orc_search_frac_costs.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int] + [C.c_int] * 6 + \
            [u32p, C.POINTER(C.c_int)]
        L.orc_cost_nxn_many.restype = None
        L.orc_cost_nxn_many.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_transform_many.restype = None
        L.orc_transform_many.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_quantize_residual_many.restype = None
        L.orc_quantize_residual_many.argtypes = [C.POINTER(QuantParams)] + [C.c_int] * 5 + [C.c_void_p] * 5 + [C.c_size_t]
        _LIB = L
    return _LIB


def host_threads():
    try:
        return max(1, min(32, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(32, os.cpu_count() or 1))


def run_ranges(count, fn, threads=None):
    """fn(lo, hi) over [0, count) cut into one contiguous range per host thread (ctypes calls release the GIL)"""
    import concurrent.futures
    threads = threads or host_threads()
    step = (count + threads - 1) // threads if count else 1
    ranges = [(lo, min(count, lo + step)) for lo in range(0, count, step)]
    if len(ranges) <= 1:
        for lo, hi in ranges:
            fn(lo, hi)
        return
    with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda r: fn(*r), ranges))


def cost_nxn_many(kind, n, blk1, blk2, threads=None):
    """cost_nxn_batch for whole launches: the loop runs in C, one range of blocks per host thread"""
    blk1, blk2 = _u8(blk1).reshape(-1, n * n), _u8(blk2).reshape(-1, n * n)
    out = np.empty(blk1.shape[0], dtype=np.uint32)
    L, bs = lib(), n * n
    run_ranges(blk1.shape[0], lambda lo, hi: L.orc_cost_nxn_many(int(kind == "satd"), n, blk1.ctypes.data + lo * bs, blk2.ctypes.data + lo * bs,
                                                                 hi - lo, out.ctypes.data + 4 * lo), threads)
    return out


def transform_many(kind, n, blocks, threads=None):
    blocks = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, n * n)
    out = np.empty_like(blocks)
    L, bs, k = lib(), 2 * n * n, {"dct": 0, "idct": 1, "dst": 2, "idst": 3}[kind]
    L.orc_dct_matrix(n)                    # builds the oracle's lazily generated matrices before the threads start
    run_ranges(blocks.shape[0], lambda lo, hi: L.orc_transform_many(k, n, blocks.ctypes.data + lo * bs, out.ctypes.data + lo * bs, hi - lo), threads)
    return out


def _p(a, t):
    return a.ctypes.data_as(t)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


# ---------------------------------------------------------------- picture
def reg_sad(a, b, off1, off2, w, h, s1, s2):
    a, b = _u8(a).ravel(), _u8(b).ravel()
    pa = C.cast(a.ctypes.data + off1, u8p)
    pb = C.cast(b.ctypes.data + off2, u8p)
    return lib().orc_reg_sad(pa, pb, w, h, s1, s2)


def cost_nxn_batch(kind, n, blk1, blk2):
    """kind 'sad'|'satd'; blk1/blk2: uint8 [count, n*n] -> uint32[count]"""
    blk1, blk2 = _u8(blk1).reshape(-1, n * n), _u8(blk2).reshape(-1, n * n)
    f = lib().orc_sad_nxn if kind == "sad" else lib().orc_satd_nxn
    out = np.empty(blk1.shape[0], dtype=np.uint32)
    for i in range(blk1.shape[0]):
        out[i] = f(n, _p(blk1[i], u8p), _p(blk2[i], u8p))
    return out


def cost_nxn_dual_batch(kind, n, preds, orig, pred_stride=1024):
    """preds: uint8 [count, 2*pred_stride] (pred k at k*pred_stride); orig [count, n*n]"""
    preds = _u8(preds).reshape(orig.shape[0], -1)
    orig = _u8(orig).reshape(-1, n * n)
    f = lib().orc_sad_nxn_dual if kind == "sad" else lib().orc_satd_nxn_dual
    out = np.empty((orig.shape[0], 2), dtype=np.uint32)
    for i in range(orig.shape[0]):
        f(n, _p(preds[i], u8p), pred_stride, _p(orig[i], u8p), _p(out[i], u32p))
    return out


def satd_any_size(w, h, a, off1, s1, b, off2, s2):
    a, b = _u8(a).ravel(), _u8(b).ravel()
    return lib().orc_satd_any_size(w, h, C.cast(a.ctypes.data + off1, u8p), s1,
                                   C.cast(b.ctypes.data + off2, u8p), s2)


def satd_any_size_quad(w, h, preds4, stride, orig, orig_off, orig_stride):
    """preds4: list of 4 uint8 arrays (each at least h*stride); orig uint8 array"""
    ps = [_u8(p).ravel() for p in preds4]
    arr = (u8p * 4)(*[_p(p, u8p) for p in ps])
    orig = _u8(orig).ravel()
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_satd_any_size_quad(w, h, arr, stride, C.cast(orig.ctypes.data + orig_off, u8p), orig_stride,
                                 _p(out, u32p))
    return out


def pixels_calc_ssd(a, off1, b, off2, s1, s2, w):
    a, b = _u8(a).ravel(), _u8(b).ravel()
    return lib().orc_pixels_calc_ssd(C.cast(a.ctypes.data + off1, u8p), C.cast(b.ctypes.data + off2, u8p), s1, s2, w)


def image_calc(kind, pic, ref, pic_x, pic_y, ref_x, ref_y, bw, bh):
    pic, ref = _u8(pic), _u8(ref)
    f = lib().orc_image_calc_sad if kind == "sad" else lib().orc_image_calc_satd
    return f(_p(pic, u8p), pic.shape[1], _p(ref, u8p), ref.shape[1], ref.shape[1], ref.shape[0],
             pic_x, pic_y, ref_x, ref_y, bw, bh)


def bipred_blend_plane(w, h, hi0, s0, hi1, s1):
    """s0/s1: int16 [h,w] when hi, else uint8 [h,w]"""
    dst = np.zeros((h, w), dtype=np.uint8)
    z16 = np.zeros(1, dtype=np.int16)
    z8 = np.zeros(1, dtype=np.uint8)
    a16 = np.ascontiguousarray(s0, dtype=np.int16) if hi0 else z16
    a8 = z8 if hi0 else _u8(s0)
    b16 = np.ascontiguousarray(s1, dtype=np.int16) if hi1 else z16
    b8 = z8 if hi1 else _u8(s1)
    lib().orc_bipred_blend_plane(w, h, int(hi0), _p(a16, i16p), _p(a8, u8p), w,
                                 int(hi1), _p(b16, i16p), _p(b8, u8p), w, _p(dst, u8p), w)
    return dst


# ---------------------------------------------------------------- dct
KINDS = {"dct": 0, "idct": 1, "dst": 2, "idst": 3}


def transform_batch(kind, n, blocks):
    blocks = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, n * n)
    out = np.empty_like(blocks)
    k = KINDS[kind]
    for i in range(blocks.shape[0]):
        lib().orc_transform(k, n, _p(blocks[i], i16p), _p(out[i], i16p))
    return out


def dct_matrix(n):
    p = lib().orc_dct_matrix(n)
    return np.ctypeslib.as_array(p, shape=(n, n)).copy()


def scan_order(scan_idx, log2):
    p = lib().orc_scan_order(scan_idx, log2)
    return np.ctypeslib.as_array(p, shape=(1 << (2 * log2),)).copy()


# ---------------------------------------------------------------- quant
def _qp(qp, slice_is_intra=0, signhide=0, quant_coeff=None, dequant_coeff=None):
    p = QuantParams()
    p.qp, p.slice_is_intra, p.signhide = qp, int(slice_is_intra), int(signhide)
    keep = []
    if quant_coeff is not None or dequant_coeff is not None:
        p.scaling_list = 1
        if quant_coeff is not None:
            q = np.ascontiguousarray(quant_coeff, dtype=np.int32); keep.append(q)
            p.quant_coeff = _p(q, C.POINTER(C.c_int32))
        if dequant_coeff is not None:
            d = np.ascontiguousarray(dequant_coeff, dtype=np.int32); keep.append(d)
            p.dequant_coeff = _p(d, C.POINTER(C.c_int32))
    return p, keep


def quant_batch(coef, w, qp, type_, scan_idx, slice_is_intra=0, signhide=0, block_is_intra=0, quant_coeff=None):
    coef = np.ascontiguousarray(coef, dtype=np.int16).reshape(-1, w * w)
    out = np.empty_like(coef)
    p, keep = _qp(qp, slice_is_intra, signhide, quant_coeff=quant_coeff)
    for i in range(coef.shape[0]):
        lib().orc_quant(C.byref(p), _p(coef[i], i16p), _p(out[i], i16p), w, w, type_, scan_idx, block_is_intra)
    return out


def dequant_batch(q_coef, w, qp, type_, block_is_intra=0, dequant_coeff=None):
    q_coef = np.ascontiguousarray(q_coef, dtype=np.int16).reshape(-1, w * w)
    out = np.empty_like(q_coef)
    p, keep = _qp(qp, dequant_coeff=dequant_coeff)
    for i in range(q_coef.shape[0]):
        lib().orc_dequant(C.byref(p), _p(q_coef[i], i16p), _p(out[i], i16p), w, w, type_, block_is_intra)
    return out


def coeff_abs_sum(c):
    c = np.ascontiguousarray(c, dtype=np.int16).ravel()
    return lib().orc_coeff_abs_sum(_p(c, i16p), c.size)


def quantize_residual_batch(ref_in, pred_in, w, qp, color, scan_order_, cu_is_intra, slice_is_intra=0,
                            signhide=0, use_trskip=0):
    """ref_in/pred_in uint8 [count, w*w] (stride w) -> (rec [count,w*w], coeff [count,w*w], has [count])"""
    ref_in, pred_in = _u8(ref_in).reshape(-1, w * w), _u8(pred_in).reshape(-1, w * w)
    rec = np.zeros_like(ref_in)
    coeff = np.zeros(ref_in.shape, dtype=np.int16)
    has = np.zeros(ref_in.shape[0], dtype=np.int32)
    p, keep = _qp(qp, slice_is_intra, signhide)
    for i in range(ref_in.shape[0]):
        has[i] = lib().orc_quantize_residual(C.byref(p), int(cu_is_intra), w, color, scan_order_, int(use_trskip),
                                             w, w, _p(ref_in[i], u8p), _p(pred_in[i], u8p), _p(rec[i], u8p),
                                             _p(coeff[i], i16p))
    return rec, coeff, has


def quantize_residual_many(ref_in, pred_in, w, qp, color, scan_order_, cu_is_intra, slice_is_intra=0, signhide=0, use_trskip=0,
                           threads=None):
    """quantize_residual_batch for whole launches (C loop, one range of TUs per host thread)"""
    ref_in, pred_in = _u8(ref_in).reshape(-1, w * w), _u8(pred_in).reshape(-1, w * w)
    rec = np.zeros_like(ref_in)
    coeff = np.zeros(ref_in.shape, dtype=np.int16)
    has = np.zeros(ref_in.shape[0], dtype=np.int32)
    p, keep = _qp(qp, slice_is_intra, signhide)
    L, bs = lib(), w * w
    L.orc_dct_matrix(w)
    L.orc_scan_order(scan_order_, {4: 2, 8: 3, 16: 4, 32: 5}[w])
    run_ranges(ref_in.shape[0], lambda lo, hi: L.orc_quantize_residual_many(
        C.byref(p), int(cu_is_intra), w, color, scan_order_, int(use_trskip), ref_in.ctypes.data + lo * bs, pred_in.ctypes.data + lo * bs,
        rec.ctypes.data + lo * bs, coeff.ctypes.data + 2 * lo * bs, has.ctypes.data + 4 * lo, hi - lo), threads)
    return rec, coeff, has


# ---------------------------------------------------------------- ipol
def sample(kind, frame, x, y, w, h, mvx, mvy):
    """kind: luma|luma14|chroma|chroma14.  src = &frame[y][x] (window around it must be inside)."""
    frame = _u8(frame)
    stride = frame.shape[1]
    src = frame.ctypes.data + y * stride + x
    mv = np.array([mvx, mvy], dtype=np.int16)
    if kind in ("luma", "chroma"):
        dst = np.zeros((h, w), dtype=np.uint8)
        f = lib().orc_sample_quarterpel_luma if kind == "luma" else lib().orc_sample_octpel_chroma
        f(src, stride, w, h, _p(dst, u8p), w, _p(mv, i16p))
    else:
        dst = np.zeros((h, w), dtype=np.int16)
        f = lib().orc_sample_14bit_quarterpel_luma if kind == "luma14" else lib().orc_sample_14bit_octpel_chroma
        f(src, stride, w, h, _p(dst, i16p), w, _p(mv, i16p))
    return dst


IPOL_STATE_BYTES = 2 * (5 * 72 * 64 + 5 * 72)


def filter_frac_steps(frame, x, y, w, h, offs, fme_level=4):
    """Runs steps 0..3 with src = &frame[y][x]; offs = (hpel_off_x, hpel_off_y) for the
    qpel steps.  Returns uint8 [4 steps, 4 blocks, 64, 64] (only [:h,:w] defined)."""
    frame = _u8(frame)
    stride = frame.shape[1]
    src = frame.ctypes.data + y * stride + x
    st = np.zeros(IPOL_STATE_BYTES, dtype=np.uint8)
    out = np.zeros((4, 4, 64, 64), dtype=np.uint8)
    for step in range(4):
        ox, oy = (0, 0) if step < 2 else offs
        lib().orc_filter_frac_blocks(step, src, stride, w, h, _p(out[step], u8p), st.ctypes.data, fme_level, ox, oy)
    return out


def search_frac_costs(pic, ref, x, y, w, h, mvx, mvy):
    pic, ref = _u8(pic), _u8(ref)
    costs = np.zeros(17, dtype=np.uint32)
    best = (C.c_int * 2)()
    lib().orc_search_frac_costs(_p(pic, u8p), pic.shape[1], _p(ref, u8p), ref.shape[1], ref.shape[0],
                                x, y, w, h, mvx, mvy, _p(costs, u32p), best)
    return costs, (best[0], best[1])


def ctu_sad_grid(pic, ref, ctus, mv_offsets):
    """ctus: (x, y, mvx, mvy) rows; mv_offsets: (dx, dy) rows -> uint32 [n_ctu, n_mv, 85]"""
    pic, ref = _u8(pic), _u8(ref)
    mv = np.ascontiguousarray(mv_offsets, dtype=np.int16).reshape(-1, 2)
    ctus = np.asarray(ctus, dtype=np.int32).reshape(-1, 4)
    out = np.zeros((ctus.shape[0], mv.shape[0], 85), dtype=np.uint32)
    for i, (x, y, mvx, mvy) in enumerate(ctus):
        lib().orc_ctu_sad_grid(_p(pic, u8p), pic.shape[1], pic.shape[1], pic.shape[0], _p(ref, u8p), ref.shape[1],
                               ref.shape[1], ref.shape[0], int(x), int(y), int(mvx), int(mvy), _p(mv, i16p), mv.shape[0],
                               _p(out[i], u32p))
    return out


# ---- intra group.  refs arrays are (count, 130) uint8 = kvz_intra_ref {left[65], top[65]} ----
def _intra_sigs():
    L = lib()
    if getattr(L, "_intra_done", False):
        return L
    L.orc_angular_pred.restype = None
    L.orc_angular_pred.argtypes = [C.c_int, C.c_int, u8p, u8p, u8p]
    L.orc_intra_pred_planar.restype = None
    L.orc_intra_pred_planar.argtypes = [C.c_int, u8p, u8p, u8p]
    L.orc_intra_predict.restype = None
    L.orc_intra_predict.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
    L.orc_intra_rough_costs.restype = None
    L.orc_intra_rough_costs.argtypes = [u8p, C.c_int, C.c_int, u8p, u32p, u32p]
    L.orc_intra_build_reference_many.restype = None
    L.orc_intra_build_reference_many.argtypes = [C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, u8p]
    L._intra_done = True
    return L


def angular_pred(log2_width, mode, above, left):
    L = _intra_sigs()
    n = 1 << log2_width
    above, left = _u8(above), _u8(left)
    dst = np.zeros(n * n, dtype=np.uint8)
    L.orc_angular_pred(log2_width, mode, _p(above, u8p), _p(left, u8p), _p(dst, u8p))
    return dst


def intra_pred_planar(log2_width, top, left):
    L = _intra_sigs()
    n = 1 << log2_width
    top, left = _u8(top), _u8(left)
    dst = np.zeros(n * n, dtype=np.uint8)
    L.orc_intra_pred_planar(log2_width, _p(top, u8p), _p(left, u8p), _p(dst, u8p))
    return dst


def intra_predict_batch(refs, log2_width, modes, is_luma=1, filter_boundary=1):
    """-> (count, len(modes), N*N)"""
    L = _intra_sigs()
    refs = _u8(refs).reshape(-1, 130)
    n = 1 << log2_width
    out = np.zeros((refs.shape[0], len(modes), n * n), dtype=np.uint8)
    for i in range(refs.shape[0]):
        r = np.ascontiguousarray(refs[i])
        for j, m in enumerate(modes):
            d = out[i, j]
            L.orc_intra_predict(_p(r, u8p), log2_width, int(m), is_luma, filter_boundary, _p(d, u8p))
    return out


def intra_build_reference_batch(log2_width, color, plane, pic_w, pic_h, xy):
    """kvz_intra_build_reference of every PU at the luma positions xy (count, 2) from the 2-D reconstruction plane of
    `color` -> (count, 130) uint8, entries past 2N zero"""
    L = _intra_sigs()
    plane = np.ascontiguousarray(plane, dtype=np.uint8)
    xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
    out = np.zeros((xy.shape[0], 130), dtype=np.uint8)
    L.orc_intra_build_reference_many(log2_width, color, _p(plane, u8p), plane.shape[1], pic_w, pic_h, xy.ctypes.data, xy.shape[0],
                                     _p(out, u8p))
    return out


def intra_rough_costs_batch(refs, log2_width, orig, filter_boundary=1):
    """-> (satd (count, 35), sad (count, 35)) uint32"""
    L = _intra_sigs()
    refs = _u8(refs).reshape(-1, 130)
    n = 1 << log2_width
    orig = _u8(orig).reshape(-1, n * n)
    satd = np.zeros((refs.shape[0], 35), dtype=np.uint32)
    sad = np.zeros((refs.shape[0], 35), dtype=np.uint32)
    for i in range(refs.shape[0]):
        r, o = np.ascontiguousarray(refs[i]), np.ascontiguousarray(orig[i])
        L.orc_intra_rough_costs(_p(r, u8p), log2_width, filter_boundary, _p(o, u8p), _p(satd[i], u32p), _p(sad[i], u32p))
    return satd, sad


# ---- motion search ----
def search_pu_batch(pic, ref, pus, params, cabac=None, cost_to_beat=None):
    """orc_search_pu_many over a structured array of patterns.ME_PU; returns patterns.ME_RESULT array"""
    from patterns import ME_RESULT
    L = lib()
    L.orc_search_pu_many.restype = None
    L.orc_search_pu_many.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    pic, ref = _u8(pic), _u8(ref)
    pus = np.ascontiguousarray(pus)
    params = np.ascontiguousarray(params).copy()
    assert params.nbytes == 96
    if cabac is not None:                       # --mv-rdo: ME_CABAC snapshots; pus["reserved"] indexes them
        cabac = np.ascontiguousarray(cabac)
        params["cabac"] = cabac.ctypes.data
    if cost_to_beat is not None:
        cost_to_beat = np.ascontiguousarray(cost_to_beat, dtype=np.uint32)
        params["cost_to_beat"] = cost_to_beat.ctypes.data
    out = np.zeros(len(pus), dtype=ME_RESULT)
    L.orc_search_pu_many(_p(pic, u8p), pic.shape[1], _p(ref, u8p), ref.shape[1], ref.shape[0], pus.ctypes.data, len(pus),
                         params.ctypes.data, out.ctypes.data)
    return out


# ---- SAO group.  sao records are 14 int32: type, eo_class, band_position[2], offsets[10] ----
def _sao17(sao14):
    """-> orc_sao_info layout {type, eo_class, ddistortion, merge_left, merge_up, band_position[2], offsets[10]}"""
    s = np.asarray(sao14, dtype=np.int32)
    return np.ascontiguousarray(np.concatenate([s[:2], [0, 0, 0], s[2:]]).astype(np.int32))


def _sao_sigs():
    L = lib()
    if getattr(L, "_sao_done", False):
        return L
    i32p = C.POINTER(C.c_int32)
    L.orc_sao_edge_ddistortion.restype = C.c_int
    L.orc_sao_edge_ddistortion.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, i32p]
    L.orc_calc_sao_edge_dir.restype = None
    L.orc_calc_sao_edge_dir.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, i32p]
    L.orc_sao_band_ddistortion.restype = C.c_int
    L.orc_sao_band_ddistortion.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, i32p]
    L.orc_calc_sao_bands.restype = None
    L.orc_calc_sao_bands.argtypes = [u8p, u8p, C.c_int, C.c_int, i32p]
    L.orc_sao_reconstruct_color.restype = None
    L.orc_sao_reconstruct_color.argtypes = [C.c_void_p, C.c_void_p, i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L._sao_done = True
    return L


def sao_edge_ddistortion(orig, rec, bw, bh, eo_class, offsets):
    L = _sao_sigs()
    orig, rec = _u8(orig), _u8(rec)
    o = np.ascontiguousarray(offsets, dtype=np.int32)
    return L.orc_sao_edge_ddistortion(_p(orig, u8p), _p(rec, u8p), bw, bh, eo_class, _p(o, C.POINTER(C.c_int32)))


def calc_sao_edge_dir(orig, rec, eo_class, bw, bh):
    L = _sao_sigs()
    orig, rec = _u8(orig), _u8(rec)
    out = np.zeros((2, 5), dtype=np.int32)
    L.orc_calc_sao_edge_dir(_p(orig, u8p), _p(rec, u8p), eo_class, bw, bh, _p(out, C.POINTER(C.c_int32)))
    return out


def sao_band_ddistortion(orig, rec, bw, bh, band_pos, bands):
    L = _sao_sigs()
    orig, rec = _u8(orig), _u8(rec)
    b = np.ascontiguousarray(bands, dtype=np.int32)
    return L.orc_sao_band_ddistortion(_p(orig, u8p), _p(rec, u8p), bw, bh, band_pos, _p(b, C.POINTER(C.c_int32)))


def calc_sao_bands(orig, rec, bw, bh):
    L = _sao_sigs()
    orig, rec = _u8(orig), _u8(rec)
    out = np.zeros((2, 32), dtype=np.int32)
    L.orc_calc_sao_bands(_p(orig, u8p), _p(rec, u8p), bw, bh, _p(out, C.POINTER(C.c_int32)))
    return out


def sao_reconstruct_color(plane, x, y, bw, bh, sao14, color):
    """plane: 2-D uint8 with at least one pixel around the block; returns the bw x bh filtered block"""
    L = _sao_sigs()
    plane = _u8(plane)
    stride = plane.shape[1]
    out = np.zeros((bh, bw), dtype=np.uint8)
    s = _sao17(sao14)
    L.orc_sao_reconstruct_color(plane.ctypes.data + y * stride + x, out.ctypes.data, _p(s, C.POINTER(C.c_int32)), stride, bw, bw, bh, color)
    return out


# ---- bi-prediction candidate cost ----
def bipred_luma_satd(pic, ref0, ref1, x, y, w, h, mv0, mv1):
    """-> (cost, w x h prediction)"""
    L = lib()
    L.orc_bipred_luma_satd.restype = C.c_uint
    L.orc_bipred_luma_satd.argtypes = [u8p, C.c_int, u8p, u8p, C.c_int, C.c_int] + [C.c_int] * 4 + [i16p, i16p, u8p]
    pic, ref0, ref1 = _u8(pic), _u8(ref0), _u8(ref1)
    a, b = np.ascontiguousarray(mv0, dtype=np.int16), np.ascontiguousarray(mv1, dtype=np.int16)
    out = np.zeros((h, w), dtype=np.uint8)
    c = L.orc_bipred_luma_satd(_p(pic, u8p), pic.shape[1], _p(ref0, u8p), _p(ref1, u8p), ref0.shape[1], ref0.shape[0], x, y, w, h,
                               _p(a, i16p), _p(b, i16p), _p(out, u8p))
    return c, out


# ---- deblocking ----
def deblock_frame(y, u, v, cus, prm):
    """-> filtered copies of the three planes (u, v may be None with prm['chroma'] == 0)"""
    L = lib()
    L.orc_deblock_frame.restype = None
    L.orc_deblock_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    y = np.array(y, dtype=np.uint8, order="C")
    u = np.array(u, dtype=np.uint8, order="C") if u is not None else None
    v = np.array(v, dtype=np.uint8, order="C") if v is not None else None
    cus = np.ascontiguousarray(cus)
    prm = np.ascontiguousarray(prm)
    assert cus.dtype.itemsize == 20 and prm.nbytes == 64 and cus.shape == (y.shape[0] // 4, (y.shape[1] + 3) // 4)
    L.orc_deblock_frame(y.ctypes.data, y.shape[1], u.ctypes.data if u is not None else None, v.ctypes.data if v is not None else None,
                        u.shape[1] if u is not None else 0, y.shape[1], y.shape[0], cus.ctypes.data, prm.ctypes.data)
    return y, u, v


# ---- AMVP / merge candidate derivation ----
def inter_candidates(params, cus, col_cus, ref_cus, pus):
    """orc_inter_candidates: -> (pus with mv_cand / extra_mv / num_merge_cand / merge filled, merge lists [count, 5] MERGE_CAND)"""
    from patterns import MERGE_CAND
    L = lib()
    L.orc_inter_candidates.restype = None
    L.orc_inter_candidates.argtypes = [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]
    cus, col_cus = np.ascontiguousarray(cus), np.ascontiguousarray(col_cus)
    ref_cus = None if ref_cus is None else np.ascontiguousarray(ref_cus)
    pus = np.ascontiguousarray(pus).copy()
    out = np.zeros((len(pus), 5), dtype=MERGE_CAND)
    L.orc_inter_candidates(cus.ctypes.data, col_cus.ctypes.data, None if ref_cus is None else ref_cus.ctypes.data,
                           np.ascontiguousarray(params).ctypes.data, pus.ctypes.data, len(pus), out.ctypes.data)
    return pus, out


def mv_cand_helpers(geoms, pic_w, pic_h):
    """orc_is_a0_cand_coded / orc_is_b0_cand_coded / orc_spatial_merge_candidate_indices for PUs (x, y, w, h)
    -> (a0 [n], b0 [n], indices [n, 5] = b0 b1 b2 a0 a1 in lcu_t.cu)"""
    L = lib()
    geoms = np.asarray(geoms, dtype=np.int32).reshape(-1, 4)
    a0 = np.zeros(len(geoms), np.int32)
    b0 = np.zeros(len(geoms), np.int32)
    idx = np.zeros((len(geoms), 5), np.int32)
    out = (C.c_int * 5)()
    for i, (x, y, w, h) in enumerate(geoms.tolist()):
        a0[i] = L.orc_is_a0_cand_coded(x, y, w, h)
        b0[i] = L.orc_is_b0_cand_coded(x, y, w, h)
        L.orc_spatial_merge_candidate_indices(x, y, w, h, pic_w, pic_h, out)
        idx[i] = list(out)
    return a0, b0, idx
