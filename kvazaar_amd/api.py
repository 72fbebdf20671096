"""Host-side Python mirror of the batched C ABI (include/kvz_hip.h).

Two levels:
  * `DeviceBuffer` + the raw `*_batch` calls of `_lib.load()` work on device
    pointers (what bench.py times);
  * the convenience functions below take numpy arrays, stage them to HBM, run
    the HIP kernel and copy the result back -- used by the parity tests, which
    therefore always go through the C ABI and the GPU.
Names follow the reference's strategy types (strategies-picture.h:174-199,
strategies-dct.h:55-69, strategies-quant.h:58-62, strategies-ipol.h:65-74)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BlockPair, IpolBlock, QuantParams, KvzHipError, check

KINDS = {"dct": 0, "idct": 1, "dst": 2, "idst": 3, "trskip": 4, "itrskip": 5}


class DeviceBuffer:
    """HBM allocation owned through kvz_hip_malloc/kvz_hip_free."""

    def __init__(self, nbytes):
        self.lib = _lib.init()
        self.nbytes = int(nbytes)
        self.ptr = self.lib.kvz_hip_malloc(max(self.nbytes, 16))
        if not self.ptr:
            raise KvzHipError("kvz_hip_malloc(%d) failed: %s" % (nbytes, self.lib.kvz_hip_last_error().decode()))

    @classmethod
    def from_numpy(cls, a, stream=None):
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes)
        if a.nbytes:
            check(buf.lib.kvz_hip_memcpy_h2d(buf.ptr, a.ctypes.data, a.nbytes, stream), "memcpy_h2d")
            check(buf.lib.kvz_hip_stream_sync(stream), "stream_sync")
        return buf

    def to_numpy(self, dtype, shape, stream=None):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= max(self.nbytes, 16)
        if out.nbytes:
            check(self.lib.kvz_hip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, stream), "memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            self.lib.kvz_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _pairs_array(pairs):
    """pairs: iterable of (x1, y1, x2, y2, w, h) -> contiguous BlockPair array as numpy int32 [n,6]"""
    a = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 6))
    return a


# ------------------------------------------------------------------ picture
def cost_nxn_batch(kind, n, blk1, blk2):
    """sad_NxN / satd_NxN over [count, n*n] uint8 block pairs -> uint32[count]"""
    L = _lib.init()
    blk1 = np.ascontiguousarray(blk1, dtype=np.uint8).reshape(-1, n * n)
    blk2 = np.ascontiguousarray(blk2, dtype=np.uint8).reshape(-1, n * n)
    count = blk1.shape[0]
    a, b, o = DeviceBuffer.from_numpy(blk1), DeviceBuffer.from_numpy(blk2), DeviceBuffer(4 * count)
    f = L.kvz_hip_sad_nxn_batch if kind == "sad" else L.kvz_hip_satd_nxn_batch
    check(f(n, a.ptr, b.ptr, count, o.ptr, None), "%s_%dx%d batch" % (kind, n, n))
    return o.to_numpy(np.uint32, (count,))


def cost_nxn_dual_batch(kind, n, preds, orig, pred_stride=1024, item_stride=2048):
    L = _lib.init()
    orig = np.ascontiguousarray(orig, dtype=np.uint8).reshape(-1, n * n)
    count = orig.shape[0]
    preds = np.ascontiguousarray(preds, dtype=np.uint8).reshape(count, item_stride)
    p, g, o = DeviceBuffer.from_numpy(preds), DeviceBuffer.from_numpy(orig), DeviceBuffer(8 * count)
    f = L.kvz_hip_sad_nxn_dual_batch if kind == "sad" else L.kvz_hip_satd_nxn_dual_batch
    check(f(n, p.ptr, pred_stride, item_stride, g.ptr, count, o.ptr, None), "%s_%dx%d_dual batch" % (kind, n, n))
    return o.to_numpy(np.uint32, (count, 2))


def _pair_call(fname, plane1, plane2, pairs, clamp):
    L = _lib.init()
    plane1 = np.ascontiguousarray(plane1, dtype=np.uint8)
    plane2 = np.ascontiguousarray(plane2, dtype=np.uint8)
    pa = _pairs_array(pairs)
    count = pa.shape[0]
    a, b, d, o = (DeviceBuffer.from_numpy(plane1), DeviceBuffer.from_numpy(plane2), DeviceBuffer.from_numpy(pa),
                  DeviceBuffer(4 * count))
    f = getattr(L, fname)
    if clamp:
        rc = f(a.ptr, plane1.shape[1], b.ptr, plane2.shape[1], plane2.shape[1], plane2.shape[0], d.ptr, count, o.ptr, None)
    else:
        rc = f(a.ptr, plane1.shape[1], b.ptr, plane2.shape[1], d.ptr, count, o.ptr, None)
    check(rc, fname)
    return o.to_numpy(np.uint32, (count,))


def reg_sad_batch(plane1, plane2, pairs):
    """reg_sad over (x1,y1,x2,y2,w,h) pairs inside 2-D uint8 planes"""
    return _pair_call("kvz_hip_reg_sad_batch", plane1, plane2, pairs, False)


def image_calc_sad_batch(pic, ref, pairs):
    return _pair_call("kvz_hip_image_calc_sad_batch", pic, ref, pairs, True)


def image_calc_satd_batch(pic, ref, pairs):
    return _pair_call("kvz_hip_image_calc_satd_batch", pic, ref, pairs, True)


def pixels_calc_ssd_batch(plane1, plane2, pairs):
    return _pair_call("kvz_hip_pixels_calc_ssd_batch", plane1, plane2, pairs, False)


def satd_any_size_quad_batch(preds, orig, pairs, pred_stride=64, pred_item_stride=64 * 64):
    """preds: uint8 [count*4, pred_item_stride]; orig: 2-D plane; pairs use (x1,y1,w,h)"""
    L = _lib.init()
    preds = np.ascontiguousarray(preds, dtype=np.uint8)
    orig = np.ascontiguousarray(orig, dtype=np.uint8)
    pa = _pairs_array(pairs)
    count = pa.shape[0]
    p, g, d, o = (DeviceBuffer.from_numpy(preds), DeviceBuffer.from_numpy(orig), DeviceBuffer.from_numpy(pa),
                  DeviceBuffer(16 * count))
    check(L.kvz_hip_satd_any_size_quad_batch(p.ptr, pred_stride, pred_item_stride, g.ptr, orig.shape[1], d.ptr, count,
                                             o.ptr, None), "satd_any_size_quad batch")
    return o.to_numpy(np.uint32, (count, 4))


def bipred_blend_batch(w, h, hi0, s0, hi1, s1):
    """s0/s1: [count, h, w] int16 (hi precision) or uint8 -> uint8 [count, h, w]"""
    L = _lib.init()
    s0 = np.ascontiguousarray(s0, dtype=np.int16 if hi0 else np.uint8).reshape(-1, h, w)
    s1 = np.ascontiguousarray(s1, dtype=np.int16 if hi1 else np.uint8).reshape(-1, h, w)
    count = s0.shape[0]
    a, b, o = DeviceBuffer.from_numpy(s0), DeviceBuffer.from_numpy(s1), DeviceBuffer(count * h * w)
    check(L.kvz_hip_bipred_blend_batch(w, h, int(hi0), a.ptr, int(hi1), b.ptr, o.ptr, count, None), "bipred blend")
    return o.to_numpy(np.uint8, (count, h, w))


def ctu_sad_grid_batch(pic, ref, ctus, mv_offsets):
    """ctus: (x, y, mvx, mvy) rows; mv_offsets: (dx, dy) rows -> uint32 [n_ctu, n_mv, 85]"""
    L = _lib.init()
    pic = np.ascontiguousarray(pic, dtype=np.uint8)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    c = np.ascontiguousarray(np.asarray(ctus, dtype=np.int32).reshape(-1, 4))
    mv = np.ascontiguousarray(np.asarray(mv_offsets, dtype=np.int16).reshape(-1, 2))
    n, k = c.shape[0], mv.shape[0]
    a, b, dc, dm, o = (DeviceBuffer.from_numpy(pic), DeviceBuffer.from_numpy(ref), DeviceBuffer.from_numpy(c),
                       DeviceBuffer.from_numpy(mv), DeviceBuffer(4 * 85 * n * k))
    check(L.kvz_hip_ctu_sad_grid_batch(a.ptr, pic.shape[1], pic.shape[1], pic.shape[0], b.ptr, ref.shape[1], ref.shape[1],
                                       ref.shape[0], dc.ptr, n, dm.ptr, k, o.ptr, None), "ctu_sad_grid batch")
    return o.to_numpy(np.uint32, (n, k, 85))


# ------------------------------------------------------------------ dct
def transform_batch(kind, n, blocks):
    L = _lib.init()
    blocks = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, n * n)
    count = blocks.shape[0]
    a, o = DeviceBuffer.from_numpy(blocks), DeviceBuffer(blocks.nbytes)
    check(L.kvz_hip_transform_batch(KINDS[kind], n, a.ptr, o.ptr, count, None), "%s %d batch" % (kind, n))
    return o.to_numpy(np.int16, blocks.shape)


# ------------------------------------------------------------------ quant
def _qparams(qp, slice_is_intra=0, signhide=0, quant_coeff=None, dequant_coeff=None):
    p = QuantParams()
    p.qp, p.slice_is_intra, p.signhide = int(qp), int(slice_is_intra), int(signhide)
    keep = []
    if quant_coeff is not None or dequant_coeff is not None:
        p.scaling_list = 1
        if quant_coeff is not None:
            q = DeviceBuffer.from_numpy(np.ascontiguousarray(quant_coeff, dtype=np.int32)); keep.append(q)
            p.quant_coeff = q.ptr
        if dequant_coeff is not None:
            d = DeviceBuffer.from_numpy(np.ascontiguousarray(dequant_coeff, dtype=np.int32)); keep.append(d)
            p.dequant_coeff = d.ptr
    return p, keep


def quant_batch(coef, w, qp, type_, scan_idx, slice_is_intra=0, signhide=0, quant_coeff=None):
    L = _lib.init()
    coef = np.ascontiguousarray(coef, dtype=np.int16).reshape(-1, w * w)
    count = coef.shape[0]
    p, keep = _qparams(qp, slice_is_intra, signhide, quant_coeff=quant_coeff)
    a, o = DeviceBuffer.from_numpy(coef), DeviceBuffer(coef.nbytes)
    check(L.kvz_hip_quant_batch(C.byref(p), a.ptr, o.ptr, w, type_, scan_idx, count, None), "quant batch")
    return o.to_numpy(np.int16, coef.shape)


def dequant_batch(q_coef, w, qp, type_, dequant_coeff=None):
    L = _lib.init()
    q_coef = np.ascontiguousarray(q_coef, dtype=np.int16).reshape(-1, w * w)
    count = q_coef.shape[0]
    p, keep = _qparams(qp, dequant_coeff=dequant_coeff)
    a, o = DeviceBuffer.from_numpy(q_coef), DeviceBuffer(q_coef.nbytes)
    check(L.kvz_hip_dequant_batch(C.byref(p), a.ptr, o.ptr, w, type_, count, None), "dequant batch")
    return o.to_numpy(np.int16, q_coef.shape)


def coeff_abs_sum_batch(coeffs, length):
    L = _lib.init()
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16).reshape(-1, length)
    count = coeffs.shape[0]
    a, o = DeviceBuffer.from_numpy(coeffs), DeviceBuffer(4 * count)
    check(L.kvz_hip_coeff_abs_sum_batch(a.ptr, length, count, o.ptr, None), "coeff_abs_sum batch")
    return o.to_numpy(np.uint32, (count,))


def quantize_residual_batch(ref_in, pred_in, w, qp, color, scan_order, cu_is_intra, slice_is_intra=0, signhide=0,
                            use_trskip=0, alias_rec=False, with_costs=False):
    """-> (rec, coeff, has_coeffs) and, with_costs, (+ ssd(ref, rec), coeff_abs_sum) from the same launch"""
    L = _lib.init()
    ref_in = np.ascontiguousarray(ref_in, dtype=np.uint8).reshape(-1, w * w)
    pred_in = np.ascontiguousarray(pred_in, dtype=np.uint8).reshape(-1, w * w)
    count = ref_in.shape[0]
    p, keep = _qparams(qp, slice_is_intra, signhide)
    r, pr = DeviceBuffer.from_numpy(ref_in), DeviceBuffer.from_numpy(pred_in)
    rec = pr if alias_rec else DeviceBuffer(ref_in.nbytes)
    co, has = DeviceBuffer(2 * ref_in.size), DeviceBuffer(4 * count)
    if with_costs:
        ssd, sab = DeviceBuffer(4 * count), DeviceBuffer(4 * count)
        check(L.kvz_hip_quantize_residual_cost_batch(C.byref(p), int(cu_is_intra), w, color, scan_order, int(use_trskip),
                                                     r.ptr, pr.ptr, rec.ptr, co.ptr, has.ptr, ssd.ptr, sab.ptr, count, None),
              "quantize_residual_cost")
        return (rec.to_numpy(np.uint8, ref_in.shape), co.to_numpy(np.int16, ref_in.shape), has.to_numpy(np.int32, (count,)),
                ssd.to_numpy(np.uint32, (count,)), sab.to_numpy(np.uint32, (count,)))
    check(L.kvz_hip_quantize_residual_batch(C.byref(p), int(cu_is_intra), w, color, scan_order, int(use_trskip),
                                            r.ptr, pr.ptr, rec.ptr, co.ptr, has.ptr, count, None), "quantize_residual")
    return (rec.to_numpy(np.uint8, ref_in.shape), co.to_numpy(np.int16, ref_in.shape), has.to_numpy(np.int32, (count,)))


# ------------------------------------------------------------------ ipol
def sample_batch(kind, ref, blocks):
    """kind: luma|luma14|chroma|chroma14; blocks: (x, y, frac_x, frac_y, w, h); returns list of arrays"""
    L = _lib.init()
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    b = np.ascontiguousarray(np.asarray(blocks, dtype=np.int32).reshape(-1, 6))
    count = b.shape[0]
    sizes = (b[:, 4].astype(np.int64) * b[:, 5])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    out14 = kind.endswith("14")
    esize = 2 if out14 else 1
    r, d, o = DeviceBuffer.from_numpy(ref), DeviceBuffer.from_numpy(b), DeviceBuffer.from_numpy(offs[:-1].copy())
    dst = DeviceBuffer(int(offs[-1]) * esize)
    f = L.kvz_hip_sample_luma_batch if kind.startswith("luma") else L.kvz_hip_sample_chroma_batch
    check(f(r.ptr, ref.shape[1], ref.shape[1], ref.shape[0], d.ptr, o.ptr, count, int(out14), dst.ptr, None),
          "sample %s batch" % kind)
    flat = dst.to_numpy(np.int16 if out14 else np.uint8, (int(offs[-1]),))
    return [flat[int(offs[i]):int(offs[i + 1])].reshape(int(b[i, 5]), int(b[i, 4])) for i in range(count)]


def search_frac_batch(pic, ref, pairs):
    """pairs: (x1, y1, x2, y2, w, h) with (x2,y2) the integer-pel position in ref.
    Returns (costs uint32 [count,17], best int32 [count,2])"""
    L = _lib.init()
    pic = np.ascontiguousarray(pic, dtype=np.uint8)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    pa = _pairs_array(pairs)
    count = pa.shape[0]
    a, b, d = DeviceBuffer.from_numpy(pic), DeviceBuffer.from_numpy(ref), DeviceBuffer.from_numpy(pa)
    co, be = DeviceBuffer(4 * 17 * count), DeviceBuffer(8 * count)
    check(L.kvz_hip_search_frac_batch(a.ptr, pic.shape[1], b.ptr, ref.shape[1], ref.shape[1], ref.shape[0], d.ptr,
                                      count, co.ptr, be.ptr, None), "search_frac batch")
    return co.to_numpy(np.uint32, (count, 17)), be.to_numpy(np.int32, (count, 2))


# ---- intra group ----
INTRA_LUMA, INTRA_FILTER_BOUNDARY, INTRA_RAW = 1, 2, 4


def intra_build_reference_batch(log2_width, color, plane, pic_w, pic_h, xy):
    """kvz_intra_build_reference for the PUs of `color` at the luma positions xy (count, 2), gathered on the device from
    the 2-D reconstruction plane of that colour; returns uint8 [count, 130] = kvz_intra_ref {left[65], top[65]}."""
    L = _lib.init()
    plane = np.ascontiguousarray(plane, dtype=np.uint8)
    xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
    count = xy.shape[0]
    p, q = DeviceBuffer.from_numpy(plane), DeviceBuffer.from_numpy(xy)
    out = DeviceBuffer(max(1, 130 * count))
    check(L.kvz_hip_intra_build_reference_batch(log2_width, color, p.ptr, plane.shape[1], pic_w, pic_h, q.ptr, count, out.ptr, None),
          "intra_build_reference batch")
    return out.to_numpy(np.uint8, (count, 130))


def intra_predict_batch(refs, log2_width, modes, flags=INTRA_LUMA | INTRA_FILTER_BOUNDARY):
    """refs: (count, 130) uint8 = kvz_intra_ref {left[65], top[65]}.  kvz_intra_predict for every PU x every mode of
    `modes`; returns uint8 [count, len(modes), N*N]."""
    L = _lib.init()
    refs = np.ascontiguousarray(refs, dtype=np.uint8).reshape(-1, 130)
    count, n = refs.shape[0], 1 << log2_width
    m = np.ascontiguousarray(modes, dtype=np.int8)
    r = DeviceBuffer.from_numpy(refs)
    out = DeviceBuffer(max(1, count * len(m) * n * n))
    check(L.kvz_hip_intra_predict_batch(log2_width, flags, r.ptr, count, m.ctypes.data, len(m), out.ptr, None), "intra_predict batch")
    return out.to_numpy(np.uint8, (count, len(m), n * n))


def intra_rough_batch(refs, log2_width, orig, flags=INTRA_LUMA | INTRA_FILTER_BOUNDARY, with_sad=False):
    """All 35 mode costs of search_intra_rough per PU: returns satd uint32 [count, 35] (and sad if with_sad)."""
    L = _lib.init()
    refs = np.ascontiguousarray(refs, dtype=np.uint8).reshape(-1, 130)
    count, n = refs.shape[0], 1 << log2_width
    orig = np.ascontiguousarray(orig, dtype=np.uint8).reshape(count, n * n)
    r, o = DeviceBuffer.from_numpy(refs), DeviceBuffer.from_numpy(orig)
    satd = DeviceBuffer(max(1, 4 * 35 * count))
    sad = DeviceBuffer(max(1, 4 * 35 * count)) if with_sad else None
    check(L.kvz_hip_intra_rough_batch(log2_width, flags, r.ptr, o.ptr, count, satd.ptr, sad.ptr if sad else None, None), "intra_rough batch")
    a = satd.to_numpy(np.uint32, (count, 35))
    return (a, sad.to_numpy(np.uint32, (count, 35))) if with_sad else a


# ---- AMVP / merge candidates of whole PUs ----
def inter_candidates_batch(params, cus, col_cus, ref_cus, pus):
    """kvz_hip_inter_candidates_batch: params = one kvz_hip_inter_params record (252 bytes), cus / col_cus / ref_cus = 2-D arrays of
    kvz_hip_cu_info records (20 bytes), pus = kvz_hip_me_pu records with x, y, width, height, pad set.  Returns (the completed
    descriptors as bytes [count, 64], the merge lists as bytes [count, 5, 12])."""
    L = _lib.init()
    params = np.ascontiguousarray(params)
    assert params.nbytes == 252
    pus = np.ascontiguousarray(pus)
    count = pus.shape[0]
    a, u = DeviceBuffer.from_numpy(np.ascontiguousarray(cus)), DeviceBuffer.from_numpy(pus)
    b = DeviceBuffer.from_numpy(np.ascontiguousarray(col_cus)) if col_cus is not None else None
    c = DeviceBuffer.from_numpy(np.ascontiguousarray(ref_cus)) if ref_cus is not None else None
    m = DeviceBuffer(max(1, 60 * count))
    check(L.kvz_hip_inter_candidates_batch(a.ptr, b.ptr if b else None, c.ptr if c else None, params.ctypes.data, u.ptr, count, m.ptr, None),
          "inter_candidates batch")
    return u.to_numpy(np.uint8, (count, 64)), m.to_numpy(np.uint8, (count, 5, 12))


def inter_candidates_multi_batch(pictures, pus):
    """kvz_hip_inter_candidates_multi_batch: pictures = list of (params, cus, col_cus, ref_cus) as inter_candidates_batch takes them;
    pus carry their picture in pad >> 2.  Returns (descriptors as bytes [count, 64], merge lists as bytes [count, 5, 12])."""
    import struct
    L = _lib.init()
    pus = np.ascontiguousarray(pus)
    count = pus.shape[0]
    keep, rec = [], b""
    for (params, cus, col_cus, ref_cus) in pictures:
        params = np.ascontiguousarray(params)
        assert params.nbytes == 252
        bufs = [DeviceBuffer.from_numpy(np.ascontiguousarray(m)) if m is not None else None for m in (cus, col_cus, ref_cus)]
        keep += bufs
        rec += struct.pack("<3Q", *[(b.ptr or 0) if b else 0 for b in bufs]) + params.tobytes() + struct.pack("<i", 0)
    assert len(rec) == 280 * len(pictures)
    table = DeviceBuffer.from_numpy(np.frombuffer(rec, dtype=np.uint8))
    u = DeviceBuffer.from_numpy(pus)
    m = DeviceBuffer(max(1, 60 * count))
    check(L.kvz_hip_inter_candidates_multi_batch(table.ptr, len(pictures), u.ptr, count, m.ptr, None), "inter_candidates multi batch")
    return u.to_numpy(np.uint8, (count, 64)), m.to_numpy(np.uint8, (count, 5, 12))


# ---- motion search of whole PUs ----
def search_pu_batch(pic, ref, pus, params, cabac=None, cost_to_beat=None):
    """pus: structured array laid out as kvz_hip_me_pu (64 bytes each), params: one kvz_hip_me_params record (96 bytes).
    cabac: kvz_hip_me_cabac snapshots (--mv-rdo); cost_to_beat: uint32 per PU (the best cost of the pictures searched before).
    Returns the raw results as int32 [count, 8] (= kvz_hip_me_result)."""
    L = _lib.init()
    pic = np.ascontiguousarray(pic, dtype=np.uint8)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    pus = np.ascontiguousarray(pus)
    params = np.ascontiguousarray(params)
    assert pus.dtype.itemsize == 64 and params.nbytes == 96
    count = pus.shape[0]
    cb = tb = None
    if cost_to_beat is not None:
        tb = DeviceBuffer.from_numpy(np.ascontiguousarray(cost_to_beat, dtype=np.uint32))
        params = params.copy()
        params.view(np.uint8).reshape(-1)[88:96] = np.frombuffer(np.uint64(tb.ptr).tobytes(), dtype=np.uint8)
    if cabac is not None:                       # --mv-rdo: kvz_hip_me_cabac snapshots, staged to the device
        cb = DeviceBuffer.from_numpy(np.ascontiguousarray(cabac).view(np.uint8))
        params = params.copy()
        params.view(np.uint8).reshape(-1)[80:88] = np.frombuffer(np.uint64(cb.ptr).tobytes(), dtype=np.uint8)
        if int(params.view(np.int32).reshape(-1)[19]) == 0:      # n_cabac: the number of snapshots handed over
            params.view(np.int32).reshape(-1)[19] = len(cabac)
    a, b, d = DeviceBuffer.from_numpy(pic), DeviceBuffer.from_numpy(ref), DeviceBuffer.from_numpy(pus.view(np.uint8))
    out = DeviceBuffer(max(1, 32 * count))
    check(L.kvz_hip_search_pu_batch(a.ptr, pic.shape[1], pic.shape[1], pic.shape[0], b.ptr, ref.shape[1], ref.shape[1], ref.shape[0],
                                    d.ptr, count, params.ctypes.data, out.ptr, None), "search_pu batch")
    return out.to_numpy(np.int32, (count, 8))


def search_pu_multi_batch(pics, refs, pus, params):
    """kvz_hip_search_pu_multi_batch: pics / refs = lists of equally sized 2-D planes, pus carry their pair in pad >> 2.
    Returns int32 [count, 8]."""
    L = _lib.init()
    pics = [np.ascontiguousarray(p, dtype=np.uint8) for p in pics]
    refs = [np.ascontiguousarray(r, dtype=np.uint8) for r in refs]
    assert len(pics) == len(refs) and all(p.shape == pics[0].shape for p in pics) and all(r.shape == refs[0].shape for r in refs)
    pus = np.ascontiguousarray(pus)
    params = np.ascontiguousarray(params)
    assert pus.dtype.itemsize == 64 and params.nbytes == 96
    count = pus.shape[0]
    dp, dr = [DeviceBuffer.from_numpy(p) for p in pics], [DeviceBuffer.from_numpy(r) for r in refs]
    tp = DeviceBuffer.from_numpy(np.array([b.ptr for b in dp], dtype=np.uint64))
    tr = DeviceBuffer.from_numpy(np.array([b.ptr for b in dr], dtype=np.uint64))
    d = DeviceBuffer.from_numpy(pus.view(np.uint8))
    out = DeviceBuffer(max(1, 32 * count))
    h, w = pics[0].shape
    rh, rw = refs[0].shape
    check(L.kvz_hip_search_pu_multi_batch(tp.ptr, w, w, h, tr.ptr, rw, rw, rh, len(pics), d.ptr, count, params.ctypes.data, out.ptr, None),
          "search_pu multi batch")
    return out.to_numpy(np.int32, (count, 8))


# ---- SAO group ----
def sao_edge_stats_batch(orig, rec, bw, bh):
    """-> int32 [count, 4 classes, 2 (sum, count), 5 categories]"""
    L = _lib.init()
    orig = np.ascontiguousarray(orig, dtype=np.uint8).reshape(-1, bw * bh)
    rec = np.ascontiguousarray(rec, dtype=np.uint8).reshape(-1, bw * bh)
    count = orig.shape[0]
    a, b, o = DeviceBuffer.from_numpy(orig), DeviceBuffer.from_numpy(rec), DeviceBuffer(max(1, 160 * count))
    check(L.kvz_hip_sao_edge_stats_batch(a.ptr, b.ptr, bw, bh, count, o.ptr, None), "sao_edge_stats")
    return o.to_numpy(np.int32, (count, 4, 2, 5))


def sao_edge_ddistortion_batch(orig, rec, bw, bh, offsets):
    """offsets int32 [count, 4, 5] -> int32 [count, 4]"""
    L = _lib.init()
    orig = np.ascontiguousarray(orig, dtype=np.uint8).reshape(-1, bw * bh)
    rec = np.ascontiguousarray(rec, dtype=np.uint8).reshape(-1, bw * bh)
    count = orig.shape[0]
    offs = np.ascontiguousarray(offsets, dtype=np.int32).reshape(count, 4, 5)
    a, b, f, o = DeviceBuffer.from_numpy(orig), DeviceBuffer.from_numpy(rec), DeviceBuffer.from_numpy(offs), DeviceBuffer(max(1, 16 * count))
    check(L.kvz_hip_sao_edge_ddistortion_batch(a.ptr, b.ptr, bw, bh, count, f.ptr, o.ptr, None), "sao_edge_ddistortion")
    return o.to_numpy(np.int32, (count, 4))


def sao_band_stats_batch(orig, rec, bw, bh):
    L = _lib.init()
    orig = np.ascontiguousarray(orig, dtype=np.uint8).reshape(-1, bw * bh)
    rec = np.ascontiguousarray(rec, dtype=np.uint8).reshape(-1, bw * bh)
    count = orig.shape[0]
    a, b, o = DeviceBuffer.from_numpy(orig), DeviceBuffer.from_numpy(rec), DeviceBuffer(max(1, 256 * count))
    check(L.kvz_hip_sao_band_stats_batch(a.ptr, b.ptr, bw, bh, count, o.ptr, None), "sao_band_stats")
    return o.to_numpy(np.int32, (count, 2, 32))


def sao_band_ddistortion_batch(orig, rec, bw, bh, band_pos, bands):
    L = _lib.init()
    orig = np.ascontiguousarray(orig, dtype=np.uint8).reshape(-1, bw * bh)
    rec = np.ascontiguousarray(rec, dtype=np.uint8).reshape(-1, bw * bh)
    count = orig.shape[0]
    bp = np.ascontiguousarray(band_pos, dtype=np.int32).reshape(count)
    bd = np.ascontiguousarray(bands, dtype=np.int32).reshape(count, 4)
    a, b, p, d, o = (DeviceBuffer.from_numpy(orig), DeviceBuffer.from_numpy(rec), DeviceBuffer.from_numpy(bp), DeviceBuffer.from_numpy(bd),
                     DeviceBuffer(max(1, 4 * count)))
    check(L.kvz_hip_sao_band_ddistortion_batch(a.ptr, b.ptr, bw, bh, count, p.ptr, d.ptr, o.ptr, None), "sao_band_ddistortion")
    return o.to_numpy(np.int32, (count,))


def sao_reconstruct_color_batch(plane, blocks, infos, color):
    """plane uint8 2-D; blocks int32 [count, 5] (x, y, w, h, sao_index); infos int32 [n, 14]; returns the filtered plane
    (pixels outside every block are copied from `plane`)"""
    L = _lib.init()
    plane = np.ascontiguousarray(plane, dtype=np.uint8)
    blocks = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 5)
    infos = np.ascontiguousarray(infos, dtype=np.int32).reshape(-1, 14)
    a, d = DeviceBuffer.from_numpy(plane), DeviceBuffer.from_numpy(plane)
    b, f = DeviceBuffer.from_numpy(blocks), DeviceBuffer.from_numpy(infos)
    check(L.kvz_hip_sao_reconstruct_color_batch(a.ptr, plane.shape[1], plane.shape[1], plane.shape[0], d.ptr, plane.shape[1],
                                                b.ptr, blocks.shape[0], f.ptr, infos.shape[0], color, None), "sao_reconstruct")
    return d.to_numpy(np.uint8, plane.shape)


def bipred_cost_batch(pic, ref0, ref1, cands):
    """cands: iterable of (x, y, w, h, mv0x, mv0y, mv1x, mv1y) (quarter-pel vectors) -> uint32 [count] SATD costs"""
    L = _lib.init()
    pic = np.ascontiguousarray(pic, dtype=np.uint8)
    ref0 = np.ascontiguousarray(ref0, dtype=np.uint8)
    ref1 = np.ascontiguousarray(ref1, dtype=np.uint8)
    assert ref0.shape == ref1.shape
    rec = np.zeros(len(cands), dtype=np.dtype([("g", "<i4", (4,)), ("mv", "<i2", (4,))]))
    for i, c in enumerate(cands):
        rec[i]["g"] = c[:4]
        rec[i]["mv"] = c[4:8]
    a, b, d, e = DeviceBuffer.from_numpy(pic), DeviceBuffer.from_numpy(ref0), DeviceBuffer.from_numpy(ref1), DeviceBuffer.from_numpy(rec.view(np.uint8))
    out = DeviceBuffer(max(1, 4 * len(rec)))
    check(L.kvz_hip_bipred_cost_batch(a.ptr, pic.shape[1], pic.shape[1], pic.shape[0], b.ptr, ref0.shape[1], d.ptr, ref1.shape[1],
                                      ref0.shape[1], ref0.shape[0], e.ptr, len(rec), out.ptr, None), "bipred_cost batch")
    return out.to_numpy(np.uint32, (len(rec),))


# ---- deblocking ----
def deblock_frame(y, u, v, cus, prm):
    """y, u, v: uint8 planes (u, v None with prm['chroma'] == 0); cus: kvz_hip_cu_info records [h/4, w/4] (20 bytes each);
    prm: one kvz_hip_deblock_params record (64 bytes).  Returns the filtered planes."""
    L = _lib.init()
    y = np.ascontiguousarray(y, dtype=np.uint8)
    cus = np.ascontiguousarray(cus)
    prm = np.ascontiguousarray(prm)
    assert cus.dtype.itemsize == 20 and prm.nbytes == 64
    dy, dc = DeviceBuffer.from_numpy(y), DeviceBuffer.from_numpy(cus.view(np.uint8))
    du = DeviceBuffer.from_numpy(np.ascontiguousarray(u, dtype=np.uint8)) if u is not None else None
    dv = DeviceBuffer.from_numpy(np.ascontiguousarray(v, dtype=np.uint8)) if v is not None else None
    check(L.kvz_hip_deblock_frame(dy.ptr, y.shape[1], du.ptr if du else None, dv.ptr if dv else None, u.shape[1] if u is not None else 0,
                                  y.shape[1], y.shape[0], dc.ptr, prm.ctypes.data, None), "deblock_frame")
    return (dy.to_numpy(np.uint8, y.shape), du.to_numpy(np.uint8, u.shape) if du else None, dv.to_numpy(np.uint8, v.shape) if dv else None)


# ---- search service: requests of many host threads in shared launches (include/kvz_hip.h, "search service") ----
class MeService:
    """kvz_hip_me_service: resident luma planes in numbered slots + kvz_hip_me_service_search, callable from many threads."""

    def __init__(self, width, height, max_pictures=8, max_threads=64):
        self.lib = _lib.init()
        cfg = np.zeros(8, dtype=np.int32)
        cfg[:4] = (width, height, max_pictures, max_threads)
        self.w, self.h = int(width), int(height)
        self.ptr = self.lib.kvz_hip_me_service_create(cfg.ctypes.data)
        if not self.ptr:
            raise KvzHipError("kvz_hip_me_service_create failed: %s" % self.lib.kvz_hip_last_error().decode())

    def put_plane(self, slot, plane):
        plane = np.ascontiguousarray(plane, dtype=np.uint8)
        assert plane.shape == (self.h, self.w)
        check(self.lib.kvz_hip_me_service_put_rect(self.ptr, slot, plane.ctypes.data, self.w, 0, 0, self.w, self.h), "me_service_put_rect")

    def put_rect(self, slot, plane, x, y, w, h):
        """rectangle (x, y, w, h) of a full host plane"""
        plane = np.ascontiguousarray(plane, dtype=np.uint8)
        check(self.lib.kvz_hip_me_service_put_rect(self.ptr, slot, plane.ctypes.data + y * plane.shape[1] + x, plane.shape[1], x, y, w, h),
              "me_service_put_rect")

    def search(self, request):
        """request: one record laid out as kvz_hip_me_request (1200 bytes).  Returns int32 [n_refs, 8] (= kvz_hip_me_result)."""
        request = np.ascontiguousarray(request)
        assert request.nbytes == 1200
        n = int(request.view(np.int32).reshape(-1)[1])
        out = np.zeros((max(n, 1), 8), dtype=np.int32)
        check(self.lib.kvz_hip_me_service_search(self.ptr, request.ctypes.data, out.ctypes.data), "me_service_search")
        return out[:n]

    def stats(self):
        s = np.zeros(11, dtype=np.uint64)
        check(self.lib.kvz_hip_me_service_get_stats(self.ptr, s.ctypes.data), "me_service_get_stats")
        return dict(zip(("requests", "units", "batches", "launches", "max_batch_units", "rects", "rect_bytes", "wait_ns", "tables", "table_bytes", "table_ns"),
                        (int(v) for v in s)))

    def sad_tables(self, pic_slot, ref_slots, ctu_x, ctu_y, rng):
        """kvz_hip_me_service_sad_tables -> uint32 [n_refs, 2 rng + 1 (dy), 2 rng + 1 (dx), 85] (a copy of the thread's table)"""
        import ctypes as C
        refs = np.ascontiguousarray(ref_slots, dtype=np.int32)
        p = self.lib.kvz_hip_me_service_sad_tables(self.ptr, pic_slot, len(refs), refs.ctypes.data, ctu_x, ctu_y, rng)
        if not p:
            raise KvzHipError("kvz_hip_me_service_sad_tables failed: %s" % self.lib.kvz_hip_last_error().decode())
        side = 2 * rng + 1
        n = len(refs) * side * side * 85
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n,)).reshape(len(refs), side, side, 85).copy()

    def close(self):
        if self.ptr:
            self.lib.kvz_hip_me_service_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
