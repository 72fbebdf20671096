"""ctypes binding of oracle/_ref/libkvzref.so -- the REFERENCE compiled from
/root/reference by oracle/Makefile plus our harness glue (oracle/ref_harness.c).
TEST INFRASTRUCTURE: used to pin the oracle, to generate tests/golden/, and as
bench.py's cpu_baseline (kind "reference").  The .so is prebuilt in the build
container and travels to the GPU box; /root/reference itself is never read at
run time."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(os.path.dirname(_HERE), "oracle", "_ref", "libkvzref.so")
_LIB = None

u8p = C.POINTER(C.c_uint8)
i16p = C.POINTER(C.c_int16)
u32p = C.POINTER(C.c_uint32)
S = C.c_char_p


def available():
    return os.path.exists(REF_SO)


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(REF_SO, mode=C.RTLD_GLOBAL)
        L.ref_init.restype = C.c_int
        L.ref_strategy.restype = C.c_void_p
        L.ref_strategy.argtypes = [S, S]
        L.ref_list.restype = C.c_void_p
        L.ref_strategy_type.restype = S
        L.ref_strategy_name.restype = S
        L.ref_reg_sad.restype = C.c_uint
        L.ref_reg_sad.argtypes = [S, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_uint]
        L.ref_cost_nxn.restype = C.c_uint
        L.ref_cost_nxn.argtypes = [S, S, C.c_void_p, C.c_void_p]
        L.ref_cost_nxn_dual.restype = None
        L.ref_cost_nxn_dual.argtypes = [S, S, C.c_void_p, C.c_void_p, u32p]
        L.ref_satd_any_size.restype = C.c_uint
        L.ref_satd_any_size.argtypes = [S, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.ref_satd_any_size_quad.restype = None
        L.ref_satd_any_size_quad.argtypes = [S, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_int, u32p]
        L.ref_pixels_calc_ssd.restype = C.c_uint
        L.ref_pixels_calc_ssd.argtypes = [S, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ref_transform.restype = None
        L.ref_transform.argtypes = [S, S, i16p, i16p]
        L.ref_coeff_abs_sum.restype = C.c_uint32
        L.ref_coeff_abs_sum.argtypes = [S, i16p, C.c_size_t]
        L.ref_quant.restype = None
        L.ref_quant.argtypes = [S] + [C.c_int] * 4 + [i16p, i16p] + [C.c_int] * 5
        L.ref_dequant.restype = None
        L.ref_dequant.argtypes = [S, C.c_int, C.c_int, i16p, i16p] + [C.c_int] * 4
        L.ref_quantize_residual.restype = C.c_int
        L.ref_quantize_residual.argtypes = [S] + [C.c_int] * 11 + [u8p, u8p, u8p, i16p]
        L.ref_quant_coeff_table.restype = C.POINTER(C.c_int32)
        L.ref_quant_coeff_table.argtypes = [C.c_int] * 3
        L.ref_dequant_coeff_table.restype = C.POINTER(C.c_int32)
        L.ref_dequant_coeff_table.argtypes = [C.c_int] * 3
        for f, dt in (("ref_sample_luma", u8p), ("ref_sample_luma_14bit", i16p),
                      ("ref_sample_chroma", u8p), ("ref_sample_chroma_14bit", i16p)):
            getattr(L, f).restype = None
            getattr(L, f).argtypes = [S, C.c_void_p, C.c_int, C.c_int, C.c_int, dt, C.c_int, C.c_int, C.c_int]
        for f in ("ref_image_calc_sad", "ref_image_calc_satd"):
            getattr(L, f).restype = C.c_uint
            getattr(L, f).argtypes = [S, u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int] + [C.c_int] * 6
        L.ref_search_frac_costs.restype = None
        L.ref_search_frac_costs.argtypes = [S, u8p, C.c_int, u8p, C.c_int, C.c_int] + [C.c_int] * 6 + \
            [u32p, C.POINTER(C.c_int)]
        L.ref_filter_step.restype = None
        L.ref_filter_step.argtypes = [S, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, u8p, i16p, i16p,
                                      C.c_int, C.c_int, C.c_int]
        L.ref_bipred.restype = None
        L.ref_bipred.argtypes = [S] + [C.c_int] * 8 + [i16p] * 6 + [u8p] * 6
        L.ref_bench_cost_nxn.restype = C.c_double
        L.ref_bench_cost_nxn.argtypes = [S, S, C.c_int, u8p, u8p, C.c_size_t, C.c_double, u32p]
        L.ref_bench_transform.restype = C.c_double
        L.ref_bench_transform.argtypes = [S, S, C.c_int, i16p, i16p, C.c_size_t, C.c_double]
        L.ref_bench_reg_sad.restype = C.c_double
        L.ref_bench_reg_sad.argtypes = [S, u8p, u8p] + [C.c_int] * 5 + [C.c_double, u32p]
        L.ref_cost_nxn_many.restype = C.c_int
        L.ref_cost_nxn_many.argtypes = [S, S, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ref_transform_many.restype = C.c_int
        L.ref_transform_many.argtypes = [S, S, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        # the selector prints its "Available/In use" banner on stderr
        if not L.ref_init():
            raise RuntimeError("reference strategyselector init failed")
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _aligned(a, align=64):
    """copy into a 64-byte aligned buffer (avx2 sad_NxN uses aligned loads, picture-avx2.c:44)"""
    a = np.ascontiguousarray(a)
    raw = np.empty(a.nbytes + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    out = raw[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def has_strategy(type_, name):
    return bool(lib().ref_strategy(type_.encode(), name.encode()))


def strategies():
    L = lib()
    return [(L.ref_strategy_type(i).decode(), L.ref_strategy_name(i).decode(), L.ref_strategy_priority(i))
            for i in range(L.ref_strategy_count())]


def reg_sad(a, b, off1, off2, w, h, s1, s2, name="generic"):
    a, b = _u8(a).ravel(), _u8(b).ravel()
    return lib().ref_reg_sad(name.encode(), a.ctypes.data + off1, b.ctypes.data + off2, w, h, s1, s2)


def cost_nxn_batch(kind, n, blk1, blk2, name="generic"):
    blk1 = _aligned(_u8(blk1).reshape(-1, n * n))
    blk2 = _aligned(_u8(blk2).reshape(-1, n * n))
    t = ("%s_%dx%d" % (kind, n, n)).encode()
    out = np.empty(blk1.shape[0], dtype=np.uint32)
    for i in range(blk1.shape[0]):
        out[i] = lib().ref_cost_nxn(t, name.encode(), blk1[i].ctypes.data, blk2[i].ctypes.data)
    return out


def cost_nxn_many(kind, n, blk1, blk2, name="generic", threads=None):
    """whole launches: the strategy function looped in C, one range of blocks per host thread"""
    from oracle_lib import run_ranges
    blk1 = _aligned(_u8(blk1).reshape(-1, n * n))
    blk2 = _aligned(_u8(blk2).reshape(-1, n * n))
    t, nm, bs = ("%s_%dx%d" % (kind, n, n)).encode(), name.encode(), n * n
    out = np.empty(blk1.shape[0], dtype=np.uint32)
    L = lib()

    def part(lo, hi):
        assert L.ref_cost_nxn_many(t, nm, n, blk1.ctypes.data + lo * bs, blk2.ctypes.data + lo * bs, hi - lo, out.ctypes.data + 4 * lo) == 0
    run_ranges(blk1.shape[0], part, threads)
    return out


def transform_many(kind, n, blocks, name="generic", threads=None):
    from oracle_lib import run_ranges
    blocks = _aligned(np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, n * n))
    out = _aligned(np.zeros_like(blocks))
    t, nm, bs = _TR_TYPE[(kind, n)].encode(), name.encode(), 2 * n * n
    L = lib()

    def part(lo, hi):
        assert L.ref_transform_many(t, nm, n, blocks.ctypes.data + lo * bs, out.ctypes.data + lo * bs, hi - lo) == 0
    run_ranges(blocks.shape[0], part, threads)
    return out


def cost_nxn_dual_batch(kind, n, preds, orig, name="generic"):
    """preds uint8 [count, 2048] (pred_buffer: pred k at k*1024)"""
    orig = _aligned(_u8(orig).reshape(-1, n * n))
    preds = _aligned(_u8(preds).reshape(orig.shape[0], 2048))
    t = ("%s_%dx%d_dual" % (kind, n, n)).encode()
    out = np.empty((orig.shape[0], 2), dtype=np.uint32)
    for i in range(orig.shape[0]):
        lib().ref_cost_nxn_dual(t, name.encode(), preds[i].ctypes.data, orig[i].ctypes.data, _p(out[i], u32p))
    return out


def satd_any_size(w, h, a, off1, s1, b, off2, s2, name="generic"):
    a, b = _u8(a).ravel(), _u8(b).ravel()
    return lib().ref_satd_any_size(name.encode(), w, h, a.ctypes.data + off1, s1, b.ctypes.data + off2, s2)


def satd_any_size_quad(w, h, preds4, stride, orig, orig_off, orig_stride, name="generic"):
    ps = [_u8(p).ravel() for p in preds4]
    orig = _u8(orig).ravel()
    out = np.zeros(4, dtype=np.uint32)
    lib().ref_satd_any_size_quad(name.encode(), w, h, *[p.ctypes.data for p in ps], stride,
                                 orig.ctypes.data + orig_off, orig_stride, _p(out, u32p))
    return out


def pixels_calc_ssd(a, off1, b, off2, s1, s2, w, name="generic"):
    a, b = _u8(a).ravel(), _u8(b).ravel()
    return lib().ref_pixels_calc_ssd(name.encode(), a.ctypes.data + off1, b.ctypes.data + off2, s1, s2, w)


def image_calc(kind, pic, ref, pic_x, pic_y, ref_x, ref_y, bw, bh, name="generic"):
    pic, ref = _u8(pic), _u8(ref)
    f = lib().ref_image_calc_sad if kind == "sad" else lib().ref_image_calc_satd
    return f(name.encode(), _p(pic, u8p), pic.shape[1], pic.shape[0], _p(ref, u8p), ref.shape[1], ref.shape[0],
             pic_x, pic_y, ref_x, ref_y, bw, bh)


_TR_TYPE = {("dct", 4): "dct_4x4", ("dct", 8): "dct_8x8", ("dct", 16): "dct_16x16", ("dct", 32): "dct_32x32",
            ("idct", 4): "idct_4x4", ("idct", 8): "idct_8x8", ("idct", 16): "idct_16x16", ("idct", 32): "idct_32x32",
            ("dst", 4): "fast_forward_dst_4x4", ("idst", 4): "fast_inverse_dst_4x4"}


def transform_batch(kind, n, blocks, name="generic"):
    blocks = _aligned(np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, n * n))
    out = _aligned(np.zeros_like(blocks))
    t = _TR_TYPE[(kind, n)].encode()
    for i in range(blocks.shape[0]):
        lib().ref_transform(t, name.encode(), _p(blocks[i], i16p), _p(out[i], i16p))
    return np.array(out)


def coeff_abs_sum(c, name="generic"):
    c = _aligned(np.ascontiguousarray(c, dtype=np.int16).ravel())
    return lib().ref_coeff_abs_sum(name.encode(), _p(c, i16p), c.size)


def quant_batch(coef, w, qp, type_, scan_idx, slice_is_intra=0, signhide=0, block_is_intra=0, sl=0, name="generic"):
    coef = _aligned(np.ascontiguousarray(coef, dtype=np.int16).reshape(-1, w * w))
    out = _aligned(np.zeros_like(coef))
    bt = 1 if block_is_intra else 2   # CU_INTRA = 1, CU_INTER = 2 (cu.h:39-41)
    for i in range(coef.shape[0]):
        lib().ref_quant(name.encode(), qp, int(slice_is_intra), int(signhide), sl, _p(coef[i], i16p), _p(out[i], i16p),
                        w, w, type_, scan_idx, bt)
    return np.array(out)


def dequant_batch(q_coef, w, qp, type_, block_is_intra=0, sl=0, name="generic"):
    q_coef = _aligned(np.ascontiguousarray(q_coef, dtype=np.int16).reshape(-1, w * w))
    out = _aligned(np.zeros_like(q_coef))
    bt = 1 if block_is_intra else 2
    for i in range(q_coef.shape[0]):
        lib().ref_dequant(name.encode(), qp, sl, _p(q_coef[i], i16p), _p(out[i], i16p), w, w, type_, bt)
    return np.array(out)


def scaling_tables(log2_tr, list_type, qp_rem, n):
    L = lib()
    q = np.ctypeslib.as_array(L.ref_quant_coeff_table(log2_tr, list_type, qp_rem), shape=(n * n,)).copy()
    d = np.ctypeslib.as_array(L.ref_dequant_coeff_table(log2_tr, list_type, qp_rem), shape=(n * n,)).copy()
    return q, d


def quantize_residual_batch(ref_in, pred_in, w, qp, color, scan_order_, cu_is_intra, slice_is_intra=0,
                            signhide=0, use_trskip=0, name="generic"):
    ref_in, pred_in = _u8(ref_in).reshape(-1, w * w), _u8(pred_in).reshape(-1, w * w)
    rec = np.zeros_like(ref_in)
    coeff = _aligned(np.zeros(ref_in.shape, dtype=np.int16))
    has = np.zeros(ref_in.shape[0], dtype=np.int32)
    for i in range(ref_in.shape[0]):
        has[i] = lib().ref_quantize_residual(name.encode(), qp, int(slice_is_intra), int(signhide), 0,
                                             int(cu_is_intra), w, color, scan_order_, int(use_trskip), w, w,
                                             _p(ref_in[i], u8p), _p(pred_in[i], u8p), _p(rec[i], u8p),
                                             _p(coeff[i], i16p))
    return rec, np.array(coeff), has


def sample(kind, frame, x, y, w, h, mvx, mvy, name="generic"):
    frame = _u8(frame)
    stride = frame.shape[1]
    src = frame.ctypes.data + y * stride + x
    if kind in ("luma", "chroma"):
        dst = np.zeros((h, w), dtype=np.uint8)
        f = lib().ref_sample_luma if kind == "luma" else lib().ref_sample_chroma
        f(name.encode(), src, stride, w, h, _p(dst, u8p), w, mvx, mvy)
    else:
        dst = np.zeros((h, w), dtype=np.int16)
        f = lib().ref_sample_luma_14bit if kind == "luma14" else lib().ref_sample_chroma_14bit
        f(name.encode(), src, stride, w, h, _p(dst, i16p), w, mvx, mvy)
    return dst


def get_extended_block(frame, xpos, ypos, mv_x, mv_y, filter_size, width, height, off_x=0, off_y=0, name="generic"):
    """epol_func of the named strategy -> (window [height + fs, width + fs], malloc_used, (stride, buffer - ref or -1, topleft - buffer))"""
    L = lib()
    L.ref_get_extended_block.restype = C.c_int
    L.ref_get_extended_block.argtypes = [S] + [C.c_int] * 6 + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_long)]
    frame = _u8(frame)
    half = filter_size >> 1
    win = np.zeros((height + 2 * half, width + 2 * half), dtype=np.uint8)
    info = (C.c_long * 3)()
    used = L.ref_get_extended_block(name.encode(), xpos, ypos, mv_x, mv_y, off_x, off_y, frame.ctypes.data, frame.shape[1], frame.shape[0],
                                    filter_size, width, height, win.ctypes.data, info)
    assert used >= 0
    return win, used, tuple(info)


def filter_frac_steps(frame, x, y, w, h, offs, fme_level=4, name="generic"):
    frame = _u8(frame)
    stride = frame.shape[1]
    src = frame.ctypes.data + y * stride + x
    inter = _aligned(np.zeros(5 * 72 * 64, dtype=np.int16))
    cols = np.zeros(5 * 72, dtype=np.int16)
    out = _aligned(np.zeros((4, 4, 64, 64), dtype=np.uint8))
    for step in range(4):
        ox, oy = (0, 0) if step < 2 else offs
        lib().ref_filter_step(name.encode(), step, src, stride, w, h, _p(out[step], u8p), _p(inter, i16p),
                              _p(cols, i16p), fme_level, ox, oy)
    return np.array(out)


def search_frac_costs(pic, ref, x, y, w, h, mvx, mvy, name="generic"):
    pic, ref = _u8(pic), _u8(ref)
    costs = np.zeros(17, dtype=np.uint32)
    best = (C.c_int * 2)()
    lib().ref_search_frac_costs(name.encode(), _p(pic, u8p), pic.shape[1], _p(ref, u8p), ref.shape[1], ref.shape[0],
                                x, y, w, h, mvx, mvy, _p(costs, u32p), best)
    return costs, (best[0], best[1])


def bipred(hi, height, width, ypos, xpos, hp0, hp1, rec, tmp, name="generic"):
    """hi = (luma0, luma1, chroma0, chroma1); hp0/hp1 = (y[4096], u[1024], v[1024]) int16;
    rec/tmp = (y, u, v) uint8.  Returns the updated rec planes."""
    h0 = [np.ascontiguousarray(a, dtype=np.int16).copy() for a in hp0]
    h1 = [np.ascontiguousarray(a, dtype=np.int16).copy() for a in hp1]
    r = [np.ascontiguousarray(a, dtype=np.uint8).copy() for a in rec]
    t = [np.ascontiguousarray(a, dtype=np.uint8).copy() for a in tmp]
    lib().ref_bipred(name.encode(), *[int(v) for v in hi], height, width, ypos, xpos,
                     *[_p(a, i16p) for a in h0 + h1], *[_p(a, u8p) for a in r + t])
    return r


def synthetic_sequence(w, h, n, seed=5):
    """n 4:2:0 frames (n, h*3/2, w) with half-pel global motion over a band-limited texture plus noise, so that
    intra, integer ME, fractional ME, residual coding and skip all occur."""
    g = np.random.default_rng(seed)
    big = g.integers(0, 256, (2 * h + 64, 2 * w + 64)).astype(np.float64)
    for _ in range(3):
        big = (big + np.roll(big, 1, 0) + np.roll(big, 1, 1) + np.roll(big, (1, 1), (0, 1))) / 4
    big = np.clip((big - big.mean()) * 6 + 128, 0, 255)
    frames = np.zeros((n, h * 3 // 2, w), dtype=np.uint8)
    for i in range(n):
        k = i % 24
        k = k if k <= 12 else 24 - k          # the pan turns round after 12 frames (the texture has a 64-pixel margin)
        oy, ox = 3 * k, 5 * k
        y = big[oy:oy + 2 * h:2, ox:ox + 2 * w:2] + g.normal(0, 1.5, (h, w))
        frames[i, :h] = np.clip(y, 0, 255).astype(np.uint8)
        c = big[oy:oy + 2 * h:4, ox:ox + 2 * w:4]
        frames[i, h:h + h // 4] = np.clip(c * 0.5 + 64, 0, 255).astype(np.uint8).reshape(h // 4, w)
        frames[i, h + h // 4:] = np.clip(255 - c * 0.5, 0, 255).astype(np.uint8).reshape(h // 4, w)
    return frames


def encode(frames, w, h, opts, strategy=None, cap=None):
    """Run the reference encoder (kvz_api) over the frames; strategy=None keeps the selector's own choice,
    otherwise every type registered under that name is installed.  Returns (bitstream bytes, n installed)."""
    L = lib()
    L.ref_encode.restype = C.c_long
    L.ref_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, C.c_long,
                             C.POINTER(C.c_int)]
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    if cap is None:
        cap = max(1 << 22, 2 * w * h * frames.shape[0])          # noise-like test sequences compress badly
    out = np.zeros(cap, dtype=np.uint8)
    inst = C.c_int(0)
    n = L.ref_encode(frames.ctypes.data, w, h, frames.shape[0], opts.encode(), strategy.encode() if strategy else None,
                     out.ctypes.data, cap, C.byref(inst))
    assert 0 < n <= cap, "ref_encode failed (%d)" % n
    return out[:n].tobytes(), inst.value


def record_inter_searches(frames, w, h, opts, max_records=400000, snapshots=0):
    """Runs the reference encoder over the frames with the harness recorder on: every 2Nx2N inter search the encoder
    itself performed on a single-reference P frame, with the candidates the encoder derived for it and what the
    reference search decided.  -> dict(pus [n] ME_PU, results [n] ME_RESULT, meta [n] (frame, lcu_x, lcu_y, seq, lambda_cost,
    depth), params ME_PARAMS (lambda_cost 0: per record), pic / ref uint8 [frames, h, w], skipped).
    snapshots > 0 also keeps, for the first that many searches, what the encoder's candidate derivation read: snap_index [m] (record
    of each snapshot), snap_cus [m, 290] CU_INFO (lcu->cu, cu.h:324), and per frame snap_col [frames, rows, stride] CU_INFO (the
    collocated picture's CU array) and snap_params [frames] INTER_PARAMS."""
    from patterns import CU_INFO, INTER_PARAMS, ME_PARAMS, ME_PU, ME_RESULT
    L = lib()
    L.ref_record_begin.restype = C.c_int
    L.ref_record_begin.argtypes = [C.c_int] * 4
    L.ref_record_end.restype = C.c_int
    L.ref_record_end.argtypes = [C.c_void_p] * 6 + [C.POINTER(C.c_int)]
    nf = len(frames)
    assert L.ref_record_begin(max_records, nf, w, h) == 0
    snap = {}
    if snapshots:
        L.ref_record_snapshots.restype = C.c_int
        L.ref_record_snapshots.argtypes = [C.c_int]
        L.ref_record_snapshots_get.restype = C.c_int
        L.ref_record_snapshots_get.argtypes = [C.c_void_p] * 4 + [C.POINTER(C.c_int)]
        assert L.ref_record_snapshots(snapshots) == 0
    try:
        bitstream, _ = encode(frames, w, h, opts)
    finally:
        if snapshots:
            rows, stride = ((h + 63) // 64) * 16, ((w + 63) // 64) * 16
            s_idx = np.zeros(snapshots, dtype=np.int32)
            s_cu = np.zeros((snapshots, 290), dtype=CU_INFO)
            s_col = np.zeros((nf, rows, stride), dtype=CU_INFO)
            s_prm = np.zeros(nf, dtype=INTER_PARAMS)
            dims = (C.c_int * 2)()
            m = L.ref_record_snapshots_get(s_idx.ctypes.data, s_cu.ctypes.data, s_col.ctypes.data, s_prm.ctypes.data, dims)
            assert (dims[0], dims[1]) == (rows, stride)
            snap = dict(snap_index=s_idx[:m].copy(), snap_cus=s_cu[:m].copy(), snap_col=s_col, snap_params=s_prm)
        pus = np.zeros(max_records, dtype=ME_PU)
        res = np.zeros(max_records, dtype=ME_RESULT)
        meta = np.zeros((max_records, 6), dtype=np.int32)
        prm = np.zeros(1, dtype=ME_PARAMS)
        pic = np.zeros((nf, h, w), dtype=np.uint8)
        ref = np.zeros((nf, h, w), dtype=np.uint8)
        info = (C.c_int * 2)()
        n = L.ref_record_end(pus.ctypes.data, res.ctypes.data, meta.ctypes.data, prm.ctypes.data, pic.ctypes.data, ref.ctypes.data, info)
    out = dict(pus=pus[:n].copy(), results=res[:n].copy(), meta=meta[:n].copy(), params=prm, pic=pic[:info[0]].copy(), ref=ref[:info[0]].copy(),
               skipped=int(info[1]), bitstream=bitstream)
    if snap:
        keep = snap["snap_index"] < n                       # a last snapshot whose search was not recorded
        snap["snap_index"], snap["snap_cus"] = snap["snap_index"][keep], snap["snap_cus"][keep]
        snap["snap_col"], snap["snap_params"] = snap["snap_col"][:info[0]].copy(), snap["snap_params"][:info[0]].copy()
        out.update(snap)
    return out


# ---- intra group ----
def _intra_sigs():
    L = lib()
    if getattr(L, "_intra_done", False):
        return L
    L.ref_angular_pred.restype = None
    L.ref_angular_pred.argtypes = [S, C.c_int, C.c_int, u8p, u8p, u8p]
    L.ref_intra_pred_planar.restype = None
    L.ref_intra_pred_planar.argtypes = [S, C.c_int, u8p, u8p, u8p]
    L.ref_intra_predict.restype = None
    L.ref_intra_predict.argtypes = [S, u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
    L.ref_intra_build_reference.restype = None
    L.ref_intra_build_reference.argtypes = [C.c_int] * 6 + [u8p, u8p, u8p, C.c_int, u8p]
    L._intra_done = True
    return L


def angular_pred(log2_width, mode, above, left, name="generic"):
    L = _intra_sigs()
    n = 1 << log2_width
    above, left = _u8(above), _u8(left)
    dst = _aligned(np.zeros(n * n, dtype=np.uint8))
    L.ref_angular_pred(name.encode(), log2_width, mode, _p(above, u8p), _p(left, u8p), _p(dst, u8p))
    return dst.copy()


def intra_pred_planar(log2_width, top, left, name="generic"):
    L = _intra_sigs()
    n = 1 << log2_width
    top, left = _u8(top), _u8(left)
    dst = _aligned(np.zeros(n * n, dtype=np.uint8))
    L.ref_intra_pred_planar(name.encode(), log2_width, _p(top, u8p), _p(left, u8p), _p(dst, u8p))
    return dst.copy()


def intra_predict(ref130, log2_width, mode, color=0, filter_boundary=1, name="generic"):
    """kvz_intra_predict (color 0 = COLOR_Y) on one kvz_intra_ref {left[65], top[65]}"""
    L = _intra_sigs()
    n = 1 << log2_width
    r = _u8(ref130)
    dst = _aligned(np.zeros(n * n, dtype=np.uint8))
    L.ref_intra_predict(name.encode(), _p(r, u8p), log2_width, mode, color, filter_boundary, _p(dst, u8p))
    return dst.copy()


def intra_build_reference(log2_width, x, y, pic_w, pic_h, rec, top, left, top_left, color=0):
    """kvz_intra_build_reference for the PU of `color` at luma picture position (x, y).  rec / top / left are the LCU's
    planes of that colour: rec 64x64 (32x32 chroma), top / left 97 (49) entries of which entry 0 is replaced by top_left."""
    L = _intra_sigs()
    out = np.zeros(130, dtype=np.uint8)
    rec, top, left = _u8(rec), _u8(top), _u8(left)
    w, nref = (32, 49) if color else (64, 97)
    assert rec.size == w * w and top.size >= nref and left.size >= nref
    L.ref_intra_build_reference(log2_width, color, x, y, pic_w, pic_h, _p(rec, u8p), _p(top, u8p), _p(left, u8p),
                                int(top_left), _p(out, u8p))
    return out


def intra_build_reference_from_plane(log2_width, color, plane, pic_w, pic_h, x, y, poison=None):
    """kvz_intra_build_reference for the PU at luma position (x, y), its lcu_t filled from a whole reconstruction plane
    of `color` the way init_lcu_t hands it over (search.c:761-835: rec of the LCU, top_ref / left_ref = the row
    above / the column left of it).  With a poison generator, every 4x4 unit of the LCU that follows the PU in coding
    order is overwritten first, so a read of a not yet coded pixel shows."""
    c = 1 if color else 0
    plane = np.asarray(plane, dtype=np.uint8)
    ph, pw = plane.shape
    w, nref = (32, 49) if c else (64, 97)
    lx, ly = (x // 64) * 64 >> c, (y // 64) * 64 >> c
    pad = np.zeros((ph + 2 * w + 1, pw + 2 * w + 1), dtype=np.uint8)
    pad[1:ph + 1, 1:pw + 1] = plane
    rec = pad[1 + ly:1 + ly + w, 1 + lx:1 + lx + w].copy()
    if poison is not None:
        def z(ux, uy):
            return sum((((ux >> b) & 1) << (2 * b)) | (((uy >> b) & 1) << (2 * b + 1)) for b in range(4))
        ux0, uy0, u = (x % 64) // 4, (y % 64) // 4, 4 >> c
        for uy in range(16):
            for ux in range(16):
                if z(ux, uy) >= z(ux0, uy0):
                    rec[uy * u:(uy + 1) * u, ux * u:(ux + 1) * u] = poison.integers(0, 256, (u, u), dtype=np.uint8)
    top = pad[ly, lx:lx + nref].copy()
    left = pad[ly:ly + nref, lx].copy()
    return intra_build_reference(log2_width, x, y, pic_w, pic_h, rec, top, left, pad[ly, lx], color=color)


# ---- motion search: the reference's static hexagon_search + search_frac (oracle/ref_me_harness.c) ----
def search_pu_batch(pic, ref, pus, params, cabac=None, cost_to_beat=None):
    from patterns import ME_RESULT
    L = lib()
    L.ref_me_search_pu.restype = None
    L.ref_me_search_pu.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    pic, ref = _u8(pic), _u8(ref)
    assert pic.shape == ref.shape
    pus = np.ascontiguousarray(pus)
    params = np.ascontiguousarray(params).copy()
    if cabac is not None:                       # --mv-rdo: ME_CABAC snapshots; pus["reserved"] indexes them
        cabac = np.ascontiguousarray(cabac)
        params["cabac"] = cabac.ctypes.data
    assert params.nbytes == 96
    if cost_to_beat is not None:
        cost_to_beat = np.ascontiguousarray(cost_to_beat, dtype=np.uint32)
    out = np.zeros(len(pus), dtype=ME_RESULT)
    for i in range(len(pus)):
        if cost_to_beat is not None:
            params["cost_to_beat"] = cost_to_beat.ctypes.data + 4 * i          # the harness reads this PU's entry
        L.ref_me_search_pu(_p(pic, u8p), _p(ref, u8p), pic.shape[1], pic.shape[0],
                           pus.ctypes.data + 64 * i, params.ctypes.data, out.ctypes.data + 32 * i)
    return out


# ---- SAO group ----
def _sao_sigs():
    L = lib()
    if getattr(L, "_sao_done", False):
        return L
    i32p = C.POINTER(C.c_int32)
    L.ref_sao_edge_ddistortion.restype = C.c_int
    L.ref_sao_edge_ddistortion.argtypes = [S, u8p, u8p, C.c_int, C.c_int, C.c_int, i32p]
    L.ref_calc_sao_edge_dir.restype = None
    L.ref_calc_sao_edge_dir.argtypes = [S, u8p, u8p, C.c_int, C.c_int, C.c_int, i32p]
    L.ref_sao_band_ddistortion.restype = C.c_int
    L.ref_sao_band_ddistortion.argtypes = [S, u8p, u8p, C.c_int, C.c_int, C.c_int, i32p]
    L.ref_sao_reconstruct_color.restype = None
    L.ref_sao_reconstruct_color.argtypes = [S, C.c_void_p, C.c_void_p, i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ref_sizeof_sao_info.restype = C.c_int
    L._sao_done = True
    return L


def sao_edge_ddistortion(orig, rec, bw, bh, eo_class, offsets, name="generic"):
    L = _sao_sigs()
    orig, rec = _aligned(_u8(orig)), _aligned(_u8(rec))
    o = np.ascontiguousarray(offsets, dtype=np.int32)
    return L.ref_sao_edge_ddistortion(name.encode(), _p(orig, u8p), _p(rec, u8p), bw, bh, eo_class, _p(o, C.POINTER(C.c_int32)))


def calc_sao_edge_dir(orig, rec, eo_class, bw, bh, name="generic"):
    L = _sao_sigs()
    orig, rec = _aligned(_u8(orig)), _aligned(_u8(rec))
    out = np.zeros((2, 5), dtype=np.int32)
    L.ref_calc_sao_edge_dir(name.encode(), _p(orig, u8p), _p(rec, u8p), eo_class, bw, bh, _p(out, C.POINTER(C.c_int32)))
    return out


def sao_band_ddistortion(orig, rec, bw, bh, band_pos, bands, name="generic"):
    L = _sao_sigs()
    orig, rec = _aligned(_u8(orig)), _aligned(_u8(rec))
    b = np.ascontiguousarray(bands, dtype=np.int32)
    return L.ref_sao_band_ddistortion(name.encode(), _p(orig, u8p), _p(rec, u8p), bw, bh, band_pos, _p(b, C.POINTER(C.c_int32)))


def sao_reconstruct_color(plane, x, y, bw, bh, sao14, color, name="generic"):
    L = _sao_sigs()
    plane = _u8(plane)
    stride = plane.shape[1]
    out = np.zeros((bh, bw), dtype=np.uint8)
    s = np.ascontiguousarray(sao14, dtype=np.int32)
    L.ref_sao_reconstruct_color(name.encode(), plane.ctypes.data + y * stride + x, out.ctypes.data, _p(s, C.POINTER(C.c_int32)),
                                stride, bw, bw, bh, color)
    return out


# ---- bi-prediction candidate cost: kvz_inter_recon_bipred + kvz_satd_any_size (oracle/ref_me_harness.c) ----
def bipred_luma_satd(pic, ref0, ref1, x, y, w, h, mv0, mv1):
    L = lib()
    L.ref_bipred_luma_satd.restype = C.c_uint
    L.ref_bipred_luma_satd.argtypes = [u8p, u8p, u8p, C.c_int, C.c_int] + [C.c_int] * 4 + [i16p, i16p, u8p]
    pic, ref0, ref1 = _u8(pic), _u8(ref0), _u8(ref1)
    a, b = np.ascontiguousarray(mv0, dtype=np.int16), np.ascontiguousarray(mv1, dtype=np.int16)
    out = np.zeros((h, w), dtype=np.uint8)
    c = L.ref_bipred_luma_satd(_p(pic, u8p), _p(ref0, u8p), _p(ref1, u8p), pic.shape[1], pic.shape[0], x, y, w, h,
                               _p(a, i16p), _p(b, i16p), _p(out, u8p))
    return c, out


# ---- deblocking: kvz_filter_deblock_lcu over every LCU (oracle/ref_harness.c: ref_deblock_frame) ----
def deblock_frame(y, u, v, cus, prm):
    L = lib()
    L.ref_deblock_frame.restype = None
    L.ref_deblock_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    y = np.array(y, dtype=np.uint8, order="C")
    u = np.array(u, dtype=np.uint8, order="C") if u is not None else None
    v = np.array(v, dtype=np.uint8, order="C") if v is not None else None
    cus = np.ascontiguousarray(cus)
    prm = np.ascontiguousarray(prm)
    L.ref_deblock_frame(y.ctypes.data, y.shape[1], u.ctypes.data if u is not None else None, v.ctypes.data if v is not None else None,
                        u.shape[1] if u is not None else 0, y.shape[1], y.shape[0], cus.ctypes.data, prm.ctypes.data)
    return y, u, v


# ---- AMVP / merge candidate derivation: the reference's kvz_inter_get_merge_cand / kvz_inter_get_mv_cand (oracle/ref_harness.c) ----
def inter_candidates(params, cus, col_cus, ref_cus, pus):
    from patterns import MERGE_CAND
    L = lib()
    L.ref_inter_candidates.restype = None
    L.ref_inter_candidates.argtypes = [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]
    cus, col_cus = np.ascontiguousarray(cus), np.ascontiguousarray(col_cus)
    ref_cus = None if ref_cus is None else np.ascontiguousarray(ref_cus)
    pus = np.ascontiguousarray(pus).copy()
    out = np.zeros((len(pus), 5), dtype=MERGE_CAND)
    L.ref_inter_candidates(cus.ctypes.data, col_cus.ctypes.data, None if ref_cus is None else ref_cus.ctypes.data,
                           np.ascontiguousarray(params).ctypes.data, pus.ctypes.data, len(pus), out.ctypes.data)
    return pus, out


def encode_with_gpu_search(frames, w, h, opts, lib_path, strategy=None, deblock=False):
    """ref_encode with the harness serving the encoder's 2Nx2N inter searches (kvz_hip_inter_candidates_batch +
    kvz_hip_search_pu_batch; oracle/ref_harness.c: gpu_search_serve) and its rough intra searches
    (kvz_hip_intra_build_reference_batch + kvz_hip_intra_rough_batch; gpu_intra_serve) through the GPU chain.
    deblock=True: the per-LCU deblocking calls are skipped as well and every picture is filtered by one kvz_hip_deblock_frame call
    before the next picture is searched (gpu_flush_deblock; SAO must be off).
    -> (bitstream, dict(inter_served, inter_passed_on, failed, launch_pairs, intra_served, intra_passed_on, bipred_pairs,
    deblocked_pictures, deblock_lcu_calls_skipped))"""
    L = lib()
    L.ref_gpu_search_begin.restype = C.c_int
    L.ref_gpu_search_begin.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.ref_gpu_search_end.restype = None
    L.ref_gpu_search_end.argtypes = [C.POINTER(C.c_long)]
    L.ref_gpu_serve_deblock.restype = C.c_int
    L.ref_gpu_serve_deblock.argtypes = [C.c_int]
    L.ref_gpu_serve_deblock_end.restype = None
    L.ref_gpu_serve_deblock_end.argtypes = [C.POINTER(C.c_long)]
    assert L.ref_gpu_search_begin(lib_path.encode(), w, h) == 0
    assert L.ref_gpu_serve_deblock(1 if deblock else 0) == 0
    out, dbk = (C.c_long * 8)(), (C.c_long * 2)()
    try:
        bitstream, _ = encode(frames, w, h, opts, strategy)
    finally:
        L.ref_gpu_serve_deblock_end(dbk)               # filters a last pending picture: before the device buffers go
        L.ref_gpu_search_end(out)
    c = dict(zip(("inter_served", "inter_passed_on", "failed", "launch_pairs", "intra_served", "intra_passed_on", "bipred_pairs",
                  "bipred_launches"), (int(v) for v in out)))
    c["deblocked_pictures"], c["deblock_lcu_calls_skipped"] = int(dbk[0]), int(dbk[1])
    return bitstream, c


def encode_with_service(frames, w, h, opts, lib_path, max_threads=64, min_size=8, shadow=False, probe=False, table_range=0, spec_probe=False, host_only=False,
                        upload_only=False):
    """ref_encode with every 2Nx2N inter search of the encoder answered by the product's search service
    (kvz_hip_me_service_search; oracle/ref_serve.c), from all of the encoder's own worker threads at once.
    min_size: PUs narrower than this run the reference's own search.  shadow: every served search is repeated by the
    reference's search and compared (the reference's result is kept).  table_range > 0: nothing of the search is served; each worker
    fetches kvz_hip_me_service_sad_tables (+-table_range) when it starts a CTU and the encoder's kvz_image_calc_sad calls are answered
    from them (table_hits / table_range_misses / table_other_calls, tables, table_bytes, table_ns in the result).
    upload_only: the pictures are uploaded as for a served run (at the first 2Nx2N search of at least min_size of a CTU), every search stays with the reference.
    host_only: no device library is opened and nothing is served (probe / spec_probe only; runs without a GPU).  spec_probe: per CU size, how
    many searches' candidates (merge list, AMVP pairs, start vectors) derived from the lcu as it was when the CTU started equal the real ones.
    -> (bitstream, dict(served, passed_on, failed, shadow_mismatch, upload_rects, search_wait_ns, upload_ns, cand_ns,
    requests, units, batches, launches, max_batch_units, rects, rect_bytes, wait_ns))"""
    L = lib()
    L.ref_service_begin.restype = C.c_int
    L.ref_service_begin.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ref_service_end.restype = None
    L.ref_service_end.argtypes = [C.POINTER(C.c_longlong)]
    assert L.ref_service_begin(lib_path.encode(), w, h, max_threads, min_size, (1 if shadow else 0) | (2 if probe or host_only else 0) | (8 if spec_probe else 0) | (16 if host_only else 0) | (32 if upload_only else 0) | (int(table_range) << 8)) == 0
    out = (C.c_longlong * 32)()
    import time as _time
    t_enc = _time.perf_counter()
    try:
        bitstream, _ = encode(frames, w, h, opts)
    finally:
        t_enc = _time.perf_counter() - t_enc           # the encode alone: a session creates its service once (device planes, page-locked areas)
        L.ref_service_end(out)
    keys = ("served", "passed_on", "failed", "shadow_mismatch", "upload_rects", "search_wait_ns", "upload_ns", "cand_ns",
            "requests", "units", "batches", "launches", "max_batch_units", "rects", "rect_bytes", "wait_ns")
    c = dict(zip(keys, (int(v) for v in out)))
    if table_range:
        c.update(table_hits=int(out[24]), table_range_misses=int(out[25]), table_other_calls=int(out[26]), tables=int(out[28]),
                 table_bytes=int(out[29]), table_ns=int(out[30]))
    if spec_probe and host_only:
        c = {}
        c["spec_searches"] = {64 >> i: int(out[8 + i]) for i in range(4)}
        c["spec_same"] = {64 >> i: int(out[12 + i]) for i in range(4)}
    c["encode_s"] = t_enc
    if probe or host_only:       # nothing served: the reference's own kvz_search_cu_inter timed per CU size
        c["probe_us_per_search"] = {64 >> i: round(out[16 + i] / 1e3 / max(1, out[20 + i]), 2) for i in range(4)}
        c["probe_searches"] = {64 >> i: int(out[20 + i]) for i in range(4)}
    return bitstream, c


def mv_cand_helpers(geoms, pic_w, pic_h):
    """the reference's file-local is_a0_cand_coded / is_b0_cand_coded / get_spatial_merge_candidates (inter.c:566-875, reached the way
    tests/mv_cand_tests.c does: oracle/ref_cand_harness.c) for PUs (x, y, w, h) -> (a0 [n], b0 [n], indices [n, 5] = b0 b1 b2 a0 a1 in lcu_t.cu)"""
    L = lib()
    geoms = np.asarray(geoms, dtype=np.int32).reshape(-1, 4)
    a0 = np.zeros(len(geoms), np.int32)
    b0 = np.zeros(len(geoms), np.int32)
    idx = np.zeros((len(geoms), 5), np.int32)
    out = (C.c_int * 5)()
    for i, (x, y, w, h) in enumerate(geoms.tolist()):
        a0[i] = L.ref_is_a0_cand_coded(x, y, w, h)
        b0[i] = L.ref_is_b0_cand_coded(x, y, w, h)
        L.ref_spatial_merge_candidate_indices(x, y, w, h, pic_w, pic_h, out)
        idx[i] = list(out)
    return a0, b0, idx
