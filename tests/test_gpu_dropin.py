"""GPU: the drop-in boundary.  libkvzhip.so registers its "hip" strategies into the
COMPILED REFERENCE's own registry (kvz_strategyselector_register, via the glue of
oracle/ref_harness.c == INTEGRATION.md), and the reference's callers / the
reference's unit-test vectors are run against them by strategy name -- the way
tests/test_strategies.c:29-52 iterates every registered implementation.
Needs the prebuilt oracle/_ref/libkvzref.so (travels with gpurun)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import ref_lib as R
from patterns import (SAD_EDGE_KAT, SATD_GOLDEN_BW, SATD_GOLDEN_GRADIENT, REG_SAD_DIMS, coeff_sum_input,
                      dct_test_input, intra_sad_gradient, rng, sad_test_frames, satd_test_bufs)

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    L = R.lib()
    L.ref_register_hip.restype = C.c_int
    L.ref_register_hip.argtypes = [C.c_char_p]
    n = L.ref_register_hip(os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so").encode())
    assert n > 0, "hip strategies failed to register"
    return n


def test_registration_and_selection(hip):
    regs = [(t, n, p) for (t, n, p) in R.strategies() if n == "hip"]
    types = {t for (t, _, _) in regs}
    # every picture cost function, every transform, the quant group and the sample filters
    for t in ["reg_sad", "satd_any_size", "satd_any_size_quad", "pixels_calc_ssd", "coeff_abs_sum", "quant", "dequant",
              "quantize_residual", "sample_quarterpel_luma", "sample_octpel_chroma", "sample_14bit_quarterpel_luma",
              "sample_14bit_octpel_chroma", "fast_forward_dst_4x4", "fast_inverse_dst_4x4", "inter_recon_bipred",
              "filter_hpel_blocks_hor_ver_luma", "filter_hpel_blocks_diag_luma", "filter_qpel_blocks_hor_ver_luma",
              "filter_qpel_blocks_diag_luma"] + \
             ["%s_%dx%d" % (k, n, n) for k in ("sad", "satd", "dct", "idct") for n in (4, 8, 16, 32)] + \
             ["sad_64x64", "satd_64x64"] + ["%s_%dx%d_dual" % (k, n, n) for k in ("sad", "satd") for n in (4, 8, 16, 32, 64)]:
        assert t in types, t
    assert all(p == 50 for (_, _, p) in regs)
    assert hip == len(regs)
    # the selector's rule (highest priority wins, strategyselector.c:258-302) now picks hip over avx2 (40)
    L = R.lib()
    for t in ("sad_8x8", "satd_8x8", "dct_32x32", "reg_sad"):
        assert L.ref_strategy(t.encode(), b"best") == L.ref_strategy(t.encode(), b"hip")


@pytest.mark.parametrize("log_w", [2, 3, 4, 5, 6])
def test_satd_and_sad_reference_unit_vectors(hip, log_w):
    n = 1 << log_w
    bw, ck, gr = satd_test_bufs(log_w)
    for (x, y), want in ((bw, SATD_GOLDEN_BW[log_w]), (ck, SATD_GOLDEN_BW[log_w]), (gr, SATD_GOLDEN_GRADIENT[log_w])):
        assert R.cost_nxn_batch("satd", n, x[None], y[None], "hip")[0] == want
        assert R.cost_nxn_batch("satd", n, y[None], x[None], "hip")[0] == want
    z, m = np.zeros(n * n, np.uint8), np.full(n * n, 255, np.uint8)
    assert R.cost_nxn_batch("sad", n, z[None], m[None], "hip")[0] == 255 * n * n
    ga, gb = intra_sad_gradient(n)
    assert R.cost_nxn_batch("sad", n, ga[None], gb[None], "hip")[0] == int(np.abs(ga.astype(int) - gb.astype(int)).sum())


def test_image_calc_sad_through_reference_caller(hip):
    """kvz_image_calc_sad (image.c:455) with kvz_reg_sad = hip: tests/sad_tests.c:121-259 closed forms"""
    pic, ref, big_pic, big_ref = sad_test_frames()
    for (x, y), want in SAD_EDGE_KAT.items():
        assert R.image_calc("sad", pic, ref, 0, 0, x, y, 8, 8, "hip") == want, (x, y)
    for (w, h) in REG_SAD_DIMS:
        want = int(np.abs(big_pic[:h, :w].astype(int) - big_ref[:h, :w].astype(int)).sum())
        assert R.reg_sad(big_pic, big_ref, 0, 0, w, h, 64, 64, "hip") == want


def test_picture_group_vs_generic(hip):
    g = rng(5)
    for n in (4, 8, 16, 32):
        orig = g.integers(0, 256, (3, n * n), dtype=np.uint8)
        preds = g.integers(0, 256, (3, 2048), dtype=np.uint8)
        for kind in ("sad", "satd"):
            np.testing.assert_array_equal(R.cost_nxn_dual_batch(kind, n, preds, orig, "hip"),
                                          R.cost_nxn_dual_batch(kind, n, preds, orig, "generic"))
    a = g.integers(0, 256, 80 * 80, dtype=np.uint8)
    b = g.integers(0, 256, 100 * 80, dtype=np.uint8)
    for (w, h) in ((8, 8), (16, 16), (12, 8), (8, 12), (4, 4), (64, 64), (24, 16)):
        assert R.satd_any_size(w, h, a, 3, 80, b, 7, 100, "hip") == R.satd_any_size(w, h, a, 3, 80, b, 7, 100, "generic")
    preds = [g.integers(0, 256, 64 * 72, dtype=np.uint8) for _ in range(4)]
    for (w, h) in ((8, 8), (16, 16), (64, 64), (12, 16), (16, 12), (4, 8)):
        np.testing.assert_array_equal(R.satd_any_size_quad(w, h, preds, 64, b, 11, 100, "hip"),
                                      R.satd_any_size_quad(w, h, preds, 64, b, 11, 100, "generic"))
    for w in (4, 8, 16, 32, 64):
        assert R.pixels_calc_ssd(a, 2, b, 5, 80, 100, w, "hip") == R.pixels_calc_ssd(a, 2, b, 5, 80, 100, w, "generic")
    assert R.image_calc("satd", a.reshape(80, 80), b.reshape(80, 100), 8, 8, -3, 70, 16, 16, "hip") == \
        O.image_calc("satd", a.reshape(80, 80), b.reshape(80, 100), 8, 8, -3, 70, 16, 16)


def test_dct_group_reference_unit_vectors(hip):
    """tests/dct_tests.c: every registered dct/idct implementation must equal generic on the gradient input"""
    src = dct_test_input()
    g = rng(9)
    for n in (4, 8, 16, 32):
        x = np.concatenate([src[:n * n][None], g.integers(-32768, 32768, (2, n * n)).astype(np.int16)])
        for kind in ("dct", "idct") + (("dst", "idst") if n == 4 else ()):
            np.testing.assert_array_equal(R.transform_batch(kind, n, x, "hip"), R.transform_batch(kind, n, x, "generic"))


def test_quant_group_through_encoder_state(hip):
    """quant / dequant / quantize_residual receive the reference's encoder_state_t and read it through the
    accessor glue; flat and default scaling lists; coeff_abs_sum KAT (tests/coeff_sum_tests.c)"""
    c, expected = coeff_sum_input()
    assert R.coeff_abs_sum(c, "hip") == expected
    g = rng(31)
    for w in (4, 8, 16, 32):
        coef = g.integers(-2500, 2501, (3, w * w)).astype(np.int16)
        for sl in (0, 1):
            for sh in (0, 1):
                q_h = R.quant_batch(coef, w, 27, 0, 0, 1, sh, 1, sl, "hip")
                q_g = R.quant_batch(coef, w, 27, 0, 0, 1, sh, 1, sl, "generic")
                np.testing.assert_array_equal(q_h, q_g, err_msg="quant w=%d sl=%d sh=%d" % (w, sl, sh))
            np.testing.assert_array_equal(R.dequant_batch(q_g, w, 27, 0, 1, sl, "hip"), R.dequant_batch(q_g, w, 27, 0, 1, sl, "generic"))
        ref_in = g.integers(0, 256, (3, w * w), dtype=np.uint8)
        pred = np.clip(ref_in.astype(int) + g.integers(-30, 31, ref_in.shape), 0, 255).astype(np.uint8)
        for intra in (0, 1):
            h = R.quantize_residual_batch(ref_in, pred, w, 24, 0, 0, intra, intra, 0, 0, "hip")
            r = R.quantize_residual_batch(ref_in, pred, w, 24, 0, 0, intra, intra, 0, 0, "generic")
            for a, b in zip(h, r):
                np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("kind", ["luma", "luma14", "chroma", "chroma14"])
def test_ipol_sample_filters(hip, kind):
    g = rng(60)
    frame = g.integers(0, 256, (96, 96), dtype=np.uint8)
    luma = kind.startswith("luma")
    for (w, h) in (((8, 8), (16, 16), (64, 64)) if luma else ((4, 4), (8, 8), (32, 32))):
        for (fx, fy) in ((0, 0), (1, 2), (3, 3)) if luma else ((0, 0), (3, 5), (7, 1)):
            np.testing.assert_array_equal(R.sample(kind, frame, 12, 10, w, h, fx, fy, "hip"),
                                          R.sample(kind, frame, 12, 10, w, h, fx, fy, "generic"))


def test_concurrent_calls_from_worker_threads(hip):
    """the strategy pointers are called concurrently from all threadqueue workers (encoderstate.c:781):
    per-thread streams + staging, no shared state"""
    import threading
    g = rng(77)
    a = g.integers(0, 256, (64, 64), dtype=np.uint8)
    b = g.integers(0, 256, (64, 64), dtype=np.uint8)
    want = O.cost_nxn_batch("satd", 8, a, b)
    errs = []

    def work():
        try:
            for _ in range(3):
                got = R.cost_nxn_batch("satd", 8, a, b, "hip")
                if not (got == want).all():
                    errs.append("mismatch")
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))
    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs


@pytest.mark.parametrize("pattern", ["random", "extreme"])
def test_frac_filter_steps_write_the_callers_scratch_like_generic(hip, pattern):
    """the four ipol_blocks_func steps: filtered blocks AND the state the next steps read; steps may be
    mixed between strategies, so run hip steps and generic steps on the same arrays' contents"""
    g = rng(70)
    frame = g.integers(0, 256, (96, 96), dtype=np.uint8)
    if pattern == "extreme":
        frame = np.where(g.integers(0, 2, (96, 96)) > 0, 255, 0).astype(np.uint8)
    for (w, h) in ((8, 8), (16, 16), (64, 64), (16, 8), (8, 4), (4, 8), (16, 12), (12, 16)):       # incl. SMP / AMP shapes
        for off in ((0, 0), (-1, 1), (1, -1), (1, 1), (-1, -1), (0, 1)):
            hipr = R.filter_frac_steps(frame, 10, 9, w, h, off, 4, "hip")
            gen = R.filter_frac_steps(frame, 10, 9, w, h, off, 4, "generic")
            np.testing.assert_array_equal(hipr[:, :, :h, :w], gen[:, :, :h, :w], err_msg="%dx%d off=%s" % (w, h, off))


def test_inter_recon_bipred(hip):
    g = rng(13)
    for (w, h, x, y) in ((16, 16, 0, 0), (8, 8, 24, 40), (64, 64, 0, 0), (32, 16, 96, 72)):
        for hi in ((1, 1, 1, 1), (0, 0, 0, 0), (1, 0, 0, 1), (0, 1, 1, 0)):
            hp0 = [g.integers(-2000, 18000, n).astype(np.int16) for n in (4096, 1024, 1024)]
            hp1 = [g.integers(-2000, 18000, n).astype(np.int16) for n in (4096, 1024, 1024)]
            rec = [g.integers(0, 256, n, dtype=np.uint8) for n in (4096, 1024, 1024)]
            tmp = [g.integers(0, 256, n, dtype=np.uint8) for n in (4096, 1024, 1024)]
            a = R.bipred(hi, h, w, y, x, hp0, hp1, rec, tmp, "hip")
            b = R.bipred(hi, h, w, y, x, hp0, hp1, rec, tmp, "generic")
            for p, q in zip(a, b):
                np.testing.assert_array_equal(p, q)


ENCODER_CONFIGS = [
    # (w, h, frames, options).  With RDOQ on (presets medium and slower) the hip quantize_residual runs residual,
    # transforms, dequantisation and reconstruction on the GPU around the encoder's own kvz_rdoq, like the avx2 strategy.
    (128, 128, 4, "preset=medium,qp=30,threads=2"),
    (128, 64, 4, "preset=slow,qp=24,threads=0"),
    (128, 64, 3, "preset=medium,rdoq-skip=1,qp=36,signhide=1,threads=0"),
    (128, 64, 4, "preset=ultrafast,qp=27,threads=0"),
    (128, 64, 4, "preset=medium,rdoq=0,qp=32,threads=0"),
    (64, 64, 5, "preset=fast,rdoq=0,signhide=1,bipred=1,gop=8,qp=22,threads=0"),
    (128, 128, 3, "preset=veryfast,rdoq=0,qp=37,threads=4,owf=2,wpp=1"),
    (256, 128, 6, "preset=slow,rdoq=0,signhide=1,qp=27,threads=4"),
    (192, 128, 4, "preset=medium,rdoq=0,me=tz,subme=4,smp=1,amp=1,bipred=1,gop=8,qp=24,threads=2"),
    (128, 128, 4, "preset=medium,rdoq=0,rd=2,qp=30,threads=2"),                       # full RD: SSD / coefficient costs everywhere
    (128, 64, 4, "preset=fast,rdoq=0,transform-skip=1,rd=1,qp=26,threads=0"),          # 4x4 transform skip + its SAD test
    # transform skip with RDOQ quantising the skipped 4x4 blocks (quant-generic.c:206-221: kvz_transformskip, then kvz_rdoq)
    (128, 64, 4, "preset=medium,transform-skip=1,rdoq=1,rdoq-skip=0,rd=1,qp=26,threads=0"),
    (128, 128, 3, "preset=medium,rdoq=0,scaling-list=default,qp=28,threads=2"),        # scaling-list tables through the accessors
    (128, 64, 4, "preset=fast,rdoq=0,me=dia,full-intra-search=1,mv-rdo=1,qp=35,threads=0"),
    (64, 64, 3, "preset=medium,rdoq=0,lossless=1,threads=0"),
    (192, 128, 3, "preset=medium,rdoq=0,tiles=2x2,qp=31,threads=3"),
]


@pytest.mark.parametrize("w,h,n,opts", ENCODER_CONFIGS)
def test_reference_encoder_bitstream_is_identical_with_hip_strategies(hip, w, h, n, opts):
    """End to end: the reference encoder itself (kvz_api, search + transform + quant + inter recon all calling through
    the strategy table) produces the same HEVC bitstream byte for byte whether its table holds the generic C
    strategies or every "hip" strategy."""
    frames = R.synthetic_sequence(w, h, n)
    hl = C.CDLL(os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so"))
    hl.kvz_hip_dropin_calls.restype = C.c_ulonglong
    gen, n_gen = R.encode(frames, w, h, opts, "generic")
    before = hl.kvz_hip_dropin_calls()
    hipb, n_hip = R.encode(frames, w, h, opts, "hip")
    calls = hl.kvz_hip_dropin_calls() - before
    assert n_hip >= 30, "only %d hip strategies installed" % n_hip
    assert calls > 1000, "the encoder made only %d calls into the hip strategies" % calls
    print("hip strategy calls:", calls)
    assert len(gen) > 200
    assert hipb == gen, "bitstreams differ (%d vs %d bytes)" % (len(hipb), len(gen))


def test_intra_group_through_registry_and_kvz_intra_predict(hip):
    """angular_pred / intra_pred_planar registered as "hip": called by name like tests/test_strategies.c does, and
    installed under the reference's own kvz_intra_predict (intra.c:281)"""
    from patterns import intra_ref_cases
    assert R.has_strategy("angular_pred", "hip") and R.has_strategy("intra_pred_planar", "hip")
    for log2_width in (2, 3, 4, 5):
        refs = intra_ref_cases(log2_width, 5, 800 + log2_width)
        for r in refs:
            left, top = r[:65], r[65:]
            for mode in (2, 9, 10, 11, 17, 18, 25, 26, 27, 34):
                np.testing.assert_array_equal(R.angular_pred(log2_width, mode, top, left, "hip"),
                                              R.angular_pred(log2_width, mode, top, left, "generic"))
            np.testing.assert_array_equal(R.intra_pred_planar(log2_width, top, left, "hip"),
                                          R.intra_pred_planar(log2_width, top, left, "generic"))
            for mode in (0, 1, 2, 10, 18, 26, 30):
                for color, fb in ((0, 1), (0, 0), (1, 1)):
                    np.testing.assert_array_equal(R.intra_predict(r, log2_width, mode, color, fb, "hip"),
                                                  R.intra_predict(r, log2_width, mode, color, fb, "generic"))


def test_sao_group_through_registry(hip):
    """the four SAO strategies registered as "hip", called by name like tests/test_strategies.c does"""
    from patterns import sao_blocks, sao_records
    for t in ("sao_edge_ddistortion", "calc_sao_edge_dir", "sao_reconstruct_color", "sao_band_ddistortion"):
        assert R.has_strategy(t, "hip")
    g = rng(3)
    for (bw, bh) in ((64, 64), (32, 32), (64, 40), (8, 16)):
        orig, rec = sao_blocks(bw, bh, 4, bw + bh)
        for i in range(4):
            for eo in range(4):
                np.testing.assert_array_equal(R.calc_sao_edge_dir(orig[i], rec[i], eo, bw, bh, "hip"), R.calc_sao_edge_dir(orig[i], rec[i], eo, bw, bh))
                offs = g.integers(-7, 8, 5)
                assert R.sao_edge_ddistortion(orig[i], rec[i], bw, bh, eo, offs, "hip") == R.sao_edge_ddistortion(orig[i], rec[i], bw, bh, eo, offs)
            bands, bp = g.integers(-7, 8, 4), int(g.integers(0, 32))
            assert R.sao_band_ddistortion(orig[i], rec[i], bw, bh, bp, bands, "hip") == R.sao_band_ddistortion(orig[i], rec[i], bw, bh, bp, bands)
    plane = g.integers(0, 256, (80, 96), dtype=np.uint8)
    for color in (0, 2):
        for s in sao_records(8, 4 + color):
            for (x, y, bw, bh) in ((1, 1, 64, 64), (5, 3, 32, 32), (1, 7, 61, 13)):
                np.testing.assert_array_equal(R.sao_reconstruct_color(plane, x, y, bw, bh, s, color, "hip"),
                                              R.sao_reconstruct_color(plane, x, y, bw, bh, s, color))


def test_device_staging_mode_in_a_child_process():
    """KVZ_HIP_ZEROCOPY=0 (explicit copies into device memory instead of kernels operating on the pinned staging buffer)
    is read when a thread's context is created, so it gets its own process"""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "import ref_lib as R\n"
        "L = R.lib(); L.ref_register_hip.restype = C.c_int; L.ref_register_hip.argtypes = [C.c_char_p]\n"
        "assert L.ref_register_hip(%r.encode()) > 0\n"
        "g = np.random.default_rng(4)\n"
        "a = g.integers(0, 256, (6, 1024), dtype=np.uint8); b = g.integers(0, 256, (6, 1024), dtype=np.uint8)\n"
        "assert (R.cost_nxn_batch('satd', 32, a, b, 'hip') == R.cost_nxn_batch('satd', 32, a, b, 'generic')).all()\n"
        "x = g.integers(-255, 256, (5, 256)).astype(np.int16)\n"
        "assert (R.transform_batch('dct', 16, x, 'hip') == R.transform_batch('dct', 16, x, 'generic')).all()\n"
        "f = g.integers(0, 256, (64, 64), dtype=np.uint8)\n"
        "assert (R.sample('luma', f, 20, 20, 16, 16, 1, 3, 'hip') == R.sample('luma', f, 20, 20, 16, 16, 1, 3, 'generic')).all()\n"
        "print('child ok')\n"
    ) % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so"))
    env = dict(os.environ, KVZ_HIP_ZEROCOPY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "child ok" in out.stdout, out.stderr[-2000:]


def test_quant_group_registers_only_what_its_accessors_support():
    """kvz_strategy_register_quant_hip with partial accessor tables: a function whose state accessors are missing is not
    registered (the host keeps its next-best strategy) instead of aborting or diverging at run time.  Own process: the
    accessor table and the registrar are process-global."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from kvazaar_amd import _lib\n"
        "L = _lib.init(0)\n"
        "seen = []\n"
        "REG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p)\n"
        "cb = REG(lambda o, t, n, p, f: (seen.append(t.decode()), 1)[1])\n"
        "L.kvz_hip_set_registrar(C.cast(cb, C.c_void_p))\n"
        "F = C.CFUNCTYPE(C.c_int, C.c_void_p)\n"
        "one = F(lambda s: 1)\n"
        "ptr = lambda f: C.cast(f, C.c_void_p).value\n"
        "def run(fields):\n"
        "    tab = (C.c_void_p * 18)()\n"
        "    for i in fields: tab[i] = ptr(one)\n"
        "    L.kvz_hip_set_state_accessors(C.cast(tab, C.c_void_p))\n"
        "    del seen[:]\n"
        "    assert L.kvz_strategy_register_quant_hip(None, 8) == 1\n"
        "    return sorted(seen)\n"
        "base = [0, 1, 2, 7]                      # qp, slice_is_intra, signhide_enable, cu_is_intra\n"
        "assert run([]) == ['coeff_abs_sum']\n"
        "assert run(base) == ['coeff_abs_sum', 'dequant', 'quant', 'quantize_residual']\n"
        "assert run(base + [3]) == ['coeff_abs_sum']                        # scaling_list_enable without the two tables\n"
        "assert run(base + [3, 4, 5]) == ['coeff_abs_sum', 'dequant', 'quant', 'quantize_residual']\n"
        "assert run(base + [6]) == ['coeff_abs_sum', 'dequant', 'quant']    # rdoq_enable without kvz_rdoq and its accessors\n"
        "assert run(base + [6, 14, 15, 16, 17]) == ['coeff_abs_sum', 'dequant', 'quant', 'quantize_residual']\n"
        "print('child ok')\n"
    ) % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "child ok" in out.stdout, out.stderr[-2000:]


def test_get_extended_block_through_registry(hip):
    """the ninth ipol type (strategies-ipol.h:74): windows inside the plane come back as pointers into it, like generic's;
    windows that leave it are edge replicated (on the GPU) into a malloc'ed buffer the caller frees"""
    assert R.has_strategy("get_extended_block", "hip")
    g = rng(77)
    frame = g.integers(0, 256, (72, 96), dtype=np.uint8)
    cases = []
    for fs in (8, 4):                                    # KVZ_LUMA_FILTER_TAPS, KVZ_CHROMA_FILTER_TAPS
        for (w, h) in ((9, 9), (17, 17), (65, 65), (8, 4), (4, 16)):
            for (x, y, mx, my) in ((20, 20, 0, 0), (0, 0, -1, -1), (90, 60, 3, 5), (40, -30, 0, 0), (-70, 30, 2, 2), (95, 71, 40, 40),
                                   (10, 10, -200, 3), (30, 200, 0, 0), (5, 62, 1, 1), (0, 30, 4, 0)):
                cases.append((x, y, mx, my, fs, w, h))
    n_malloc = 0
    for (x, y, mx, my, fs, w, h) in cases:
        a, ua, ia = R.get_extended_block(frame, x, y, mx, my, fs, w, h, name="hip")
        b, ub, ib = R.get_extended_block(frame, x, y, mx, my, fs, w, h, name="generic")
        assert ua == ub and ia == ib, (x, y, mx, my, fs, w, h, ua, ub, ia, ib)
        np.testing.assert_array_equal(a, b, err_msg=str((x, y, mx, my, fs, w, h)))
        n_malloc += ua
    assert 0 < n_malloc < len(cases)
    a, _, _ = R.get_extended_block(frame, 3, 4, 0, 0, 8, 9, 9, off_x=16, off_y=8, name="hip")      # tile offsets (search_inter.c:1007-1012)
    b, _, _ = R.get_extended_block(frame, 3, 4, 0, 0, 8, 9, 9, off_x=16, off_y=8, name="generic")
    np.testing.assert_array_equal(a, b)


GPU_SEARCH_CONFIGS = [
    # single-reference P frames; everything else about the encode is the preset's (RDOQ, SAO, deblocking, TMVP, WPP)
    (192, 128, 4, "preset=medium,ref=1,bipred=0,gop=0,qp=30,threads=0,period=0"),
    (192, 128, 4, "preset=medium,ref=1,bipred=0,gop=0,rdoq=0,qp=24,threads=0,me=tz,me-early-termination=sensitive,mv-constraint=frametilemargin,period=0"),
    (128, 128, 5, "preset=fast,ref=1,bipred=0,gop=0,rdoq=0,qp=37,threads=0,me=dia,subme=2,deblock=1,sao=off,owf=0,wpp=0,period=0"),
    (168, 104, 4, "preset=veryfast,ref=1,bipred=0,gop=0,qp=33,threads=0,tmvp=0,period=0"),       # ragged LCUs, no temporal candidates
    (128, 64, 6, "preset=slow,ref=1,bipred=0,gop=0,rdoq=0,qp=27,threads=0,smp=1,amp=1,period=0"),   # + every SMP / AMP PU (kvz_search_cu_smp)
    (192, 128, 6, "preset=medium,smp=1,amp=1,qp=31,threads=0"),                                      # SMP / AMP in B slices with four references
    # several reference pictures: every picture searched in turn, the best cost so far as the cost to beat (search_inter.c:1239)
    (192, 128, 7, "preset=medium,ref=3,bipred=0,gop=0,qp=29,threads=0,period=0"),
    # preset medium as it is: B slices in a GOP of 8, four reference pictures in two lists, uni-prediction
    (192, 128, 10, "preset=medium,qp=30,threads=0"),
    (128, 128, 9, "preset=fast,gop=lp-g4d3t1,qp=34,threads=0,rdoq=0"),                               # low-delay P GOP
    (1920, 1080, 3, "preset=medium,qp=32,threads=0"),                                                # BASELINE's 1080p medium, two B pictures
    # BASELINE's largest picture: one 4K P frame (intra PUs of 8x8 and 16x16 only, to keep the test short)
    (3840, 2160, 2, "preset=medium,ref=1,bipred=0,gop=0,qp=34,threads=0,period=0,pu-depth-intra=2-3"),
    # bi-prediction: pairs of merge candidates scored by kvz_hip_bipred_cost_batch (search_pu_inter_bipred)
    (192, 128, 10, "preset=medium,bipred=1,qp=30,threads=0"),
    (128, 128, 9, "preset=slow,qp=26,threads=0"),
    # tiles: the candidate derivation and the search work in the tile's picture, descriptors and reference planes in the whole one
    (192, 128, 5, "preset=medium,tiles=2x2,qp=31,threads=0"),
    # CTU-row tiles with vectors kept inside the tile: the partition bench.py's shard leg cuts a frame by, in a real encode
    (256, 256, 5, "preset=medium,ref=2,bipred=0,gop=0,tiles=1x2,mv-constraint=frametilemargin,qp=29,threads=0,period=0"),
    (1920, 1080, 2, "preset=medium,ref=1,bipred=0,gop=0,tiles=1x4,mv-constraint=frametilemargin,qp=33,threads=0,period=0,pu-depth-intra=2-3"),
    # rd 2: the searches are followed by the reference's own full-reconstruction refinements (kvz_cu_cost_inter_rd2, search_intra_rdo)
    (128, 128, 6, "preset=medium,rd=2,qp=28,threads=0"),
    (128, 64, 9, "preset=slower,qp=31,threads=0"),
    # intra pictures only; 4x4 transform skip (its SAD test in get_cost); every mode in the first pass
    (192, 128, 3, "preset=medium,period=1,qp=27,threads=0"),
    (128, 128, 3, "preset=fast,period=1,transform-skip=1,rd=1,qp=24,threads=0"),
    (168, 104, 3, "preset=medium,period=1,full-intra-search=1,qp=35,threads=0"),
]


@pytest.mark.parametrize("w,h,n,opts", GPU_SEARCH_CONFIGS)
def test_reference_encoder_with_its_searches_served_by_the_gpu_chain(hip, w, h, n, opts):
    """The batched entries inside a live encode.  Every inter search of the reference encoder (2Nx2N, and the PUs of the SMP / AMP partitions) is answered by
    kvz_hip_inter_candidates_batch (candidates from the encoder's lcu->cu, copied into a device CU array) followed by
    kvz_hip_search_pu_batch, once per reference picture; every rough intra search by kvz_hip_intra_build_reference_batch (from
    lcu->rec and its borders laid out as a picture) followed by kvz_hip_intra_rough_batch, the harness walking the 35-cost table in
    search_intra_rough's order.  The encoder carries on with those decisions -- mode decision, reconstruction, the neighbours'
    candidates, the next pictures' temporal candidates all consume them.  The bitstream must be the untouched encoder's."""
    import time
    frames = R.synthetic_sequence(w, h, n, seed=5)
    t0 = time.perf_counter()
    plain, _ = R.encode(frames, w, h, opts)
    t1 = time.perf_counter()
    served_bs, c = R.encode_with_gpu_search(frames, w, h, opts, os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so"))
    t2 = time.perf_counter()
    # one search per launch and three host round trips each: a correctness path, its time is printed for the record only
    print("%dx%d x %d frames: inter searches served by the GPU chain: %d (%d candidate + search launch pairs), left to the reference: %d; "
          "bi-prediction pairs scored: %d in %d calls; intra searches served: %d, left to the reference: %d; whole encode %.2f s untouched, %.2f s served"
          % (w, h, n, c["inter_served"], c["launch_pairs"], c["inter_passed_on"], c["bipred_pairs"], c["bipred_launches"], c["intra_served"], c["intra_passed_on"],
             t1 - t0, t2 - t1))
    assert c["bipred_pairs"] > 100 or "bipred=1" not in opts
    if (w, h) == (1920, 1080) and "ref=1" not in opts:
        # for the record beside it: the untouched encoder with its thread pool on this host (BASELINE's "encoder fps 1080p medium")
        t3 = time.perf_counter()
        R.encode(frames, w, h, opts.replace("threads=0", "threads=16"))
        print("%dx%d x %d frames, untouched encoder with threads=16: %.2f s (%.1f frames/s)" % (w, h, n, time.perf_counter() - t3, n / (time.perf_counter() - t3)))
    assert c["failed"] == 0 and c["inter_served"] + c["intra_served"] >= 40 * (n - 1) and c["intra_served"] > 0
    assert served_bs == plain, "bitstreams differ (%d vs %d bytes)" % (len(served_bs), len(plain))


def test_reference_encoder_with_searches_served_and_hip_strategies_installed(hip):
    """both halves of the boundary at once: the per-call "hip" strategies in the encoder's function table (SAD / SATD, transforms,
    quantisation, interpolation, intra prediction, SAO) AND the inter / intra searches served by the batched entries -- the
    encoder then does none of SURVEY 8(a)'s arithmetic on the host, and its bitstream is still the generic C encoder's"""
    w, h, n, opts = 192, 128, 5, "preset=medium,qp=29,threads=0"
    frames = R.synthetic_sequence(w, h, n, seed=9)
    gen, _ = R.encode(frames, w, h, opts, "generic")
    both, c = R.encode_with_gpu_search(frames, w, h, opts, os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so"), strategy="hip")
    assert c["failed"] == 0 and c["inter_served"] > 500 and c["intra_served"] > 500
    assert both == gen, "bitstreams differ (%d vs %d bytes)" % (len(both), len(gen))


@pytest.mark.parametrize("w,h,n,opts", [
    (192, 128, 6, "preset=medium,sao=off,deblock=1,qp=34,threads=0"),                                   # B slices, four references
    (168, 104, 5, "preset=fast,ref=2,bipred=0,gop=0,sao=off,deblock=-2:3,qp=38,threads=0,period=0"),     # ragged LCUs, filter offsets
    (1920, 1080, 3, "preset=ultrafast,ref=1,gop=0,sao=off,deblock=1,qp=36,threads=0,period=0"),
])
def test_reference_encoder_with_pictures_deblocked_by_one_gpu_call(hip, w, h, n, opts):
    """the in-loop filter as a whole-picture entry inside a live encode: the encoder's per-LCU kvz_filter_deblock_lcu calls are
    skipped and each picture is filtered by ONE kvz_hip_deblock_frame call (from the encoder's own CU array: TU / PU edges, cbf,
    QPs, vectors) before the next picture is searched; every later picture predicts from those pixels.  Searches served as well."""
    frames = R.synthetic_sequence(w, h, n, seed=13)
    plain, _ = R.encode(frames, w, h, opts)
    served_bs, c = R.encode_with_gpu_search(frames, w, h, opts, os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so"), deblock=True)
    print("%dx%d x %d frames: %d pictures deblocked by one GPU call each (%d per-LCU calls skipped), inter / intra searches served: %d / %d"
          % (w, h, n, c["deblocked_pictures"], c["deblock_lcu_calls_skipped"], c["inter_served"], c["intra_served"]))
    assert c["failed"] == 0 and c["deblocked_pictures"] == n and c["deblock_lcu_calls_skipped"] == n * ((w + 63) // 64) * ((h + 63) // 64)
    assert served_bs == plain, "bitstreams differ (%d vs %d bytes)" % (len(served_bs), len(plain))
