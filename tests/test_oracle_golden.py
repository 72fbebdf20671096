"""CPU: the oracle against (1) the known-answer values the reference's own unit
tests hard-code and (2) the committed golden fixtures generated from the compiled
reference's generic strategy (oracle/gen_golden.py).  Runs anywhere -- does not
need /root/reference or oracle/_ref."""
import os

import numpy as np
import pytest

import oracle_lib as O
from patterns import (SAD_EDGE_KAT, SATD_GOLDEN_BW, SATD_GOLDEN_GRADIENT, REG_SAD_DIMS, coeff_sum_input,
                      intra_sad_gradient, lcg_bytes, sad_test_frames, satd_test_bufs)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.mark.parametrize("log_w", [2, 3, 4, 5, 6])
def test_satd_known_answers(log_w):
    """tests/satd_tests.c:109,127,146"""
    n = 1 << log_w
    bw, ck, gr = satd_test_bufs(log_w)
    for (x, y), want in ((bw, SATD_GOLDEN_BW[log_w]), (ck, SATD_GOLDEN_BW[log_w]), (gr, SATD_GOLDEN_GRADIENT[log_w])):
        assert O.cost_nxn_batch("satd", n, x[None], y[None])[0] == want
        assert O.cost_nxn_batch("satd", n, y[None], x[None])[0] == want


@pytest.mark.parametrize("log_w", [2, 3, 4, 5, 6])
def test_intra_sad_patterns(log_w):
    """tests/intra_sad_tests.c:124-167"""
    n = 1 << log_w
    z, m = np.zeros(n * n, np.uint8), np.full(n * n, 255, np.uint8)
    assert O.cost_nxn_batch("sad", n, z[None], m[None])[0] == 255 * n * n
    ga, gb = intra_sad_gradient(n)
    want = int(np.abs(ga.astype(np.int64) - gb.astype(np.int64)).sum())
    assert O.cost_nxn_batch("sad", n, ga[None], gb[None])[0] == want == O.cost_nxn_batch("sad", n, gb[None], ga[None])[0]


def test_image_calc_sad_known_answers():
    """tests/sad_tests.c:121-259"""
    pic, ref, _, _ = sad_test_frames()
    for (x, y), want in SAD_EDGE_KAT.items():
        assert O.image_calc("sad", pic, ref, 0, 0, x, y, 8, 8) == want, (x, y)


def test_reg_sad_shapes_and_overflow():
    """tests/sad_tests.c:261-320,369-376"""
    _, _, big_pic, big_ref = sad_test_frames()
    z, m = np.zeros((64, 64), np.uint8), np.full((64, 64), 255, np.uint8)
    for (w, h) in REG_SAD_DIMS:
        want = int(np.abs(big_pic[:h, :w].astype(np.int64) - big_ref[:h, :w].astype(np.int64)).sum())
        assert O.reg_sad(big_pic, big_ref, 0, 0, w, h, 64, 64) == want
        assert O.reg_sad(z, m, 0, 0, w, h, 64, 64) == 255 * w * h


def test_coeff_abs_sum_known_answer():
    c, expected = coeff_sum_input()
    assert O.coeff_abs_sum(c) == expected


def test_lcg_bytes_is_stable():
    assert lcg_bytes(8).tolist() == [(((1664525 * 12345 + 1013904223) & 0xFFFFFFFF) >> 8) & 0xFF] + lcg_bytes(8).tolist()[1:]


def test_scan_tables_match_structure():
    for scan in (0, 1, 2):
        for log2 in (2, 3, 4, 5):
            s = O.scan_order(scan, log2)
            assert sorted(s.tolist()) == list(range(1 << (2 * log2)))      # a permutation
    assert O.scan_order(0, 2).tolist() == [0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15]
    assert O.scan_order(0, 3).tolist()[:20] == [0, 8, 1, 16, 9, 2, 24, 17, 10, 3, 25, 18, 11, 26, 19, 27, 32, 40, 33, 48]
    assert O.scan_order(2, 3).tolist()[:8] == [0, 8, 16, 24, 1, 9, 17, 25]


# ---------------------------------------------------------------- golden fixtures
def test_golden_picture():
    d = gold("picture.npz")
    for n in (4, 8, 16, 32, 64):
        a, b = d["a%d" % n], d["b%d" % n]
        np.testing.assert_array_equal(O.cost_nxn_batch("sad", n, a, b), d["sad%d" % n])
        np.testing.assert_array_equal(O.cost_nxn_batch("satd", n, a, b), d["satd%d" % n])
        if n <= 32:
            np.testing.assert_array_equal(O.cost_nxn_dual_batch("sad", n, d["dual_preds%d" % n], a[:8]), d["sad_dual%d" % n])
            np.testing.assert_array_equal(O.cost_nxn_dual_batch("satd", n, d["dual_preds%d" % n], a[:8]), d["satd_dual%d" % n])
    pic, ref = d["frame_pic"], d["frame_ref"]
    for p, s, t in zip(d["pairs"], d["image_sad"], d["image_satd"]):
        assert O.image_calc("sad", pic, ref, *[int(v) for v in p]) == s
        assert O.image_calc("satd", pic, ref, *[int(v) for v in p]) == t
    for (w, h), c in zip(d["quad_dims"], d["quad_costs"]):
        np.testing.assert_array_equal(O.satd_any_size_quad(int(w), int(h), list(d["quad_preds"]), 64, pic, 0, 64), c)
    for w, v in zip((4, 8, 16, 32), d["ssd"]):
        assert O.pixels_calc_ssd(pic, 0, ref, 0, 64, 64, w) == v


def test_golden_dct():
    d = gold("dct.npz")
    for n in (4, 8, 16, 32):
        for kind in ("dct", "idct") + (("dst", "idst") if n == 4 else ()):
            np.testing.assert_array_equal(O.transform_batch(kind, n, d["in%d" % n]), d["%s%d" % (kind, n)])


def test_golden_quant():
    d = gold("quant.npz")
    for w in (4, 8, 16, 32):
        coef = d["coef%d" % w]
        for qp in (22, 37):
            for sh in (0, 1):
                np.testing.assert_array_equal(O.quant_batch(coef, w, qp, 0, 0, 1, sh), d["quant%d_qp%d_sh%d" % (w, qp, sh)])
            np.testing.assert_array_equal(O.dequant_batch(d["quant%d_qp%d_sh0" % (w, qp)], w, qp, 0), d["dequant%d_qp%d" % (w, qp)])
        for intra in (0, 1):
            rec, co, has = O.quantize_residual_batch(d["qr_ref%d" % w], d["qr_pred%d" % w], w, 22, 0, 0, intra, intra)
            np.testing.assert_array_equal(rec, d["qr_rec%d_i%d" % (w, intra)])
            np.testing.assert_array_equal(co, d["qr_coeff%d_i%d" % (w, intra)])
            np.testing.assert_array_equal(has, d["qr_has%d_i%d" % (w, intra)])


def test_golden_ipol():
    d = gold("ipol.npz")
    frame, pic = d["frame"], d["pic"]
    for kind, blocks in (("luma", d["luma_blocks"]), ("luma14", d["luma_blocks"]),
                         ("chroma", d["chroma_blocks"]), ("chroma14", d["chroma_blocks"])):
        got = np.concatenate([O.sample(kind, frame, int(b[0]), int(b[1]), int(b[4]), int(b[5]), int(b[2]), int(b[3])).ravel()
                              for b in blocks])
        np.testing.assert_array_equal(got, d[kind])
    for c, costs, best in zip(d["sf_cases"], d["sf_costs"], d["sf_best"]):
        oc, ob = O.search_frac_costs(pic, frame, *[int(v) for v in c])
        np.testing.assert_array_equal(oc, costs)
        assert ob == tuple(int(v) for v in best)
    for i, (ox, oy) in enumerate(((0, 0), (-1, 1), (1, -1))):
        got = O.filter_frac_steps(frame, 20, 18, 16, 16, (ox, oy))[:, :, :16, :16]
        np.testing.assert_array_equal(got, d["filter_steps"][i])


# ---- fixtures of the rows added after the core path: intra, SAO, motion search ----
def test_golden_intra():
    d = gold("intra.npz")
    for lg in (2, 3, 4, 5):
        refs, orig = d["refs%d" % lg], d["orig%d" % lg]
        for fb in (0, 1):
            np.testing.assert_array_equal(O.intra_predict_batch(refs, lg, list(range(35)), 1, fb), d["pred%d_fb%d" % (lg, fb)])
            satd, sad = O.intra_rough_costs_batch(refs, lg, orig, fb)
            np.testing.assert_array_equal(satd, d["satd%d_fb%d" % (lg, fb)])
            np.testing.assert_array_equal(sad, d["sad%d_fb%d" % (lg, fb)])
        np.testing.assert_array_equal(O.intra_predict_batch(refs, lg, list(range(35)), 0, 1), d["pred%d_chroma" % lg])


def test_golden_intra_ref():
    d = gold("intra_ref.npz")
    pic_w, pic_h = (int(v) for v in d["size"])
    for color in (0, 1, 2):
        for lg in (2, 3, 4, 5):
            got = O.intra_build_reference_batch(lg, color, d["plane%d" % color], pic_w, pic_h, d["xy%d_c%d" % (lg, color)])
            np.testing.assert_array_equal(got, d["refs%d_c%d" % (lg, color)], err_msg="color %d log2 %d" % (color, lg))


def test_golden_inter_candidates():
    """inputs are regenerated from the seeds (patterns.inter_cand_case), the fixture holds what the reference derived from them"""
    from patterns import INTER_CAND_CONFIGS, inter_cand_case
    d = gold("inter_cand.npz")
    for (name, *_rest) in INTER_CAND_CONFIGS:
        p, cus, col, refm, pus = inter_cand_case(name, 0)
        got_pus, got_merge = O.inter_candidates(p, cus, col, refm, pus)
        np.testing.assert_array_equal(got_pus.view(np.uint8), d[name + "_out_pus"].view(np.uint8), err_msg=name)
        np.testing.assert_array_equal(got_merge.view(np.uint8), d[name + "_out_merge"].view(np.uint8), err_msg=name)


def test_reference_known_answers_of_the_candidate_helpers():
    """tests/mv_cand_tests.c: the spatial candidates' places in lcu_t.cu (:26-49) and the truth tables of is_a0_cand_coded (:51-133)
    and is_b0_cand_coded (:135-213); then every PU of every partition mode against what the reference's functions returned
    (tests/golden/mv_cand.npz, oracle/gen_golden.py: mv_cand)"""
    from patterns import MV_CAND_KAT_A0, MV_CAND_KAT_B0, MV_CAND_KAT_SPATIAL
    (x, y, w, h, pw, ph), want = MV_CAND_KAT_SPATIAL
    assert tuple(O.mv_cand_helpers([(x, y, w, h)], pw, ph)[2][0]) == want
    a0 = O.mv_cand_helpers([g for g, _ in MV_CAND_KAT_A0], 1920, 1080)[0]
    assert [bool(v) for v in a0] == [e for _, e in MV_CAND_KAT_A0]
    b0 = O.mv_cand_helpers([g for g, _ in MV_CAND_KAT_B0], 1920, 1080)[1]
    assert [bool(v) for v in b0] == [e for _, e in MV_CAND_KAT_B0]
    d = gold("mv_cand.npz")
    a0, b0, idx = O.mv_cand_helpers(d["geoms"], 192, 192)
    np.testing.assert_array_equal(a0, d["a0"])
    np.testing.assert_array_equal(b0, d["b0"])
    np.testing.assert_array_equal(idx, d["idx"])


def test_candidate_places_show_in_the_derived_merge_lists():
    """the same places through the full derivation: in a picture whose units all carry a vector naming them, the first merge
    candidates of every PU are the units at A1, B1, B0, A0, B2 (inter.c:1314-1446 order) that mv_cand.npz says exist"""
    from patterns import mv_cand_unique_map_case
    d = gold("mv_cand.npz")
    p, cus, pus = mv_cand_unique_map_case(192)
    out_pus, out_merge = O.inter_candidates(p, cus, None, None, pus)
    from patterns import check_unique_map_merge_lists
    check_unique_map_merge_lists(d, pus, np.asarray(out_merge))


def test_reference_bipred_unit_test_configuration():
    """tests/inter_recon_bipred_tests.c: its configuration (16x16 at the LCU origin, both vectors fractional, zero buffers) and seeded
    variants -- the test file's own restatement of the blend (:74-121, patterns.bipred_expected), the oracle's blend and what the
    compiled reference's generic strategy wrote (tests/golden/bipred.npz) must all agree"""
    from patterns import BIPRED_CASES, bipred_case_blocks, bipred_case_inputs, bipred_expected
    d = gold("bipred.npz")
    for k, (seed, w, h, x, y, hi) in enumerate(BIPRED_CASES):
        hp0, hp1, rec, tmp = bipred_case_inputs(seed)
        want = [d["y%d" % k], d["u%d" % k], d["v%d" % k]]
        for a, b in zip(bipred_expected(hi, w, h, x, y, hp0, hp1, rec, tmp), want):
            np.testing.assert_array_equal(a, b, err_msg="case %d: the test file's formula" % k)
        for plane, (bw, bh, hi0, s0, hi1, s1, (rows, cols)) in enumerate(bipred_case_blocks(k)):
            stride = 64 if plane == 0 else 32
            np.testing.assert_array_equal(O.bipred_blend_plane(bw, bh, hi0, s0, hi1, s1), want[plane].reshape(stride, stride)[rows, cols],
                                          err_msg="case %d plane %d" % (k, plane))
            outside = want[plane].reshape(stride, stride).copy()
            outside[rows, cols] = rec[plane].reshape(stride, stride)[rows, cols]
            np.testing.assert_array_equal(outside.ravel(), rec[plane], err_msg="case %d: pixels outside the block are untouched" % k)


def test_golden_recorded_candidates():
    """candidates the reference encoder derived during a real encode, from snapshots of the state its functions read"""
    from patterns import recorded_cand_fixture
    got, want = recorded_cand_fixture(gold("recorded_cand.npz"), lambda *a: O.inter_candidates(*a)[0])
    for fld in ("num_merge_cand", "merge", "mv_cand", "extra_mv"):
        np.testing.assert_array_equal(got[fld], want[fld], err_msg=fld)


def test_golden_sao():
    d = gold("sao.npz")
    for (bw, bh) in ((64, 64), (32, 32), (64, 40), (8, 16)):
        key = "%dx%d" % (bw, bh)
        orig, rec = d["orig" + key], d["rec" + key]
        for i in range(len(orig)):
            for eo in range(4):
                np.testing.assert_array_equal(O.calc_sao_edge_dir(orig[i], rec[i], eo, bw, bh), d["edge" + key][i, eo])
                assert O.sao_edge_ddistortion(orig[i], rec[i], bw, bh, eo, d["offs" + key][i, eo]) == d["edge_dd" + key][i, eo]
            assert O.sao_band_ddistortion(orig[i], rec[i], bw, bh, int(d["band_pos" + key][i]), d["band_offs" + key][i]) == d["band_dd" + key][i]
    plane, recs, blocks = d["plane"], d["records"], d["blocks"]
    for color in (0, 2):
        got = np.concatenate([O.sao_reconstruct_color(plane, int(x), int(y), int(w), int(h), s, color).ravel()
                              for s in recs for (x, y, w, h) in blocks])
        np.testing.assert_array_equal(got, d["recon_c%d" % color])


def test_golden_motion_search():
    from patterns import ME_PARAMS, ME_PU, me_params_from, me_pus_in_tile
    d = gold("me.npz")
    pus = np.ascontiguousarray(d["pus"]).view(ME_PU).reshape(-1)
    n_cfg = sum(1 for k in d.files if k.startswith("params"))
    assert n_cfg >= 12
    from patterns import ME_CABAC
    cab = np.ascontiguousarray(d["cabac"]).view(ME_CABAC).reshape(-1)
    for i in range(n_cfg):
        prm = me_params_from(d["params%d" % i])
        kw = dict(cabac=cab) if int(prm["mv_rdo"][0]) else {}
        got = O.search_pu_batch(d["pic"], d["ref"], me_pus_in_tile(pus, prm), prm, **kw).view(np.int32).reshape(len(pus), 8)
        np.testing.assert_array_equal(got[:, :7], d["results%d" % i][:, :7])


def test_golden_deblock():
    from patterns import CU_INFO, DEBLOCK_PARAMS
    d = gold("deblock.npz")
    for i in range(3):
        cus = np.ascontiguousarray(d["cus%d" % i]).view(CU_INFO).reshape(d["cus%d" % i].shape[:2])
        prm = np.ascontiguousarray(d["params%d" % i]).view(DEBLOCK_PARAMS)
        chroma = bool(prm["chroma"][0])
        got = O.deblock_frame(d["y%d" % i], d["u%d" % i] if chroma else None, d["v%d" % i] if chroma else None, cus, prm)
        np.testing.assert_array_equal(got[0], d["out_y%d" % i])
        if chroma:
            np.testing.assert_array_equal(got[1], d["out_u%d" % i])
            np.testing.assert_array_equal(got[2], d["out_v%d" % i])


def test_golden_front_replay():
    """searches recorded from real encodes by the reference encoder itself (candidates the encoder derived, decisions the
    reference search took): the oracle must take the same decisions -- every search of the small sequence, a strided sample of
    the 1080p frame"""
    from patterns import fronts_fixture
    d = gold("fronts.npz")
    total = 0
    for which, stride in (("small", 1), ("hd", 97)):
        for (pic, ref, pus, want, meta, prm) in fronts_fixture(d, which):
            got = O.search_pu_batch(pic, ref, pus[::stride], prm)
            for f in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
                np.testing.assert_array_equal(got[f], want[::stride][f], err_msg="%s %s" % (which, f))
            total += len(got)
    assert total > 1900
