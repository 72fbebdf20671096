#!/usr/bin/env python3
"""Randomised differential run of the batched entries against the oracle, larger than the test suite
(a checker, not collected by pytest; run on the GPU box: python tests/fuzz_parity.py --seed 1 --scale 1)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
from kvazaar_amd import api, _lib  # noqa: E402
from patterns import (ME_RESULT, intra_ref_cases, me_cabac_states, me_frames, me_params, me_pus_in_tile, me_random_pus, sao_blocks)  # noqa: E402


def check(name, ok, detail=""):
    if not ok:
        print("MISMATCH", name, detail)
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scale", type=int, default=1)
    a = ap.parse_args()
    g = np.random.default_rng(a.seed)
    _lib.init(0)
    t0 = time.time()

    # motion search: random settings x random PUs x random motion
    n = 0
    for it in range(12 * a.scale):
        prm = me_params(lambda_cost=int(g.integers(0, 120)), early_termination=int(g.integers(0, 3)),
                        max_steps=int(g.choice([0xFFFFFFFF, 0, 1, 3, 8])), fme_level=int(g.integers(0, 5)),
                        wpp_owf=int(g.integers(0, 2)), ref_delay_px=int(g.choice([0, 8, 10])),
                        max_ref_lcu_down=int(g.integers(0, 3)), max_ref_lcu_right=int(g.integers(0, 3)),
                        algorithm=int(g.integers(0, 4)), search_range=int(g.integers(1, 13)))
        w, h = int(g.choice([128, 192, 320])), int(g.choice([64, 128, 200]))
        motion = (int(g.integers(-20, 21)), int(g.integers(-20, 21)))
        pic, ref = me_frames(w, h, int(g.integers(0, 1 << 30)), motion)
        sizes = tuple((bw, bh) for bw in (8, 16, 24, 32, 48, 64) for bh in (8, 16, 24, 32, 48, 64) if bw <= w and bh <= h)
        sizes += ((8, 4), (4, 8), (16, 4), (4, 16), (16, 12), (12, 16)) * 2          # the AMP / SMP shapes
        pus = me_random_pus(w, h, 150, int(g.integers(0, 1 << 30)), hint=(-4 * motion[0] + 2, -4 * motion[1]), sizes=sizes)
        # round 2: the mv_constraint branches with and without a real tile, and --mv-rdo from random CABAC snapshots
        kw = {}
        if it % 3 == 1:
            prm["mv_constraint"] = int(g.integers(1, 5))
            if g.integers(0, 2):
                tx, ty = int(g.integers(0, w // 64)) * 64, int(g.integers(0, max(1, h // 64))) * 64
                tw, th = int(g.integers(1, (w - tx) // 8 + 1)) * 8, int(g.integers(1, (h - ty) // 8 + 1)) * 8
                prm["tile_x"], prm["tile_y"], prm["tile_w"], prm["tile_h"] = tx, ty, tw, th
                pus = me_pus_in_tile(pus, prm)
                pus = pus[(pus["width"] <= tw) & (pus["height"] <= th) & (pus["x"] + pus["width"] <= tx + tw) & (pus["y"] + pus["height"] <= ty + th)]
        if it % 3 == 2:
            prm["mv_rdo"], prm["refs_before"] = 1, int(g.integers(1, 6))
            prm["ref_idx"] = int(g.integers(0, prm["refs_before"][0]))
            cab = me_cabac_states(11, int(g.integers(0, 1 << 30)))
            pus["reserved"] = g.integers(0, 11, len(pus))
            kw = dict(cabac=cab)
        if len(pus) == 0:
            continue
        if it % 2 == 1:                                  # a later picture of a multi-reference search: a cost to beat per PU
            from patterns import cost_to_beat_case
            kw["cost_to_beat"] = cost_to_beat_case(O.search_pu_batch(pic, ref, pus, prm, **kw)["cost"], int(g.integers(0, 1 << 30)))
        got = api.search_pu_batch(pic, ref, pus, prm, **kw).view(ME_RESULT).reshape(-1)
        want = O.search_pu_batch(pic, ref, pus, prm, **kw)
        for f in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
            check("search_pu." + f, np.array_equal(got[f], want[f]), "iter %d prm %s" % (it, prm))
        n += len(pus)
    print("search_pu: %d PUs ok (%.0f s)" % (n, time.time() - t0))

    # intra: every size, random + structured references
    for lg in (2, 3, 4, 5):
        nn = 1 << lg
        refs = intra_ref_cases(lg, 300 * a.scale, int(g.integers(0, 1 << 30)))
        orig = g.integers(0, 256, (len(refs), nn * nn), dtype=np.uint8)
        for fb in (0, 1):
            satd, sad = api.intra_rough_batch(refs, lg, orig, 1 | (fb << 1), with_sad=True)
            ws, wd = O.intra_rough_costs_batch(refs, lg, orig, fb)
            check("intra_rough.satd", np.array_equal(satd, ws), "log2 %d fb %d" % (lg, fb))
            check("intra_rough.sad", np.array_equal(sad, wd), "log2 %d fb %d" % (lg, fb))
        for flags in (0, 1, 3):
            got = api.intra_predict_batch(refs[:60], lg, list(range(35)), flags)
            want = O.intra_predict_batch(refs[:60], lg, list(range(35)), is_luma=flags & 1, filter_boundary=(flags >> 1) & 1)
            check("intra_predict", np.array_equal(got, want), "log2 %d flags %d" % (lg, flags))
    # intra references from a reconstruction plane: random picture sizes (multiples of 8), strides, colours, every PU position
    from patterns import intra_ref_positions
    for it in range(6 * a.scale):
        pw, ph, color, lg = 8 * int(g.integers(1, 40)), 8 * int(g.integers(1, 30)), int(g.integers(0, 3)), int(g.integers(2, 6))
        c = 1 if color else 0
        plane = g.integers(0, 256, (ph >> c, (pw >> c) + int(g.integers(0, 9))), dtype=np.uint8)
        xy = intra_ref_positions(lg, color, pw, ph)
        if len(xy):
            check("intra_build_reference", np.array_equal(api.intra_build_reference_batch(lg, color, plane, pw, ph, xy),
                                                          O.intra_build_reference_batch(lg, color, plane, pw, ph, xy)),
                  "%dx%d color %d log2 %d" % (pw, ph, color, lg))
    print("intra ok (%.0f s)" % (time.time() - t0))

    # AMVP / merge candidate derivation: every configuration with fresh seeds
    from patterns import INTER_CAND_CONFIGS, ME_PU, inter_cand_case
    n = 0
    for (name, *_rest) in INTER_CAND_CONFIGS:
        for _ in range(2 * a.scale):
            sd = int(g.integers(0, 1 << 20))
            p, cus, col, refm, pus = inter_cand_case(name, sd)
            want_pus, want_merge = O.inter_candidates(p, cus, col, refm, pus)
            got_pus, got_merge = api.inter_candidates_batch(p, cus, col, refm, pus)
            check("inter_candidates.pus", np.array_equal(got_pus.view(ME_PU).ravel(), want_pus), "%s seed %d" % (name, sd))
            check("inter_candidates.merge", np.array_equal(got_merge.ravel(), want_merge.view(np.uint8).ravel()), "%s seed %d" % (name, sd))
            n += len(pus)
    print("inter candidates: %d PUs ok (%.0f s)" % (n, time.time() - t0))

    # fused TU: random qp / flags / sizes, with the rd=0 costs
    for it in range(40 * a.scale):
        w = int(g.choice([4, 8, 16, 32]))
        cnt = int(g.integers(1, 40)) if it % 4 else int(g.integers(40, 200))      # every tail of the 8-TU wave step / 4-TU tile
        ref_in = g.integers(0, 256, (cnt, w * w), dtype=np.uint8)
        amp = int(g.choice([2, 10, 60, 255]))
        pred = np.clip(ref_in.astype(np.int32) + g.integers(-amp, amp + 1, ref_in.shape), 0, 255).astype(np.uint8)
        qp, intra, sh = int(g.integers(0, 52)), int(g.integers(0, 2)), int(g.integers(0, 2))
        color = 0 if w == 32 else int(g.integers(0, 3))
        ts = int(g.integers(0, 2)) if w == 4 else 0
        scan = int(g.integers(0, 3)) if w in (4, 8) else 0
        got = api.quantize_residual_batch(ref_in, pred, w, qp, color, scan, intra, intra, sh, ts, with_costs=True)
        want = O.quantize_residual_batch(ref_in, pred, w, qp, color, scan, intra, intra, sh, ts)
        for x, y, nm in zip(got[:3], want, ("rec", "coeff", "has")):
            check("quantize_residual." + nm, np.array_equal(x, y), "w %d qp %d color %d intra %d sh %d ts %d scan %d" % (w, qp, color, intra, sh, ts, scan))
        for i in range(cnt):
            check("qr.ssd", got[3][i] == O.pixels_calc_ssd(ref_in[i], 0, want[0][i], 0, w, w, w))
            check("qr.abs_sum", got[4][i] == O.coeff_abs_sum(want[1][i]))
    print("quantize_residual ok (%.0f s)" % (time.time() - t0))

    # SAO statistics over odd block shapes
    for it in range(30 * a.scale):
        bw, bh = int(g.integers(1, 65)), int(g.integers(1, 65))
        orig, rec = sao_blocks(bw, bh, 6, int(g.integers(0, 1 << 30)))
        stats = api.sao_edge_stats_batch(orig, rec, bw, bh)
        bands = api.sao_band_stats_batch(orig, rec, bw, bh)
        for i in range(6):
            for eo in range(4):
                check("sao_edge_stats", np.array_equal(stats[i, eo], O.calc_sao_edge_dir(orig[i], rec[i], eo, bw, bh)), "%dx%d" % (bw, bh))
            check("sao_band_stats", np.array_equal(bands[i], O.calc_sao_bands(orig[i], rec[i], bw, bh)), "%dx%d" % (bw, bh))
    print("sao ok (%.0f s)" % (time.time() - t0))

    # interpolation: random blocks incl. far outside the frame
    frame = g.integers(0, 256, (120, 136), dtype=np.uint8)
    PAD = 300                                            # the oracle reads a plain window: give it the edge-replicated plane
    padded = np.pad(frame, PAD, mode="edge")             # (kvz_get_extended_block semantics, ipol-generic.c:731-784)
    for kind, nfrac, sizes in (("luma", 4, (8, 16, 24, 32, 64)), ("luma14", 4, (8, 16, 64)), ("chroma", 8, (2, 4, 8, 16, 32)), ("chroma14", 8, (4, 8, 32))):
        blocks = [(int(g.integers(-90, 200)), int(g.integers(-90, 190)), int(g.integers(0, nfrac)), int(g.integers(0, nfrac)),
                   int(g.choice(sizes)), int(g.choice(sizes))) for _ in range(120 * a.scale)]
        got = api.sample_batch(kind, frame, blocks)
        for b, o in zip(blocks, got):
            x, y, fx, fy, w, h = b
            check("sample." + kind, np.array_equal(o, O.sample(kind, padded, x + PAD, y + PAD, w, h, fx, fy)), str(b))
    print("sample ok (%.0f s)" % (time.time() - t0))

    # picture group: contiguous blocks of every size, frame-level descriptors incl. vectors that leave the frame
    for n in (4, 8, 16, 32, 64):
        cnt = int(g.integers(1, 40))
        a8 = g.integers(0, 256, (cnt, n * n), dtype=np.uint8)
        b8 = np.clip(a8.astype(np.int32) + g.integers(-30, 31, a8.shape), 0, 255).astype(np.uint8)
        for kind in ("sad", "satd"):
            check("%s_%d" % (kind, n), np.array_equal(api.cost_nxn_batch(kind, n, a8, b8), O.cost_nxn_batch(kind, n, a8, b8)))
    picf = g.integers(0, 256, (96, 160), dtype=np.uint8)
    reff = g.integers(0, 256, (96, 160), dtype=np.uint8)
    pairs = []
    for _ in range(150 * a.scale):
        bw, bh = int(g.choice([4, 8, 12, 16, 24, 32, 64])), int(g.choice([4, 8, 12, 16, 32, 64]))
        x1, y1 = int(g.integers(0, 160 - bw + 1)), int(g.integers(0, 96 - bh + 1))
        pairs.append((x1, y1, x1 + int(g.integers(-200, 201)), y1 + int(g.integers(-120, 121)), bw, bh))
    got_sad, got_satd = api.image_calc_sad_batch(picf, reff, pairs), api.image_calc_satd_batch(picf, reff, pairs)
    for pr, s1, s2 in zip(pairs, got_sad, got_satd):
        check("image_calc_sad", s1 == O.image_calc("sad", picf, reff, *pr), str(pr))
        check("image_calc_satd", s2 == O.image_calc("satd", picf, reff, *pr), str(pr))
    print("picture ok (%.0f s)" % (time.time() - t0))

    # transforms / quant / dequant on full-range inputs
    for n in (4, 8, 16, 32):
        x = g.integers(-32768, 32768, (int(g.integers(1, 300)), n * n)).astype(np.int16)      # tails of the 4- and 64-block tiles
        r = g.integers(-255, 256, x.shape).astype(np.int16)
        for kind, src in (("dct", r), ("idct", x)) + ((("dst", r), ("idst", x)) if n == 4 else ()):
            check("%s_%d" % (kind, n), np.array_equal(api.transform_batch(kind, n, src), O.transform_batch(kind, n, src)))
        qp, sh, intra = int(g.integers(0, 52)), int(g.integers(0, 2)), int(g.integers(0, 2))
        tp = 0 if n == 32 else int(g.choice([0, 2]))
        check("quant_%d" % n, np.array_equal(api.quant_batch(x, n, qp, tp, 0, intra, sh), O.quant_batch(x, n, qp, tp, 0, intra, sh)),
              "qp %d sh %d intra %d type %d" % (qp, sh, intra, tp))
        check("dequant_%d" % n, np.array_equal(api.dequant_batch(x, n, qp, tp), O.dequant_batch(x, n, qp, tp)))
    print("transform / quant ok (%.0f s)" % (time.time() - t0))

    # bi-prediction candidate costs and the fractional search, PUs inside an LCU
    pic, ref0 = me_frames(192, 128, int(g.integers(0, 1 << 30)), (2, -1))
    _, ref1 = me_frames(192, 128, int(g.integers(0, 1 << 30)), (-3, 2))
    cands, sf = [], []
    for _ in range(120 * a.scale):
        w, h = int(g.choice([8, 16, 24, 32, 64])), int(g.choice([8, 16, 32, 64]))
        x = int(g.integers(0, 3)) * 64 + int(g.integers(0, (64 - w) // 8 + 1)) * 8
        y = int(g.integers(0, 2)) * 64 + int(g.integers(0, (64 - h) // 8 + 1)) * 8
        big = int(g.choice([12, 60, 500]))
        mv = g.integers(-big, big + 1, 4)
        cands.append((x, y, w, h, int(mv[0]), int(mv[1]), int(mv[2]), int(mv[3])))
        sf.append((x, y, x + int(mv[0]) // 4, y + int(mv[1]) // 4, w, h))
    got = api.bipred_cost_batch(pic, ref0, ref1, cands)
    for c, v in zip(cands, got):
        check("bipred_cost", v == O.bipred_luma_satd(pic, ref0, ref1, c[0], c[1], c[2], c[3], c[4:6], c[6:8])[0], str(c))
    costs, best = api.search_frac_batch(pic, ref0, sf)
    for k, d in enumerate(sf):
        oc, ob = O.search_frac_costs(pic, ref0, d[0], d[1], d[4], d[5], d[2] - d[0], d[3] - d[1])
        check("search_frac.costs", np.array_equal(costs[k], oc), str(d))
        check("search_frac.best", tuple(int(v) for v in best[k]) == tuple(ob), str(d))
    print("bipred / search_frac ok (%.0f s)" % (time.time() - t0))

    # SAO distortion deltas and reconstruction
    for it in range(12 * a.scale):
        bw, bh = int(g.integers(3, 65)), int(g.integers(3, 65))
        orig, rec = sao_blocks(bw, bh, 5, int(g.integers(0, 1 << 30)))
        offs = g.integers(-7, 8, (5, 4, 5)).astype(np.int32)
        dd = api.sao_edge_ddistortion_batch(orig, rec, bw, bh, offs)
        bp, bo = g.integers(0, 32, 5).astype(np.int32), g.integers(-7, 8, (5, 4)).astype(np.int32)
        bd = api.sao_band_ddistortion_batch(orig, rec, bw, bh, bp, bo)
        for i in range(5):
            for eo in range(4):
                check("sao_edge_dd", dd[i, eo] == O.sao_edge_ddistortion(orig[i], rec[i], bw, bh, eo, offs[i, eo]))
            check("sao_band_dd", bd[i] == O.sao_band_ddistortion(orig[i], rec[i], bw, bh, int(bp[i]), bo[i]))
    plane = g.integers(0, 256, (90, 130), dtype=np.uint8)
    from patterns import sao_records
    infos = sao_records(40, int(g.integers(0, 1 << 30)))
    blocks = [(int(g.integers(1, 60)), int(g.integers(1, 40)), int(g.integers(1, 65)), int(g.integers(1, 45)), k) for k in range(40)]
    for color in (0, 1, 2):
        for b, info in zip(blocks, infos):
            got = api.sao_reconstruct_color_batch(plane, [b[:4] + (0,)], info[None], color)
            want = plane.copy()
            want[b[1]:b[1] + b[3], b[0]:b[0] + b[2]] = O.sao_reconstruct_color(plane, b[0], b[1], b[2], b[3], info, color)
            check("sao_reconstruct", np.array_equal(got, want), str(b))
    print("sao dd / reconstruct ok (%.0f s)" % (time.time() - t0))

    # ---- deblocking: random frame sizes, QPs, offsets, slice types ----
    t0 = time.time()
    from patterns import deblock_case, deblock_params
    for _ in range(6 * a.scale):
        w, h = int(g.integers(1, 40)) * 8, int(g.integers(1, 30)) * 8
        prm = deblock_params(qp=int(g.integers(18, 52)), beta=int(g.integers(-6, 7)), tc=int(g.integers(-6, 7)), per_cu_qp=int(g.integers(0, 2)),
                             slice_is_b=int(g.integers(0, 2)), chroma=int(g.integers(0, 4) > 0))
        chroma = bool(prm["chroma"][0])
        y, u, v, cus = deblock_case(w, h, int(g.integers(0, 1 << 30)), intra_share=float(g.random()), slice_is_b=int(prm["slice_is_b"][0]),
                                    qp=int(prm["qp"][0]))
        want = O.deblock_frame(y, u if chroma else None, v if chroma else None, cus, prm)
        got = api.deblock_frame(y, u if chroma else None, v if chroma else None, cus, prm)
        for k in range(3 if chroma else 1):
            check("deblock_frame", np.array_equal(got[k], want[k]), "%dx%d plane %d" % (w, h, k))
    print("deblock ok (%.0f s)" % (time.time() - t0))
    print("FUZZ OK seed %d scale %d" % (a.seed, a.scale))


if __name__ == "__main__":
    main()
