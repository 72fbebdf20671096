#!/usr/bin/env python3
"""Randomised differential run of the batched entries against the oracle, larger than the test suite
(development tool; run on the GPU box: python tools/fuzz_parity.py --seed 1 --scale 1)."""
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
from patterns import (ME_RESULT, intra_ref_cases, me_frames, me_params, me_random_pus, sao_blocks)  # noqa: E402


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
        pus = me_random_pus(w, h, 150, int(g.integers(0, 1 << 30)), hint=(-4 * motion[0] + 2, -4 * motion[1]), sizes=sizes)
        got = api.search_pu_batch(pic, ref, pus, prm).view(ME_RESULT).reshape(-1)
        want = O.search_pu_batch(pic, ref, pus, prm)
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
    print("intra ok (%.0f s)" % (time.time() - t0))

    # fused TU: random qp / flags / sizes, with the rd=0 costs
    for it in range(40 * a.scale):
        w = int(g.choice([4, 8, 16, 32]))
        cnt = int(g.integers(1, 40))
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
    print("FUZZ OK seed %d scale %d" % (a.seed, a.scale))


if __name__ == "__main__":
    main()
