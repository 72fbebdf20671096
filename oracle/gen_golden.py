#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the COMPILED REFERENCE's `generic` strategy
(oracle/_ref/libkvzref.so, built from /root/reference by oracle/Makefile).

Run in the build container:  python oracle/gen_golden.py
The fixtures are data only (seeded inputs + the reference's outputs) and are
committed; tests/test_oracle_golden.py checks the oracle against them on any
machine (the GPU box has no /root/reference), tests/test_gpu_golden.py checks the
HIP kernels against them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_lib as R  # noqa: E402
from patterns import (dct_test_input, intra_ref_cases, me_frames, me_params, me_pus_in_tile, me_random_pus, rng, sao_blocks,  # noqa: E402
                      sao_records)

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 20261004


def picture():
    g = rng(SEED)
    d = {}
    for n in (4, 8, 16, 32, 64):
        cnt = 24 if n < 64 else 6
        a = g.integers(0, 256, (cnt, n * n), dtype=np.uint8)
        b = np.clip(a.astype(np.int32) + g.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
        b[::3] = g.integers(0, 256, (len(b[::3]), n * n), dtype=np.uint8)
        d["a%d" % n], d["b%d" % n] = a, b
        d["sad%d" % n] = R.cost_nxn_batch("sad", n, a, b)
        d["satd%d" % n] = R.cost_nxn_batch("satd", n, a, b)
        if n <= 32:
            preds = g.integers(0, 256, (8, 2048), dtype=np.uint8)
            d["dual_preds%d" % n] = preds
            d["sad_dual%d" % n] = R.cost_nxn_dual_batch("sad", n, preds, a[:8])
            d["satd_dual%d" % n] = R.cost_nxn_dual_batch("satd", n, preds, a[:8])
    pic = g.integers(0, 256, (48, 64), dtype=np.uint8)
    ref = g.integers(0, 256, (48, 64), dtype=np.uint8)
    pairs = []
    for (bw, bh) in ((8, 8), (16, 16), (32, 16), (12, 8), (4, 4), (64, 48)):
        for (dx, dy) in ((0, 0), (-3, -3), (5, -70), (70, 9), (3, 0)):
            px, py = (64 - bw) // 2, (48 - bh) // 2
            pairs.append((px, py, px + dx, py + dy, bw, bh))
    d["frame_pic"], d["frame_ref"] = pic, ref
    d["pairs"] = np.array(pairs, dtype=np.int32)
    d["image_sad"] = np.array([R.image_calc("sad", pic, ref, *p) for p in pairs], dtype=np.uint32)
    d["image_satd"] = np.array([R.image_calc("satd", pic, ref, *p) for p in pairs], dtype=np.uint32)
    # quad incl. the non-multiple-of-8 behaviour
    qp = g.integers(0, 256, (4, 64 * 64), dtype=np.uint8)
    dims = [(8, 8), (16, 16), (64, 48), (12, 16), (16, 12), (4, 8), (24, 24)]
    d["quad_preds"] = qp
    d["quad_dims"] = np.array(dims, dtype=np.int32)
    d["quad_costs"] = np.array([R.satd_any_size_quad(w, h, list(qp), 64, pic, 0, 64) for (w, h) in dims], dtype=np.uint32)
    d["ssd"] = np.array([R.pixels_calc_ssd(pic, 0, ref, 0, 64, 64, w) for w in (4, 8, 16, 32)], dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "picture.npz"), **d)


def dct():
    g = rng(SEED + 1)
    d = {"gradient": dct_test_input()}
    for n in (4, 8, 16, 32):
        res = g.integers(-255, 256, (6, n * n)).astype(np.int16)
        full = g.integers(-32768, 32768, (4, n * n)).astype(np.int16)
        x = np.concatenate([res, full, d["gradient"][:n * n][None]])
        d["in%d" % n] = x
        for kind in ("dct", "idct") + (("dst", "idst") if n == 4 else ()):
            d["%s%d" % (kind, n)] = R.transform_batch(kind, n, x)
    np.savez_compressed(os.path.join(OUT, "dct.npz"), **d)


def quant():
    g = rng(SEED + 2)
    d = {}
    for w in (4, 8, 16, 32):
        coef = g.integers(-1500, 1501, (6, w * w)).astype(np.int16)
        coef[0] = g.integers(-32768, 32768, w * w)
        d["coef%d" % w] = coef
        for qp in (22, 37):
            for sh in (0, 1):
                q = R.quant_batch(coef, w, qp, 0, 0, 1, sh)
                d["quant%d_qp%d_sh%d" % (w, qp, sh)] = q
            d["dequant%d_qp%d" % (w, qp)] = R.dequant_batch(d["quant%d_qp%d_sh0" % (w, qp)], w, qp, 0)
        ref_in = g.integers(0, 256, (6, w * w), dtype=np.uint8)
        pred = np.clip(ref_in.astype(np.int32) + g.integers(-30, 31, ref_in.shape), 0, 255).astype(np.uint8)
        pred[0] = ref_in[0]
        d["qr_ref%d" % w], d["qr_pred%d" % w] = ref_in, pred
        for intra in (0, 1):
            rec, co, has = R.quantize_residual_batch(ref_in, pred, w, 22, 0, 0, intra, intra, 0, 0)
            d["qr_rec%d_i%d" % (w, intra)], d["qr_coeff%d_i%d" % (w, intra)], d["qr_has%d_i%d" % (w, intra)] = rec, co, has
    np.savez_compressed(os.path.join(OUT, "quant.npz"), **d)


def ipol():
    g = rng(SEED + 3)
    frame = g.integers(0, 256, (80, 96), dtype=np.uint8)
    frame[30:50, 30:50] = np.where(g.integers(0, 2, (20, 20)) > 0, 255, 0)
    d = {"frame": frame}
    lb = [(12, 10, fx, fy, w, h) for (w, h) in ((8, 8), (16, 16), (32, 16)) for fx in range(4) for fy in range(4)]
    cb = [(12, 10, fx, fy, w, h) for (w, h) in ((4, 4), (8, 8), (16, 8)) for fx in range(8) for fy in range(0, 8, 3)]
    d["luma_blocks"], d["chroma_blocks"] = np.array(lb, np.int32), np.array(cb, np.int32)
    d["luma"] = np.concatenate([R.sample("luma", frame, *b[:2], b[4], b[5], b[2], b[3]).ravel() for b in lb])
    d["luma14"] = np.concatenate([R.sample("luma14", frame, *b[:2], b[4], b[5], b[2], b[3]).ravel() for b in lb])
    d["chroma"] = np.concatenate([R.sample("chroma", frame, *b[:2], b[4], b[5], b[2], b[3]).ravel() for b in cb])
    d["chroma14"] = np.concatenate([R.sample("chroma14", frame, *b[:2], b[4], b[5], b[2], b[3]).ravel() for b in cb])
    pic = ((frame.astype(np.int32) + np.roll(frame, 1, axis=1)) // 2).astype(np.uint8)
    sf = [(x, y, w, h, mvx, mvy) for (w, h) in ((8, 8), (16, 16), (32, 32))
          for (x, y) in ((0, 0), (40, 32)) for (mvx, mvy) in ((0, 0), (-3, 2), (-60, -60), (80, 70))]
    d["pic"] = pic
    d["sf_cases"] = np.array(sf, np.int32)
    costs, bests = [], []
    for (x, y, w, h, mvx, mvy) in sf:
        c, b = R.search_frac_costs(pic, frame, x, y, w, h, mvx, mvy)
        costs.append(c); bests.append(b)
    d["sf_costs"], d["sf_best"] = np.array(costs, np.uint32), np.array(bests, np.int32)
    # the four filter steps' blocks for one case per hpel offset
    fs = []
    for (ox, oy) in ((0, 0), (-1, 1), (1, -1)):
        fs.append(R.filter_frac_steps(frame, 20, 18, 16, 16, (ox, oy))[:, :, :16, :16])
    d["filter_steps"] = np.array(fs)
    np.savez_compressed(os.path.join(OUT, "ipol.npz"), **d)


def intra():
    """kvz_intra_predict (with the generic angular / planar strategies) for every mode, and the rough-search costs
    satd_NxN / sad_NxN of those predictions (search_intra.c:99-172)"""
    g = rng(SEED + 4)
    d = {}
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        refs = intra_ref_cases(lg, 10, SEED + 40 + lg)
        orig = g.integers(0, 256, (len(refs), n * n), dtype=np.uint8)
        d["refs%d" % lg], d["orig%d" % lg] = refs, orig
        for fb in (0, 1):
            pred = np.array([[R.intra_predict(r, lg, m, 0, fb) for m in range(35)] for r in refs], dtype=np.uint8)
            d["pred%d_fb%d" % (lg, fb)] = pred
            satd = np.array([[R.cost_nxn_batch("satd", n, pred[i, m][None], orig[i][None])[0] for m in range(35)]
                             for i in range(len(refs))], dtype=np.uint32)
            sad = np.array([[R.cost_nxn_batch("sad", n, pred[i, m][None], orig[i][None])[0] for m in range(35)]
                            for i in range(len(refs))], dtype=np.uint32)
            d["satd%d_fb%d" % (lg, fb)], d["sad%d_fb%d" % (lg, fb)] = satd, sad
        d["pred%d_chroma" % lg] = np.array([[R.intra_predict(r, lg, m, 1, 1) for m in range(35)] for r in refs], dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "intra.npz"), **d)


def intra_ref():
    """kvz_intra_build_reference (intra.c:334-588) at every PU position of a 104 x 88 picture (one whole LCU, a ragged
    column and row of them), each PU on an lcu_t cut from the planes as init_lcu_t does, with everything that is coded
    after the PU overwritten with noise"""
    from patterns import intra_ref_positions
    g = rng(SEED + 45)
    pic_w, pic_h = 104, 88
    d = {"size": np.array([pic_w, pic_h], np.int32)}
    for color in (0, 1, 2):
        plane = g.integers(0, 256, (pic_h >> (color > 0), pic_w >> (color > 0)), dtype=np.uint8)
        d["plane%d" % color] = plane
        for lg in (2, 3, 4, 5):
            xy = intra_ref_positions(lg, color, pic_w, pic_h)
            d["xy%d_c%d" % (lg, color)] = xy
            refs = np.array([R.intra_build_reference_from_plane(lg, color, plane, pic_w, pic_h, int(x), int(y), poison=g) for (x, y) in xy])
            n2 = 2 << lg
            refs[:, n2 + 1:65] = 0
            refs[:, 65 + n2 + 1:] = 0
            d["refs%d_c%d" % (lg, color)] = refs
    np.savez_compressed(os.path.join(OUT, "intra_ref.npz"), **d)


def inter_cand():
    """kvz_inter_get_merge_cand / kvz_inter_get_mv_cand / the start vector of search_pu_inter_ref (oracle/ref_harness.c:
    ref_inter_candidates) for every inter PU of seeded random CU maps: P and B slices, one to four references, POC distances up
    to the scaling clamp, TMVP on and off, a tile"""
    from patterns import INTER_CAND_CONFIGS, inter_cand_case
    d = {}
    for (name, *_rest) in INTER_CAND_CONFIGS:
        p, cus, col, refm, pus = inter_cand_case(name, 0)
        out_pus, out_merge = R.inter_candidates(p, cus, col, refm, pus)
        d[name + "_out_pus"], d[name + "_out_merge"] = out_pus, out_merge
    np.savez_compressed(os.path.join(OUT, "inter_cand.npz"), **d)


def mv_cand():
    """the reference's file-local candidate helpers (inter.c:566-875; oracle/ref_cand_harness.c reaches them the way
    tests/mv_cand_tests.c does) for every PU of every partition mode in a 192 x 192 picture"""
    from patterns import valid_pu_geometries
    geoms = valid_pu_geometries(192)
    a0, b0, idx = R.mv_cand_helpers(geoms, 192, 192)
    np.savez_compressed(os.path.join(OUT, "mv_cand.npz"), geoms=geoms, a0=a0.astype(np.uint8), b0=b0.astype(np.uint8), idx=idx.astype(np.int16))


def bipred():
    """inter_recon_bipred of the compiled reference's generic strategy (picture-generic.c:538-588) on the configuration of
    tests/inter_recon_bipred_tests.c (16x16 at the LCU origin, all sources 14-bit, zero buffers) and seeded variants"""
    from patterns import BIPRED_CASES, bipred_case_inputs
    d = {}
    for k, (seed, w, h, x, y, hi) in enumerate(BIPRED_CASES):
        hp0, hp1, rec, tmp = bipred_case_inputs(seed)
        out = R.bipred(hi, h, w, y, x, hp0, hp1, rec, tmp, "generic")
        d["y%d" % k], d["u%d" % k], d["v%d" % k] = out
    np.savez_compressed(os.path.join(OUT, "bipred.npz"), **d)


def recorded_cand():
    """What the reference ENCODER's candidate derivation read and produced during a real encode (harness recorder with snapshots,
    oracle/ref_harness.c): for 600 of the 2Nx2N inter searches of four 192 x 128 frames the lcu->cu array as it stood, the collocated
    picture's CU array and POC tables per frame, and the search descriptor the encoder's kvz_inter_get_merge_cand /
    kvz_inter_get_mv_cand calls filled"""
    frames = R.synthetic_sequence(192, 128, 4, seed=11)
    rec = R.record_inter_searches(frames, 192, 128, "preset=medium,ref=1,bipred=0,gop=0,rdoq=0,qp=30,threads=0,smp=0,amp=0,period=0", snapshots=4000)
    pick = np.linspace(0, len(rec["snap_index"]) - 1, 600).astype(np.int64)
    idx = rec["snap_index"][pick]
    np.savez_compressed(os.path.join(OUT, "recorded_cand.npz"), snap_cus=rec["snap_cus"][pick], meta=rec["meta"][idx], pus=rec["pus"][idx],
                        snap_col=rec["snap_col"], snap_params=rec["snap_params"])


def sao():
    g = rng(SEED + 5)
    d = {}
    for (bw, bh) in ((64, 64), (32, 32), (64, 40), (8, 16)):
        orig, rec = sao_blocks(bw, bh, 6, SEED + bw + bh)
        key = "%dx%d" % (bw, bh)
        d["orig" + key], d["rec" + key] = orig, rec
        d["edge" + key] = np.array([[R.calc_sao_edge_dir(orig[i], rec[i], eo, bw, bh) for eo in range(4)] for i in range(6)], dtype=np.int32)
        offs = g.integers(-7, 8, (6, 4, 5)).astype(np.int32)
        offs[:, :, 0] = 0
        d["offs" + key] = offs
        d["edge_dd" + key] = np.array([[R.sao_edge_ddistortion(orig[i], rec[i], bw, bh, eo, offs[i, eo]) for eo in range(4)] for i in range(6)], dtype=np.int32)
        bp, bo = g.integers(0, 32, 6).astype(np.int32), g.integers(-7, 8, (6, 4)).astype(np.int32)
        d["band_pos" + key], d["band_offs" + key] = bp, bo
        d["band_dd" + key] = np.array([R.sao_band_ddistortion(orig[i], rec[i], bw, bh, int(bp[i]), bo[i]) for i in range(6)], dtype=np.int32)
    plane = g.integers(0, 256, (72, 88), dtype=np.uint8)
    recs = sao_records(8, SEED + 6)
    blocks = [(1, 1, 64, 64), (5, 3, 32, 32), (1, 7, 61, 13)]
    d["plane"], d["records"], d["blocks"] = plane, recs, np.array(blocks, dtype=np.int32)
    for color in (0, 2):
        d["recon_c%d" % color] = np.concatenate([R.sao_reconstruct_color(plane, x, y, w, h, s, color).ravel()
                                                 for s in recs for (x, y, w, h) in blocks])
    np.savez_compressed(os.path.join(OUT, "sao.npz"), **d)


def me():
    """the reference's static hexagon_search + search_frac (oracle/ref_me_harness.c) for three encoder settings"""
    d = {}
    cfgs = [dict(), dict(early_termination=2, fme_level=2, lambda_cost=35), dict(wpp_owf=1, ref_delay_px=10, lambda_cost=9, early_termination=0),
            dict(algorithm=1, lambda_cost=25), dict(algorithm=2, lambda_cost=15),
            dict(algorithm=3, search_range=8, lambda_cost=30),
            # the mv_constraint branches of fracmv_within_tile (search_inter.c:142-171), frame = one tile and a real tile
            dict(mv_constraint=1, lambda_cost=12), dict(mv_constraint=4, lambda_cost=12, early_termination=0),
            dict(mv_constraint=3, tile=(64, 0, 128, 128), lambda_cost=18),
            dict(mv_constraint=4, tile=(0, 64, 192, 64), wpp_owf=1, ref_delay_px=10, lambda_cost=9, algorithm=1),
            # --mv-rdo: kvz_calc_mvd_cost_cabac as the cost model, from the CABAC snapshots stored beside the results
            dict(mv_rdo=1, lambda_cost=14), dict(mv_rdo=1, refs_before=3, ref_idx=1, algorithm=2, lambda_cost=22)]
    pic, ref = me_frames(192, 128, SEED + 7, (5, -3))
    pus = me_random_pus(192, 128, 48, SEED + 8, hint=(-18, 12))
    d["pic"], d["ref"], d["pus"] = pic, ref, pus.view(np.uint8).reshape(len(pus), 64)
    from patterns import me_cabac_states
    cab = me_cabac_states(6, SEED + 9)
    d["cabac"] = cab.view(np.uint8).reshape(-1, 16)
    pus["reserved"] = np.arange(len(pus)) % 6                       # read only with mv_rdo
    d["pus"] = pus.view(np.uint8).reshape(len(pus), 64)
    for i, c in enumerate(cfgs):
        prm = me_params(**c)
        d["params%d" % i] = prm.view(np.uint8).reshape(-1)           # the pointer field stays 0: loaders attach the snapshots
        d["results%d" % i] = R.search_pu_batch(pic, ref, me_pus_in_tile(pus, prm), prm,
                                               cabac=cab if c.get("mv_rdo") else None).view(np.int32).reshape(len(pus), 8)
    np.savez_compressed(os.path.join(OUT, "me.npz"), **d)


FRONT_OPTS = "preset=medium,ref=1,bipred=0,gop=0,rdoq=0,qp=32,threads=0,smp=0,amp=0,period=0"


def fronts():
    """What the reference ENCODER searched: every 2Nx2N inter search of real encodes (harness recorder, oracle/ref_harness.c:
    __wrap_kvz_search_cu_inter) with the AMVP / merge candidates and start vectors the encoder derived and the decisions the
    reference search took.  Small sequence: planes stored whole.  One 1080p P frame (42 900 searches, BASELINE config 4): the
    source planes are regenerated by ref_lib.synthetic_sequence, the reference picture (the encoder's reconstruction of frame
    0) is stored as its int8 difference from the source frame 0."""
    d = {}
    w, h = 192, 128
    rec = R.record_inter_searches(R.synthetic_sequence(w, h, 4), w, h, FRONT_OPTS)
    assert rec["skipped"] == 0 and len(rec["pus"]) == 3 * 510
    d["small_pic"], d["small_ref"] = rec["pic"], rec["ref"]
    d["small_pus"] = rec["pus"].view(np.uint8).reshape(-1, 64)
    d["small_results"] = rec["results"].view(np.int32).reshape(-1, 8)
    d["small_meta"], d["small_params"] = rec["meta"], rec["params"].view(np.uint8).reshape(-1)
    w, h = 1920, 1080
    frames = R.synthetic_sequence(w, h, 2)
    rec = R.record_inter_searches(frames, w, h, FRONT_OPTS, max_records=60000)
    assert rec["skipped"] == 0 and len(rec["pus"]) == 42900 and (rec["pic"][0] == frames[1, :h]).all()
    delta = rec["ref"][0].astype(np.int16) - frames[0, :h].astype(np.int16)
    assert np.abs(delta).max() < 128
    d["hd_ref_delta"] = delta.astype(np.int8)
    d["hd_pus"] = rec["pus"].view(np.uint8).reshape(-1, 64)
    d["hd_results"] = rec["results"].view(np.int32).reshape(-1, 8)
    d["hd_meta"], d["hd_params"] = rec["meta"], rec["params"].view(np.uint8).reshape(-1)
    np.savez_compressed(os.path.join(OUT, "fronts.npz"), **d)


def deblock():
    """kvz_filter_deblock_lcu over every LCU (oracle/ref_harness.c: ref_deblock_frame) on three fabricated frames"""
    from patterns import deblock_case, deblock_params
    d = {}
    cfgs = [dict(w=192, h=128, qp=34), dict(w=136, h=72, qp=40, beta=2, tc=-1, per_cu_qp=1, slice_is_b=1), dict(w=128, h=64, qp=28, chroma=0)]
    for i, c in enumerate(cfgs):
        c = dict(c)
        w, h = c.pop("w"), c.pop("h")
        prm = deblock_params(**c)
        y, u, v, cus = deblock_case(w, h, SEED + 20 + i, slice_is_b=int(prm["slice_is_b"][0]), qp=int(prm["qp"][0]))
        oy, ou, ov = R.deblock_frame(y, u, v, cus, prm)
        d["y%d" % i], d["u%d" % i], d["v%d" % i] = y, u, v
        d["cus%d" % i] = cus.view(np.uint8).reshape(cus.shape[0], cus.shape[1], 20)
        d["params%d" % i] = prm.view(np.uint8).reshape(64)
        d["out_y%d" % i], d["out_u%d" % i], d["out_v%d" % i] = oy, ou, ov
    np.savez_compressed(os.path.join(OUT, "deblock.npz"), **d)


if __name__ == "__main__":
    if not R.available():
        sys.exit("oracle/_ref/libkvzref.so missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(OUT, exist_ok=True)
    groups = dict(picture=picture, dct=dct, quant=quant, ipol=ipol, intra=intra, intra_ref=intra_ref, inter_cand=inter_cand, mv_cand=mv_cand, bipred=bipred, recorded_cand=recorded_cand, sao=sao, me=me, deblock=deblock, fronts=fronts)
    for name in (sys.argv[1:] or list(groups)):          # python oracle/gen_golden.py [group ...]
        groups[name]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
