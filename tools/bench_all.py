#!/usr/bin/env python3
"""Per-kernel throughput of every batched entry of the C ABI (HIP-event timed, inputs
resident in HBM, working sets larger than the 256 MiB Infinity Cache where the kernel
streams).  Development tool; bench.py is the contract benchmark.

  python tools/bench_all.py                       # table of all kernels
  python tools/bench_all.py --only dct --tune dct32_wgs_per_cu=3,4,5,6,8   # interleaved A/B in one process (--only a,b: several substrings)
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from kvazaar_amd import _lib  # noqa: E402
from kvazaar_amd._lib import QuantParams  # noqa: E402


def timed(L, stream, fn, iters=20, warm=3):
    e0, e1 = L.kvz_hip_event_create(), L.kvz_hip_event_create()
    for _ in range(warm):
        fn()
    L.kvz_hip_event_record(e0, stream)
    for _ in range(iters):
        fn()
    L.kvz_hip_event_record(e1, stream)
    ms = C.c_float()
    _lib.check(L.kvz_hip_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
    L.kvz_hip_event_destroy(e0); L.kvz_hip_event_destroy(e1)
    return ms.value / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--tune", default="", help="key=v1,v2,... (interleaved rounds)")
    ap.add_argument("--mb", type=int, default=512, help="bytes per operand array (MiB)")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.init(0)
    st = L.kvz_hip_stream_create()
    nbytes = args.mb << 20
    g = torch.Generator(device=dev); g.manual_seed(1)
    a8 = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev, generator=g)
    b8 = (a8.to(torch.int16) + torch.randint(-8, 9, (nbytes,), dtype=torch.int16, device=dev, generator=g)).clamp_(0, 255).to(torch.uint8)
    r16 = torch.randint(-255, 256, (nbytes // 2,), dtype=torch.int16, device=dev, generator=g)
    o16 = torch.empty_like(r16)
    o8 = torch.empty_like(a8)
    o32 = torch.empty(nbytes // 16, dtype=torch.int32, device=dev)
    has = torch.empty(nbytes // 16, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    cases = []   # (name, blocks, algorithmic bytes per block, fn)
    for n in (4, 8, 16, 32, 64):
        cnt = nbytes // (n * n)
        cases.append(("sad_%dx%d" % (n, n), cnt, 2 * n * n + 4,
                      lambda n=n, cnt=cnt: L.kvz_hip_sad_nxn_batch(n, a8.data_ptr(), b8.data_ptr(), cnt, o32.data_ptr(), st)))
        cases.append(("satd_%dx%d" % (n, n), cnt, 2 * n * n + 4,
                      lambda n=n, cnt=cnt: L.kvz_hip_satd_nxn_batch(n, a8.data_ptr(), b8.data_ptr(), cnt, o32.data_ptr(), st)))
    for n in (4, 8, 16, 32):
        cnt = (nbytes // 2) // (n * n)
        for kind, kid in (("dct", 0), ("idct", 1)):
            cases.append(("%s_%dx%d" % (kind, n, n), cnt, 4 * n * n,
                          lambda n=n, cnt=cnt, kid=kid: L.kvz_hip_transform_batch(kid, n, r16.data_ptr(), o16.data_ptr(), cnt, st)))
    qp = QuantParams(); qp.qp = 27
    for n in (4, 8, 16, 32):
        cnt = (nbytes // 2) // (n * n)
        cases.append(("quant_%dx%d" % (n, n), cnt, 4 * n * n,
                      lambda n=n, cnt=cnt: L.kvz_hip_quant_batch(C.byref(qp), r16.data_ptr(), o16.data_ptr(), n, 0, 0, cnt, st)))
        cases.append(("dequant_%dx%d" % (n, n), cnt, 4 * n * n,
                      lambda n=n, cnt=cnt: L.kvz_hip_dequant_batch(C.byref(qp), r16.data_ptr(), o16.data_ptr(), n, 0, cnt, st)))
        cnt2 = (nbytes // 4) // (n * n)
        cases.append(("quantize_residual_%dx%d" % (n, n), cnt2, 5 * n * n,
                      lambda n=n, cnt2=cnt2: L.kvz_hip_quantize_residual_batch(C.byref(qp), 0, n, 0, 0, 0, a8.data_ptr(), b8.data_ptr(),
                                                                              o8.data_ptr(), o16.data_ptr(), has.data_ptr(), cnt2, st)))

    # batched ME: 1080p CTU grid x frames, the speed_tests.c +-6 grid (25 candidates) -> 85 PU costs each
    import numpy as np
    W, H, F = 1920, 1080, 16
    picf = torch.randint(0, 256, (F * H, W), dtype=torch.uint8, device=dev, generator=g)
    reff = torch.roll(picf, shifts=(1, 2), dims=(0, 1)).contiguous()
    ctu_list = np.array([(x, f * H + y, 0, 0) for f in range(F) for y in range(0, H - 63, 64) for x in range(0, W, 64)], dtype=np.int32)
    ctus_d = torch.from_numpy(ctu_list).to(dev)
    for label, offs in (("grid25", [(dx, dy) for dy in (-6, -3, 0, 3, 6) for dx in (-6, -3, 0, 3, 6)]),
                        ("full8", [(dx, dy) for dy in range(-8, 9) for dx in range(-8, 9)])):
        mv_d = torch.tensor(offs, dtype=torch.int16, device=dev)
        out_d = torch.empty((len(ctu_list), len(offs), 85), dtype=torch.int32, device=dev)
        n_sad8 = len(ctu_list) * len(offs) * 64
        # bytes per 8x8-candidate: the HBM traffic of this kernel divided by the 8x8 SADs it produces
        bpb = (len(ctu_list) * (4096 + (64 + 16) ** 2) + out_d.numel() * 4) / n_sad8
        cases.append(("ctu_sad_%s(8x8 cands)" % label, n_sad8, bpb,
                      lambda mv_d=mv_d, out_d=out_d, offs=offs: L.kvz_hip_ctu_sad_grid_batch(
                          picf.data_ptr(), W, W, F * H, reff.data_ptr(), W, W, F * H, ctus_d.data_ptr(), len(ctu_list),
                          mv_d.data_ptr(), len(offs), out_d.data_ptr(), st)))

    # frame-level kernels on one 1080p frame pair x F frames: every 8x8 / 16x16 / 64x64 block with a small MV
    rs = np.random.default_rng(3)
    for n in (8, 16, 64):
        prs = np.array([(x, f * H + y, min(max(x + int(dx), 0), W - n), f * H + min(max(y + int(dy), 0), H - n), n, n)
                        for f in range(4) for y in range(0, H - n + 1, n) for x in range(0, W - n + 1, n)
                        for (dx, dy) in [rs.integers(-6, 7, 2)]], dtype=np.int32)
        prs_d = torch.from_numpy(prs).to(dev)
        outp = torch.empty(len(prs), dtype=torch.int32, device=dev)
        cases.append(("reg_sad_frame_%dx%d" % (n, n), len(prs), 2 * n * n + 4 + 24,
                      lambda prs_d=prs_d, outp=outp, k=len(prs): L.kvz_hip_reg_sad_batch(picf.data_ptr(), W, reff.data_ptr(), W, prs_d.data_ptr(), k, outp.data_ptr(), st)))
        cases.append(("image_satd_frame_%dx%d" % (n, n), len(prs), 2 * n * n + 4 + 24,
                      lambda prs_d=prs_d, outp=outp, k=len(prs): L.kvz_hip_image_calc_satd_batch(picf.data_ptr(), W, reff.data_ptr(), W, W, F * H, prs_d.data_ptr(), k, outp.data_ptr(), st)))
    # ipol: quarter-pel luma samples and the fused fractional search, all blocks of 4 frames
    for n in (8, 16, 32, 64):
        blks = np.array([(x, f * H + y, int(fx), int(fy), n, n) for f in range(4) for y in range(0, H - n + 1, n) for x in range(0, W - n + 1, n)
                         for (fx, fy) in [rs.integers(0, 4, 2)]], dtype=np.int32)
        blks_d = torch.from_numpy(blks).to(dev)
        offs_d = torch.arange(len(blks), dtype=torch.int64, device=dev) * (n * n)
        dst = torch.empty(len(blks) * n * n, dtype=torch.uint8, device=dev)
        cases.append(("sample_luma_%dx%d" % (n, n), len(blks), (n + 7) * (n + 7) + n * n,
                      lambda blks_d=blks_d, offs_d=offs_d, dst=dst, k=len(blks): L.kvz_hip_sample_luma_batch(
                          reff.data_ptr(), W, W, F * H, blks_d.data_ptr(), offs_d.data_ptr(), k, 0, dst.data_ptr(), st)))
        prs = np.array([(x, f * H + y, x + 1, f * H + y + 1, n, n) for f in range(4) for y in range(0, H - n + 1, n) for x in range(0, W - n + 1, n)], dtype=np.int32)
        prs_d = torch.from_numpy(prs).to(dev)
        co = torch.empty(len(prs) * 17, dtype=torch.int32, device=dev)
        be = torch.empty(len(prs) * 2, dtype=torch.int32, device=dev)
        cases.append(("search_frac_%dx%d" % (n, n), len(prs), (n + 8) * (n + 8) + n * n + 76,
                      lambda prs_d=prs_d, co=co, be=be, k=len(prs): L.kvz_hip_search_frac_batch(
                          picf.data_ptr(), W, reff.data_ptr(), W, W, F * H, prs_d.data_ptr(), k, co.data_ptr(), be.data_ptr(), st)))

    # whole-PU motion search (hexagon + fractional, with MV costs): every n x n PU of 4 frames
    me_prm = np.zeros(24, dtype=np.int32); me_prm[:8] = (20, 1, -1, 4, 0, 0, 1, 1)
    for n in (8, 16, 32, 64):
        rows = [(x, f * H + y) for f in range(4) for y in range(0, H - n + 1, n) for x in range(0, W - n + 1, n)]
        pus = np.zeros((len(rows), 16), dtype=np.int32)
        pus[:, 0] = [r[0] for r in rows]; pus[:, 1] = [r[1] for r in rows]; pus[:, 2] = n; pus[:, 3] = n
        pus_d = torch.from_numpy(pus).to(dev)
        res_d = torch.empty((len(rows), 8), dtype=torch.int32, device=dev)
        cls_prm = me_prm.copy(); cls_prm[10] = 1 if n <= 16 else (2 if n <= 32 else 4)     # size_classes hint: one launch instead of three
        cases.append(("search_pu_%dx%d" % (n, n), len(rows), 2 * n * n + 96,
                      lambda pus_d=pus_d, res_d=res_d, k=len(rows), cls_prm=cls_prm: L.kvz_hip_search_pu_batch(
                          picf.data_ptr(), W, W, F * H, reff.data_ptr(), W, W, F * H, pus_d.data_ptr(), k,
                          cls_prm.ctypes.data, res_d.data_ptr(), st)))
        if n == 16:
            for alg_name, alg in (("dia", 1), ("tz", 2)):                      # --me dia / --me tz
                alg_prm = cls_prm.copy(); alg_prm[8] = alg
                cases.append(("search_pu_16x16_%s" % alg_name, len(rows), 2 * n * n + 96,
                              lambda pus_d=pus_d, res_d=res_d, k=len(rows), alg_prm=alg_prm: L.kvz_hip_search_pu_batch(
                                  picf.data_ptr(), W, W, F * H, reff.data_ptr(), W, W, F * H, pus_d.data_ptr(), k,
                                  alg_prm.ctypes.data, res_d.data_ptr(), st)))
            full_prm = me_prm.copy(); full_prm[8] = 3; full_prm[9] = 16        # --me full16: 1089 positions per PU
            kk = len(rows) // 8
            cases.append(("search_pu_16x16_full16", kk, 2 * n * n + 96,
                          lambda pus_d=pus_d, res_d=res_d, kk=kk, full_prm=full_prm: L.kvz_hip_search_pu_batch(
                              picf.data_ptr(), W, W, F * H, reff.data_ptr(), W, W, F * H, pus_d.data_ptr(), kk,
                              full_prm.ctypes.data, res_d.data_ptr(), st)))

    # SAO statistics: every 64x64 luma LCU of 16 frames (blocks contiguous, as sao.c blits them)
    sao_cnt = min(nbytes // 4096, 510 * 16)
    sao_stats = torch.empty(sao_cnt * 64, dtype=torch.int32, device=dev)
    cases.append(("sao_edge_stats_64x64(4 classes)", sao_cnt, 2 * 4096 + 160,
                  lambda: L.kvz_hip_sao_edge_stats_batch(a8.data_ptr(), b8.data_ptr(), 64, 64, sao_cnt, sao_stats.data_ptr(), st)))
    cases.append(("sao_band_stats_64x64", sao_cnt, 2 * 4096 + 256,
                  lambda: L.kvz_hip_sao_band_stats_batch(a8.data_ptr(), b8.data_ptr(), 64, 64, sao_cnt, sao_stats.data_ptr(), st)))

    # deblocking of a whole 1080p (1920x1080 -> 1080 is not a multiple of 8 in the reference either: 1088 coded rows) frame
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from patterns import deblock_case, deblock_params
    DW, DH = 1920, 1088
    ty, tu, tv, tcus = deblock_case(256, 128, 11, qp=36)
    reps = (DH // 128 + 1, DW // 256 + 1)
    dby = torch.from_numpy(np.tile(ty, reps)[:DH, :DW].copy()).to(dev)
    dbu = torch.from_numpy(np.tile(tu, reps)[:DH // 2, :DW // 2].copy()).to(dev)
    dbv = torch.from_numpy(np.tile(tv, reps)[:DH // 2, :DW // 2].copy()).to(dev)
    dbc = torch.from_numpy(np.tile(tcus, reps)[:DH // 4, :DW // 4].copy().view(np.uint8)).to(dev)
    dbp = deblock_params(qp=36)
    # bytes per frame: luma + chroma read and written once per pass at most + the SCU map
    cases.append(("deblock_frame_1080p(frames)", 1, 2 * 2 * (DW * DH * 3 // 2) + DW * DH // 16 * 20 * 2,
                  lambda: L.kvz_hip_deblock_frame(dby.data_ptr(), DW, dbu.data_ptr(), dbv.data_ptr(), DW // 2, DW, DH, dbc.data_ptr(),
                                                  dbp.ctypes.data, st)))

    # intra rough search: all 35 modes per PU; bytes per PU = refs 130 + orig N^2 + 35 costs
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        cnt = min((nbytes // 4) // (n * n), 1 << 20)
        refs_d = torch.randint(0, 256, (cnt * 130,), dtype=torch.uint8, device=dev, generator=g)
        costs_d = torch.empty(cnt * 35, dtype=torch.int32, device=dev)
        cases.append(("intra_rough_%dx%d(PUs)" % (n, n), cnt, 130 + n * n + 140,
                      lambda lg=lg, cnt=cnt, refs_d=refs_d, costs_d=costs_d: L.kvz_hip_intra_rough_batch(
                          lg, 3, refs_d.data_ptr(), a8.data_ptr(), cnt, costs_d.data_ptr(), None, st)))

    # candidate derivation + intra reference building of one 1080p frame's 8x8 PUs (the glue kernels between dependency fronts)
    from patterns import ME_PU, inter_cu_map, inter_params
    ip = inter_params(1920, 1080, poc=8, ref_pocs=(7,), l0=(0,), col_ref_pocs=(6,), col_l0=(0,))
    cu_now, _ = inter_cu_map(1920, 1080, 1)
    cu_col, _ = inter_cu_map(1920, 1080, 2)
    grid = np.zeros(32400, dtype=ME_PU)
    grid["x"], grid["y"] = (np.arange(32400) % 240) * 8, (np.arange(32400) // 240) * 8
    grid["width"] = grid["height"] = 8
    cu_now_d = torch.from_numpy(cu_now.view(np.uint8).copy()).to(dev)
    cu_col_d = torch.from_numpy(cu_col.view(np.uint8).copy()).to(dev)
    grid_d = torch.from_numpy(grid.view(np.uint8).copy()).to(dev)
    merge_d = torch.empty(32400 * 60, dtype=torch.uint8, device=dev)
    ip_host = np.ascontiguousarray(ip)
    cases.append(("inter_candidates_8x8(PUs)", 32400, 64 + 60 + 7 * 20,
                  lambda: L.kvz_hip_inter_candidates_batch(cu_now_d.data_ptr(), cu_col_d.data_ptr(), cu_col_d.data_ptr(), ip_host.ctypes.data,
                                                           grid_d.data_ptr(), 32400, merge_d.data_ptr(), st)))
    pos = np.stack([grid["x"], grid["y"]], axis=1).astype(np.int32)
    pos_d = torch.from_numpy(pos.copy()).to(dev)
    rec_d = torch.randint(0, 256, (1080 * 1920,), dtype=torch.uint8, device=dev, generator=g)
    refs8_d = torch.empty(32400 * 130, dtype=torch.uint8, device=dev)
    cases.append(("intra_build_reference_8x8(PUs)", 32400, 130 + 33,
                  lambda: L.kvz_hip_intra_build_reference_batch(3, 0, rec_d.data_ptr(), 1920, 1920, 1080, pos_d.data_ptr(), 32400, refs8_d.data_ptr(), st)))

    tune_key, tune_vals = None, [None]
    if args.tune:
        tune_key, vals = args.tune.split("=")
        tune_vals = [int(v) for v in vals.split(",")]
    torch.cuda.synchronize()         # the operands above were written on torch's stream; the library's stream does not wait for it
    print("%-26s %10s %12s %10s %8s" % ("kernel", "tune", "Mblocks/s", "GB/s", "ms"))
    for name, blocks, bpb, fn in cases:
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        best = {}
        for _ in range(args.rounds):
            for v in tune_vals:
                if tune_key:
                    _lib.check(L.kvz_hip_set_tuning(tune_key.encode(), v), "set_tuning")
                ms = timed(L, st, lambda: _lib.check(fn(), name))
                best[v] = min(best.get(v, 1e9), ms)
        for v in tune_vals:
            ms = best[v]
            print("%-26s %10s %12.1f %10.1f %8.4f" % (name, "-" if v is None else v, blocks / ms / 1e3, blocks * bpb / ms / 1e6, ms))
    if tune_key:
        L.kvz_hip_set_tuning(tune_key.encode(), -1)


if __name__ == "__main__":
    main()
