#!/usr/bin/env python3
"""One 1080p frame's worth of the batched entries, launched the three ways a host can launch them:

  eager    every entry enqueued on one stream, frame after frame
  graph    the frame's launch sequence captured once (kvz_hip_graph_begin/_end), replayed per frame
  graph4   the same, with the four independent stages (motion search, intra rough search, TU
           reconstruction + cost, SAO statistics) on forked streams = parallel branches of the graph
  graph7   the four motion-search launches (one per PU size class) on a branch each as well
  eager4/7 the forked streams without the graph

The per-frame batch sizes are the real ones (every PU / TU / LCU of one 1920x1080 frame, each size
once), so launch overhead, partial waves and tails count the way they do in an encoder -- unlike
tools/bench_all.py, whose batches are sized to hide them.  It is a measurement of the kernels on a
frame-sized workload, not of an encoder: the mode decision around them is not here.

  python tools/frame_pipeline.py [--frames 200]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from kvazaar_amd import _lib  # noqa: E402
from kvazaar_amd._lib import QuantParams  # noqa: E402

W, H = 1920, 1080


def build_stages(L, dev):
    """-> {stage: [(name, units, launch(stream))]}, plus the tensors kept alive"""
    g = torch.Generator(device=dev); g.manual_seed(7)
    keep = []
    cur = torch.randint(0, 256, (H, W), dtype=torch.uint8, device=dev, generator=g)
    ref = torch.roll(cur, shifts=(1, 2), dims=(0, 1)).contiguous()
    noise = torch.randint(-6, 7, (H, W), dtype=torch.int16, device=dev, generator=g)
    pred = (cur.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    keep += [cur, ref, pred]
    stages = {"me": [], "intra": [], "tu": [], "sao": []}      # "sao": the in-loop filter stage, deblocking + SAO statistics

    # the CU arrays the candidate derivation reads: this frame's (as if every neighbour were decided) and the collocated one's
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from patterns import inter_cu_map, inter_params
    ip = np.ascontiguousarray(inter_params(W, H, poc=8, ref_pocs=(7,), l0=(0,), col_ref_pocs=(6,), col_l0=(0,)))
    cu_now = torch.from_numpy(inter_cu_map(W, H, 1)[0].view(np.uint8).copy()).to(dev)
    cu_col = torch.from_numpy(inter_cu_map(W, H, 2)[0].view(np.uint8).copy()).to(dev)
    keep += [ip, cu_now, cu_col]
    me_prm = np.zeros(24, dtype=np.int32); me_prm[:8] = (20, 1, -1, 4, 0, 0, 1, 1)
    for n in (8, 16, 32, 64):
        xy = [(x, y) for y in range(0, H - n + 1, n) for x in range(0, W - n + 1, n)]
        pus = np.zeros((len(xy), 16), dtype=np.int32)
        pus[:, 0] = [p[0] for p in xy]; pus[:, 1] = [p[1] for p in xy]; pus[:, 2] = n; pus[:, 3] = n
        pus_d = torch.from_numpy(pus).to(dev)
        res_d = torch.empty((len(xy), 8), dtype=torch.int32, device=dev)
        prm = me_prm.copy(); prm[10] = 1 if n <= 16 else (2 if n <= 32 else 4)
        keep += [pus_d, res_d, prm]
        stages["me"].append(("inter_candidates_%dx%d" % (n, n), len(xy),
                             lambda s, pus_d=pus_d, k=len(xy): L.kvz_hip_inter_candidates_batch(
                                 cu_now.data_ptr(), cu_col.data_ptr(), cu_col.data_ptr(), ip.ctypes.data, pus_d.data_ptr(), k, None, s)))
        stages["me"].append(("search_pu_%dx%d" % (n, n), len(xy),
                             lambda s, pus_d=pus_d, res_d=res_d, k=len(xy), prm=prm: L.kvz_hip_search_pu_batch(
                                 cur.data_ptr(), W, W, H, ref.data_ptr(), W, W, H, pus_d.data_ptr(), k, prm.ctypes.data, res_d.data_ptr(), s)))

    flat_cur, flat_pred = cur.reshape(-1), pred.reshape(-1)
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        cnt = (W // n) * (H // n)
        refs_d = torch.empty(cnt * 130, dtype=torch.uint8, device=dev)
        costs_d = torch.empty(cnt * 35, dtype=torch.int32, device=dev)
        pos_d = torch.tensor([(x, y) for y in range(0, H - n + 1, n) for x in range(0, W - n + 1, n)], dtype=torch.int32, device=dev)
        keep += [refs_d, costs_d, pos_d]
        stages["intra"].append(("intra_build_reference_%dx%d" % (n, n), cnt,
                                lambda s, lg=lg, cnt=cnt, refs_d=refs_d, pos_d=pos_d: L.kvz_hip_intra_build_reference_batch(
                                    lg, 0, pred.data_ptr(), W, W, H, pos_d.data_ptr(), cnt, refs_d.data_ptr(), s)))
        stages["intra"].append(("intra_rough_%dx%d" % (n, n), cnt,
                                lambda s, lg=lg, cnt=cnt, refs_d=refs_d, costs_d=costs_d: L.kvz_hip_intra_rough_batch(
                                    lg, 3, refs_d.data_ptr(), flat_cur.data_ptr(), cnt, costs_d.data_ptr(), None, s)))

    qp = QuantParams(); qp.qp = 27
    keep.append(qp)
    for n in (4, 8, 16, 32):
        cnt = (W // n) * (H // n)
        rec = torch.empty(cnt * n * n, dtype=torch.uint8, device=dev)
        coef = torch.empty(cnt * n * n, dtype=torch.int16, device=dev)
        has = torch.empty(cnt, dtype=torch.int32, device=dev)
        ssd = torch.empty(cnt, dtype=torch.int32, device=dev)
        cas = torch.empty(cnt, dtype=torch.int32, device=dev)
        keep += [rec, coef, has, ssd, cas]
        stages["tu"].append(("quantize_residual_cost_%dx%d" % (n, n), cnt,
                             lambda s, n=n, cnt=cnt, rec=rec, coef=coef, has=has, ssd=ssd, cas=cas: L.kvz_hip_quantize_residual_cost_batch(
                                 C.byref(qp), 0, n, 0, 0, 0, flat_cur.data_ptr(), flat_pred.data_ptr(), rec.data_ptr(), coef.data_ptr(),
                                 has.data_ptr(), ssd.data_ptr(), cas.data_ptr(), cnt, s)))

    # SAO: every luma LCU (64x64) and both chroma planes (32x32 per LCU) of the frame
    for label, n, cnt in (("luma", 64, (W // 64) * (H // 64)), ("chroma", 32, 2 * (W // 64) * (H // 64))):
        edge = torch.empty(cnt * 40, dtype=torch.int32, device=dev)
        band = torch.empty(cnt * 64, dtype=torch.int32, device=dev)
        keep += [edge, band]
        stages["sao"].append(("sao_edge_stats_%s" % label, cnt,
                              lambda s, n=n, cnt=cnt, edge=edge: L.kvz_hip_sao_edge_stats_batch(
                                  flat_cur.data_ptr(), flat_pred.data_ptr(), n, n, cnt, edge.data_ptr(), s)))
        stages["sao"].append(("sao_band_stats_%s" % label, cnt,
                              lambda s, n=n, cnt=cnt, band=band: L.kvz_hip_sao_band_stats_batch(
                                  flat_cur.data_ptr(), flat_pred.data_ptr(), n, n, cnt, band.data_ptr(), s)))
    # deblocking of the reconstructed frame (1088 coded rows), both passes
    from patterns import deblock_case, deblock_params
    DH = 1088
    ty, tu, tv, tcus = deblock_case(256, 128, 11, qp=36)
    reps = (DH // 128 + 1, W // 256 + 1)
    dby = torch.from_numpy(np.tile(ty, reps)[:DH, :W].copy()).to(dev)
    dbu = torch.from_numpy(np.tile(tu, reps)[:DH // 2, :W // 2].copy()).to(dev)
    dbv = torch.from_numpy(np.tile(tv, reps)[:DH // 2, :W // 2].copy()).to(dev)
    dbc = torch.from_numpy(np.tile(tcus, reps)[:DH // 4, :W // 4].copy().view(np.uint8)).to(dev)
    dbp = deblock_params(qp=36)
    keep += [dby, dbu, dbv, dbc, dbp]
    stages["sao"].insert(0, ("deblock_frame", 1, lambda s: L.kvz_hip_deblock_frame(
        dby.data_ptr(), W, dbu.data_ptr(), dbv.data_ptr(), W // 2, W, DH, dbc.data_ptr(), dbp.ctypes.data, s)))
    return stages, keep


def chk(rc, what):
    _lib.check(rc, what)


def enqueue_serial(stages, s):
    for st in stages.values():
        for name, _, fn in st:
            chk(fn(s), name)


def by_launch(stages):
    """finer branches: every motion-search launch on its own (they are latency bound and, at 480..8040 PUs, none of the
    larger sizes fills the chip), the other three stages as before"""
    out = {}
    for name, units, fn in stages["me"]:                 # a size's candidate derivation and its search stay on one branch, in order
        out.setdefault(name.rsplit("_", 1)[1], []).append((name, units, fn))
    out.update({k: v for k, v in stages.items() if k != "me"})
    return out


def enqueue_forked(L, stages, s, side, events):
    """stage k on side[k]; fork from s, join back into s"""
    fork = events[0]
    chk(L.kvz_hip_event_record(fork, s), "fork")
    for k, st in enumerate(stages.values()):
        chk(L.kvz_hip_stream_wait_event(side[k], fork), "fork wait")
        for name, _, fn in st:
            chk(fn(side[k]), name)
        chk(L.kvz_hip_event_record(events[1 + k], side[k]), "join record")
        chk(L.kvz_hip_stream_wait_event(s, events[1 + k]), "join wait")


def run(L, s, frames, per_frame):
    """wall-clock ms per frame, frames enqueued back to back, one sync at the end"""
    for _ in range(5):
        per_frame()
    chk(L.kvz_hip_stream_sync(s), "sync")
    t0 = time.perf_counter()
    for _ in range(frames):
        per_frame()
    t_enq = time.perf_counter() - t0
    chk(L.kvz_hip_stream_sync(s), "sync")
    return (time.perf_counter() - t0) * 1e3 / frames, t_enq * 1e3 / frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--tune", default="", help="key=value[,key=value...] passed to kvz_hip_set_tuning")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.init(0)
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        chk(L.kvz_hip_set_tuning(k.encode(), int(v)), "set_tuning " + k)
    s = L.kvz_hip_stream_create()
    side = [L.kvz_hip_stream_create() for _ in range(7)]
    events = [L.kvz_hip_event_create() for _ in range(8)]
    stages, keep = build_stages(L, dev)
    torch.cuda.synchronize()
    n_launch = sum(len(v) for v in stages.values())

    # per-entry device time at frame-sized batches (HIP events around 20 back-to-back launches)
    print("%-34s %9s %10s" % ("entry (one 1080p frame)", "units", "us/launch"))
    total = 0.0
    for st in stages.values():
        for name, units, fn in st:
            e0, e1 = L.kvz_hip_event_create(), L.kvz_hip_event_create()
            for _ in range(3):
                chk(fn(s), name)
            L.kvz_hip_event_record(e0, s)
            for _ in range(20):
                chk(fn(s), name)
            L.kvz_hip_event_record(e1, s)
            ms = C.c_float()
            chk(L.kvz_hip_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
            L.kvz_hip_event_destroy(e0); L.kvz_hip_event_destroy(e1)
            total += ms.value / 20
            print("%-34s %9d %10.1f" % (name, units, ms.value / 20 * 1e3))
    print("%-34s %9s %10.1f   (%d launches)" % ("sum of the entries", "", total * 1e3, n_launch))

    results = {}
    results["eager"] = run(L, s, args.frames, lambda: enqueue_serial(stages, s))
    results["eager4"] = run(L, s, args.frames, lambda: enqueue_forked(L, stages, s, side, events))

    graph = C.c_void_p()
    chk(L.kvz_hip_graph_begin(s), "graph_begin")
    enqueue_serial(stages, s)
    chk(L.kvz_hip_graph_end(s, C.byref(graph)), "graph_end")
    results["graph"] = run(L, s, args.frames, lambda: chk(L.kvz_hip_graph_launch(graph, s), "graph_launch"))
    L.kvz_hip_graph_destroy(graph)

    graph4 = C.c_void_p()
    chk(L.kvz_hip_graph_begin(s), "graph_begin")
    enqueue_forked(L, stages, s, side, events)
    chk(L.kvz_hip_graph_end(s, C.byref(graph4)), "graph_end")
    results["graph4"] = run(L, s, args.frames, lambda: chk(L.kvz_hip_graph_launch(graph4, s), "graph_launch"))
    L.kvz_hip_graph_destroy(graph4)

    fine = by_launch(stages)
    results["eager7"] = run(L, s, args.frames, lambda: enqueue_forked(L, fine, s, side, events))
    graph7 = C.c_void_p()
    chk(L.kvz_hip_graph_begin(s), "graph_begin")
    enqueue_forked(L, fine, s, side, events)
    chk(L.kvz_hip_graph_end(s, C.byref(graph7)), "graph_end")
    results["graph7"] = run(L, s, args.frames, lambda: chk(L.kvz_hip_graph_launch(graph7, s), "graph_launch"))
    L.kvz_hip_graph_destroy(graph7)

    print()
    print("%-8s %14s %18s %10s" % ("mode", "ms per frame", "host enqueue ms", "frames/s"))
    for k, (ms, enq) in results.items():
        print("%-8s %14.3f %18.3f %10.0f" % (k, ms, enq, 1e3 / ms))


if __name__ == "__main__":
    main()
