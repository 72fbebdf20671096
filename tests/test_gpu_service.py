"""GPU: the search service (include/kvz_hip.h "search service", kvazaar_amd/csrc/serve.hip).

1. Requests posted from many host threads -- one PU with all its reference pictures each -- against the oracle's
   search run picture after picture under the running cost, i.e. the loop of search_pu_inter
   (search_inter.c:1502-1507 with :1239-1252).
2. The compiled reference ENCODER with its own thread pool (WPP, frames in flight under --owf), every 2Nx2N inter
   search of every worker answered by the service: the bitstream must be the untouched encoder's, byte for byte
   (needs oracle/_ref/libkvzref.so, which travels with gpurun)."""
import concurrent.futures
import os
import time

import numpy as np
import pytest

import oracle_lib as O
import ref_lib as R
from patterns import ME_PARAMS, ME_PU, ME_REQUEST, me_frames, me_params, me_pus_in_tile, me_random_pus

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so")
MAX_INT = 2147483647


@pytest.fixture(scope="module")
def api():
    from kvazaar_amd import api as a, _lib
    _lib.init(0)
    return a


def _sequential_loop(pic, refs, pus_per_ref, prm, start=MAX_INT):
    """the oracle's search of every PU, picture after picture under the running cost -> [n_pus, n_refs] ME_RESULT records as int32"""
    n = len(pus_per_ref[0])
    running = np.full(n, start, dtype=np.uint32)
    out = np.zeros((n, len(refs), 8), dtype=np.int32)
    for r, ref in enumerate(refs):
        res = O.search_pu_batch(pic, ref, pus_per_ref[r], prm, cost_to_beat=running)
        out[:, r] = np.asarray(res).view(np.int32).reshape(n, 8)
        running = np.minimum(running, np.asarray(res["cost"], dtype=np.uint32))
    return out


SERVICE_CASES = [
    dict(),                                                     # hexbs, fme 4, early termination
    dict(fme_level=2, early_termination=2, lambda_cost=37),
    dict(algorithm=1, fme_level=0),                             # dia, no fractional stage: both outcomes coincide
    dict(algorithm=2, early_termination=0, lambda_cost=9),      # tz
    dict(wpp_owf=1, ref_delay_px=10, lambda_cost=25),           # the availability rule of --owf with WPP
    dict(mv_constraint=4, tile=(64, 0, 128, 128)),
    dict(algorithm=3, search_range=8, fme_level=1),
    # the exhaustive search on the whole workgroup (full_search_wg): overlapping windows, several staging chunks, the tile and
    # availability rules inside the windows, the largest range (one window of a 64x64 PU fills the staging area)
    dict(algorithm=3, search_range=16, fme_level=4, lambda_cost=30),
    dict(algorithm=3, search_range=33, fme_level=0, wpp_owf=1, ref_delay_px=10, lambda_cost=14),
    dict(algorithm=3, search_range=20, fme_level=2, mv_constraint=4, tile=(64, 0, 128, 128)),
    dict(algorithm=3, search_range=64, fme_level=1, lambda_cost=51),
    dict(algorithm=3, search_range=16, fme_level=3, lambda_cost=22, tune=(b"full_qsad", 0)),
]


@pytest.mark.parametrize("workers", [0, 40], ids=["launches", "resident_workers"])
@pytest.mark.parametrize("case", range(len(SERVICE_CASES)))
def test_service_requests_from_many_threads_equal_the_sequential_loop(api, case, workers):
    """workers > 0: the same requests through the ring and the resident workgroups (tuning "service_workers", read when the service is
    created) instead of a launch per batch; short linger and life so that workers come and go while the test runs"""
    w, h, n_refs = 192, 128, 4
    spec = dict(SERVICE_CASES[case])
    tune = spec.pop("tune", None)
    from kvazaar_amd import _lib as _l
    for key, value in ((b"service_workers", workers), (b"service_linger_us", 60 if case % 2 else 2000), (b"service_life_ms", 1 if case % 3 == 0 else 20)):
        _l.check(_l.load().kvz_hip_set_tuning(key, value), "tuning")             # workers = 0: a launch per batch
    prm = me_params(**spec)
    planes = [me_frames(w, h, 300 + 7 * case + k, motion) for k, motion in enumerate(((3, -2), (-5, 4), (0, 0), (9, 7)))]
    pic = planes[0][0]
    refs = [p[1] for p in planes]
    base = me_pus_in_tile(me_random_pus(w, h, 90, 4100 + case, hint=(-10, 8)), prm)
    g = np.random.default_rng(77 + case)
    pus_per_ref = []
    for r in range(n_refs):
        q = base.copy()
        if r:                                                   # what differs per picture: AMVP pair, start vector, same_ref
            q["mv_cand"] = g.integers(-40, 41, q["mv_cand"].shape)
            q["extra_mv"] = g.integers(-24, 25, q["extra_mv"].shape)
            q["merge"]["same_ref"] = g.integers(0, 2, q["merge"]["same_ref"].shape)
        pus_per_ref.append(q)
    want = _sequential_loop(pic, refs, pus_per_ref, prm)
    # a second start value: something the first pictures cannot beat for many PUs
    low = int(np.median(want[:, 0, 2].astype(np.uint32)))
    want_low = _sequential_loop(pic, refs, pus_per_ref, prm, start=low)

    from kvazaar_amd import _lib
    svc = api.MeService(w, h, max_pictures=6, max_threads=32)
    try:
        if tune:
            _lib.check(_lib.load().kvz_hip_set_tuning(tune[0], tune[1]), "tuning")
        svc.put_plane(0, pic)
        for r in range(n_refs):                                 # rectangles: the way a picture under reconstruction arrives
            svc.put_rect(1 + r, refs[r], 0, 0, w, 64)
            svc.put_rect(1 + r, refs[r], 0, 64, 64, h - 64)
            svc.put_rect(1 + r, refs[r], 64, 64, w - 64, h - 64)

        def one(job):
            i, start = job
            nr = 1 + (i % n_refs) if start != MAX_INT else n_refs        # requests of 1..4 pictures
            req = np.zeros(1, dtype=ME_REQUEST)
            req["pic_slot"], req["n_refs"], req["cost_to_beat"] = 0, nr, start
            req["ref_slot"][0, :n_refs] = 1 + np.arange(n_refs)
            req["params"] = prm[0]
            for r in range(nr):
                req["pu"][0, r] = pus_per_ref[r][i]
            return i, start, svc.search(req)

        jobs = [(i, MAX_INT) for i in range(len(base))] + [(i, low) for i in range(len(base))]
        with concurrent.futures.ThreadPoolExecutor(max_workers=16) as ex:
            for i, start, got in ex.map(one, jobs):
                ref_tab = want if start == MAX_INT else want_low
                np.testing.assert_array_equal(got, ref_tab[i, :len(got)], err_msg="case %d PU %d start %d" % (case, i, start))
        st = svc.stats()
        assert st["requests"] == len(jobs) and st["units"] >= st["requests"] and st["batches"] <= st["requests"]
        print("case %d (%s): %d requests (%d units) in %d batches / %d launches, largest batch %d units, mean wait %.1f us"
              % (case, "workers" if workers else "launches", st["requests"], st["units"], st["batches"], st["launches"], st["max_batch_units"], st["wait_ns"] / 1e3 / st["requests"]))
    finally:
        if tune:
            _lib.load().kvz_hip_set_tuning(tune[0], -1)
        for key in (b"service_workers", b"service_linger_us", b"service_life_ms"):
            _lib.load().kvz_hip_set_tuning(key, -1)
        svc.close()


@pytest.mark.parametrize("algorithm", [0, 3], ids=["hexbs", "full8"])
@pytest.mark.parametrize("workers", [0, 64, -64], ids=["launches", "resident_workers", "resident_workers_no_push"])
def test_pictures_replaced_between_requests_are_seen(api, workers, algorithm):
    """The pictures in the slots -- the one being coded and the reference -- are overwritten between requests, dozens of times, while the
    same workgroups stay on the device: a search posted after put_rect returned must read the new pixels (nothing stale in a CU's vector
    cache, in the scalar cache the exhaustive search reads the current block through, or in an XCD's L2), whole pictures and rectangles
    alike.  The same PUs every time, so the same addresses are read over and over."""
    from kvazaar_amd import _lib
    for key, value in ((b"service_workers", abs(workers)), (b"service_linger_us", 5000), (b"service_push", 0 if workers < 0 else 1)):
        _lib.check(_lib.load().kvz_hip_set_tuning(key, value), "tuning")
    w, h = 256, 192
    prm = me_params(lambda_cost=21, algorithm=algorithm, search_range=8)
    pics = [me_frames(w, h, 700 + k, motion) for k, motion in enumerate(((2, 1), (-4, 3), (6, -5)))]
    srcs = [pics[0][0], pics[1][0]]
    refs = [p[1] for p in pics]
    mixed = refs[0].copy()                                      # half one reference, half another (the lower part arrives as a rectangle)
    mixed[96:] = refs[1][96:]
    refs.append(mixed)
    pus = me_random_pus(w, h, 20, 5151, hint=(8, 4))
    big = np.full(len(pus), MAX_INT, np.uint32)
    want = {(a, b): np.asarray(O.search_pu_batch(srcs[a], refs[b], pus, prm, cost_to_beat=big)).view(np.int32).reshape(len(pus), 8)
            for a in range(2) for b in range(4)}
    svc = api.MeService(w, h, max_pictures=3, max_threads=4)
    try:
        req = np.zeros(1, dtype=ME_REQUEST)
        req["pic_slot"], req["n_refs"], req["cost_to_beat"] = 0, 1, MAX_INT
        req["ref_slot"][0, 0] = 1
        req["params"] = prm[0]
        for it in range(32):
            a, b = (it // 2) % 2, (it * 3 + it // 5) % 4
            svc.put_plane(0, srcs[a])
            if b < 3:
                svc.put_plane(1, refs[b])
            else:
                svc.put_plane(1, refs[0])
                svc.put_rect(1, refs[1], 0, 96, w, h - 96)
            for i in range(len(pus)):
                req["pu"][0, 0] = pus[i]
                np.testing.assert_array_equal(svc.search(req)[0], want[(a, b)][i], err_msg="iteration %d (source %d, reference %d) PU %d" % (it, a, b, i))
    finally:
        for key in (b"service_workers", b"service_linger_us", b"service_push"):
            _lib.load().kvz_hip_set_tuning(key, -1)
        svc.close()


@pytest.mark.parametrize("workers", [0, 32], ids=["launches", "resident_workers"])
def test_exhaustive_search_of_amp_and_smp_shapes(api, workers):
    """the service's exhaustive search on the PU shapes of --smp / --amp (widths that are 4 mod 8: the any-width row loop, 4- and 12-pixel
    rows) and on 48-wide ones, against the oracle"""
    from kvazaar_amd import _lib
    _lib.check(_lib.load().kvz_hip_set_tuning(b"service_workers", workers), "tuning")
    w, h = 192, 128
    shapes = ((12, 16), (16, 12), (4, 8), (8, 4), (16, 4), (4, 16), (48, 64), (64, 48), (24, 32), (32, 24), (64, 16), (16, 64))
    pic, ref = me_frames(w, h, 911, (-3, 2))
    svc = api.MeService(w, h, max_pictures=2, max_threads=4)
    try:
        svc.put_plane(0, pic)
        svc.put_plane(1, ref)
        for rng, fme in ((8, 4), (19, 1)):
            prm = me_params(algorithm=3, search_range=rng, fme_level=fme, lambda_cost=23)
            pus = me_random_pus(w, h, 48, 7300 + rng, hint=(-12, 8), sizes=shapes)
            want = np.asarray(O.search_pu_batch(pic, ref, pus, prm, cost_to_beat=np.full(len(pus), MAX_INT, np.uint32))).view(np.int32).reshape(len(pus), 8)
            req = np.zeros(1, dtype=ME_REQUEST)
            req["pic_slot"], req["n_refs"], req["cost_to_beat"] = 0, 1, MAX_INT
            req["ref_slot"][0, 0] = 1
            req["params"] = prm[0]
            for i in range(len(pus)):
                req["pu"][0, 0] = pus[i]
                np.testing.assert_array_equal(svc.search(req)[0], want[i], err_msg="range %d PU %d (%dx%d)" % (rng, i, pus[i]["width"], pus[i]["height"]))
    finally:
        _lib.load().kvz_hip_set_tuning(b"service_workers", -1)
        svc.close()


def test_ring_tickets_across_two_to_the_32(api):
    """the ring's ticket counters are 64-bit, the slots' sequence words 32-bit and never 0 ("free"): requests posted while the ticket count
    passes 2^32 (the service started just below it) are answered like any other"""
    from kvazaar_amd import _lib
    _lib.check(_lib.load().kvz_hip_set_tuning(b"service_ticket_base_k", 4194303), "tuning")      # 2^32 - 1024
    try:
        w, h = 192, 128
        prm = me_params(lambda_cost=19)
        pic, ref = me_frames(w, h, 931, (2, -3))
        pus = me_random_pus(w, h, 40, 7700, hint=(8, -12))
        want = np.asarray(O.search_pu_batch(pic, ref, pus, prm, cost_to_beat=np.full(len(pus), MAX_INT, np.uint32))).view(np.int32).reshape(len(pus), 8)
        svc = api.MeService(w, h, max_pictures=2, max_threads=8)
        try:
            svc.put_plane(0, pic)
            svc.put_plane(1, ref)

            def one(i):
                req = np.zeros(1, dtype=ME_REQUEST)
                req["pic_slot"], req["n_refs"], req["cost_to_beat"] = 0, 2, MAX_INT
                req["ref_slot"][0, :2] = 1
                req["params"] = prm[0]
                req["pu"][0, 0] = pus[i % len(pus)]
                req["pu"][0, 1] = pus[i % len(pus)]
                got = svc.search(req)
                np.testing.assert_array_equal(got[0], want[i % len(pus)], err_msg="request %d" % i)
                return 1

            with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
                assert sum(ex.map(one, range(1200))) == 1200            # 2400 units: tickets 2^32 - 1024 .. 2^32 + 1376
            assert svc.stats()["units"] == 2400
        finally:
            svc.close()
    finally:
        _lib.load().kvz_hip_set_tuning(b"service_ticket_base_k", -1)


def test_two_services_at_once_and_the_fallback_for_many_threads(api):
    """Two services alive on one device, each with its own resident workers, used alternately from several threads; a third one sized
    for more calling threads than the ring serves (> 128) answers through a launch per batch by itself."""
    prm = me_params(lambda_cost=17)
    sizes = ((192, 128), (256, 192))
    data = []
    for k, (w, h) in enumerate(sizes):
        pic, ref = me_frames(w, h, 810 + k, (3 - 5 * k, 2))
        pus = me_random_pus(w, h, 16, 6000 + k, hint=(4, -4))
        want = np.asarray(O.search_pu_batch(pic, ref, pus, prm, cost_to_beat=np.full(len(pus), MAX_INT, np.uint32))).view(np.int32).reshape(len(pus), 8)
        data.append((pic, ref, pus, want))
    svcs = [api.MeService(w, h, max_pictures=2, max_threads=8) for (w, h) in sizes]
    big = api.MeService(192, 128, max_pictures=2, max_threads=200)
    try:
        for svc, (pic, ref, _, _) in zip(svcs + [big], data + [data[0]]):
            svc.put_plane(0, pic)
            svc.put_plane(1, ref)

        def one(job):
            which, i = job
            svc = (svcs + [big])[which]
            _, _, pus, want = data[which if which < 2 else 0]
            req = np.zeros(1, dtype=ME_REQUEST)
            req["pic_slot"], req["n_refs"], req["cost_to_beat"] = 0, 1, MAX_INT
            req["ref_slot"][0, 0] = 1
            req["params"] = prm[0]
            req["pu"][0, 0] = pus[i]
            np.testing.assert_array_equal(svc.search(req)[0], want[i], err_msg="service %d PU %d" % (which, i))
            return which

        jobs = [(which, i) for rep in range(6) for i in range(16) for which in range(3)]
        with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
            assert sum(1 for _ in ex.map(one, jobs)) == len(jobs)
        st = [s.stats() for s in svcs + [big]]
        assert st[0]["max_batch_units"] == 0 and st[1]["max_batch_units"] == 0          # resident workers: no batches
        assert st[2]["max_batch_units"] >= 1 and st[2]["launches"] >= 1                   # the fallback: a launch per batch
    finally:
        for s in svcs + [big]:
            s.close()


def test_service_refuses_bad_requests(api):
    from kvazaar_amd._lib import KvzHipError
    svc = api.MeService(64, 64, max_pictures=2, max_threads=2)
    try:
        plane = np.zeros((64, 64), np.uint8)
        svc.put_plane(0, plane)
        svc.put_plane(1, plane)
        req = np.zeros(1, dtype=ME_REQUEST)
        req["pic_slot"], req["n_refs"], req["cost_to_beat"] = 0, 1, MAX_INT
        req["ref_slot"][0, 0] = 1
        req["params"] = me_params()[0]
        req["pu"][0, 0]["width"] = req["pu"][0, 0]["height"] = 16
        assert svc.search(req).shape == (1, 8)
        for field, value in (("n_refs", 0), ("n_refs", 17), ("pic_slot", 2)):
            bad = req.copy()
            bad[field] = value
            with pytest.raises(KvzHipError):
                svc.search(bad)
        bad = req.copy()
        bad["ref_slot"][0, 0] = 5
        with pytest.raises(KvzHipError):
            svc.search(bad)
        bad = req.copy()
        bad["params"]["mv_rdo"] = 1
        with pytest.raises(KvzHipError):
            svc.search(bad)
        bad = req.copy()
        bad["params"]["algorithm"], bad["params"]["search_range"] = 3, 8
        assert svc.search(bad).shape == (1, 8)
        bad["pu"][0, 0]["x"] = 2                                # the exhaustive search reads the block with dword-aligned scalar loads
        with pytest.raises(KvzHipError):
            svc.search(bad)
        bad = req.copy()
        bad["pu"][0, 0]["x"] = 56                               # the PU leaves the picture: flagged by the kernel
        with pytest.raises(KvzHipError):
            svc.search(bad)
        with pytest.raises(KvzHipError):
            svc.put_rect(0, plane, 32, 32, 40, 8)
    finally:
        svc.close()


ENCODE_CASES = [
    # WPP workers and frames in flight: every worker posts its searches, reference pictures arrive as staircases
    (320, 192, 8, "preset=medium,qp=30,threads=6,owf=2", 8),
    (320, 192, 10, "preset=medium,qp=27,threads=8", 8),                                 # --owf auto, B pyramid, four references
    (256, 256, 6, "preset=medium,ref=2,gop=0,qp=33,threads=4,owf=3,period=0", 8),       # P frames only, long chains of pictures in flight
    (320, 192, 6, "preset=medium,me=tz,qp=29,threads=5,owf=1,sao=off", 8),              # deblocking delay only
    (320, 192, 6, "preset=medium,me=full8,qp=31,threads=4,owf=2,deblock=0,sao=off", 16),  # no filter delay; small PUs stay with the reference
    (320, 192, 5, "preset=medium,qp=32,threads=4,owf=0", 8),                            # complete reference pictures only
    (320, 192, 5, "preset=medium,qp=32,threads=0", 8),                                  # no thread pool at all
]


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("w,h,n,opts,min_size", ENCODE_CASES)
def test_reference_encoder_with_worker_threads_served_by_the_service(w, h, n, opts, min_size):
    frames = R.synthetic_sequence(w, h, n, seed=11)
    t0 = time.perf_counter()
    plain, _ = R.encode(frames, w, h, opts)
    t1 = time.perf_counter()
    served, c = R.encode_with_service(frames, w, h, opts, LIB, max_threads=32, min_size=min_size)
    t2 = time.perf_counter()
    print("%dx%d x %d (%s): %d searches served (%d left to the reference) in %d batches / %d launches, largest batch %d units, "
          "%d rectangles uploaded; %.2f s untouched, %.2f s served"
          % (w, h, n, opts, c["served"], c["passed_on"], c["batches"], c["launches"], c["max_batch_units"], c["upload_rects"], t1 - t0, t2 - t1))
    assert c["failed"] == 0 and c["served"] > 100
    assert served == plain, "bitstreams differ (%d vs %d bytes)" % (len(served), len(plain))


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
def test_shadow_mode_agrees_search_by_search():
    """every served search repeated by the reference's own search on the worker that posted it, from the same encoder state"""
    w, h, n, opts = 320, 192, 6, "preset=medium,qp=28,threads=6,owf=2"
    frames = R.synthetic_sequence(w, h, n, seed=3)
    plain, _ = R.encode(frames, w, h, opts)
    served, c = R.encode_with_service(frames, w, h, opts, LIB, max_threads=32, min_size=8, shadow=True)
    assert c["failed"] == 0 and c["served"] > 100 and c["shadow_mismatch"] == 0
    assert served == plain


def test_sad_tables_equal_image_calc_sad(api):
    """kvz_hip_me_service_sad_tables against the oracle's kvz_image_calc_sad for every PU of a CTU, every vector of the window and two
    pictures, at an inner CTU, at the picture's corner CTUs (edge replication) and at a ragged last row (PUs outside the picture: 0xFFFFFFFF)"""
    w, h, rng_ = 200, 152, 5                                    # 4 x 3 CTUs, the last column 8 and the last row 24 pixels
    planes = [me_frames(w, h, 900 + k, motion)[1] for k, motion in enumerate(((0, 0), (2, -1), (-3, 2)))]
    svc = api.MeService(w, h, max_pictures=4, max_threads=4)
    try:
        for k, p in enumerate(planes):
            svc.put_plane(k, p)
        for (cx, cy) in ((64, 64), (0, 0), (192, 128), (128, 128), (192, 0)):
            tab = svc.sad_tables(0, [1, 2], cx, cy, rng_)
            assert tab.shape == (2, 2 * rng_ + 1, 2 * rng_ + 1, 85)
            for i, ref in enumerate((planes[1], planes[2])):
                for k, (n, bx, by) in enumerate([(64, 0, 0)] + [(32, x, y) for y in (0, 32) for x in (0, 32)] +
                                                [(16, x, y) for y in range(0, 64, 16) for x in range(0, 64, 16)] +
                                                [(8, x, y) for y in range(0, 64, 8) for x in range(0, 64, 8)]):
                    px, py = cx + bx, cy + by
                    inside = px + n <= w and py + n <= h
                    for dy in (-rng_, -1, 0, 2, rng_):
                        for dx in (-rng_, 0, 1, rng_):
                            got = int(tab[i, dy + rng_, dx + rng_, k])
                            want = O.image_calc("sad", planes[0], ref, px, py, px + dx, py + dy, n, n) if inside else 0xFFFFFFFF
                            assert got == want, ((cx, cy), i, n, bx, by, dx, dy, got, want)
        st = svc.stats()
        assert st["tables"] == 10 and st["table_bytes"] == 5 * 2 * (2 * rng_ + 1) ** 2 * 85 * 4
    finally:
        svc.close()


TABLE_ENCODES = [
    (320, 192, 6, "preset=medium,qp=30,threads=6,owf=2", 16),                            # hexbs: most vectors inside +-16
    (320, 192, 5, "preset=medium,me=full8,qp=31,threads=4,owf=1", 8),                    # exhaustive +-8: the first window is the table
    (256, 256, 5, "preset=medium,me=tz,ref=2,gop=0,qp=33,threads=3,owf=0,period=0", 12),
]


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("w,h,n,opts,table_range", TABLE_ENCODES)
def test_reference_encoder_with_its_sads_answered_from_tables(w, h, n, opts, table_range):
    """the reference's own searches, their kvz_image_calc_sad calls answered from kvz_hip_me_service_sad_tables fetched once per CTU
    by the worker that starts it: same SADs, hence same decisions and the same bitstream"""
    frames = R.synthetic_sequence(w, h, n, seed=21)
    plain, _ = R.encode(frames, w, h, opts)
    served, c = R.encode_with_service(frames, w, h, opts, LIB, max_threads=32, table_range=table_range)
    print("%dx%d x %d (%s), +-%d: %d SADs from tables, %d outside the range, %d other calls; %d tables (%.1f MB) in %.1f ms of worker time"
          % (w, h, n, opts, table_range, c["table_hits"], c["table_range_misses"], c["table_other_calls"], c["tables"], c["table_bytes"] / 1e6, c["table_ns"] / 1e6))
    assert c["failed"] == 0 and c["table_hits"] > 1000 and c["served"] == 0
    assert served == plain, "bitstreams differ (%d vs %d bytes)" % (len(served), len(plain))
