#!/usr/bin/env python3
"""Round-trip time of ONE search-service request (kvz_hip_me_service_search) by PU size and search algorithm, one caller, nothing else
in flight: what a worker of the encoder waits for when its request shares no launch.  Synthetic 1080p planes with global motion.

    python3 tools/service_latency.py [--refs 4] [--n 200] [--algos hexbs,full8,full16,full32,full64] [--tune full_qsad=0]
prints one JSON line per (algorithm, PU size): mean / median / p95 microseconds per request.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--refs", type=int, default=4)
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--algos", default="hexbs,full8,full16,full32,full64")
    ap.add_argument("--sizes", default="8,16,32,64")
    ap.add_argument("--tune", default="")
    ap.add_argument("--plain", action="store_true", help="PUs without merge candidates and start vector: the exhaustive search has one window")
    a = ap.parse_args()
    from kvazaar_amd import api, _lib
    from patterns import ME_REQUEST, me_frames, me_params, me_random_pus
    _lib.init(0)
    L = _lib.load()
    for kv in [t for t in a.tune.split(",") if t]:
        k, v = kv.split("=")
        _lib.check(L.kvz_hip_set_tuning(k.encode(), int(v)), "tuning")
    w, h = 1920, 1080
    planes = [me_frames(w, h, 900 + k, motion) for k, motion in enumerate(((3, -2), (-5, 4), (0, 0), (9, 7), (1, 1), (-7, 2), (4, 4), (2, -9))[:a.refs])]
    svc = api.MeService(w, h, max_pictures=a.refs + 2, max_threads=4)
    try:
        svc.put_plane(0, planes[0][0])
        for r in range(a.refs):
            svc.put_plane(1 + r, planes[r][1])
        for algo in a.algos.split(","):
            if algo.startswith("full"):
                prm = me_params(algorithm=3, search_range=int(algo[4:]), fme_level=4, lambda_cost=30)
            else:
                prm = me_params(algorithm={"hexbs": 0, "dia": 1, "tz": 2}[algo], fme_level=4, lambda_cost=30)
            for size in [int(s) for s in a.sizes.split(",")]:
                pus = me_random_pus(w, h, a.n, 31 + size, hint=(-10, 8), sizes=((size, size),))
                if a.plain:
                    pus["num_merge_cand"] = 0
                    pus["extra_mv"] = 0
                req = np.zeros(1, dtype=ME_REQUEST)
                req["pic_slot"], req["n_refs"], req["cost_to_beat"] = 0, a.refs, 2147483647
                req["ref_slot"][0, :a.refs] = 1 + np.arange(a.refs)
                req["params"] = prm[0]
                times = []
                for i in range(a.n):
                    for r in range(a.refs):
                        req["pu"][0, r] = pus[i]
                    t0 = time.perf_counter_ns()
                    svc.search(req)
                    times.append(time.perf_counter_ns() - t0)
                t = np.asarray(times[a.n // 10:], dtype=np.float64) / 1e3          # the first tenth warms up
                print(json.dumps(dict(algorithm=algo, pu=size, refs=a.refs, tune=a.tune, plain=bool(a.plain), mean_us=round(float(t.mean()), 1),
                                      median_us=round(float(np.median(t)), 1), p95_us=round(float(np.percentile(t, 95)), 1))), flush=True)
    finally:
        svc.close()


if __name__ == "__main__":
    main()
