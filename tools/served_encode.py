#!/usr/bin/env python3
"""The reference ENCODER (oracle/_ref/libkvzref.so: Kvazaar compiled from /root/reference by oracle/Makefile) timed untouched
and with its inter searches answered by the product's search service (kvz_hip_me_service_*, kvazaar_amd/csrc/serve.hip)
from all of its worker threads, same options, same frames; the two bitstreams must be identical.

    python3 tools/served_encode.py --size 1920x1080 --frames 24 --opts preset=medium,qp=32 --threads 16,64 --min-size 8,16,32

Prints one JSON object per (threads, min_size) and a final summary line.  Measurement infrastructure: it drives the
reference host, it is not part of the product path.  BASELINE.json's "encoder fps 1080p medium" is what `untouched` is."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(w, h, n, opts, threads, min_sizes, seed=5, repeat=1, shadow=False, probe=False, table_ranges=(), upload_only=False):
    import ref_lib as R
    lib = os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so")
    frames = R.synthetic_sequence(w, h, n, seed=seed)
    rows = []
    for t in threads:
        o = "%s,threads=%d" % (opts, t)
        best_plain, plain = None, None
        for _ in range(repeat):
            t0 = time.perf_counter()
            plain, _ = R.encode(frames, w, h, o)
            dt = time.perf_counter() - t0
            best_plain = dt if best_plain is None else min(best_plain, dt)
        if probe:
            _, c = R.encode_with_service(frames, w, h, o, lib, max_threads=max(64, t + 8), probe=True)
            print(json.dumps(dict(size="%dx%d" % (w, h), opts=o, cpu_search_us=c["probe_us_per_search"], cpu_searches=c["probe_searches"])), flush=True)
        if upload_only:      # what the picture uploads alone cost the encode: nothing is served
            t0 = time.perf_counter()
            served, c = R.encode_with_service(frames, w, h, o, lib, max_threads=max(64, t + 8), min_size=64, upload_only=True)
            dt = time.perf_counter() - t0
            print(json.dumps(dict(size="%dx%d" % (w, h), frames=n, opts=o, threads=t, mode="uploads_only", fps_untouched=round(n / best_plain, 3),
                                  fps_with_uploads=round(n / c["encode_s"], 3), service_setup_s=round(dt - c["encode_s"], 3), identical_bitstream=bool(served == plain), upload_rects=c["upload_rects"],
                                  upload_MB=round(c["rect_bytes"] / 1e6, 1), worker_ms_uploading=round(c["upload_ns"] / 1e6, 1))), flush=True)
        for tr in table_ranges:
            t0 = time.perf_counter()
            served, c = R.encode_with_service(frames, w, h, o, lib, max_threads=max(64, t + 8), table_range=tr)
            dt = c["encode_s"]                 # the encode alone; creating the service (device planes, page-locked areas) is a per-session cost
            looked = c["table_hits"] + c["table_range_misses"]
            row = dict(size="%dx%d" % (w, h), frames=n, opts=o, threads=t, mode="sad_tables", table_range=tr, fps_untouched=round(n / best_plain, 3),
                       fps_with_tables=round(n / dt, 3), identical_bitstream=bool(served == plain), failed=c["failed"],
                       sad_calls_answered_from_tables=c["table_hits"], sad_calls_outside_the_range=c["table_range_misses"], other_sad_calls=c["table_other_calls"],
                       hit_rate=round(c["table_hits"] / max(1, looked + c["table_other_calls"]), 4), tables=c["tables"],
                       table_MB=round(c["table_bytes"] / 1e6, 1), table_KB_per_ctu_and_picture=round(c["table_bytes"] / 1e3 / max(1, c["tables"]), 1),
                       mean_us_per_ctu_fetch=round(c["table_ns"] / 1e3 / max(1, c["tables"]) * (c["tables"] / max(1, c["tables"])), 1),
                       worker_ms_fetching_tables=round(c["table_ns"] / 1e6, 1), upload_MB=round(c["rect_bytes"] / 1e6, 1))
            print(json.dumps(row), flush=True)
            rows.append(row)
        for ms in min_sizes:
            best, c, same = None, None, True
            for _ in range(repeat):
                t0 = time.perf_counter()
                served, c = R.encode_with_service(frames, w, h, o, lib, max_threads=max(64, t + 8), min_size=ms, shadow=shadow)
                setup = time.perf_counter() - t0 - c["encode_s"]
                dt = c["encode_s"]             # the encode alone; creating the service (device planes, page-locked areas) is a per-session cost
                best = dt if best is None else min(best, dt)
                same = same and served == plain
            tune = dict(kv.split("=") for kv in os.environ.get("KVZ_HIP_TUNE", "").split(",") if "=" in kv)
            way = "launch_per_batch" if tune.get("service_workers") == "0" else "resident_workers"   # (`launches` then counts worker start-ups)
            row = dict(size="%dx%d" % (w, h), frames=n, opts=o, threads=t, min_pu_served=ms, service=way,
                       fps_untouched=round(n / best_plain, 3), fps_served=round(n / best, 3), service_setup_s=round(setup, 3),
                       identical_bitstream=bool(same), searches_served=c["served"], searches_left_to_cpu=c["passed_on"], failed=c["failed"],
                       batches=c["batches"], launches=c["launches"],
                       mean_requests_per_batch=round(c["requests"] / max(1, c["batches"]), 2),
                       mean_units_per_launch=round(c["units"] / max(1, c["launches"]), 2), max_batch_units=c["max_batch_units"],
                       mean_wait_us=round(c["wait_ns"] / 1e3 / max(1, c["requests"]), 1),
                       upload_rects=c["upload_rects"], upload_MB=round(c["rect_bytes"] / 1e6, 1),
                       worker_time_ms=dict(upload=round(c["upload_ns"] / 1e6, 1), candidates=round(c["cand_ns"] / 1e6, 1),
                                           search_wait=round(c["search_wait_ns"] / 1e6, 1)))
            if shadow:
                row["shadow_mismatch"] = c["shadow_mismatch"]
            print(json.dumps(row), flush=True)
            rows.append(row)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--opts", default="preset=medium,qp=32")
    ap.add_argument("--threads", default="16")
    ap.add_argument("--min-size", default="8")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--shadow", action="store_true")
    ap.add_argument("--tables", default="", help="comma-separated ranges: SAD-table mode runs (kvz_hip_me_service_sad_tables answering kvz_image_calc_sad)")
    ap.add_argument("--upload-only", action="store_true", help="also a run in which only the picture uploads happen (every search stays with the reference)")
    ap.add_argument("--probe", action="store_true", help="also time the reference's own inter searches per CU size (nothing served)")
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    rows = run(w, h, a.frames, a.opts, [int(v) for v in a.threads.split(",")], [int(v) for v in a.min_size.split(",") if v],
               seed=a.seed, repeat=a.repeat, shadow=a.shadow, probe=a.probe, upload_only=a.upload_only,
               table_ranges=[int(v) for v in a.tables.split(",") if v])
    ok = all(r["identical_bitstream"] and r["failed"] == 0 for r in rows)
    print(json.dumps(dict(summary="served_encode", all_identical=ok, runs=len(rows))))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
