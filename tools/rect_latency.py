import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from kvazaar_amd import api, _lib
_lib.init(0)
L = _lib.load()
w, h = 1920, 1080
plane = np.random.default_rng(1).integers(0, 256, (h, w), dtype=np.uint8)
for workers in (64, 0):
    _lib.check(L.kvz_hip_set_tuning(b"service_workers", workers), "t")
    svc = api.MeService(w, h, max_pictures=2, max_threads=4)
    svc.put_plane(0, plane)
    # a search first so that workers are alive
    for (rw, rh) in ((64, 64), (256, 64), (1920, 64), (64, 16)):
        ts = []
        for i in range(300):
            x = (i * 64) % (w - rw + 1) // 64 * 64
            t0 = time.perf_counter_ns()
            svc.put_rect(1, plane, x, 128, rw, rh)
            ts.append(time.perf_counter_ns() - t0)
        t = np.asarray(ts[30:]) / 1e3
        print("workers", workers, "rect %dx%d" % (rw, rh), "mean %.1f us median %.1f p95 %.1f" % (t.mean(), np.median(t), np.percentile(t, 95)), flush=True)
    svc.close()
