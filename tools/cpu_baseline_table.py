#!/usr/bin/env python3
"""The reference's CPU strategies beside the GPU numbers: generic and avx2, one thread and all host threads,
L2-resident and frame-streaming working sets, for sad_8x8 / satd_8x8 / dct_32x32 in Mblocks/s.  Uses the compiled reference
(oracle/_ref); states the core count and CPU model.  python3 tools/cpu_baseline_table.py [--seconds 1.0]"""
import argparse
import ctypes as C
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import ref_lib as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.0)
    a = ap.parse_args()
    L = R.lib()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads_all = min(cores, 16)
    model = "?"
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    print("host: %s, %d logical CPUs available, all-threads column uses %d" % (model, cores, threads_all))
    g = np.random.default_rng(12345)
    u8p, i16p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16)
    sets = {"L2 (256 blocks)": (256, 64), "streaming (4 x 1080p)": (129600, 7920)}
    print("%-10s %-9s %-22s %12s %12s" % ("function", "strategy", "working set", "1 thread", "%d threads" % threads_all))
    for wname, (n8, n32) in sets.items():
        bufs = []
        for i in range(threads_all):
            x8 = R._aligned(g.integers(0, 256, n8 * 64, dtype=np.uint8))
            y8 = R._aligned(g.integers(0, 256, n8 * 64, dtype=np.uint8))
            x16 = R._aligned(g.integers(-255, 256, n32 * 1024).astype(np.int16))
            y16 = R._aligned(np.zeros(n32 * 1024, np.int16))
            bufs.append((x8, y8, x16, y16))
        for strat in ("generic", "avx2"):
            if not R.has_strategy("satd_8x8", strat):
                continue
            for fn in ("sad_8x8", "satd_8x8", "dct_32x32"):
                def run(i):
                    b = bufs[i]
                    if fn == "dct_32x32":
                        return L.ref_bench_transform(fn.encode(), strat.encode(), 32, b[2].ctypes.data_as(i16p), b[3].ctypes.data_as(i16p), n32, a.seconds)
                    return L.ref_bench_cost_nxn(fn.encode(), strat.encode(), 8, b[0].ctypes.data_as(u8p), b[1].ctypes.data_as(u8p), n8, a.seconds, None)
                one = run(0)
                out = [0.0] * threads_all
                ts = [threading.Thread(target=lambda i=i: out.__setitem__(i, run(i))) for i in range(threads_all)]
                [t.start() for t in ts]; [t.join() for t in ts]
                print("%-10s %-9s %-22s %12.1f %12.1f" % (fn, strat, wname, one / 1e6, sum(out) / 1e6))


if __name__ == "__main__":
    main()
