#!/usr/bin/env python3
"""Latency of the per-call strategy drop-in (host buffers in, result on return) measured through the
compiled reference's registry -- the PCIe-inclusive figure DESIGN.md quotes.  Development tool."""
import os
import sys
import time
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ref_lib as R

L = R.lib()
L.ref_register_hip.restype = C.c_int
L.ref_register_hip.argtypes = [C.c_char_p]
assert L.ref_register_hip(os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so").encode()) > 0
g = np.random.default_rng(0)
for n in (8, 32, 64):
    a = R._aligned(g.integers(0, 256, n * n, dtype=np.uint8)); b = R._aligned(g.integers(0, 256, n * n, dtype=np.uint8))
    for kind in ("sad", "satd"):
        t = ("%s_%dx%d" % (kind, n, n)).encode()
        for name in (b"hip", b"avx2"):
            for _ in range(50):
                L.ref_cost_nxn(t, name, a.ctypes.data, b.ctypes.data)
            k = 2000
            t0 = time.perf_counter()
            for _ in range(k):
                L.ref_cost_nxn(t, name, a.ctypes.data, b.ctypes.data)
            dt = (time.perf_counter() - t0) / k
            print("%-12s %-5s %8.2f us/call" % (t.decode(), name.decode(), dt * 1e6))
x = R._aligned(g.integers(-255, 256, 1024).astype(np.int16)); y = R._aligned(np.zeros(1024, np.int16))
i16p = C.POINTER(C.c_int16)
for name in (b"hip", b"avx2"):
    k = 2000
    t0 = time.perf_counter()
    for _ in range(k):
        L.ref_transform(b"dct_32x32", name, x.ctypes.data_as(i16p), y.ctypes.data_as(i16p))
    print("%-12s %-5s %8.2f us/call" % ("dct_32x32", name.decode(), (time.perf_counter() - t0) / k * 1e6))
