#!/usr/bin/env python3
"""One 1080p frame's PUs of all four square sizes in ONE kvz_hip_search_pu_batch call (no size hint: three kernels scan the
whole list) against four hinted calls; development probe."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from kvazaar_amd import _lib
from bench_all import timed
dev = torch.device("cuda", 0); L = _lib.init(0); st = L.kvz_hip_stream_create()
g = torch.Generator(device=dev); g.manual_seed(1)
W, H = 1920, 1080
pic = torch.randint(0, 256, (H, W), dtype=torch.uint8, device=dev, generator=g)
ref = torch.roll(pic, shifts=(1, 2), dims=(0, 1)).contiguous()
lists = {}
for n in (8, 16, 32, 64):
    xy = [(x, y) for y in range(0, H - n + 1, n) for x in range(0, W - n + 1, n)]
    p = np.zeros((len(xy), 16), dtype=np.int32)
    p[:, 0] = [q[0] for q in xy]; p[:, 1] = [q[1] for q in xy]; p[:, 2] = n; p[:, 3] = n
    lists[n] = p
allp = np.concatenate([lists[n] for n in (64, 32, 16, 8)])
def run(pus, cls):
    d = torch.from_numpy(pus).to(dev); out = torch.empty((len(pus), 8), dtype=torch.int32, device=dev)
    prm = np.zeros(24, dtype=np.int32); prm[:8] = (20, 1, -1, 4, 0, 0, 1, 1); prm[10] = cls
    torch.cuda.synchronize()
    return min(timed(L, st, lambda: _lib.check(L.kvz_hip_search_pu_batch(pic.data_ptr(), W, W, H, ref.data_ptr(), W, W, H, d.data_ptr(), len(pus),
                                                                        prm.ctypes.data, out.data_ptr(), st), "x")) for _ in range(3)) * 1e3
tot = 0.0
for n, cls in ((8, 1), (16, 1), (32, 2), (64, 4)):
    us = run(lists[n], cls); tot += us
    print("%2dx%-2d hinted  %7.1f us" % (n, n, us))
print("sum of the four hinted calls %7.1f us" % tot)
print("one mixed call, no hint      %7.1f us   (%d PUs)" % (run(allp, 0), len(allp)))
print("one mixed call, hint 7       %7.1f us" % run(allp, 7))
