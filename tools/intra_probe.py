#!/usr/bin/env python3
"""Launch only the intra rough-search kernel a few times (profiling target for rocprofv3 --pmc).
  python3 tools/intra_probe.py --log2 3 --count 262144 --iters 3"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kvazaar_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=3)
    ap.add_argument("--count", type=int, default=262144)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.init(0)
    n = 1 << a.log2
    g = torch.Generator(device=dev); g.manual_seed(3)
    refs = torch.randint(0, 256, (a.count * 130,), dtype=torch.uint8, device=dev, generator=g)
    orig = torch.randint(0, 256, (a.count * n * n,), dtype=torch.uint8, device=dev, generator=g)
    costs = torch.empty(a.count * 35, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    st = L.kvz_hip_stream_create()
    for _ in range(a.iters):
        _lib.check(L.kvz_hip_intra_rough_batch(a.log2, 3, refs.data_ptr(), orig.data_ptr(), a.count, costs.data_ptr(), None, st), "rough")
    _lib.check(L.kvz_hip_stream_sync(st), "sync")
    print("done", int(costs[:35].sum().item()))


if __name__ == "__main__":
    main()
