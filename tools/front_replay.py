#!/usr/bin/env python3
"""Search-only timing of kvz_hip_search_pu_batch on searches RECORDED from real encodes of the reference encoder, issued front
by front in the encoder's dependency order (tools/front_replay.c does the timed loop in C; this script prepares its input
from tests/golden/fronts.npz -- recorded by oracle/gen_golden.py, group `fronts` -- builds the C program and prints its JSON line).  Not an encoder: labelled "search only, fronts" wherever it is quoted.

  python3 tools/front_replay.py [--repeats N] [--sessions 2,4,8,16] [--merged 4,16,64,256]"""
import argparse
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build():
    exe = os.path.join(ROOT, "tools", "front_replay")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "front_replay.c"),
                           "-o", exe, "-L" + os.path.join(ROOT, "kvazaar_amd"), "-lkvzhip", "-Wl,-rpath," + os.path.join(ROOT, "kvazaar_amd")])
    return exe


def write_case(path, pic, ref, pus, res, meta, prm):
    from patterns import front_groups
    groups = front_groups(meta)
    order = np.concatenate(groups)
    off = np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.int32)
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", pic.shape[1], pic.shape[0], len(order), len(groups)))
        f.write(np.ascontiguousarray(pic).tobytes()); f.write(np.ascontiguousarray(ref).tobytes())
        f.write(np.ascontiguousarray(prm).tobytes())
        f.write(np.ascontiguousarray(pus[order]).tobytes()); f.write(np.ascontiguousarray(res[order]).tobytes())
        f.write(off.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--keep-case", default="", help="also leave the first frame's input file here (to run tools/front_replay under rocprofv3)")
    ap.add_argument("--merged", default="", help="comma list, e.g. 4,16,64: also replay with that many sessions merged into one launch per front")
    ap.add_argument("--sessions", default="", help="comma list, e.g. 2,4,8,16: also replay with that many host threads at once")
    args = ap.parse_args()
    from patterns import fronts_fixture
    exe = build()
    cases = []
    d = np.load(os.path.join(ROOT, "tests", "golden", "fronts.npz"))
    cases = fronts_fixture(d, "hd")
    for i, c in enumerate(cases):
        path = "/tmp/kvz_front_case_%d.bin" % i
        write_case(path, *c)
        for hint in (1, 0):
            sys.stdout.write(subprocess.check_output([exe, path, str(args.repeats), str(hint)], text=True))
        for k in [int(v) for v in args.sessions.split(",") if v]:
            sys.stdout.write(subprocess.check_output([exe, path, str(args.repeats), "1", str(k)], text=True))
            sys.stdout.flush()
        for k in [int(v) for v in args.merged.split(",") if v]:
            sys.stdout.write(subprocess.check_output([exe, path, str(args.repeats), "1", "1", str(k)], text=True))
            sys.stdout.flush()
        if i == 0 and args.keep_case:
            os.replace(path, args.keep_case)
        else:
            os.remove(path)


if __name__ == "__main__":
    main()
