#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, as
MI355X_MICROARCH.md prescribes: they do not fit one TCC pass) of bench.py into
profiles/pmc_traffic.json: HBM bytes per launch of each hot kernel.

gfx950 corrections (MI355X_MICROARCH.md, HBM): the counters are in KiB; FETCH_SIZE
reports exactly half the bytes of a wide coalesced (16 B/lane) streaming read, so it
is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores (the 4-byte cost
stores of the SAD/SATD kernels are < 4 % of their traffic and uncalibrated).

The result is tagged with the commit, the date and a digest of the kernel sources it was captured on; bench.py quotes it as
`roofline.traffic` only while the kernel sources are unchanged.

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [commit]"""
import csv
import glob
import json
import os
import sys

NAMES = {"sad_nxn_kernel<8": "sad_8x8", "satd8_kernel": "satd_8x8", "dct32_mfma_kernel<32, false": "dct_32x32"}


def collect(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = next((v for p, v in NAMES.items() if p in row["Kernel_Name"]), None)
            if k:
                out.setdefault(k, []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def main():
    fd, wd, outp = sys.argv[1:4]
    fetch, nf = collect(fd, "FETCH_SIZE")
    write, nw = collect(wd, "WRITE_SIZE")
    res, detail = {}, {}
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * fetch.get(k, 0.0) * 1024.0
        wr = write.get(k, 0.0) * 1024.0
        res[k] = round(rd + wr)
        detail[k] = {"read_bytes_corrected": round(rd), "write_bytes": round(wr), "FETCH_SIZE_KiB_raw": fetch.get(k),
                     "WRITE_SIZE_KiB_raw": write.get(k), "dispatches": [nf.get(k, 0), nw.get(k, 0)]}
    res["_detail"] = detail
    import datetime
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_sources_digest
    res["_captured"] = {"commit": sys.argv[4] if len(sys.argv) > 4 else "unknown", "date": datetime.date.today().isoformat(),
                        "kernel_sources_sha1_16": kernel_sources_digest()}
    res["_method"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py; KiB*1024; FETCH_SIZE doubled (gfx950)"
    json.dump(res, open(outp, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
