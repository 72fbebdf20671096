#!/usr/bin/env python3
"""Per-launch timeline of the three headline kernels from a rocprofv3 --kernel-trace CSV: duration of every launch, the gap to the
previous launch on the device, and the spread over the run -- what `ms_per_step` is made of when the kernels take 0.1-0.2 ms.
usage: trace_launches.py <p_kernel_trace.csv>"""
import csv
import sys

KEYS = (("sad_nxn_kernel<8", "sad_8x8"), ("satd8_kernel", "satd_8x8"), ("dct32_mfma_kernel<32, false", "dct_32x32"))


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        name = r.get("Kernel_Name", "")
        k = next((v for p, v in KEYS if p in name), None)
        if k:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
    rows.sort()
    per = {}
    prev_end = None
    for (s, e, k) in rows:
        d = per.setdefault(k, {"dur": [], "gap": []})
        d["dur"].append((e - s) / 1e3)
        if prev_end is not None:
            d["gap"].append((s - prev_end) / 1e3)
        prev_end = e
    print("kernel        launches   duration us: mean  median     min     max    p90 | gap before launch us: mean  median     max")
    for k, d in per.items():
        du, ga = sorted(d["dur"]), sorted(d["gap"]) or [0.0]
        print("%-12s %9d %19.1f %7.1f %7.1f %7.1f %6.1f | %27.1f %7.1f %7.1f"
              % (k, len(du), sum(du) / len(du), du[len(du) // 2], du[0], du[-1], du[int(len(du) * 0.9)], sum(ga) / len(ga), ga[len(ga) // 2], ga[-1]))
    # the launches in the order they ran (warm-up first): is the spread a drift, an alternation, or scattered?
    for k, d in per.items():
        print("%-10s in order: %s" % (k, " ".join("%.0f" % v for v in d["dur"])))
    if rows:
        first, last = rows[0][0], rows[-1][1]
        busy = sum(e - s for (s, e, _) in rows)
        print("device busy with these kernels %.1f %% of the %.2f ms between the first start and the last end" % (100.0 * busy / (last - first), (last - first) / 1e6))


if __name__ == "__main__":
    main()
