#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) of ONE command into a per-kernel counter table.

  python3 tools/pmc_kernels.py OUT.txt KERNEL_SUBSTR[,SUBSTR...] PASS_DIR [PASS_DIR ...]

For every kernel whose name contains one of the substrings: the mean of each counter over its dispatches, plus
derived figures -- vector / LDS / scalar instructions per wave, the share of wave-cycles spent issuing / waiting,
LDS bank-conflict share, and HBM bytes per dispatch (FETCH_SIZE, WRITE_SIZE in KiB; FETCH_SIZE doubled on gfx950 as
MI355X_MICROARCH.md prescribes).  SQ_* cycle counters are in quad-cycles (same guide)."""
import collections
import csv
import glob
import os
import sys


def main():
    out, subs, dirs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if not any(s in k for s in subs):
                    continue
                short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[short] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"])
    lines = []
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        g = meta[k]
        lines.append("%s  grid %s wg %s lds %s vgpr %s agpr %s sgpr %s  (%d dispatches)" % (k, g[0], g[1], g[2], g[3], g[4], g[5],
                                                                                          max(len(v) for v in acc[k].values())))
        for n in sorted(c):
            lines.append("    %-28s %16.0f" % (n, c[n]))
        w = c.get("SQ_WAVES")
        if w:
            for n, lab in (("SQ_INSTS_VALU", "vector instr / wave"), ("SQ_INSTS_SALU", "scalar instr / wave"), ("SQ_INSTS_LDS", "LDS instr / wave"),
                           ("SQ_INSTS_VMEM_RD", "vmem loads / wave"), ("SQ_INSTS_VMEM_WR", "vmem stores / wave"), ("SQ_INSTS_MFMA", "MFMA instr / wave"),
                           ("SQ_INSTS_VALU_MFMA_I8", "i8 MFMA instr / wave")):
                if n in c:
                    lines.append("    -> %-26s %12.1f" % (lab, c[n] / w))
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n, lab in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_ACTIVE_INST_VALU", "issuing VALU"), ("SQ_ACTIVE_INST_LDS", "issuing LDS"),
                           ("SQ_WAIT_ANY", "parked (s_waitcnt / barrier)"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_WAIT_INST_LDS", "issue-stalled on LDS")):
                if n in c:
                    lines.append("    -> %-32s %5.1f %% of wave-cycles" % (lab, 100.0 * c[n] / wc))
            if w and "SQ_BUSY_CYCLES" in c:
                lines.append("    -> mean resident waves (wave-cycles / busy-cycles, all SEs) %.1f" % (wc / c["SQ_BUSY_CYCLES"]))
        if c.get("SQ_LDS_IDX_ACTIVE"):
            lines.append("    -> LDS bank conflicts %.1f %% of LDS-array cycles" % (100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]))
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rd, wr = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0, c.get("WRITE_SIZE", 0.0) * 1024.0
            lines.append("    -> HBM bytes per dispatch: read %.0f (FETCH_SIZE x2) + write %.0f = %.0f" % (rd, wr, rd + wr))
        lines.append("")
    open(out, "w").write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
