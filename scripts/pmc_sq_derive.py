"""Derived figures from a pmc_sq_summary.py listing + the kernel-stats CSV of the same command:
    python scripts/pmc_sq_derive.py <summary.txt> <kernel_stats.csv>
per kernel instantiation: MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), effective clock = GRBM_GUI_ACTIVE / 8 /
mean launch duration (the duration comes from the separate kernel-trace run), share of resident wave cycles issuing / waiting, VALU per MFMA."""
import csv, sys
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Name"].split("::")[-1].split("(")[0]] = float(r["AverageNs"]) / 1e3
cur, vals = None, {}
def flush():
    if cur and "GRBM_GUI_ACTIVE" in vals and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        g = vals["GRBM_GUI_ACTIVE"] / 8
        us = dur.get(cur)
        line = f"   -> MFMA pipe busy {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * g) * 100:.1f} %"
        if us:
            line += f"; mean launch {us:.1f} us -> effective clock {g / us / 1e3:.2f} GHz"
        if vals.get("SQ_WAVE_CYCLES"):
            line += f"; of the resident wave cycles: issuing {vals['SQ_ACTIVE_INST_ANY'] / vals['SQ_WAVE_CYCLES'] * 100:.0f} %, waiting on a counter or barrier {vals['SQ_WAIT_ANY'] / vals['SQ_WAVE_CYCLES'] * 100:.0f} %"
        if vals.get("SQ_INSTS_MFMA"):
            line += f"; VALU per MFMA instruction {vals['SQ_INSTS_VALU'] / vals['SQ_INSTS_MFMA']:.1f}"
        print(line)
for ln in open(sys.argv[1]):
    if not ln.startswith(" "):
        flush()
        cur, vals = ln.strip(), {}
        print(cur)
    else:
        print(ln.rstrip())
        p = ln.split()
        vals[p[0]] = float(p[1])
flush()
