#!/usr/bin/env python3
"""SQ counter passes of profiles/collect_issue.sh -> a table per kernel: instructions by pipe, the share of wave time spent issuing vs parked
(SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES, disjoint per MI355X_MICROARCH.md) and the VALU issue floor:
a wave64 VALU instruction holds its SIMD16 for 4 cycles, so   floor = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x clock)   is the least time the
kernel's vector instructions need when every SIMD of the chip is busy all the time."""
import csv
import glob
import os
import sys
from collections import defaultdict

CLOCK_HZ = 2.4e9
SIMDS = 256 * 4


def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0].strip()


def main():
    src, dst, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
    for p in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    dur = {}
    st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        for r in csv.DictReader(open(st[0])):
            dur[short(r["Name"])] = float(r["AverageNs"]) / 1e3
    rows = []
    for k in acc:
        a = {c: acc[k][c] / max(1, cnt[k][c]) for c in acc[k]}
        if k not in dur or dur[k] < 20:
            continue
        wc = a.get("SQ_WAVE_CYCLES", 0) or 1
        floor_us = a.get("SQ_INSTS_VALU", 0) * 4 / (SIMDS * CLOCK_HZ) * 1e6
        rows.append((dur[k], k, a, floor_us, wc))
    rows.sort(reverse=True)
    with open(dst, "w") as f:
        f.write("# SQ counters per launch, `bench.py --workload %s --parity none --ctx-per-gpu 1 --steps 5 --warmup 1` (MI355X; rocprofv3 --pmc, 4 counters per pass)\n\n" % wl)
        f.write("VALU floor = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz): the time the vector instructions alone need with every SIMD busy.\n")
        f.write("issuing / parked / stalled = SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES.\n\n")
        f.write("| kernel | avg us | VALU | SALU | LDS | VMEM rd | VMEM wr | VALU floor us | floor / measured | issuing | parked (waitcnt) | issue-stalled | LDS conflict / LDS active |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for d, k, a, fl, wc in rows[:14]:
            f.write("| `%s` | %.1f | %.3g | %.3g | %.3g | %.3g | %.3g | %.1f | %.2f | %.2f | %.2f | %.2f | %.2f |\n" % (
                k, d, a.get("SQ_INSTS_VALU", 0), a.get("SQ_INSTS_SALU", 0), a.get("SQ_INSTS_LDS", 0), a.get("SQ_INSTS_VMEM_RD", 0), a.get("SQ_INSTS_VMEM_WR", 0),
                fl, fl / d, a.get("SQ_ACTIVE_INST_ANY", 0) / wc, a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc,
                a.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, a.get("SQ_ACTIVE_INST_LDS", 0))))
    print(open(dst).read())


if __name__ == "__main__":
    main()
