#!/usr/bin/env python3
"""Where a genome pass with C contexts side by side spends its time, from a rocprofv3 --kernel-trace CSV of bench.py (wgs_50x).
    python3 profiles/overlap_analysis.py <kernel_trace.csv> > profiles/r03_overlap_wgs_50x.md
For every instant inside the windows in which more than one queue is busy: is the GPU idle, running only small kernels, or running k_edges /
k_extract_phase (the two kernels that fill the chip by themselves)?  And: how long do those two take when they share the GPU with each other."""
import csv
import sys
from collections import defaultdict

BIG = ("k_edges", "k_extract_phase")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0][:40]


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "0"))))
    rows.sort()
    t0 = rows[0][0]
    # windows = maximal runs of kernels with gaps < 2 ms in which >= 3 queues take part (the timed regions and warm-ups of a group)
    wins = []; cur = [rows[0]]
    for r in rows[1:]:
        if r[0] - max(x[1] for x in cur[-64:]) > 2_000_000:
            wins.append(cur); cur = [r]
        else:
            cur.append(r)
    wins.append(cur)
    multi = [w for w in wins if len({x[3] for x in w}) >= 3 and any(x[2] in BIG for x in w)]
    print(f"# {path.split('/')[-1]}: {len(rows)} kernel launches, {len(wins)} busy windows, {len(multi)} with three or more queues\n")
    tot = defaultdict(float); span = 0.0
    dur_by = defaultdict(lambda: defaultdict(list))
    for w in multi:
        ev = []
        for s, e, n, q in w:
            ev.append((s, 1, n in BIG)); ev.append((e, -1, n in BIG))
        ev.sort()
        nb = ns = 0; last = ev[0][0]
        for t, d, big in ev:
            dt = t - last
            key = "idle" if nb + ns == 0 else ("big>=2" if nb >= 2 else ("big=1" if nb == 1 else "small only"))
            tot[key] += dt; span += dt; last = t
            if big: nb += d
            else: ns += d
        bigs = [(s, e, n) for s, e, n, q in w if n in BIG]
        for s, e, n in bigs:
            ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, n2 in bigs if (s2, e2) != (s, e))
            cls = "alone" if ov < 0.1 * (e - s) else ("half" if ov < 0.6 * (e - s) else "shared")
            dur_by[n][cls].append((e - s) / 1e3)
    print("| state of the GPU inside the multi-queue windows | time ms | share |\n|---|---|---|")
    for k in ("big>=2", "big=1", "small only", "idle"):
        print(f"| {k} | {tot[k] / 1e6:.2f} | {tot[k] / span:.3f} |")
    print(f"| total | {span / 1e6:.2f} | 1 |\n")
    print("| kernel | other big kernel running for <10 % of its life: n, mean us | 10-60 %: n, mean us | >60 %: n, mean us |\n|---|---|---|---|")
    for n in BIG:
        c = dur_by[n]
        f = lambda x: f"{len(x)}, {sum(x) / len(x):.0f}" if x else "0, -"
        print(f"| `{n}` | {f(c['alone'])} | {f(c['half'])} | {f(c['shared'])} |")
    # per kernel name: total busy time inside the windows (sum of durations; overlapping kernels both count)
    busy = defaultdict(float); cnt = defaultdict(int)
    for w in multi:
        for s, e, n, q in w:
            busy[n] += e - s; cnt[n] += 1
    print("\n| kernel | launches | summed duration ms | over window time |\n|---|---|---|---|")
    for n, v in sorted(busy.items(), key=lambda kv: -kv[1])[:16]:
        print(f"| `{n}` | {cnt[n]} | {v / 1e6:.2f} | {v / span:.3f} |")


if __name__ == "__main__":
    main(sys.argv[1])
