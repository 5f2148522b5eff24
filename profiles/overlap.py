#!/usr/bin/env python3
"""How much the kernels of the contexts overlap in the whole-genome bench: from a rocprofv3 --kernel-trace csv, the time covered by at least one kernel,
by at least two, ... and the sum of kernel durations (run on the GPU box: profiles/collect_overlap.sh)."""
import csv, glob, sys, collections
src = sys.argv[1]
f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []; dur = collections.defaultdict(float); per_q = collections.defaultdict(float)
rows = list(csv.DictReader(open(f)))
names = ("k_extract_phase", "k_edges", "k_graph_obs", "k_node_scatter", "k_read_correction", "k_scan_spec", "k_mark_nodes")
lib = [r for r in rows if any(n in r["Kernel_Name"] for n in ("k_",)) and "k_fill" not in r["Kernel_Name"] and "k_cigar" not in r["Kernel_Name"] and "k_aln" not in r["Kernel_Name"] and "haplotag" not in r["Kernel_Name"]]
# timed region = the last 5/6 of each group's launches is not separable here; report over all phase-kernel activity
for r in lib:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1)); dur[r["Kernel_Name"].split("(")[0]] += e - s; per_q[r.get("Queue_Id", "?")] += e - s
ev.sort()
cover = collections.defaultdict(int); depth = 0; last = ev[0][0]
for t, d in ev:
    cover[depth] += t - last; last = t; depth += d
tot = sum(v for k, v in cover.items() if k > 0)
print("phase kernels:", len(lib), " sum of durations %.1f ms" % (sum(dur.values()) / 1e6), " covered by >=1 kernel %.1f ms" % (tot / 1e6))
for k in sorted(cover):
    if k > 0: print("  exactly %d kernels in flight: %.1f ms" % (k, cover[k] / 1e6))
print("  average kernels in flight while any runs: %.2f" % (sum(dur.values()) / tot))
print("queues:", {k: round(v / 1e6, 1) for k, v in per_q.items()})
for k, v in sorted(dur.items(), key=lambda x: -x[1])[:8]: print("  %-28s %.1f ms" % (k[:28], v / 1e6))
