#!/usr/bin/env python3
"""rocprofv3 outputs of profiles/collect.sh -> the committed summaries:
   <dst>/kernel_stats.csv|.md   per-kernel calls / average duration (from --kernel-trace --stats)
   <dst>/traffic.json           HBM bytes per launch of the main kernels: FETCH_SIZE (KB, doubled: gfx950 tallies 128-B
                                requests at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE (KB), averaged over launches.
                                Keyed by bench.py stage name so that bench.py can quote it in roofline.traffic."""
import csv
import glob
import json
import os
import sys

STAGE_OF = {"k_extract_phase": "extract", "k_edges": "edges", "k_scan_spec": "vote_scan", "k_read_correction": "read_correction",
            "k_merge_multi": "merge_rows", "k_node_scatter": "node_lists", "k_haplotag_score": "haplotag_extract", "k_haplotag_stream": "haplotag_extract", "k_graph_rows": "graph_rows"}


def find(d, pat):
    m = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return m[0] if m else None


def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0].strip()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    wl = sys.argv[3] if len(sys.argv) > 3 else "chr1_50x"
    os.makedirs(dst, exist_ok=True)
    stats = find(os.path.join(src, "stats"), "*kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(dst, "kernel_stats.csv"), "w") as f:
        f.write(open(stats).read())
    with open(os.path.join(dst, "kernel_stats.md"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --parity none --ctx-per-gpu 1 --steps 10 --warmup 2 --no-cpu-baseline (MI355X)" % wl + "\n\n| kernel | calls | avg us | total % |\n|---|---|---|---|\n")
        for r in rows[:40]:
            f.write("| `%s` | %s | %.1f | %s |\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    per = {}
    for what in ("fetch", "write"):
        p = find(os.path.join(src, what), "*counter_collection.csv")
        if not p:
            continue
        acc = {}
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if k in STAGE_OF and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                a = acc.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (tot, n) in acc.items():
            per.setdefault(k, {})[what + "_kb"] = tot / n
            per[k]["launches_" + what] = n
    out = {wl: {}, "_detail": {}, "_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 5 --warmup 1), "
           "average per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads) - an upper estimate for "
           "gather-heavy kernels, raw values kept in _detail"}
    for k, v in per.items():
        fk, wk = v.get("fetch_kb", 0.0), v.get("write_kb", 0.0)
        v["hbm_bytes_raw"] = int((fk + wk) * 1024); v["hbm_bytes_fetch_x2"] = int((2 * fk + wk) * 1024)
        out["_detail"][k] = v
        out[wl][STAGE_OF[k]] = v["hbm_bytes_fetch_x2"]
    json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps(out[wl]))


if __name__ == "__main__":
    main()
