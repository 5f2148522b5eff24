#!/bin/bash
# L2 hit / miss and fabric-request counters of the bench command (one pass, never combined with a trace): bash profiles/collect_cache.sh [workload]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-chr1_50x}
OUT=$ROOT/gpurun_out/cache_$WL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/p1" -o c -- python3 "$ROOT/bench.py" --workload $WL --parity none --ctx-per-gpu 1 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/p1.json" 2> "$OUT/p1.err" || echo "pass failed"
cd "$ROOT" && python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | TCC_REQ | TCC_HIT | TCC_MISS | hit rate |\n|---|---|---|---|---|")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("TCC_REQ_sum", [0]))):
    m = lambda n: sum(c.get(n, [0])) / max(1, len(c.get(n, [0])))
    h, mi = m("TCC_HIT_sum"), m("TCC_MISS_sum")
    if m("TCC_REQ_sum") > 1e5:
        print(f"| `{k}` | {m('TCC_REQ_sum'):.3g} | {h:.3g} | {mi:.3g} | {h / max(1.0, h + mi):.2f} |")
PY
