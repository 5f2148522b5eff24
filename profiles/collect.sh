#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline object refers to.  Run ON THE GPU BOX from the repo root:
#   bash profiles/collect.sh [workload]   (writes gpurun_out/prof_<workload>/*, then profiles/summarize.py turns it into the committed files)
# Three separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: kernel trace + stats; FETCH_SIZE; WRITE_SIZE
# (the two TCC counters do not fit one pass, and counters are never combined with a runtime/HIP trace).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-chr1_50x}            # bench.py --workload: chr1_50x = the largest contig of the headline workload (roofline is quoted there)
OUT=$ROOT/gpurun_out/prof_$WL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --workload $WL --parity none --ctx-per-gpu 1 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_under_trace.json" 2> "$OUT/stats.err"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" --workload $WL --parity none --ctx-per-gpu 1 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/fetch.err"
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" --workload $WL --parity none --ctx-per-gpu 1 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/write.err"
echo "WRITE_SIZE pass done"
cd "$ROOT" && python3 profiles/summarize.py "$OUT" gpurun_out/prof_summary_$WL $WL
