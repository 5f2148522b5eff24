#!/bin/bash
# Everything the round's committed profiles come from, in one go ON THE GPU BOX (repo root): kernel stats + FETCH / WRITE passes at chr1-50x and
# chr20-30x (collect.sh), the SQ issue counters at chr1-50x (collect_issue.sh), the kernel trace of the whole-genome bench with four contexts
# (overlap_analysis.py), the kernel stats of the tumor / normal leg.  Summaries land under gpurun_out/; copy what is to be judged into profiles/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
bash profiles/collect.sh chr1_50x || echo "collect chr1 failed"
bash profiles/collect.sh chr20_30x || echo "collect chr20 failed"
bash profiles/collect_issue.sh chr1_50x || echo "issue failed"
mkdir -p gpurun_out/round
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/round/ovl" -o ovl -- python3 "$ROOT/bench.py" --parity none --no-cpu-baseline --no-somatic --steps 3 --warmup 1 > "$ROOT/gpurun_out/round/ovl_bench.json" 2> "$ROOT/gpurun_out/round/ovl.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/round/som" -o stats -- python3 "$ROOT/bench.py" --workload somatic_tn --steps 3 --parity none > "$ROOT/gpurun_out/round/som.json" 2> "$ROOT/gpurun_out/round/som.err"
cd "$ROOT"
python3 profiles/overlap_analysis.py $(find gpurun_out/round/ovl -name "*kernel_trace.csv" | head -1) > gpurun_out/round/overlap_wgs_50x.md 2> gpurun_out/round/overlap.err
echo "round collection done"
