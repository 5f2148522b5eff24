#!/bin/bash
# Instruction-issue evidence for the hot kernels: SQ counters of the same bench command as collect.sh, in separate passes of at
# most four counters (never combined with a trace).  Run ON THE GPU BOX from the repo root:  bash profiles/collect_issue.sh [workload]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-chr1_50x}
OUT=$ROOT/gpurun_out/issue_$WL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o c -- python3 "$ROOT/bench.py" --workload $WL --parity none --ctx-per-gpu 1 --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/p$i.json" 2> "$OUT/p$i.err" || echo "pass $i failed"
  echo "pass $i done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --workload $WL --parity none --ctx-per-gpu 1 --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/stats.json" 2> "$OUT/stats.err"
cd "$ROOT" && python3 profiles/summarize_issue.py "$OUT" gpurun_out/issue_summary_$WL.md $WL
