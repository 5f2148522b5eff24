ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export LPS_EXTRACT_ONLY=1
cd /tmp && export TMPDIR=/tmp
for LIB in x5.so x9.so; do
export LPS_HIP_LIB=$ROOT/longphase-s_amd/csrc/ab/$LIB
OUT=$ROOT/gpurun_out/xpmc2_${LIB%.so}; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o c -- python3 "$ROOT/profiles/extract_only.py" --child --n 4 > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for p in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if "extract_phase" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {c: round(sum(v)/len(v)) for c, v in sorted(acc.items())})
PY
done
