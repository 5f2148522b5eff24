#!/bin/bash
# Memory-system and dispatch counters of the extraction kernel ALONE (profiles/extract_only.py: lps_phase_chromosome stops after the extraction):
#   bash profiles/collect_extract_pmc.sh build.so [workload]      (build in longphase-s_amd/csrc/ab/; run ON THE GPU BOX from the repo root)
# Separate passes of at most four counters, never combined with a trace.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$1; WL=${2:-chr1_50x}
OUT=$ROOT/gpurun_out/xpmc_${LIB%.so}
rm -rf "$OUT"; mkdir -p "$OUT"
export LPS_EXTRACT_ONLY=1 LPS_HIP_LIB=$ROOT/longphase-s_amd/csrc/ab/$LIB
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_IFETCH SQ_WAIT_INST_LDS" \
           "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_CSN_BUSY SPI_CSN_WAVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o c -- python3 "$ROOT/profiles/extract_only.py" --child --workload $WL --n 4 > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
cd "$ROOT" && python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(sys.argv[1], "summary.txt"), "w") as f:
    for k in acc:
        if "extract_phase" not in k: continue
        for c in sorted(acc[k]):
            v = acc[k][c]; f.write(f"{k} {c} n={len(v)} mean={sum(v)/len(v):.6g}\n")
print(open(os.path.join(sys.argv[1], "summary.txt")).read())
PY
