set -e
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_SMEM"; do
i=$((i+1))
rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc$i.err
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,collections
for i in (1,2):
    rows=list(csv.DictReader(open(f"gpurun_out/pmc{i}/pmc_counter_collection.csv")))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k=r["Kernel_Name"][:24]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        if k.startswith(("k_extract","void k_extract","void k_haplotag")):
            print(k[:14],{c:round(sum(v)/len(v)) for c,v in agg[k].items()})
PY
