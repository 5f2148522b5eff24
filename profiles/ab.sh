#!/bin/bash
# A/B of builds of liblps_hip.so on ONE box (boxes differ by a few percent): bash profiles/ab.sh [-w workload] [-n rounds] A.so B.so [C.so ...]
# The builds lie in longphase-s_amd/csrc/ab/ and are handed to the loader through LPS_HIP_LIB: the in-tree csrc/liblps_hip.so is never touched.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; WL=chr1_50x; N=3
while getopts "w:n:" o; do case $o in w) WL=$OPTARG;; n) N=$OPTARG;; esac; done; shift $((OPTIND-1))
for i in $(seq $N); do for v in "$@"; do
  LPS_HIP_LIB="$ROOT/longphase-s_amd/csrc/ab/$v" timeout -k 10 300 python3 "$ROOT/bench.py" --workload $WL --no-cpu-baseline --parity none > /tmp/ab.json 2> /tmp/ab.err || { echo "$v failed"; tail -3 /tmp/ab.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); st=d['stages_at_largest_contig']
print('$v', 'step', round(d['ms_per_step'],3), 'solo', round(d['roofline']['solo_call_ms'],3), ' '.join(f'{k} {v[\"ms\"]:.3f}' for k,v in st.items() if v['ms']>=0.02), 'haplotag', round(d['secondary']['ms_per_step'],3), 'hap kernel', round(d['per_contig_rank0'][0].get('haplotag_kernel_ms', 0), 4))"
done; done
