#!/bin/bash
# A/B of builds on ONE box with the whole-genome workload (four contexts): bash profiles/ab_wgs.sh [-n rounds] A.so B.so [...]  (builds in csrc/ab/, see ab.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; N=3
while getopts "n:" o; do case $o in n) N=$OPTARG;; esac; done; shift $((OPTIND-1))
for i in $(seq $N); do for v in "$@"; do
  LPS_HIP_LIB="$ROOT/longphase-s_amd/csrc/ab/$v" timeout -k 10 300 python3 "$ROOT/bench.py" --no-cpu-baseline --no-somatic --parity none > /tmp/ab.json 2> /tmp/ab.err || { echo "$v failed"; tail -3 /tmp/ab.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print('$v', 'pass ms', round(d['ms_per_step'],2), 'M SNPs/s', round(d['value']/1e6,1), 'haplotag ms', round(d['secondary']['ms_per_step'],2))"
done; done
