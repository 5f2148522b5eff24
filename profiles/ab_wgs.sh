#!/bin/bash
# A/B of two builds on ONE box with the whole-genome workload (four contexts): bash profiles/ab_wgs.sh A.so B.so
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do for v in "$1" "$2"; do
  cp "$ROOT/longphase-s_amd/csrc/ab/$v" "$ROOT/longphase-s_amd/csrc/liblps_hip.so"
  timeout -k 10 300 python3 "$ROOT/bench.py" --no-cpu-baseline --parity none > /tmp/ab.json 2> /tmp/ab.err || { echo "$v failed"; tail -3 /tmp/ab.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print('$v', 'pass ms', round(d['ms_per_step'],2), 'M SNPs/s', round(d['value']/1e6,1), 'haplotag ms', round(d['secondary']['ms_per_step'],2))"
done; done
