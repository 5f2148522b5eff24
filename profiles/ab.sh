#!/bin/bash
# A/B of two builds of csrc/liblps_hip.so on ONE box (boxes differ by a few percent): bash profiles/ab.sh A.so B.so [workload] - alternates the two, three times each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; WL=${3:-chr1_50x}
for i in 1 2 3; do for v in "$1" "$2"; do
  cp "$ROOT/longphase-s_amd/csrc/ab/$v" "$ROOT/longphase-s_amd/csrc/liblps_hip.so"
  timeout -k 10 300 python3 "$ROOT/bench.py" --workload $WL --no-cpu-baseline --parity none > /tmp/ab.json 2> /tmp/ab.err || { echo "$v failed"; tail -3 /tmp/ab.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); st=d['stages_at_largest_contig']
print('$v', 'step', round(d['ms_per_step'],3), 'dominant', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), 'edges', st['edges']['ms'], 'nodes', st['nodes']['ms'], 'node_lists', st['node_lists']['ms'], 'rc', st['read_correction']['ms'], 'merge', st['merge_rows']['ms'], 'groups', st['name_groups']['ms'], 'scan', st['vote_scan']['ms'], 'extract', st['extract']['ms'], 'haplotag', round(d['secondary']['ms_per_step'],3), 'hap kernel', round(d['per_contig_rank0'][0].get('haplotag_kernel_ms', 0), 4))"
done; done
