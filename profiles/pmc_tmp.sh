ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for c in 3 4 5 6 8; do
  LPS_HIP_LIB=$ROOT/longphase-s_amd/csrc/ab/p4.so timeout -k 10 300 python3 $ROOT/bench.py --no-cpu-baseline --parity none --ctx-per-gpu $c > /tmp/c.json 2>/tmp/c.err || { echo "ctx $c failed"; tail -3 /tmp/c.err; }
  python3 -c "
import json
d=json.loads(open('/tmp/c.json').read().strip().splitlines()[-1])
print('ctx $c pass ms', round(d['ms_per_step'],2), 'haplotag ms', round(d['secondary']['ms_per_step'],2))"
done
