ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/xp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xp -o x -- python3 -m pytest $ROOT/tests/test_scale_gpu.py -m gpu -x -q -k sv_and_mod > /tmp/xp.log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/xp/x_kernel_stats.csv')):
    n=r["Name"]
    if any(k in n for k in ("k_extra","k_read_x0","k_extract_phase","k_count_ranks","k_graph_rows","k_edges")): print(n[:40], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
