ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in xf3 xf4; do
  export LPS_HIP_LIB=$ROOT/longphase-s_amd/csrc/ab/$v.so
  rm -rf /tmp/xp_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xp_$v -o x -- python3 -m pytest $ROOT/tests/test_scale_gpu.py -m gpu -x -q -k sv_and_mod > /tmp/xp_$v.log 2>&1
  echo "$v: $(grep k_extra_find /tmp/xp_$v/x_kernel_stats.csv | cut -d, -f1,2,4 | cut -c1-40,100-)"
  grep "k_extra_find" /tmp/xp_$v/x_kernel_stats.csv | awk -F, '{print $(NF-6), $(NF-5), $(NF-4)}'
done
