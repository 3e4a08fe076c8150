R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
for B in 2 3 4; do
  echo "== GPF_WSCAN_BLOCKS=$B"
  GPF_WSCAN_BLOCKS=$B python tools/bench_configs.py config5 2>/dev/null | cut -c1-230
  GPF_WSCAN_BLOCKS=$B python tools/resample_loop.py multinomial 200 1500000 2>/dev/null | tail -1
done
GPF_WSCAN_BLOCKS=4 timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
for B in 2 4; do
  D=$R/gpurun_out/prof_ws$B; rm -rf $D
  GPF_WSCAN_BLOCKS=$B rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/config_loop.py config5 40 > $D.log 2>&1
  f=$(find $D -name "*kernel_stats.csv" | head -1); echo "== $B"; grep "k_scan" $f | cut -c1-60,230-300
done
