R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 2000 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_views.py tests/test_line_model.py tests/test_resize.py -m gpu -x -q 2>&1 | tail -8
OUT=$R/gpurun_out/r03u_sort3.txt; : > $OUT
bash tools/variant_stats.sh $OUT stratified_sorted base hip base hip
cat $OUT
cd $R; python tools/bench_configs.py sorted 2>/dev/null | cut -c1-260
