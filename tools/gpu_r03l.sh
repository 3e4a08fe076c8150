R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_resize.py -m gpu -x -q -k "sort or resize or extreme" > gpurun_out/r03l_pytest.log 2>&1; tail -3 gpurun_out/r03l_pytest.log | cut -c1-200
OUT=$R/gpurun_out/r03l_sort_variants.txt; : > $OUT
bash tools/variant_stats.sh $OUT stratified_sorted hip oneticket hip oneticket
grep -v "k_iota\|k_init\|k_publish\|k_step\|k_scan\|k_search" $OUT
