R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "sort or stratified" 2>&1 | tail -4
OUT=$R/gpurun_out/r03v_wide.txt; : > $OUT
bash tools/variant_stats.sh $OUT stratified_sorted hip w2n4 w2n8 w4n8 w1n8 hip
bash tools/variant_stats.sh $OUT stratified hip
grep "==\|k_search_strat" $OUT
