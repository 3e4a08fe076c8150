R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "sort" 2>&1 | tail -3
OUT=$R/gpurun_out/r03y_sort.txt; : > $OUT
bash tools/variant_stats.sh $OUT stratified_sorted hip
GPF_SORT_TICKET=1 bash tools/variant_stats.sh $OUT stratified_sorted hip
bash tools/variant_stats.sh $OUT stratified_sorted hip
cat $OUT
for i in 1 2; do python tools/bench_configs.py "lgssm2 stratified(sorted)" 2>/dev/null | cut -c1-200; done
