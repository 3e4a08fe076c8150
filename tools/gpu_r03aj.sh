R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "sort" 2>&1 | tail -2
OUT=$R/gpurun_out/r03aj.txt; : > $OUT
bash tools/variant_stats.sh $OUT stratified_sorted hip hip
grep "==\|k_sort_keys\|k_sort_pass\|k_sort_finish" $OUT
python tools/bench_configs.py "lgssm2 stratified(sorted)" 2>/dev/null | cut -c1-120
