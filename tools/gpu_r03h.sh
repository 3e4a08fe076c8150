R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
OUT=$R/gpurun_out/r03h_sort_variants.txt; : > $OUT
bash tools/variant_stats.sh $OUT stratified_sorted hip fin512 fin1024 kh1 kh4
grep -v "k_iota\|k_init\|k_publish\|k_step\|k_scan\|k_search" $OUT
python3 tools/bench_configs.py 2>/dev/null | sed -n 3p | cut -c1-300
