R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
OUT=$R/gpurun_out/r03al.txt; : > $OUT
bash tools/variant_stats.sh $OUT multinomial hip nocdf hip nocdf
grep "==\|k_scan\|k_search_multi\|k_step" $OUT
