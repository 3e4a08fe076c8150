R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
OUT=$R/gpurun_out/r03q_finish_variants.txt; : > $OUT
bash tools/variant_stats.sh $OUT stratified_sorted hip fin256 hip fin256
grep "==\|k_sort_finish\|k_sort_pass" $OUT
