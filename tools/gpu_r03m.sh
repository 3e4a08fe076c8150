R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
OUT=$R/gpurun_out/r03m_search_variants.txt; : > $OUT
bash tools/variant_stats.sh $OUT multinomial hip logg1 ns2 hip logg1
grep -v "k_iota\|k_init\|k_publish" $OUT
