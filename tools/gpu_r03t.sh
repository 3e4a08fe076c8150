R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_views.py tests/test_line_model.py tests/test_proposal.py -m gpu -x -q 2>&1 | tail -5
OUT=$R/gpurun_out/r03t_maxslots.txt; : > $OUT
bash tools/variant_stats.sh $OUT multinomial base hip base hip
bash tools/variant_stats.sh $OUT stratified base hip
grep "==\|k_scan\|k_step\|k_search" $OUT
cd $R; for i in 1 2; do python tools/bench_configs.py config2 config3 2>/dev/null | cut -c1-200; done
