# round 3, fourth GPU call: the new tests (tree sums, sharded tempering), two more variants for the scan / search decision
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_parity.py tests/test_history.py tests/test_gpu_levels.py -m gpu -x -q -k "tempered or mean_var or history or readme or levels or transport" > gpurun_out/r03d_pytest.log 2>&1; tail -5 gpurun_out/r03d_pytest.log
OUT=$R/gpurun_out/r03d_variants.txt; : > $OUT
bash tools/variant_stats.sh $OUT multinomial hip fine_noties scan_b256 hip scan_b256
cat $OUT | grep -v "k_iota\|k_init"
