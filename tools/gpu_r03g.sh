R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_fullsize.py tests/test_resize.py tests/test_gpu_fuzz.py tests/test_views.py -m gpu -x -q > gpurun_out/r03g_pytest.log 2>&1; tail -4 gpurun_out/r03g_pytest.log | cut -c1-300
echo "== four passes + finish" > gpurun_out/r03g_sorted.txt; bash tools/sorted_quick.sh >> gpurun_out/r03g_sorted.txt 2>&1
echo "== GPF_SORT=radix8" >> gpurun_out/r03g_sorted.txt; GPF_SORT=radix8 bash tools/sorted_quick.sh >> gpurun_out/r03g_sorted.txt 2>&1
cat gpurun_out/r03g_sorted.txt | grep -v "k_iota\|k_init\|k_publish"
python3 tools/bench_configs.py 2>/dev/null | sed -n 3p | cut -c1-300
