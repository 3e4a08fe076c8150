R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "sort" > gpurun_out/r03f_pytest.log 2>&1; tail -4 gpurun_out/r03f_pytest.log | cut -c1-300
echo "== sample sort" > gpurun_out/r03f_sorted.txt; bash tools/sorted_quick.sh >> gpurun_out/r03f_sorted.txt 2>&1
cat gpurun_out/r03f_sorted.txt
