# round 3, fifth GPU call: sample sort (tests + kernel stats of the sorted-stratified loop, both sorts), proposal move tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_fullsize.py tests/test_proposal.py tests/test_line_model.py tests/test_resize.py tests/test_views.py -m gpu -x -q -k "sort or proposal or line_model or extreme or resize or views" > gpurun_out/r03e_pytest.log 2>&1; tail -8 gpurun_out/r03e_pytest.log | cut -c1-300
echo "== sample sort" > gpurun_out/r03e_sorted.txt; bash tools/sorted_quick.sh >> gpurun_out/r03e_sorted.txt 2>&1
echo "== GPF_SORT=radix" >> gpurun_out/r03e_sorted.txt; GPF_SORT=radix bash tools/sorted_quick.sh >> gpurun_out/r03e_sorted.txt 2>&1
cat gpurun_out/r03e_sorted.txt
python3 tools/bench_configs.py > gpurun_out/r03e_configs.jsonl 2>/dev/null; cut -c1-330 gpurun_out/r03e_configs.jsonl
