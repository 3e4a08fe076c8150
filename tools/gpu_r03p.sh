R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_resize.py tests/test_views.py -m gpu -x -q -k "sort or resize or extreme or views" > gpurun_out/r03p_pytest.log 2>&1; tail -3 gpurun_out/r03p_pytest.log | cut -c1-200
GPF_FUZZ_SEEDS=60 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r03p_fuzz.log 2>&1; tail -2 gpurun_out/r03p_fuzz.log
bash tools/sorted_quick.sh | grep -v "k_iota\|k_init\|k_publish"
python3 tools/bench_configs.py "lgssm2 stratified(sorted)" 2>/dev/null | cut -c1-300
