R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_views.py tests/test_line_model.py tests/test_history.py tests/test_gpu_blocks.py tests/test_resize.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python tools/bench_configs.py config4 2>/dev/null | cut -c1-120; GPF_ESS_PUBLISH=kernel python tools/bench_configs.py config4 2>/dev/null | cut -c1-120; done
python tools/small_filters.py 200 2>/dev/null | head -1 | cut -c1-160; GPF_ESS_PUBLISH=kernel python tools/small_filters.py 200 2>/dev/null | head -1 | cut -c1-160
