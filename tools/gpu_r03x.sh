R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_resize.py tests/test_views.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do python tools/bench_configs.py "lgssm2 stratified(sorted)" 2>/dev/null | cut -c1-300; done
