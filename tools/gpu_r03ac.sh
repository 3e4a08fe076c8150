R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_views.py tests/test_line_model.py tests/test_resize.py tests/test_history.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do python tools/bench_configs.py "lgssm2 residual" config4 2>/dev/null | cut -c1-260; done
GPF_RESIDUAL_HEAD=search python tools/bench_configs.py "lgssm2 residual" config4 2>/dev/null | cut -c1-260
