R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03i_pytest.log 2>&1; tail -4 gpurun_out/r03i_pytest.log | cut -c1-300
bash tools/gpu_round.sh r03i 2>&1 | grep -v "^+" | cut -c1-330
