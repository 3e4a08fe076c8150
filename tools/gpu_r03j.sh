R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_line_model.py tests/test_strata.py tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/r03j_pytest.log 2>&1; tail -4 gpurun_out/r03j_pytest.log | cut -c1-300
for M in multinomial stratified; do echo "== sharded one rank, no communicator: $M"; bash tools/sharded_quick.sh $M; done 2>&1 | cut -c1-150
