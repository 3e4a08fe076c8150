R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
GPF_FUZZ_SEEDS=240 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r03o_fuzz.log 2>&1; tail -6 gpurun_out/r03o_fuzz.log | cut -c1-400
GPF_FUZZ_SHARD_SEEDS=24 timeout 2400 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -k "random_api" > gpurun_out/r03o_fuzz_sharded.log 2>&1; tail -4 gpurun_out/r03o_fuzz_sharded.log | cut -c1-400
